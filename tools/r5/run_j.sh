#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j; mkdir -p $O
V=tools/build/variants
for b in 1 2; do
echo "== new B=$b"; timeout 300 python tools/gpu_decode_probe.py $b 2>&1 | grep -v "launch floor\|cross\|self"
echo "== r5c B=$b"; WM_LIB_PATH=$PWD/$V/r5c.so WM_DBG_LIB_PATH=$PWD/$V/r5c_dbg.so timeout 300 python tools/gpu_decode_probe.py $b 2>&1 | grep -v "launch floor\|cross\|self"
done
