#!/bin/bash
# round 5, GPU run K: row split as a compile-time choice of the residual kernels only
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k; mkdir -p $O
V=tools/build/variants
for b in 1 2; do
echo "== new B=$b"; timeout 300 python tools/gpu_decode_probe.py $b 2>&1 | grep -v "launch floor\|cross\|self"
echo "== r5c B=$b"; WM_LIB_PATH=$PWD/$V/r5c.so WM_DBG_LIB_PATH=$PWD/$V/r5c_dbg.so timeout 300 python tools/gpu_decode_probe.py $b 2>&1 | grep -v "launch floor\|cross\|self"
done
CF="tiny.en:1,tiny.en:8,tiny.en:24,base:1,base:8,small:1,large-v2:1,large-v2:2,large-v2:4,large-v2:8,large-v2:12,large-v3:15,large-v2:24"
timeout 900 python tools/gpu_latency_probe.py "" "$CF" > $O/lat_new.txt 2>&1; cat $O/lat_new.txt
WM_LIB_PATH=$PWD/$V/r5c.so WM_DBG_LIB_PATH=$PWD/$V/r5c_dbg.so timeout 900 python tools/gpu_latency_probe.py "xattn_deep8_max_pairs=0,xattn_pair_wg_max_pairs=0" "$CF" > $O/lat_r5c.txt 2>&1; cat $O/lat_r5c.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
