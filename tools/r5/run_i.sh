#!/bin/bash
# round 5, GPU run I: balanced fused query + cross-attention, shifts instead of divisions in the row split
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; mkdir -p $O
V=tools/build/variants
timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "fused_query or bit_level or flat_cross" > $O/tests_fq.txt 2>&1; echo "fq tests rc=$?"; tail -5 $O/tests_fq.txt
CF="tiny.en:1,tiny.en:8,tiny.en:24,base:1,base:8,base:16,small:1,small:8,large-v2:1,large-v2:4,large-v2:5,large-v2:8,large-v2:10,large-v2:12,large-v3:15,large-v2:24"
timeout 900 python tools/gpu_latency_probe.py ";xattn_fuse_q=0" "$CF" > $O/lat_new.txt 2>&1; cat $O/lat_new.txt
WM_LIB_PATH=$PWD/$V/r5c.so WM_DBG_LIB_PATH=$PWD/$V/r5c_dbg.so timeout 900 python tools/gpu_latency_probe.py "xattn_deep8_max_pairs=0,xattn_pair_wg_max_pairs=0" "tiny.en:1,tiny.en:8,base:1,small:1,large-v2:1,large-v2:8" > $O/lat_r5c.txt 2>&1; cat $O/lat_r5c.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
