#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
for m in small base tiny.en; do timeout 300 python tools/gpu_small_flow_probe.py $m; done > $O/r06_small_flow_parts_v4.txt 2>&1; cat $O/r06_small_flow_parts_v4.txt
for cfg in "small 1" "base 1" "tiny.en 1" "small 2" "base 2" "base 4"; do echo "== $cfg"; timeout 300 python tools/gpu_encode_only.py $cfg 20 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "tile or small_geometry or swift or language" 2>&1 | tail -3
