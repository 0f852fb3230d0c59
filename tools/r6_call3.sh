#!/bin/bash
# round 6, GPU call 3: sub-chip lanes on the SMALL models (their chains barely notice the CU count), the 64 x 64 GEMM tile and
# 2-wave attention at one chunk, the new parity tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 600 python tools/gpu_group_policy_probe.py base 16,32,48,64 gc=128,split=2,parts=2,split=3,parts=3,product > $O/r06_group_policy_base.txt 2>&1; cat $O/r06_group_policy_base.txt
timeout 600 python tools/gpu_group_policy_probe.py tiny.en 8,16,32,64 gc=128,split=2,parts=2,split=3,parts=3,product > $O/r06_group_policy_tiny.txt 2>&1; cat $O/r06_group_policy_tiny.txt
timeout 600 python tools/gpu_group_policy_probe.py small 16,32,64 gc=128,split=2,parts=2,split=3,parts=3,product > $O/r06_group_policy_small.txt 2>&1; cat $O/r06_group_policy_small.txt
for M in small tiny.en base; do
  for K in "gemm_tile=128 enc_attn_waves=4" "enc_attn_waves=4" "gemm_tile=128" "gemm_tile=0"; do
    echo "== $M x 1: $K"; timeout 300 python tools/gpu_encode_only.py $M 1 30 $K 2>&1 | tail -1
  done
done > $O/r06_single_chunk_encoder_ab.txt 2>&1; cat $O/r06_single_chunk_encoder_ab.txt
for K in "gemm_tile=0" "enc_attn_mfma_sum=1"; do echo "== large-v2 x 56: $K"; timeout 300 python tools/gpu_encode_only.py large-v2 56 3 $K 2>&1 | tail -1; done > $O/r06_enc_attn_mfma_sum_ab.txt 2>&1; cat $O/r06_enc_attn_mfma_sum_ab.txt
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "tile or fused_query or small_geometry or tiny_en or swift" 2>&1 | tail -5
