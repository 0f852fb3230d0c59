#!/bin/bash
# round 6, GPU call 5: the pipelined 64 x 64 tile (kernel tests first), single-chunk encoder A/B, new model / bench tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -5
for M in small tiny.en base; do
  for K in "gemm_tile=128" "gemm_tile=0"; do echo "== $M x 1: $K"; timeout 300 python tools/gpu_encode_only.py $M 1 30 $K 2>&1 | tail -1; done
done > $O/r06_single_chunk_encoder_ab2.txt 2>&1; cat $O/r06_single_chunk_encoder_ab2.txt
rm -rf /tmp/prof_s1
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_s1 -- python tools/gpu_encode_only.py small 1 5 > $O/r06_trace_small_b1.log 2>&1
DB=$(find /tmp/prof_s1 -name "*.db" | head -1); python tools/rocprof_summary.py $DB 40 > $O/r06_kernel_trace_small_b1_encoder_summary_v3.txt
head -14 $O/r06_kernel_trace_small_b1_encoder_summary_v3.txt | cut -c1-70,92-150
timeout 2400 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "tiny_en or base_geometry or tile or small_geometry or fused_query" 2>&1 | tail -30
timeout 2400 python -m pytest tests/test_bench_gpu.py -x -q -m gpu -k "config5 or gpus_flag" 2>&1 | tail -30
cp $O/parity_margins_tests.txt $O/r06_parity_margins_partial.txt 2>/dev/null
