#!/bin/bash
# round 6, GPU call 4: finer sub-chip-lane sweep on the narrow models, encoder traces after the round-6 kernel edits, new tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 900 python tools/gpu_group_policy_probe.py base 20,24,28,32,40,48,56,64,80,96,128 gc=128,split=2,parts=2 > $O/r06_group_policy_base_fine.txt 2>&1; cat $O/r06_group_policy_base_fine.txt
timeout 900 python tools/gpu_group_policy_probe.py tiny.en 20,24,28,32,40,48,56,64,96 gc=128,split=2,parts=2 > $O/r06_group_policy_tiny_fine.txt 2>&1; cat $O/r06_group_policy_tiny_fine.txt
rm -rf /tmp/prof_s1
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_s1 -- python tools/gpu_encode_only.py small 1 5 > $O/r06_trace_small_b1.log 2>&1
DB=$(find /tmp/prof_s1 -name "*.db" | head -1); python tools/rocprof_summary.py $DB 40 > $O/r06_kernel_trace_small_b1_encoder_summary_v2.txt
head -14 $O/r06_kernel_trace_small_b1_encoder_summary_v2.txt | cut -c1-70,92-150
rm -rf /tmp/prof_l56
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_l56 -- python tools/gpu_encode_only.py large-v2 56 2 > $O/r06_trace_large_b56.log 2>&1
DB=$(find /tmp/prof_l56 -name "*.db" | head -1); python tools/rocprof_summary.py $DB 40 > $O/r06_kernel_trace_large_v2_b56_encoder_summary.txt
head -12 $O/r06_kernel_trace_large_v2_b56_encoder_summary.txt | cut -c1-70,92-150
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "tiny_en or base_geometry" 2>&1 | tail -40
