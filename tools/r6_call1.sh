#!/bin/bash
# round 6, GPU call 1: CU-mask lab + encoder stall counters + kernel trace of the reference's own geometry (small, one chunk)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 600 tools/build/cu_mask_lab > $O/r06_cu_mask_lab.txt 2>&1; echo "lab rc $?"
tail -50 $O/r06_cu_mask_lab.txt
bash tools/profile_encoder_stalls.sh r06 large-v2 56
rm -rf /tmp/prof_s1
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_s1 -- python tools/gpu_encode_only.py small 1 5 > $O/r06_trace_small_b1.log 2>&1
DB=$(find /tmp/prof_s1 -name "*.db" | head -1); python tools/rocprof_summary.py $DB 40 > $O/r06_kernel_trace_small_b1_encoder_summary.txt
head -30 $O/r06_kernel_trace_small_b1_encoder_summary.txt
# first look at CU-masked sub-chip lanes (launch shapes sized by the lane's CU count)
timeout 900 python tools/gpu_group_policy_probe.py large-v2 8,12,15,16,24,32,48 gc=128,gc=8,parts=2,partsx=2,parts=3 > $O/r06_group_policy_first.txt 2>&1
cat $O/r06_group_policy_first.txt
