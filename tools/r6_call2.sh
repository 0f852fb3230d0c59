#!/bin/bash
# round 6, GPU call 2: WHY do CU-masked sub-chip lanes lose?  one decode chain alone on n CUs of every XCD (CU-count sensitivity),
# per-kernel traces of the same chain on 256 vs 128 CUs, masked vs unmasked pairs; + the lively-gain calibration
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 900 python tools/gpu_latency_probe.py ";lane_solo_cus=24;lane_solo_cus=16;lane_solo_cus=11;lane_solo_cus=8" large-v2:8,large-v2:4,large-v2:12,large-v2:1,base:16,tiny.en:1 > $O/r06_solo_lane_latency.txt 2>&1
cat $O/r06_solo_lane_latency.txt
for K in "" "lane_solo_cus=16"; do
  T=full; [ -n "$K" ] && T=half
  rm -rf /tmp/prof_$T
  WM_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$T -- python tools/gpu_decode_once.py large-v2 8 24 $K > $O/r06_trace_b8_$T.log 2>&1
  DB=$(find /tmp/prof_$T -name "*.db" | head -1); python tools/rocprof_summary.py $DB 30 > $O/r06_kernel_trace_b8_${T}_chip_summary.txt
  head -22 $O/r06_kernel_trace_b8_${T}_chip_summary.txt
done
timeout 900 python tools/gpu_group_policy_probe.py large-v2 16,24,48 gc=128,split=2,parts=2,parts=3 > $O/r06_group_policy_second.txt 2>&1
cat $O/r06_group_policy_second.txt
timeout 900 python tools/gpu_lively_gain_probe.py tiny.en,base,small 4,6,8,12,16 32 96 > $O/r06_lively_gain.txt 2>&1
cat $O/r06_lively_gain.txt
