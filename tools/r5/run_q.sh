#!/bin/bash
# round 5, GPU run Q: two probes (logits TN = 8 at d <= 512; arg-max rows per workgroup), then the per-lane trace retry
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q; mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "timestamp or suppress or early_stop_equals or tie or lively or policy or full_text or tiny_en or base_geometry" > $O/tests_am.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests_am.txt
timeout 900 python tools/gpu_latency_probe.py ";logits_tn=8;argmax_rows_per_wg=1;argmax_rows_per_wg=4;logits_tn=8,argmax_rows_per_wg=1" "tiny.en:1,tiny.en:8,base:1,base:8,small:1,large-v2:1,large-v2:8,large-v3:15,large-v2:24" > $O/lat.txt 2>&1; cat $O/lat.txt
for k in "" "--tuning argmax_rows_per_wg=1" "--tuning argmax_rows_per_wg=4"; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-early-stop $k 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k', 'value %.1f batch8 %.1f decode %.3f' % (d['value'], d['value_batch8'], d['stage_roofline']['decode']['frac']), {k: round(v,1) for k,v in d['roofline']['in_situ']['families_avg_us'].items() if 'argmax' in k})"
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for try in 1 2; do
rm -rf /tmp/prof_lanes
WM_NO_GRAPH=1 timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_lanes -- python bench.py --steps 20 --warmup 5 --new-tokens 28 --no-cpu-baseline --no-early-stop --no-other-configs --no-single-batch > gpurun_out/r05_lanes.json 2> gpurun_out/r05_lanes.err
DB=$(find /tmp/prof_lanes -name "*.db" 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocprof_lanes.py $DB 56 1 > gpurun_out/r05_timed_config_lanes.txt 2>&1; cat gpurun_out/r05_timed_config_lanes.txt; break; fi
echo "rocprofv3 try $try failed"
done
