#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 300 env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $2 2> gpurun_out/r5w.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', 'value %.1f batch8 %.1f decode %.3f early %.1f checks %s' % (d['value'], d['value_batch8'], d['stage_roofline']['decode']['frac'], d['early_stop']['value'], d['tokens_consistent_across_groups']))"; }
for rep in 1 2; do
run "X=1" ""
run "GPU_MAX_HW_QUEUES=8" ""
run "GPU_MAX_HW_QUEUES=2" ""
run "X=1" "--inflight 4"
run "GPU_MAX_HW_QUEUES=8" "--inflight 4"
run "GPU_MAX_HW_QUEUES=8" "--inflight 5"
done
