#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for k in "" "--tuning lane_barrier=32,lane_barrier_n=3" "--tuning lane_barrier=4,lane_barrier_n=3" "--tuning lane_barrier=1,lane_barrier_n=3"; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $k 2> gpurun_out/r5v.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k', 'value %.1f batch8 %.1f decode %.3f early %.1f checks %s' % (d['value'], d['value_batch8'], d['stage_roofline']['decode']['frac'], d['early_stop']['value'], d['tokens_consistent_across_groups']))"
done
done
