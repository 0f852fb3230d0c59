#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5y; mkdir -p $O
for rep in 1 2 3; do
for k in "" "--tuning gemv_nblk=1"; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $k 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k', 'value %.1f batch8 %.1f decode %.3f early %.1f checks %s' % (d['value'], d['value_batch8'], d['stage_roofline']['decode']['frac'], d['early_stop']['value'], d['tokens_consistent_across_groups']))"
done
done
