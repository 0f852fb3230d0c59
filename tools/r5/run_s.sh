#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s; mkdir -p $O
timeout 600 python tools/r5/fc2_probe.py 2>&1 | tee $O/fc2_probe.txt
for k in "" "--tuning gemv_ppw2_nblk=2" "" "--tuning gemv_ppw2_nblk=2"; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $k 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k', 'value %.1f batch8 %.1f decode %.3f early %.1f checks %s' % (d['value'], d['value_batch8'], d['stage_roofline']['decode']['frac'], d['early_stop']['value'], d['tokens_consistent_across_groups']))"
done
