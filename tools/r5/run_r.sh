#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
for k in "" "--tuning argmax_rows_per_wg=16"; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $k 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k', 'value %.1f batch8 %.1f decode %.3f early %.1f' % (d['value'], d['value_batch8'], d['stage_roofline']['decode']['frac'], d['early_stop']['value']))"
done
bash tools/r5/run_n.sh
