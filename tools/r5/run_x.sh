#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5x; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "gemm or whole_model or encoder or small_geometry or large_v3" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt | head -1
for rep in 1 2 3; do
for k in "" "--tuning gemm_no_tail_split=1"; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $k 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k', 'value %.1f batch8 %.1f decode %.3f enc %.4f early %.1f checks %s' % (d['value'], d['value_batch8'], d['stage_roofline']['decode']['frac'], d['stage_roofline']['encoder_xkv']['frac'], d['early_stop']['value'], d['tokens_consistent_across_groups']))"
done
done
