#!/bin/bash
# round 5, GPU run O: final sanity of the committed tree -- smoke(), the whole -m gpu suite, one driver-command bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5o; mkdir -p $O
timeout 600 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench driver rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5o/bench_driver.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f batch8 %.1f decode %.3f enc %.3f roof %.3f traffic %s checks %s" % (d["value"], d["value_batch8"], d["stage_roofline"]["decode"]["frac"], d["stage_roofline"]["encoder_xkv"]["frac"], r["frac"], r["traffic"], d["token_checks"]))
PY
