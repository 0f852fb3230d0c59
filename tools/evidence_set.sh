#!/bin/bash
# The committed evidence set of a round's FINAL build (run on the GPU box through gpurun; was tools/r5/run_n.sh):
#   1. tools/profile_round.sh <round>: rocprofv3 kernel trace of ONE decode group of 56 chunks on one lane + FETCH_SIZE and MFMA-busy
#      PMC passes (separate runs) + the literal batch-of-8 trace  -> gpurun_out/<round>_*.txt
#   2. profiles/pmc_traffic.json refreshed from the FETCH_SIZE pass (bench.py reports `roofline.traffic` only while the
#      kernel source's sha256 matches the pass)
#   3. tools/profile_tiny_en.sh <round>: the tiny.en single-chunk trace
#   4. the two bench lines: the driver's command (--steps 20 --warmup 5) and the default run
# usage: bash tools/evidence_set.sh r06      (then copy gpurun_out/<round>_* into profiles/)
R=${1:-rXX}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
bash tools/profile_round.sh $R > $O/${R}_profile_round.log 2>&1; tail -3 $O/${R}_profile_round.log
python tools/pmc_traffic_update.py $O/${R}_pmc_fetch_size_group56.txt large-v2 56 "profiles/${R}_pmc_fetch_size_group56.txt"
cp profiles/pmc_traffic.json $O/pmc_traffic.json
bash tools/profile_tiny_en.sh $R > $O/${R}_kernel_trace_tiny_en_b1_summary.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/${R}_bench_n1_driver_cmd.json 2> $O/${R}_bench_n1_driver_cmd.err; echo "bench driver rc=$?"
timeout 900 python bench.py > $O/${R}_bench_n1_default.json 2> $O/${R}_bench_n1_default.err; echo "bench default rc=$?"
python - "$R" <<'PY'
import json, sys
R = sys.argv[1]
for v in ("driver_cmd", "default"):
    d = json.loads(open("gpurun_out/%s_bench_n1_%s.json" % (R, v)).read().strip().splitlines()[-1])
    print(v, json.dumps(d.get("summary")))
PY
