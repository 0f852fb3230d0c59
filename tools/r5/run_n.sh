#!/bin/bash
# round 5, GPU run N: the committed evidence set of the FINAL build (profiles first, so that bench.py's roofline.traffic is
# the PMC pass of this very kernel source), then the two bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n; mkdir -p $O
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
python tools/pmc_traffic_update.py gpurun_out/r05_pmc_fetch_size_group56.txt large-v2 56 "profiles/r05_pmc_fetch_size_group56.txt"
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
bash tools/profile_tiny_en.sh r05 > gpurun_out/r05_kernel_trace_tiny_en_b1_summary.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench driver rc=$?"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python - <<'PY'
import json
for v in ("driver", "default"):
    d = json.loads(open("gpurun_out/r5n/bench_%s.json" % v).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(v, "value %.1f batch8 %.1f decode %.3f enc %.3f roof %.3f (%.2f us) traffic %s step %.3f early %.1f checks %s" % (d["value"], d["value_batch8"], d["stage_roofline"]["decode"]["frac"], d["stage_roofline"]["encoder_xkv"]["frac"], r["frac"], r["avg_us"], r["traffic"], d["step_roofline"]["frac"], (d.get("early_stop") or {}).get("value", 0), all(d["token_checks"].values())))
    oc = d.get("other_configs") or {}
    print({k: (round(x["value"], 1) if x.get("value") else x) for k, x in oc.items()})
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
