#!/bin/bash
# round 5, GPU run L: flat deep cross-attention back on the speculative kernel; then the round's evidence set
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; mkdir -p $O
V=tools/build/variants
CF="tiny.en:1,tiny.en:8,tiny.en:24,base:1,base:8,small:1,large-v2:1,large-v2:2,large-v2:4,large-v2:8,large-v2:12,large-v3:15,large-v2:24"
timeout 900 python tools/gpu_latency_probe.py "" "$CF" > $O/lat_new.txt 2>&1; cat $O/lat_new.txt
WM_LIB_PATH=$PWD/$V/r5c.so WM_DBG_LIB_PATH=$PWD/$V/r5c_dbg.so timeout 900 python tools/gpu_latency_probe.py "xattn_deep8_max_pairs=0,xattn_pair_wg_max_pairs=0" "tiny.en:1,base:1,small:1,large-v2:1" > $O/lat_r5c.txt 2>&1; cat $O/lat_r5c.txt
WM_LIB_PATH=$PWD/$V/base.so WM_DBG_LIB_PATH=$PWD/$V/base_dbg.so timeout 900 python tools/gpu_latency_probe.py "" "$CF" > $O/lat_base.txt 2>&1; cat $O/lat_base.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench driver rc=$?"
WM_LIB_PATH=$PWD/$V/base.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_base.json 2> $O/bench_base.err; echo "bench base rc=$?"
timeout 900 python bench.py --no-cpu-baseline --no-other-configs > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python - <<'PY'
import json
for v in ("driver", "base", "default"):
    try:
        d = json.loads(open("gpurun_out/r5l/bench_%s.json" % v).read().strip().splitlines()[-1])
        print(v, "value %.1f batch8 %.1f decode frac %.3f enc frac %.3f roof %.3f (%.2f us) step %.3f early %.1f checks %s" % (d["value"], d["value_batch8"], d["stage_roofline"]["decode"]["frac"], d["stage_roofline"]["encoder_xkv"]["frac"], d["roofline"]["frac"], d["roofline"]["avg_us"], d["step_roofline"]["frac"], (d.get("early_stop") or {}).get("value", 0), all(d["token_checks"].values())))
        if v == "driver":
            oc = d["other_configs"]
            print({k: (round(x["value"], 1) if x.get("value") else x) for k, x in oc.items()})
            print({k: round(x["step_roofline"]["frac"], 3) for k, x in oc.items() if "step_roofline" in x})
            print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
    except Exception as e:
        print(v, "failed", e)
PY
timeout 1500 bash tools/run_sanitized.sh gpu > $O/sanitized.txt 2>&1; echo "sanitized rc=$?"; grep -v "^  File\|^$" $O/sanitized.txt | tail -5
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
bash tools/profile_tiny_en.sh r05 > gpurun_out/r05_kernel_trace_tiny_en_b1_summary.txt 2>&1; head -14 gpurun_out/r05_kernel_trace_tiny_en_b1_summary.txt
