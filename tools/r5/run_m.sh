#!/bin/bash
# round 5, GPU run M: helper waves in the step-closing arg-max
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5m; mkdir -p $O
V=tools/build/variants
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "timestamp or suppress or early_stop_equals or argmax or tie or lively or policy or full_text" > $O/tests_am.txt 2>&1; echo "argmax tests rc=$?"; tail -3 $O/tests_am.txt
CF="tiny.en:1,tiny.en:2,tiny.en:8,base:1,small:1,large-v2:1,large-v2:8,large-v3:15,large-v2:24"
timeout 900 python tools/gpu_latency_probe.py "" "$CF" > $O/lat_new.txt 2>&1; cat $O/lat_new.txt
WM_LIB_PATH=$PWD/$V/r5c.so WM_DBG_LIB_PATH=$PWD/$V/r5c_dbg.so timeout 900 python tools/gpu_latency_probe.py "xattn_deep8_max_pairs=0,xattn_pair_wg_max_pairs=0" "tiny.en:1,tiny.en:8,base:1,large-v2:1" > $O/lat_r5c.txt 2>&1; cat $O/lat_r5c.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5m/bench.json").read().strip().splitlines()[-1])
print("value %.1f batch8 %.1f decode frac %.3f roof %.3f checks %s" % (d["value"], d["value_batch8"], d["stage_roofline"]["decode"]["frac"], d["roofline"]["frac"], all(d["token_checks"].values())))
print({k: (round(x["value"], 1) if x.get("value") else x) for k, x in d["other_configs"].items()})
print("alone", {k: round(x["avg_us"], 1) for k, x in d["kernel_families"].items() if k.startswith(("dec_", "argmax"))})
PY
