#!/bin/bash
# round 5, GPU run E: row-split epilogue without the stagger regression (same-box A/B against the build before it), new
# group policy, sanitizer leg retry
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
V=tools/build/variants
timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
CF="tiny.en:1,tiny.en:8,base:1,base:8,small:1,large-v2:1,large-v2:4,large-v2:8,large-v2:12,large-v3:15,large-v2:24"
timeout 900 python tools/gpu_latency_probe.py "" "$CF" > $O/lat_new.txt 2>&1; cat $O/lat_new.txt
WM_LIB_PATH=$PWD/$V/r5c.so WM_DBG_LIB_PATH=$PWD/$V/r5c_dbg.so timeout 900 python tools/gpu_latency_probe.py "xattn_deep8_max_pairs=0,xattn_pair_wg_max_pairs=0" "$CF" > $O/lat_r5c.txt 2>&1; cat $O/lat_r5c.txt
timeout 300 python tools/gpu_decode_probe.py > $O/probe.txt 2>&1; grep -v "launch floor" $O/probe.txt
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_new$i.json 2> $O/bench_new$i.err; echo "bench new rc=$?"
WM_LIB_PATH=$PWD/$V/r5c.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_r5c$i.json 2> $O/bench_r5c$i.err; echo "bench r5c rc=$?"
done
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?"
python - <<'PY'
import json
for v in ("new1", "r5c1", "new2", "r5c2", "full"):
    try:
        d = json.loads(open("gpurun_out/r5e/bench_%s.json" % v).read().strip().splitlines()[-1])
        print(v, "value %.1f batch8 %.1f decode frac %.3f enc frac %.3f roof %.3f checks %s" % (d["value"], d["value_batch8"], d["stage_roofline"]["decode"]["frac"], d["stage_roofline"]["encoder_xkv"]["frac"], d["roofline"]["frac"], all(d["token_checks"].values())))
        print("    alone", {k: round(x["avg_us"], 1) for k, x in d["kernel_families"].items() if k.startswith("dec_")})
        if v == "full":
            oc = d["other_configs"]
            print({k: (round(x["value"], 1) if x.get("value") else x) for k, x in oc.items()})
            print(json.dumps(oc.get("small_lid_reference_flow"), indent=1)); print(json.dumps(oc.get("frontend_reference_abi"), indent=1))
    except Exception as e:
        print(v, "failed", e)
PY
timeout 1500 bash tools/run_sanitized.sh gpu > $O/sanitized.txt 2>&1; echo "sanitized rc=$?"; grep -v "^  File\|^$" $O/sanitized.txt | tail -40
