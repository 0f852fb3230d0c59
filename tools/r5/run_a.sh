#!/bin/bash
# round 5, GPU run A: correctness of the re-ordered decode kernel arguments + A/B against the round-4 build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
V=tools/build/variants
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
for v in base nopreload new; do
  if [ $v = new ]; then unset WM_LIB_PATH WM_DBG_LIB_PATH; else export WM_LIB_PATH=$PWD/$V/$v.so WM_DBG_LIB_PATH=$PWD/$V/${v}_dbg.so; fi
  timeout 600 python tools/gpu_latency_probe.py > $O/lat_$v.txt 2>&1; echo "== latency $v"; cat $O/lat_$v.txt
  timeout 300 python tools/gpu_decode_probe.py > $O/probe_$v.txt 2>&1; echo "== probe $v"; cat $O/probe_$v.txt
done
unset WM_LIB_PATH WM_DBG_LIB_PATH
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_new.json 2> $O/bench_new.err; echo "bench new rc=$?"
WM_LIB_PATH=$PWD/$V/base.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err; echo "bench base rc=$?"
python - <<'PY'
import json
for v in ("new", "base"):
    try:
        d = json.loads(open("gpurun_out/r5a/bench_%s.json" % v).read().strip().splitlines()[-1])
        oc = d.get("other_configs") or {}
        print(v, "value %.1f batch8 %.1f decode frac %.3f enc frac %.3f checks %s" % (d["value"], d["value_batch8"], d["stage_roofline"]["decode"]["frac"], d["stage_roofline"]["encoder_xkv"]["frac"], d["token_checks"]),
              {k: round(x["value"], 1) for k, x in oc.items() if x.get("value")})
    except Exception as e:
        print(v, "failed", e)
PY
