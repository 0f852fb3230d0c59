#!/bin/bash
# round 5, GPU run B: the re-ordered arguments after the cross-attention fix -- tests, latency, bench A/B vs round 4
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
V=tools/build/variants
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -6 $O/tests.txt
for v in new base; do
  if [ $v = new ]; then unset WM_LIB_PATH WM_DBG_LIB_PATH; else export WM_LIB_PATH=$PWD/$V/$v.so WM_DBG_LIB_PATH=$PWD/$V/${v}_dbg.so; fi
  timeout 600 python tools/gpu_latency_probe.py > $O/lat_$v.txt 2>&1; echo "== latency $v"; cat $O/lat_$v.txt
  timeout 300 python tools/gpu_decode_probe.py > $O/probe_$v.txt 2>&1; echo "== probe $v"; grep -v "launch floor" $O/probe_$v.txt
done
unset WM_LIB_PATH WM_DBG_LIB_PATH
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_new$i.json 2> $O/bench_new$i.err; echo "bench new rc=$?"
WM_LIB_PATH=$PWD/$V/base.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_base$i.json 2> $O/bench_base$i.err; echo "bench base rc=$?"
done
python - <<'PY'
import json
for v in ("new1", "base1", "new2", "base2"):
    try:
        d = json.loads(open("gpurun_out/r5b/bench_%s.json" % v).read().strip().splitlines()[-1])
        oc = d.get("other_configs") or {}
        print(v, "value %.1f batch8 %.1f decode frac %.3f enc frac %.3f roof %.3f checks %s" % (d["value"], d["value_batch8"], d["stage_roofline"]["decode"]["frac"], d["stage_roofline"]["encoder_xkv"]["frac"], d["roofline"]["frac"], all(d["token_checks"].values())),
              {k: round(x["value"], 1) for k, x in oc.items() if x.get("value")})
        if v == "new1":
            print(json.dumps({k: oc[k] for k in ("small_lid_reference_flow", "frontend_reference_abi", "large-v3_15_chunks_product_lanes") if k in oc}, indent=1))
    except Exception as e:
        print(v, "failed", e)
PY
