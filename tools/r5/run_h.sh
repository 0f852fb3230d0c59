#!/bin/bash
# round 5, GPU run H: cross-attention family back on the round-4 body; fused query + cross-attention (96..256 pairs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
V=tools/build/variants
timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "fused_query or bit_level or flat_cross" > $O/tests_fq.txt 2>&1; echo "fq tests rc=$?"; tail -15 $O/tests_fq.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
CF="tiny.en:1,tiny.en:8,tiny.en:24,base:1,base:8,base:16,small:1,small:8,large-v2:1,large-v2:4,large-v2:5,large-v2:8,large-v2:12,large-v3:15,large-v2:24"
timeout 900 python tools/gpu_latency_probe.py ";xattn_fuse_q=0" "$CF" > $O/lat_new.txt 2>&1; cat $O/lat_new.txt
WM_LIB_PATH=$PWD/$V/r5c.so WM_DBG_LIB_PATH=$PWD/$V/r5c_dbg.so timeout 900 python tools/gpu_latency_probe.py "xattn_deep8_max_pairs=0,xattn_pair_wg_max_pairs=0" "tiny.en:1,base:1,large-v2:1,large-v2:8" > $O/lat_r5c.txt 2>&1; cat $O/lat_r5c.txt
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_new$i.json 2> $O/bench_new$i.err; echo "bench new rc=$?"
WM_LIB_PATH=$PWD/$V/base.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_base$i.json 2> $O/bench_base$i.err; echo "bench base rc=$?"
done
python - <<'PY'
import json
for v in ("new1", "base1", "new2", "base2"):
    try:
        d = json.loads(open("gpurun_out/r5h/bench_%s.json" % v).read().strip().splitlines()[-1])
        print(v, "value %.1f batch8 %.1f decode frac %.3f enc frac %.3f roof %.3f (%.2f us) checks %s" % (d["value"], d["value_batch8"], d["stage_roofline"]["decode"]["frac"], d["stage_roofline"]["encoder_xkv"]["frac"], d["roofline"]["frac"], d["roofline"]["avg_us"], all(d["token_checks"].values())))
    except Exception as e:
        print(v, "failed", e)
PY
