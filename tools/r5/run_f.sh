#!/bin/bash
# round 5, GPU run F: row-split with scalar LDS reads; tiny.en shapes; UBSan GPU leg; the round's profile set
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
V=tools/build/variants
timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
CF="tiny.en:1,tiny.en:8,base:1,base:8,small:1,large-v2:1,large-v2:4,large-v2:8,large-v2:12,large-v3:15,large-v2:24"
timeout 900 python tools/gpu_latency_probe.py ";xattn_split_below=0,xattn_deep8_max_pairs=256;xattn_split_below=0" "tiny.en:1,tiny.en:2,base:1,small:1,large-v2:1,large-v2:2" > $O/lat_flat.txt 2>&1; cat $O/lat_flat.txt
timeout 900 python tools/gpu_latency_probe.py "" "$CF" > $O/lat_new.txt 2>&1; cat $O/lat_new.txt
WM_LIB_PATH=$PWD/$V/r5c.so WM_DBG_LIB_PATH=$PWD/$V/r5c_dbg.so timeout 900 python tools/gpu_latency_probe.py "xattn_deep8_max_pairs=0,xattn_pair_wg_max_pairs=0" "$CF" > $O/lat_r5c.txt 2>&1; cat $O/lat_r5c.txt
WM_LIB_PATH=$PWD/$V/base.so WM_DBG_LIB_PATH=$PWD/$V/base_dbg.so timeout 900 python tools/gpu_latency_probe.py "" "$CF" > $O/lat_base.txt 2>&1; cat $O/lat_base.txt
timeout 300 python tools/gpu_decode_probe.py > $O/probe.txt 2>&1; grep -v "launch floor" $O/probe.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5f/bench_full.json").read().strip().splitlines()[-1])
print("value %.1f batch8 %.1f decode frac %.3f enc frac %.3f roof %.3f checks %s" % (d["value"], d["value_batch8"], d["stage_roofline"]["decode"]["frac"], d["stage_roofline"]["encoder_xkv"]["frac"], d["roofline"]["frac"], all(d["token_checks"].values())))
oc = d["other_configs"]
print({k: (round(x["value"], 1) if x.get("value") else x) for k, x in oc.items()})
PY
timeout 1500 bash tools/run_sanitized.sh gpu > $O/sanitized.txt 2>&1; echo "sanitized rc=$?"; grep -v "^  File\|^$" $O/sanitized.txt | tail -12
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
bash tools/profile_tiny_en.sh r05 > $O/profile_tiny.log 2>&1; tail -5 $O/profile_tiny.log
