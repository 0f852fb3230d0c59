#!/bin/bash
# round 5, GPU run C: 8-wave deep cross-attention (<= 256 pairs), two-per-CU shape (257..512 pairs), group policy sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
timeout 900 python tools/gpu_latency_probe.py ";xattn_deep8_max_pairs=0;xattn_pair_wg_max_pairs=0" "tiny.en:1,tiny.en:8,tiny.en:24,base:8,base:16,small:8,large-v2:1,large-v2:5,large-v2:8,large-v2:12,large-v2:13,large-v3:15,large-v2:16,large-v2:24" > $O/lat.txt 2>&1; cat $O/lat.txt
timeout 300 python tools/gpu_decode_probe.py > $O/probe.txt 2>&1; grep "attn" $O/probe.txt
timeout 900 python tools/gpu_group_policy_probe.py large-v2 "9,12,15,16,20,24,32,48" "8,12,16,24,128" > $O/policy_v2.txt 2>&1; cat $O/policy_v2.txt
timeout 900 python tools/gpu_group_policy_probe.py large-v3 "15" "5,8,16" > $O/policy_v3.txt 2>&1; cat $O/policy_v3.txt
timeout 900 python tools/gpu_group_policy_probe.py base "16,32,48,64" "8,16,32,128" > $O/policy_base.txt 2>&1; cat $O/policy_base.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5c/bench.json").read().strip().splitlines()[-1])
oc = d.get("other_configs") or {}
print("value %.1f batch8 %.1f decode frac %.3f enc frac %.3f roof %.3f checks %s" % (d["value"], d["value_batch8"], d["stage_roofline"]["decode"]["frac"], d["stage_roofline"]["encoder_xkv"]["frac"], d["roofline"]["frac"], d["token_checks"]))
print(json.dumps(oc, indent=1)[:6000])
print(json.dumps(d["cpu_baseline"], indent=1)[:1500])
PY
