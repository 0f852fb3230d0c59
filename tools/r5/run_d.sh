#!/bin/bash
# round 5, GPU run D: row-split GEMV epilogues + staggered weights; canaries; group policy at 64..128; profiles; sanitizers
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
V=tools/build/variants
timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
timeout 900 python tools/gpu_latency_probe.py ";gemv_stagger=0" > $O/lat.txt 2>&1; cat $O/lat.txt
WM_LIB_PATH=$PWD/$V/base.so WM_DBG_LIB_PATH=$PWD/$V/base_dbg.so timeout 600 python tools/gpu_latency_probe.py "" "tiny.en:1,large-v2:8,large-v3:15,large-v2:24" > $O/lat_base.txt 2>&1; cat $O/lat_base.txt
timeout 300 python tools/gpu_decode_probe.py > $O/probe.txt 2>&1; grep -v "launch floor" $O/probe.txt
timeout 900 python tools/gpu_group_policy_probe.py large-v2 "64,96,128" "22,32,43,48,64,128" > $O/policy_v2_big.txt 2>&1; cat $O/policy_v2_big.txt
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_new$i.json 2> $O/bench_new$i.err; echo "bench new rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --tuning gemv_stagger=0 > $O/bench_nostag$i.json 2> $O/bench_nostag$i.err; echo "bench nostagger rc=$?"
done
python - <<'PY'
import json
for v in ("new1", "nostag1", "new2", "nostag2"):
    try:
        d = json.loads(open("gpurun_out/r5d/bench_%s.json" % v).read().strip().splitlines()[-1])
        print(v, "value %.1f batch8 %.1f decode frac %.3f enc frac %.3f roof %.3f checks %s" % (d["value"], d["value_batch8"], d["stage_roofline"]["decode"]["frac"], d["stage_roofline"]["encoder_xkv"]["frac"], d["roofline"]["frac"], all(d["token_checks"].values())))
        print("    alone", {k: round(x["avg_us"], 1) for k, x in d["kernel_families"].items() if k.startswith("dec_")})
    except Exception as e:
        print(v, "failed", e)
PY
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; tail -20 $O/profile_round.log
# the FETCH_SIZE pass again without the staggered weights (A/B of VERDICT r4 next #4)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_ns; WM_NO_GRAPH=1 timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_ns -- python bench.py --steps 7 --warmup 1 --inflight 1 --fuse 7 --no-cpu-baseline --no-single-batch --no-early-stop --no-other-configs --new-tokens 6 --tuning gemv_stagger=0 > $O/fetch_ns.json 2> $O/fetch_ns.err
python tools/rocprof_pmc_summary.py $(find /tmp/prof_ns -name "*.db" | head -1) FETCH_SIZE > gpurun_out/r05_pmc_fetch_size_group56_no_stagger.txt; head -14 gpurun_out/r05_pmc_fetch_size_group56_no_stagger.txt; head -14 gpurun_out/r05_pmc_fetch_size_group56.txt
timeout 1500 bash tools/run_sanitized.sh gpu > $O/sanitized.txt 2>&1; echo "sanitized rc=$?"; tail -8 $O/sanitized.txt
