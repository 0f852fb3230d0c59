#!/bin/bash
# Final sanity of a committed tree on the GPU box (was tools/r5/run_o.sh): smoke(), the whole -m gpu suite, one driver-command
# bench line -- what the driver itself runs at round end.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 600 python __graft_entry__.py smoke > $O/final_smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $O/final_smoke.txt
timeout 3000 python -m pytest tests -m gpu -x -q > $O/final_tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/final_tests.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/final_bench_driver.json 2> $O/final_bench_driver.err; echo "bench driver rc=$?"
python -c "import json; d = json.loads(open('gpurun_out/final_bench_driver.json').read().strip().splitlines()[-1]); print(json.dumps(d['summary']))"
