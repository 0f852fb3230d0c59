#!/bin/bash
# round 6, GPU call 6: front end A/B, the whole GPU suite, the driver's bench command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
for b in 56 8 1; do timeout 300 python tools/gpu_frontend_probe.py $b 80; done > $O/r06_frontend_lds_twiddles.txt 2>&1
timeout 300 python tools/gpu_frontend_probe.py 15 128 >> $O/r06_frontend_lds_twiddles.txt 2>&1; cat $O/r06_frontend_lds_twiddles.txt
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -40
cp $O/parity_margins_tests.txt $O/r06_parity_margins_tests.txt 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06_bench_n1_driver_cmd.json 2> $O/r06_bench_n1_driver_cmd.err; tail -c 2100 $O/r06_bench_n1_driver_cmd.json
