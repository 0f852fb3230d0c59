#!/bin/bash
# rocprofv3 kernel trace of the literal configs[1]: tiny.en, one chunk, greedy 224 tokens (eager launches)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_tiny
WM_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_tiny -- python bench.py --model tiny.en --batch 1 --steps 3 --warmup 1 --inflight 1 --fuse 1 --no-cpu-baseline --no-single-batch --no-early-stop --no-other-configs > gpurun_out/tiny_trace.json 2> gpurun_out/tiny_trace.err
DB=$(find /tmp/prof_tiny -name "*.db" | head -1)
python tools/rocprof_summary.py $DB 30
