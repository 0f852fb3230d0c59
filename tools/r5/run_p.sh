#!/bin/bash
# round 5, GPU run P: the TIMED configuration (3 lanes x 56/56/48, eager, --new-tokens 28) traced per lane
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_lanes
WM_NO_GRAPH=1 timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_lanes -- python bench.py --steps 20 --warmup 5 --new-tokens 28 --no-cpu-baseline --no-early-stop --no-other-configs --no-single-batch > gpurun_out/r05_lanes.json 2> gpurun_out/r05_lanes.err
python tools/rocprof_lanes.py $(find /tmp/prof_lanes -name "*.db" | head -1) 56 1 > gpurun_out/r05_timed_config_lanes.txt 2>&1; cat gpurun_out/r05_timed_config_lanes.txt
