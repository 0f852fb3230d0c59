#!/bin/bash
# Round profile set (run on the GPU box through gpurun): rocprofv3 kernel trace of ONE decode group of 56 chunks on one lane
# (what bench.py's `roofline` measures), a FETCH_SIZE PMC pass and an MFMA-busy PMC pass of the same command (separate runs,
# --kernel-trace only), and the literal batch-of-8 trace.  Eager launches (WM_NO_GRAPH=1: rocprofv3 cannot trace the replays).
# usage: bash tools/profile_round.sh r04
R=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
CMD56="python bench.py --steps 7 --warmup 1 --inflight 1 --fuse 7 --no-cpu-baseline --no-single-batch --no-early-stop --no-other-configs"
run() { # name, extra rocprof args, bench extra args
  rm -rf /tmp/prof_$1
  WM_NO_GRAPH=1 timeout 900 rocprofv3 --kernel-trace $2 -d /tmp/prof_$1 -- $3 > $OUT/${R}_$1.json 2> $OUT/${R}_$1.err
  find /tmp/prof_$1 -name "*.db" | head -1
}
DB=$(run trace56 "" "$CMD56"); python tools/rocprof_summary.py $DB 40 > $OUT/${R}_kernel_trace_group56_summary.txt
DB=$(run fetch56 "--pmc FETCH_SIZE" "$CMD56 --new-tokens 6"); python tools/rocprof_pmc_summary.py $DB FETCH_SIZE > $OUT/${R}_pmc_fetch_size_group56.txt
DB=$(run mfma56 "--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "$CMD56 --new-tokens 6")
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do python tools/rocprof_pmc_summary.py $DB $c; done > $OUT/${R}_pmc_mfma_busy.txt
DB=$(run trace8 "" "python bench.py --steps 2 --warmup 1 --inflight 1 --fuse 1 --no-cpu-baseline --no-single-batch --no-early-stop --no-other-configs"); python tools/rocprof_summary.py $DB 30 > $OUT/${R}_kernel_trace_batch8_summary.txt
rm -rf /tmp/prof_*
head -12 $OUT/${R}_kernel_trace_group56_summary.txt; head -6 $OUT/${R}_pmc_fetch_size_group56.txt
