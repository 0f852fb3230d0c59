#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5u; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt | head -2
bash tools/r5/run_n.sh
