#!/bin/bash
# round 6, GPU call 7: where the small flow's 2.2 ms go; stall counters of the f32 front end; the base test again
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
for m in small tiny.en; do timeout 300 python tools/gpu_small_flow_probe.py $m; done > $O/r06_small_flow_parts.txt 2>&1; cat $O/r06_small_flow_parts.txt
CMD="python tools/gpu_frontend_probe.py 56 80"
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
 "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"
 "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
)
DBS=""; i=0
for P in "${PASSES[@]}"; do
  i=$((i+1)); rm -rf /tmp/prof_fe$i
  timeout 600 rocprofv3 --kernel-trace --pmc $P -d /tmp/prof_fe$i -- $CMD > $O/r06_fe_stall_pass$i.log 2>&1
  DB=$(find /tmp/prof_fe$i -name "*.db" | head -1); [ -n "$DB" ] && DBS="$DBS $DB"
done
python tools/rocprof_pmc_table.py --filter logmel_stage1 $DBS > $O/r06_pmc_frontend_stalls.txt; cat $O/r06_pmc_frontend_stalls.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "base_geometry or tiny_en" 2>&1 | tail -15
cp $O/parity_margins_tests.txt $O/r06_parity_margins_base_tiny.txt 2>/dev/null
