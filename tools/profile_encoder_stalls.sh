#!/bin/bash
# Stall attribution of the encoder's MFMA kernels (VERDICT r5 "Missing #5"): rocprofv3 PMC passes (8 SQ counters each, --kernel-trace
# only) over the encoder alone at large-v2 x 56 chunks, merged into one table per kernel.
# usage (GPU box, via gpurun): bash tools/profile_encoder_stalls.sh r06 [model] [B]
R=${1:-rXX}; MODEL=${2:-large-v2}; NB=${3:-56}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
rocprofv3 -L > $OUT/${R}_pmc_counter_list.txt 2>&1
CMD="python tools/gpu_encode_only.py $MODEL $NB 1"
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
 "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"
 "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
 "SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE"
)
DBS=""
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1)); rm -rf /tmp/prof_st$i
  timeout 600 rocprofv3 --kernel-trace --pmc $P -d /tmp/prof_st$i -- $CMD > $OUT/${R}_stall_pass$i.log 2>&1
  DB=$(find /tmp/prof_st$i -name "*.db" | head -1)
  [ -n "$DB" ] && DBS="$DBS $DB" || echo "pass $i produced no db (see $OUT/${R}_stall_pass$i.log)"
done
python tools/rocprof_pmc_table.py --filter gemm256 --filter enc_attn --filter gemm_bf16 --filter layernorm $DBS > $OUT/${R}_pmc_encoder_stalls_${MODEL}_b${NB}.txt
rm -rf /tmp/prof_st*
cat $OUT/${R}_pmc_encoder_stalls_${MODEL}_b${NB}.txt | head -120
