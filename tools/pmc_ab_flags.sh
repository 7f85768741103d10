#!/bin/bash
# usage (GPU box): bash tools/pmc_ab_flags.sh "0 256"  -- SQ counters of the 32k triangle launch for each value of
# TMX_RUN_FLAGS (extra flags of the attention call), two passes each
for f in $1; do
  export TMX_RUN_FLAGS=$f
  echo "== TMX_RUN_FLAGS=$f =="
  bash tools/pmc.sh pmcf_${f}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" triangle 32768 10 64 1 | grep -A9 "attn_kernel" || exit 1
  bash tools/pmc.sh pmcf_${f}_b "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" triangle 32768 10 64 1 | grep -A9 "attn_kernel" || exit 1
done
