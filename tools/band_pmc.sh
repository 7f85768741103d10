#!/bin/bash
# Tuning build: SQ / TCC counters of the attention launch with one role switched off (TMX_EXP) -- separate passes.
export TMX_EXP=${1:-1} TMX_BAND_RL=${2:-16}
tag=exp${TMX_EXP}_rl${TMX_BAND_RL}
bash tools/pmc.sh pmc_${tag}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" triangle 32768 10 64 1 || exit 1
bash tools/pmc.sh pmc_${tag}_b "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" triangle 32768 10 64 1 || exit 1
bash tools/pmc.sh pmc_${tag}_c "FETCH_SIZE TCC_HIT_sum" triangle 32768 10 64 1 || exit 1
bash tools/pmc.sh pmc_${tag}_d "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" triangle 32768 10 64 1 || exit 1
