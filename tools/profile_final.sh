#!/bin/bash
# usage (on the GPU box): bash tools/profile_final.sh TAG
# Kernel-trace statistics of bench.py, HBM traffic of the triangle launch (FETCH_SIZE / WRITE_SIZE, one pass each),
# SQ counters of the triangle launch, of the dense causal kernel and of the decode kernel (16-bit and fp8 cache):
# every counter set in a run of its own.
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py > $out/bench_line.json 2> $out/bench.err || exit 1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats.csv
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $GRAFT_REPO_ROOT/tools/run_once.py triangle 32768 5 64 1 > $out/fetch.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $GRAFT_REPO_ROOT/tools/run_once.py triangle 32768 5 64 1 > $out/write.log 2>&1 || exit 3
python3 $GRAFT_REPO_ROOT/tools/traffic.py $out/fetch $out/write 32768 $out/traffic.json
cd $GRAFT_REPO_ROOT
SQ_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ_B="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES"
{
  echo "== triangle 32768 (attn_kernel) =="
  bash tools/pmc.sh prof_${tag}/sq_tri_a "$SQ_A" triangle 32768 5 64 1 || exit 4
  bash tools/pmc.sh prof_${tag}/sq_tri_b "$SQ_B" triangle 32768 5 64 1 || exit 4
  bash tools/pmc.sh prof_${tag}/tcc_tri "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" triangle 32768 5 64 1 || exit 4
  echo "== causal 32768 (dense_causal_kernel) =="
  bash tools/pmc.sh prof_${tag}/sq_causal_a "$SQ_A" causal 32768 2 64 1 || exit 5
  bash tools/pmc.sh prof_${tag}/sq_causal_b "$SQ_B" causal 32768 2 64 1 || exit 5
  echo "== decode 64 x 8192 (decode_kernel), 16-bit cache =="
  bash tools/pmc.sh prof_${tag}/sq_dec_a "$SQ_A" decode 8192 5 64 || exit 6
  bash tools/pmc.sh prof_${tag}/sq_dec_b "$SQ_B" decode 8192 5 64 || exit 6
  bash tools/pmc.sh prof_${tag}/fetch_dec "FETCH_SIZE" decode 8192 5 64 || exit 6
  echo "== decode 64 x 8192 (decode_kernel), fp8 cache =="
  bash tools/pmc.sh prof_${tag}/sq_dec8_a "$SQ_A" decode8 8192 5 64 || exit 7
  bash tools/pmc.sh prof_${tag}/fetch_dec8 "FETCH_SIZE" decode8 8192 5 64 || exit 7
} > $out/pmc_summary.txt 2>&1
head -12 $out/kernel_stats.csv | cut -c1-160
cat $out/traffic.json
tail -c 1500 $out/bench_line.json
