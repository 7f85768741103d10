#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh TAG
# kernel-trace statistics of bench.py, then two counter passes (FETCH_SIZE, WRITE_SIZE) over run_once.py
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-baselines > $out/bench_line.json 2> $out/bench.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $GRAFT_REPO_ROOT/tools/run_once.py triangle 32768 5 64 1 > $out/fetch.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $GRAFT_REPO_ROOT/tools/run_once.py triangle 32768 5 64 1 > $out/write.log 2>&1 || exit 3
python3 $GRAFT_REPO_ROOT/tools/traffic.py $out/fetch $out/write 32768 $out/traffic.json
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats.csv
cat $out/kernel_stats.csv | head -12
cat $out/bench_line.json
