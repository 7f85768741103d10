#!/bin/bash
# usage (on the GPU box): bash tools/profile_r03.sh TAG
# Kernel-trace statistics of bench.py with the headline step ONLY (--no-baselines: every attn_kernel / combine_kernel /
# index launch in the trace is a 32k-token step), the same at 8192 tokens, the HBM traffic of the 32k triangle launch
# (FETCH_SIZE / WRITE_SIZE, one pass each), and the full bench line outside the profiler.
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace32k -- python3 $GRAFT_REPO_ROOT/bench.py --no-baselines > $out/bench_32k_under_rocprof.json 2> $out/bench32k.err || exit 1
cp $(find $out/trace32k -name "*kernel_stats.csv" | head -1) $out/kernel_stats_32k_only.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace8k -- python3 $GRAFT_REPO_ROOT/bench.py --no-baselines --seq-len 8192 > $out/bench_8k_under_rocprof.json 2> $out/bench8k.err || exit 1
cp $(find $out/trace8k -name "*kernel_stats.csv" | head -1) $out/kernel_stats_8k_only.csv
# the dense causal kernel of the layers outside the triangle set (staged form: dense_stage_kernel + dense_causal_kernel)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tracedense -- python3 $GRAFT_REPO_ROOT/tools/dense_bench.py 32768 > $out/dense_32k_under_rocprof.txt 2> $out/dense.err || exit 1
cp $(find $out/tracedense -name "*kernel_stats.csv" | head -1) $out/kernel_stats_dense_32k.csv
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $GRAFT_REPO_ROOT/tools/run_once.py triangle 32768 5 64 1 > $out/fetch.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $GRAFT_REPO_ROOT/tools/run_once.py triangle 32768 5 64 1 > $out/write.log 2>&1 || exit 3
python3 $GRAFT_REPO_ROOT/tools/traffic.py $out/fetch $out/write 32768 $out/traffic.json
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 bench.py --seq-len 8192 --no-baselines > $out/bench_8k.json 2>> $out/bench8k.err || exit 4
timeout -k 10 300 python3 bench.py > $out/bench.json 2> $out/bench.err || exit 5
rm -rf $out/trace32k $out/trace8k $out/tracedense $out/fetch $out/write
head -8 $out/kernel_stats_32k_only.csv | cut -c1-170
head -8 $out/kernel_stats_8k_only.csv | cut -c1-170
head -5 $out/kernel_stats_dense_32k.csv | cut -c1-170
cat $out/traffic.json
