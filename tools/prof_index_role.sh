#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace --stats) of the step with the index built inside the attention launch and
# with the build as a launch of its own, at 32k and 8k tokens
out=$GRAFT_REPO_ROOT/gpurun_out/prof_index_role
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for n in 32768 8192; do
  for v in inlaunch separate; do
    flag=""; [ $v = inlaunch ] && flag="--index-in-launch"
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_${n}_$v -- python3 $GRAFT_REPO_ROOT/bench.py --no-baselines --seq-len $n $flag > $out/bench_${n}_$v.json 2> $out/err_${n}_$v.txt || exit 1
    f=$(find $out/t_${n}_$v -name "*kernel_stats.csv" | head -1)
    echo "== n=$n $v: $(python3 -c "import json;print(json.load(open('$out/bench_${n}_$v.json'))['ms_per_step'])") ms per step"
    python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if any(k in r['Name'] for k in ('attn_kernel', 'index_scan', 'combine')):
        print('   %-100s calls %5s  avg %9.0f ns  min %s' % (r['Name'][:100], r['Calls'], float(r['AverageNs']), r['MinNs']))
"
    rm -rf $out/t_${n}_$v
  done
done
