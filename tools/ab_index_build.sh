#!/bin/bash
# usage (on the GPU box): bash tools/ab_index_build.sh   -- the step with the index built inside the attention launch
# against the step with the build as a launch of its own, alternating, at 32k and 8k tokens
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for n in 32768 8192; do
    for v in "--index-in-launch" ""; do
      timeout -k 10 120 python3 bench.py --no-baselines --seq-len $n $v > /tmp/ab.json 2>/tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
      python3 -c "
import json; b=json.load(open('/tmp/ab.json')); r=b['roofline']
print('n=%6d %-24s step %.4f ms  kernel %.4f  combine %.4f  index %.4f' % ($n, '$v' or 'build launch', b['ms_per_step'], r['kernel_ms'], r['combine_ms'], r['index_build_ms']))"
    done
  done
done
