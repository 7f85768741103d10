#!/bin/bash
# usage (GPU box): bash tools/sweep_8k.sh -- band run length x piece length at 8192 tokens (tuning build of the library)
cd $GRAFT_REPO_ROOT
export TMX_LIB_PATH=$GRAFT_REPO_ROOT/tools/libtrianglemix_hip_tune.so TRIANGLEMIX_BINDING=ctypes
for chunk in 0 8 16 32; do
  for rl in 0 4 6 8 12 16; do
    env=""; [ $rl -gt 0 ] && export TMX_BAND_RL=$rl || unset TMX_BAND_RL
    a=""; [ $chunk -gt 0 ] && a="--chunk-tiles $chunk"
    timeout -k 10 100 python3 bench.py --no-baselines --seq-len ${1:-8192} --steps 100 --warmup 20 $a > /tmp/sw.json 2>/tmp/sw.err || { tail -3 /tmp/sw.err; exit 1; }
    python3 -c "
import json; b=json.load(open('/tmp/sw.json')); r=b['roofline']
print('chunk %2s  run %2s   step %.2f us  kernel %.2f us' % ('$chunk' if $chunk else 'lib', '$rl' if $rl else 'lib', b['ms_per_step']*1e3, r['kernel_ms']*1e3))"
  done
done
