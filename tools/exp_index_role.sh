cd $GRAFT_REPO_ROOT
export TMX_LIB_PATH=$GRAFT_REPO_ROOT/tools/libtrianglemix_hip_tune.so TRIANGLEMIX_BINDING=ctypes
for n in 32768 8192; do
for e in 3 2 0; do
  for v in "--index-in-launch" ""; do
    TMX_EXP=$e timeout -k 10 120 python3 bench.py --no-baselines --seq-len $n $v > /tmp/ab.json 2>/tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
    python3 -c "
import json; b=json.load(open('/tmp/ab.json')); r=b['roofline']
print('n=%6d TMX_EXP=$e %-24s step %.4f ms  kernel %.4f' % ($n, '$v' or 'build launch', b['ms_per_step'], r['kernel_ms']))"
  done
done
done
