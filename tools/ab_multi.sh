#!/bin/bash
# usage (GPU box): LIBS="a.so b.so" bash tools/ab_multi.sh -> several workloads per build
for r in 1 2; do
  for lib in $LIBS; do
    for args in "--cpu-rows 64" "--seq-len 8192 --no-baselines" "--seq-len 65536 --no-baselines" "--emulate-world 8 --no-baselines"; do
      TMX_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py $args > /tmp/abm.json || exit 1
      python - "$lib $args" <<'P'
import json, sys
d = json.loads(open('/tmp/abm.json').read().strip().splitlines()[-1])
dec = d.get('extra', {}).get('decode_64x8192_ms')
print(f"{sys.argv[1]:58s} step {d['ms_per_step']*1e3:7.1f} us  kernel {d['roofline']['kernel_ms']*1e3:7.1f} us" + (f"  decode {dec*1e3:7.1f} us" if dec else ""), flush=True)
P
    done
  done
done
