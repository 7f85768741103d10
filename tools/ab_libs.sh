#!/bin/bash
# usage (GPU box): LIBS="a.so b.so" [ARGS=...] bash tools/ab_libs.sh -- bench.py through the ctypes binding for each build of the library
export TRIANGLEMIX_BINDING=ctypes
for r in 1 2 3; do
  for lib in $LIBS; do
    TMX_LIB_PATH=$PWD/$lib timeout -k 10 150 python bench.py --no-baselines $ARGS > /tmp/ab_l.json 2>/dev/null || exit 1
    python - "$lib $ARGS" <<'P'
import json, sys
d = json.loads(open('/tmp/ab_l.json').read().strip().splitlines()[-1])
print(f"{sys.argv[1]:50s} step {d['ms_per_step']*1e3:7.1f} us  kernel {d['roofline']['kernel_ms']*1e3:7.1f} us", flush=True)
P
  done
done
