#!/bin/bash
# usage (GPU box): LIBS="a.so b.so" [ARGS="--chunk-tiles 48"] bash tools/ab_bench.sh
# step and kernel time of bench.py for each build of the library, alternating, three rounds
for r in 1 2 3; do
  for lib in $LIBS; do
    TMX_LIB_PATH=$PWD/$lib timeout -k 10 150 python bench.py --no-baselines $ARGS > /tmp/ab_bench_out.json || exit 1
    python - "$lib $ARGS" <<'P'
import json, sys
d = json.loads(open('/tmp/ab_bench_out.json').read().strip().splitlines()[-1])
print(f"{sys.argv[1]:40s} step {d['ms_per_step']*1e3:7.1f} us  kernel {d['roofline']['kernel_ms']*1e3:7.1f} us  combine {d['roofline']['combine_ms']*1e3:5.1f} us  index {d['roofline']['index_build_ms']*1e3:5.1f} us  {d['value']/1e6:6.1f} M tokens/s", flush=True)
P
  done
done
