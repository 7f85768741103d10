#!/bin/bash
# usage (GPU box): VARIANTS="--no-band|-" [ARGS="--seq-len 8192"] bash tools/ab_flags.sh
# step and kernel time of bench.py for each set of extra flags ('|'-separated, '-' = none), alternating, three rounds
IFS='|' read -ra VS <<< "$VARIANTS"
for r in 1 2 3; do
  for v in "${VS[@]}"; do
    [ "$v" = "-" ] && v=""
    timeout -k 10 150 python bench.py --no-baselines $ARGS $v > /tmp/ab_flags_out.json || exit 1
    python - "[$v] $ARGS" <<'P'
import json, sys
d = json.loads(open('/tmp/ab_flags_out.json').read().strip().splitlines()[-1])
print(f"{sys.argv[1]:40s} step {d['ms_per_step']*1e3:7.1f} us  kernel {d['roofline']['kernel_ms']*1e3:7.1f} us  combine {d['roofline']['combine_ms']*1e3:5.1f} us  index {d['roofline']['index_build_ms']*1e3:5.1f} us  {d['value']/1e6:6.1f} M tokens/s", flush=True)
P
  done
done
