#!/bin/bash
# usage (GPU box): bash tools/warm_probe.sh -> bench step time against the number of warm-up / timed steps
for r in 1 2; do
  for cfg in "50 10" "50 200" "400 100" "20 5"; do
    set -- $cfg
    timeout -k 10 150 python bench.py --no-baselines --steps $1 --warmup $2 > /tmp/wp.json || exit 1
    python - "$cfg" <<'P'
import json, sys
d = json.loads(open('/tmp/wp.json').read().strip().splitlines()[-1])
print(f"steps/warmup {sys.argv[1]:8s} step {d['ms_per_step']*1e3:7.1f} us  kernel {d['roofline']['kernel_ms']*1e3:7.1f} us", flush=True)
P
  done
done
