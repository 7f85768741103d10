#!/bin/bash
# usage (GPU box): bash tools/length_sweep.sh [lengths ...]   -- bench.py step and kernel times per prompt length
for n in ${@:-4096 8192 16384 32768 65536 131072}; do
  timeout -k 10 200 python bench.py --no-baselines --seq-len $n > /tmp/length_sweep.json 2>/dev/null || exit 1
  python - $n <<'P'
import json, sys
d = json.loads(open("/tmp/length_sweep.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("seq_len %6s: step %7.1f us  %6.1f M tokens/s  attn %7.1f us  frac %.3f  index %5.1f us  merge %5.1f us (the last two with event overhead)"
      % (sys.argv[1], d["ms_per_step"] * 1e3, d["value"] / 1e6, r["kernel_ms"] * 1e3, r["frac"], r["index_build_ms"] * 1e3, r["combine_ms"] * 1e3), flush=True)
P
done
