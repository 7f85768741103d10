#!/bin/bash
# Tuning build only: time the two roles of the attention launch apart (TMX_EXP=1: band workgroups only,
# 2: work-list workgroups only) for a few run lengths.
for cfg in "0 16" "1 4" "1 16" "1 32" "2 16"; do
  set -- $cfg
  TMX_EXP=$1 TMX_BAND_RL=$2 timeout -k 10 120 python bench.py --no-baselines --steps 100 --warmup 20 ${BENCH_ARGS} > gpurun_out/exp.json 2>gpurun_out/exp.err || exit 1
  python - <<PY
import json
j=json.load(open("gpurun_out/exp.json")); r=j["roofline"]
print("exp $1 rl $2: kernel %.1f us" % (r["kernel_ms"]*1e3))
PY
done
