#!/bin/bash
# Sweep of the band workgroups' run length (tuning build: -DTMX_TUNING reads TMX_BAND_RL).
for rl in 2 4 8 16 32; do
  TMX_BAND_RL=$rl timeout -k 10 120 python bench.py --no-baselines --steps 100 --warmup 20 ${BENCH_ARGS} > gpurun_out/sweep_rl$rl.json 2>gpurun_out/sweep_rl$rl.err || exit 1
  python - <<PY
import json
j=json.load(open("gpurun_out/sweep_rl$rl.json")); r=j["roofline"]
print("rl", $rl, "Mtok/s %.1f step %.1f us kernel %.1f us frac %.3f" % (j["value"]/1e6, j["ms_per_step"]*1e3, r["kernel_ms"]*1e3, r["frac"]))
PY
done
