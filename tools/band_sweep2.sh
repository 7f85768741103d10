#!/bin/bash
# Tuning build: piece length of the dense rows x run length of the band workgroups.
for ct in 32 64 128; do for rl in 8 16 32; do
  TMX_BAND_RL=$rl timeout -k 10 120 python bench.py --no-baselines --steps 100 --warmup 20 --chunk-tiles $ct ${BENCH_ARGS} > gpurun_out/sw.json 2>gpurun_out/sw.err || exit 1
  python - <<PY
import json
j=json.load(open("gpurun_out/sw.json")); r=j["roofline"]
print("chunk $ct rl $rl: step %.1f us kernel %.1f us combine %.1f" % (j["ms_per_step"]*1e3, r["kernel_ms"]*1e3, r["combine_ms"]*1e3))
PY
done; done
