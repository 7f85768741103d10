#!/bin/bash
# Tuning build only (make -C vllm_trianglemix_amd/csrc tune): kernel time of bench.py's launch with one role
# switched off (TMX_EXP=1: band workgroups only, 2: work-list workgroups only, 0: both) per band run length.
# usage (GPU box): CFGS="1:32 1:16 0:32" bash tools/band_exp2.sh
[ -n "$RLS" ] && export TMX_BAND_RLS=$RLS
[ -n "$TAU" ] && export TMX_BAND_TAU=$TAU
[ -n "$WALK" ] && export TMX_WALK_SHIFT=$WALK
export TRIANGLEMIX_BINDING=ctypes TMX_LIB_PATH=$PWD/tools/libtrianglemix_hip_tune.so
for cfg in ${CFGS:-"0:32 1:32 1:16 1:8 2:32"}; do
  e=${cfg%%:*}; rl=${cfg##*:}
  TMX_EXP=$e TMX_BAND_RL=$rl TMX_BAND_MAP=${MAP:-0} timeout -k 10 120 python bench.py --no-baselines --steps 100 --warmup 20 ${BENCH_ARGS} > /tmp/exp.json 2>/tmp/exp.err || { echo "rc=$?"; grep -v amdgpu.ids /tmp/exp.err | tail -5; tail -c 300 /tmp/exp.json; exit 1; }
  python - "$e" "$rl" <<'PY'
import json, sys
j = json.loads(open("/tmp/exp.json").read().strip().splitlines()[-1]); r = j["roofline"]
print("exp %s rl %s: kernel %.1f us" % (sys.argv[1], sys.argv[2], r["kernel_ms"] * 1e3), flush=True)
PY
done
