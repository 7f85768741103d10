#!/bin/bash
# usage (on the GPU box): LIBS="a.so b.so" bash tools/ab.sh [probe args]
# A/B of builds of the library on the same box, alternating; default: tools/ab_base.so against the in-tree one.
LIBS=${LIBS:-"tools/ab_base.so vllm_trianglemix_amd/libtrianglemix_hip.so"}
for r in 1 2; do
  for lib in $LIBS; do
    echo "== $lib"
    TMX_LIB_PATH=$PWD/$lib timeout -k 10 150 python tools/attn_probe.py --chunks 64 --iters 40 "$@" || exit 1
  done
done
