#!/bin/bash
# usage (GPU box): LIBS="a.so b.so" bash tools/ab_dense.sh [n] -- the dense causal kernel alone for each build of the library, alternating
export TRIANGLEMIX_BINDING=ctypes
for r in 1 2 3; do
  for lib in $LIBS; do
    echo -n "$lib  "
    TMX_LIB_PATH=$PWD/$lib timeout -k 10 150 python tools/dense_bench.py ${1:-32768} 2>/dev/null | tail -1 || exit 1
  done
done
