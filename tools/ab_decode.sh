#!/bin/bash
# usage (GPU box): LIBS="a.so b.so" bash tools/ab_decode.sh [BATCHxKEYS ...]
# decode steps (tools/decode_probe.py) for each build of the library, alternating, three rounds
export TRIANGLEMIX_BINDING=ctypes
for r in 1 2 3; do
  for lib in $LIBS; do
    echo "== $lib"
    TMX_LIB_PATH=$PWD/$lib timeout -k 10 200 python -u tools/decode_probe.py "$@" 2>/dev/null | cut -c1-250 || exit 1
  done
done
