#!/bin/bash
# usage (on the GPU box): bash tools/sweep_w.sh "W values" [probe args]
ws=$1; shift
for w in $ws; do
  echo "== W=$w"
  TMX_SCHED_W=$w timeout -k 10 150 python tools/attn_probe.py --chunks 64 --iters 40 "$@" || exit 1
done
