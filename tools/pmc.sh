#!/bin/bash
# usage: tools/pmc.sh OUTDIR "COUNTERS..." -- args for run_once.py
out=$1; shift; ctrs=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/tools/run_once.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$out/log.txt 2>&1
echo "rc=$?"
f=$(find $GRAFT_REPO_ROOT/gpurun_out/$out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
