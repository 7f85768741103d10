#!/bin/bash
# usage (GPU box): bash tools/dense_counters.sh -- matrix-pipe and vector-unit busy counters of the dense causal kernel at 32k tokens,
# register form of the tile fetch against the staged form (one counter pass each; gpurun_out/dense_counters/)
out=$GRAFT_REPO_ROOT/gpurun_out/dense_counters
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for form in 0 1; do
  export TMX_DENSE_STAGE=$form
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $out/form$form -- python3 $GRAFT_REPO_ROOT/tools/dense_bench.py 32768 > $out/form$form.log 2>&1 || exit 1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "dense_counters")
for form in (0, 1):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, f"form{form}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "dense_causal_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("register form" if form == 0 else "staged form", {k: sum(v) / len(v) for k, v in sorted(acc.items())}, "launches", {k: len(v) for k, v in acc.items()})
PY
