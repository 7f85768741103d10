#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/run_once.py into
profiles-style JSON: HBM-side bytes per attn_kernel launch.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies a 128-byte request of a wide
streaming read as 64 bytes, so it is doubled (16 bytes per lane reads, LDS-DMA included);
WRITE_SIZE is exact for 16-byte-per-lane stores.
Usage: python tools/traffic.py FETCH_DIR WRITE_DIR SEQ_LEN OUT.json"""
import csv, glob, json, os, sys

def mean_counter(d, name):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if "attn_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(vals) / len(vals), len(vals)

fetch_kib, nf = mean_counter(sys.argv[1], "FETCH_SIZE")
write_kib, nw = mean_counter(sys.argv[2], "WRITE_SIZE")
out = {"kernel": "attn_kernel", "seq_len": int(sys.argv[3]), "launches_averaged": [nf, nw],
       "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
       "read_bytes_per_launch": fetch_kib * 1024 * 2, "write_bytes_per_launch": write_kib * 1024,
       "hbm_bytes_per_launch": fetch_kib * 1024 * 2 + write_kib * 1024,
       "correction": "FETCH_SIZE x 2 (gfx950 tallies 128-B streaming read requests at 64 B); WRITE_SIZE as is"}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
