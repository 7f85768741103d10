"""Reads the cycle stamps of a dense kernel built with -DTMX_DENSE_STAMPS after one 32k launch (the stamped kernel is an
experimental slotted variant of the dense kernel; the shipped kernel carries no stamps and does not export the reader)."""
import ctypes
import os
import sys

import numpy as np

sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dense_bench  # noqa: E402,F401  (runs the launches)

lib = ctypes.CDLL(os.environ["TMX_LIB_PATH"])
n = 2048 * 8
buf = (ctypes.c_longlong * n)()
rc = lib.tmx_dense_read_stamps(buf, n)
a = np.frombuffer(buf, dtype=np.int64).reshape(2048, 8).astype(np.float64)
tiles = a[:, 7]
names = ["tile total", "mask + max (no matrix instr.)", "slots 0-7 (QK next + exponentials)", "slots 8-15 (PV)", "fetch issue",
         "stage (waits for the fetch, LDS writes)", "barrier wait"]
print(f"rc {rc}; waves {len(a)}, tiles per wave {tiles.mean():.0f}")
for g, sel in (("waves 0-3 of a workgroup", np.arange(2048) % 8 < 4), ("waves 4-7", np.arange(2048) % 8 >= 4)):
    print(g)
    for i, nm in enumerate(names):
        per = a[sel, i].sum() / tiles[sel].sum()
        print(f"  {nm:42s} {per:8.0f} cycles per tile" + ("" if i not in (1, 2, 3) else f"  ({per / 2:.0f} per 32-key block)"))
