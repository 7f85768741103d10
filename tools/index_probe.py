#!/usr/bin/env python3
"""Index-build probe: kernel time of tmx_build_block_index by stream events.  Usage: python tools/index_probe.py [N ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_trianglemix_amd import ops, _lib
import ctypes
dev = torch.device("cuda:0")
for n in [int(x) for x in (sys.argv[1:] or ["8192", "32768", "131072"])]:
    cu_h, sl_h = np.asarray([0, n], np.int32), np.asarray([n], np.int32)
    cu, sl = torch.from_numpy(cu_h).to(dev), torch.from_numpy(sl_h).to(dev)
    page = 16
    bt = torch.randperm(n // page).to(torch.int32).view(1, -1).to(dev)
    lib = _lib.load()
    for pages, host in ((True, False), (False, False), (True, True)):
        idx = ops.build_block_index(cu, sl, cu_h, sl_h, 4, 32, 64, 32, 32, 2, chunk_tiles=64,
                                    block_table=bt if pages else None, page_size=page if pages else 0)
        hc = cu_h.ctypes.data if host else None
        hs = sl_h.ctypes.data if host else None
        def run():
            _lib.check(lib.tmx_build_block_index(ctypes.byref(idx.c), cu.data_ptr(), sl.data_ptr(), hc, hs, None))
        for _ in range(5):
            run()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
        for a, b in evs:
            a.record(); run(); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        # the same launch 20 times between one pair of events: launch gaps included, event overhead amortised
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"index build N={n} pages={pages} host arrays={host}: med {ts[len(ts)//2]*1e3:.1f} us min {ts[0]*1e3:.1f} us, "
              f"back to back {e0.elapsed_time(e1) / 20 * 1e3:.1f} us (q-blocks {idx.num_qblocks}, tiles {idx.nnz_tiles})")
