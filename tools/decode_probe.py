#!/usr/bin/env python3
"""Decode-step probe: B requests with one query row each and a paged context of N tokens (the dense last row
of the pattern).  Prints kernel time and K/V bytes per second.  Usage: python tools/decode_probe.py [B] [N] [chunk]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_trianglemix_amd import ops
from vllm_trianglemix_amd._lib import TMX_ATTN_SHORT_QUERIES

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
chunks = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["64"])]
H, Hkv, D, page = 32, 8, 128, 16
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
nblk = B * (N // page)
kc = torch.randn((nblk, page, Hkv, D), generator=g).to(dev).bfloat16()
vc = torch.randn((nblk, page, Hkv, D), generator=g).to(dev).bfloat16()
bt = torch.randperm(nblk, generator=g).to(torch.int32).view(B, -1).to(dev)
q = torch.randn((B, H, D), generator=g).to(dev).bfloat16()
out = torch.empty_like(q)
cu_h = np.arange(B + 1, dtype=np.int32)
sl_h = np.full((B,), N, np.int32)
cu, sl = torch.from_numpy(cu_h).to(dev), torch.from_numpy(sl_h).to(dev)
bm = ops.attention_block_m(H, Hkv)
for chunk in chunks:
    idx = ops.build_block_index(cu, sl, cu_h, sl_h, 4, 32, 64, bm, 32, ops.MODES["triangle"], chunk_tiles=chunk,
                                block_table=bt, page_size=page)
    ws = ops.alloc_workspace(idx, H, D, dev)
    import time
    t_end = time.perf_counter() + 0.15  # an idle GPU needs tens of milliseconds of load before its clocks settle
    while time.perf_counter() < t_end:
        for _ in range(10):
            ops.attention(q, kc, vc, out, idx, cu, sl, D ** -0.5, block_table=bt, workspace=ws, flags=TMX_ATTN_SHORT_QUERIES)
        torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in evs:
        a.record()
        ops.attention(q, kc, vc, out, idx, cu, sl, D ** -0.5, block_table=bt, workspace=ws, flags=TMX_ATTN_SHORT_QUERIES)
        b.record()
    torch.cuda.synchronize()
    t = np.median([a.elapsed_time(b) for a, b in evs])
    kv_bytes = 2 * B * N * Hkv * D * 2
    print(f"decode B={B} N={N} chunk={chunk}: items={idx.sizes.num_items} attention+combine {t * 1e3:8.1f} us  "
          f"K/V {kv_bytes / 1e9:.2f} GB -> {kv_bytes / (t * 1e-3) / 1e9:7.1f} GB/s  ({B / (t * 1e-3):.0f} tokens/s)")
