#!/usr/bin/env python3
"""Mixed-step probe: one whole prompt of P tokens plus B decode requests with N cached tokens each, Qwen3-8B heads,
paged K/V; attention + combine time against the piece length, and the library's advice.
Usage: python tools/mixed_probe.py [P] [B] [N] [chunk,chunk,...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_trianglemix_amd import ops

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
chunks = [int(x) for x in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["16", "32", "64", "128", "256"])]
H, Hkv, D, page = 32, 8, 128, 16
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
q_lens = [P] + [1] * B
seq_lens = [P] + [N] * B
nblk = sum((s + page - 1) // page for s in seq_lens)
kc = torch.randn((nblk, page, Hkv, D), generator=g).to(dev).bfloat16()
vc = torch.randn((nblk, page, Hkv, D), generator=g).to(dev).bfloat16()
maxb = max((s + page - 1) // page for s in seq_lens)
perm = torch.randperm(nblk, generator=g).to(torch.int32)
bt = torch.zeros((len(seq_lens), maxb), dtype=torch.int32)
o = 0
for b, s in enumerate(seq_lens):
    nb = (s + page - 1) // page
    bt[b, :nb] = perm[o:o + nb]
    o += nb
bt = bt.to(dev)
T = sum(q_lens)
q = torch.randn((T, H, D), generator=g).to(dev).bfloat16()
out = torch.empty_like(q)
cu_h = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
sl_h = np.asarray(seq_lens, np.int32)
cu, sl = torch.from_numpy(cu_h).to(dev), torch.from_numpy(sl_h).to(dev)
bm = ops.attention_block_m(H, Hkv)
advice = ops.suggest_chunk_tiles(cu_h, sl_h, 4, 32, 64, bm, 32, ops.MODES["triangle"], ops.units_per_item(H, Hkv),
                                 tri_min_len=2048)
for chunk in chunks + [advice]:
    idx = ops.build_block_index(cu, sl, cu_h, sl_h, 4, 32, 64, bm, 32, ops.MODES["triangle"], chunk_tiles=chunk,
                                tri_min_len=2048, block_table=bt, page_size=page)
    ws = ops.alloc_workspace(idx, H, D, dev)
    flags = int(os.environ.get("FLAGS", "0"))   # e.g. 32 = TMX_ATTN_NO_DECODE: decode rows on the work list
    run = lambda: ops.attention(q, kc, vc, out, idx, cu, sl, D ** -0.5, block_table=bt, workspace=ws, flags=flags)
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        for _ in range(10):
            run()
        torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in evs:
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    t = np.median([a.elapsed_time(b) for a, b in evs])
    print(f"prompt {P} + {B} x decode({N}) chunk={chunk:4d}{' (advice)' if chunk == advice else ''}: items={idx.sizes.num_items:6d} "
          f"attention+combine {t * 1e3:8.1f} us", flush=True)
