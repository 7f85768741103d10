#!/usr/bin/env python3
"""Runs a few attention launches of one configuration (for rocprofv3 --pmc / --kernel-trace).
Usage: python tools/run_once.py MODE N [ITERS] [CHUNK] [PAGED]   (MODE: triangle|causal|stream|window;
PAGED=1: paged K/V cache with 16-token pages and a shuffled block table, as bench.py uses)
       python tools/run_once.py decode|decode8 KEYS [ITERS] [BATCH]   (one-row requests over KEYS cached tokens each,
16-bit or fp8 paged cache: the decode kernel alone)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_trianglemix_amd import ops
from vllm_trianglemix_amd._lib import TMX_ATTN_SKIP_COMBINE
mode = sys.argv[1]; n = int(sys.argv[2]); iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 32
paged = int(sys.argv[5]) if len(sys.argv) > 5 else 0
H, Hkv, D = 32, 8, 128
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
if mode in ("decode", "decode8"):
    from vllm_trianglemix_amd._lib import TMX_ATTN_DECODE_ONLY, TMX_ATTN_SHORT_QUERIES
    B, PAGE = (int(sys.argv[4]) if len(sys.argv) > 4 else 64), 16
    nblk = B * (n // PAGE)
    kc = torch.randn((nblk, PAGE, Hkv, D), generator=g).to(dev).bfloat16()
    vc = torch.randn((nblk, PAGE, Hkv, D), generator=g).to(dev).bfloat16()
    bt = torch.randperm(nblk, generator=g).to(torch.int32).view(B, -1).to(dev)
    q = torch.randn((B, H, D), generator=g).to(dev).bfloat16()
    out = torch.empty_like(q)
    if mode == "decode8":
        k8, v8 = torch.zeros(kc.shape, dtype=torch.uint8, device=dev), torch.zeros(kc.shape, dtype=torch.uint8, device=dev)
        ops.reshape_and_cache_fp8(kc.view(-1, Hkv, D), vc.view(-1, Hkv, D), k8, v8, torch.arange(nblk * PAGE, device=dev), 1.0, 1.0)
        kc, vc = k8, v8
    cu_h, sl_h = np.arange(B + 1, dtype=np.int32), np.full((B,), n, np.int32)
    cu, sl = torch.from_numpy(cu_h).to(dev), torch.from_numpy(sl_h).to(dev)
    bm = ops.attention_block_m(H, Hkv)
    c = ops.suggest_chunk_tiles(cu_h, sl_h, 4, 32, 64, bm, 32, ops.MODES["triangle"], ops.units_per_item(H, Hkv))
    idx = ops.build_block_index(cu, sl, cu_h, sl_h, 4, 32, 64, bm, 32, ops.MODES["triangle"], chunk_tiles=c, block_table=bt,
                                page_size=PAGE)
    work = ops.alloc_workspace(idx, H, D, dev)
    for _ in range(iters):
        ops.attention(q, kc, vc, out, idx, cu, sl, D ** -0.5, block_table=bt, workspace=work,
                      flags=TMX_ATTN_SHORT_QUERIES | TMX_ATTN_DECODE_ONLY)
    torch.cuda.synchronize()
    print("done", mode, B, n, idx.sizes.num_items)
    sys.exit(0)
q = torch.randn((n, H, D), generator=g).to(dev).bfloat16()
k = torch.randn((n, Hkv, D), generator=g).to(dev).bfloat16()
v = torch.randn((n, Hkv, D), generator=g).to(dev).bfloat16()
out = torch.empty_like(q)
cu_h, sl_h = np.asarray([0, n], np.int32), np.asarray([n], np.int32)
cu, sl = torch.from_numpy(cu_h).to(dev), torch.from_numpy(sl_h).to(dev)
pat = {"triangle": (4, 32, 64, "triangle"), "stream": (4, 32, 0, "triangle"), "window": (-1, 32, 0, "triangle"),
       "diag": (-1, 0, 0, "triangle"), "diagstrip": (-1, 0, 64, "triangle"),
       "causal": (0, 0, 0, "causal")}[mode]
bt = None
if paged:
    PAGE = 16
    nblk = (n + PAGE - 1) // PAGE
    perm = torch.randperm(nblk, generator=g).to(torch.int32)
    kc = torch.zeros((nblk, PAGE, Hkv, D), dtype=torch.bfloat16, device=dev)
    vc = torch.zeros_like(kc)
    pos = torch.arange(n)
    ops.reshape_and_cache(k, v, kc, vc, (perm.long()[pos // PAGE] * PAGE + pos % PAGE).to(dev))
    bt = perm.view(1, -1).to(dev)
idx = ops.build_block_index(cu, sl, cu_h, sl_h, pat[0], pat[1], pat[2], ops.attention_block_m(H, Hkv), 32,
                            ops.MODES[pat[3]], chunk_tiles=chunk, block_table=bt, page_size=16 if paged else 0)
work = ops.alloc_workspace(idx, H, D, dev)
extra_flags = int(os.environ.get("TMX_RUN_FLAGS", "0"))  # e.g. 256 = TMX_ATTN_NO_STRIP, 8 = TMX_ATTN_NO_BAND
for _ in range(iters):
    if paged:
        ops.attention(q, kc, vc, out, idx, cu, sl, D ** -0.5, block_table=bt, workspace=work,
                      flags=TMX_ATTN_SKIP_COMBINE | extra_flags)
    else:
        ops.attention(q, k, v, out, idx, cu, sl, D ** -0.5, cu_seqlens_k=cu, workspace=work,
                      flags=TMX_ATTN_SKIP_COMBINE | extra_flags)
torch.cuda.synchronize()
print("done", mode, n, idx.sizes.num_items)
