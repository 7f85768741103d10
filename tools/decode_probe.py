#!/usr/bin/env python3
"""Decode steps, timed: the decode kernel (default), the same step kept on the work list (TMX_ATTN_NO_DECODE)
and the decode kernel alone (TMX_ATTN_DECODE_ONLY), 16-bit and fp8 (e4m3) caches.
Usage: python tools/decode_probe.py [BATCHxKEYS ...]   (default 64x8192 8x32768 256x2048 1x131072)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_trianglemix_amd import ops
from vllm_trianglemix_amd._lib import TMX_ATTN_DECODE_ONLY, TMX_ATTN_NO_DECODE, TMX_ATTN_SHORT_QUERIES

H, Hkv = (int(x) for x in os.environ.get("HEADS", "32,8").split(","))   # query heads, K/V heads
D, PAGE = 128, 16
dev = torch.device("cuda:0")
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(64, 8192), (8, 32768), (256, 2048), (1, 131072)]
chunks = [int(x) for x in os.environ.get("CHUNKS", "0").split(",")]   # 0: the library's advice


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B, N in shapes:
    g = torch.Generator().manual_seed(1)
    nblk = B * (N // PAGE)
    kc = torch.randn((nblk, PAGE, Hkv, D), generator=g).to(dev).bfloat16()
    vc = torch.randn((nblk, PAGE, Hkv, D), generator=g).to(dev).bfloat16()
    bt = torch.randperm(nblk, generator=g).to(torch.int32).view(B, -1).to(dev)
    q = torch.randn((B, H, D), generator=g).to(dev).bfloat16()
    out = torch.empty_like(q)
    cu_h, sl_h = np.arange(B + 1, dtype=np.int32), np.full((B,), N, np.int32)
    cu, sl = torch.from_numpy(cu_h).to(dev), torch.from_numpy(sl_h).to(dev)
    k8, v8 = torch.zeros(kc.shape, dtype=torch.uint8, device=dev), torch.zeros(kc.shape, dtype=torch.uint8, device=dev)
    ops.reshape_and_cache_fp8(kc.view(-1, Hkv, D), vc.view(-1, Hkv, D), k8, v8, torch.arange(nblk * PAGE, device=dev), 1.0, 1.0)
    bm, bn = ops.attention_block_m(H, Hkv), ops.attention_block_n()
    kv_bytes = 2 * B * N * Hkv * D * 2
    for chunk in chunks:
        c = chunk or ops.suggest_chunk_tiles(cu_h, sl_h, 4, 32, 64, bm, bn, ops.MODES["triangle"], ops.units_per_item(H, Hkv))
        idx = ops.build_block_index(cu, sl, cu_h, sl_h, 4, 32, 64, bm, bn, ops.MODES["triangle"], chunk_tiles=c,
                                    block_table=bt, page_size=PAGE)
        ws = ops.alloc_workspace(idx, H, D, dev)
        res = {}
        for name, fl in (("decode", TMX_ATTN_SHORT_QUERIES), ("alone", TMX_ATTN_SHORT_QUERIES | TMX_ATTN_DECODE_ONLY),
                         ("worklist", TMX_ATTN_SHORT_QUERIES | TMX_ATTN_NO_DECODE)):
            for tag, kk, vv in (("bf16", kc, vc), ("fp8", k8, v8)):
                fn = lambda: ops.attention(q, kk, vv, out, idx, cu, sl, D ** -0.5, block_table=bt, workspace=ws, flags=fl)
                res[(name, tag)] = timed(fn, 20)
        line = f"{B}x{N} chunk {c:4d} items {int(idx.sizes.num_items):5d}:"
        for (name, tag), t in res.items():
            byts = kv_bytes * (0.5 if tag == "fp8" else 1.0)
            line += f"  {name}/{tag} {t:7.1f} us ({byts / t / 1e6:5.2f} TB/s)"
        print(line, flush=True)
    del kc, vc, k8, v8
