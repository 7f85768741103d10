#!/usr/bin/env python3
"""Times the attention kernel alone (stream events) for pattern / size variants.
Usage: python tools/attn_probe.py [--n 32768] [--heads 32] [--kv 8] [--iters 20]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_trianglemix_amd import ops
from vllm_trianglemix_amd._lib import TMX_ATTN_SKIP_COMBINE, TMX_ATTN_ONLY_COMBINE

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32768)
ap.add_argument("--heads", type=int, default=32)
ap.add_argument("--kv", type=int, default=8)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--chunks", type=str, default="64")
ap.add_argument("--dtype", type=str, default="bf16")
ap.add_argument("--only", type=str, default="")
ap.add_argument("--paged", type=int, default=1, help="0 contiguous, 1 paged with resolved pages, 2 paged with in-kernel lookups")
args = ap.parse_args()
dev = torch.device("cuda:0")
n, H, Hkv, D = args.n, args.heads, args.kv, 128
g = torch.Generator().manual_seed(0)
DT = torch.bfloat16 if args.dtype == "bf16" else torch.float16
q = torch.randn((n, H, D), generator=g).to(dev).to(DT)
k = torch.randn((n, Hkv, D), generator=g).to(dev).to(DT)
v = torch.randn((n, Hkv, D), generator=g).to(dev).to(DT)
out = torch.empty_like(q)
cu_h, sl_h = np.asarray([0, n], np.int32), np.asarray([n], np.int32)
cu, sl = torch.from_numpy(cu_h).to(dev), torch.from_numpy(sl_h).to(dev)
PAGE = 16
nblk = (n + PAGE - 1) // PAGE
perm = torch.randperm(nblk, generator=g).to(torch.int32)
kc = torch.zeros((nblk, PAGE, Hkv, D), dtype=DT, device=dev); vc = torch.zeros_like(kc)
pos = torch.arange(n)
slots = (perm.long()[pos // PAGE] * PAGE + pos % PAGE).to(dev)
ops.reshape_and_cache(k, v, kc, vc, slots)
bt = perm.view(1, -1).to(dev)

def timed(fn, iters):
    # an idle GPU needs tens of milliseconds of load before its clocks settle
    import time
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in evs]
    return float(np.median(ts)), float(np.min(ts))

bm, bn = ops.attention_block_m(H, Hkv), ops.attention_block_n()
algo = 2 * (H + Hkv) * D * 2 * n
for name, (si, sl_, last), mode in (("triangle(4,32,64)", (4, 32, 64), "triangle"), ("stream only (last=0)", (4, 32, 0), "triangle"),
                                     ("window only (si=-1,last=0)", (-1, 32, 0), "triangle"), ("diag+strip (si=-1,sl=0)", (-1, 0, 64), "triangle"), ("diag only (si=-1,sl=0,last=0)", (-1, 0, 0), "triangle"), ("sink+diag (si=4,sl=0,last=0)", (4, 0, 0), "triangle"), ("sink31+diag (si=31,sl=0,last=0)", (31, 0, 0), "triangle"), ("causal", (0, 0, 0), "causal")):
    if args.only and not any(name.startswith(o) for o in args.only.split(",")):
        continue
    for chunk in [int(c) for c in args.chunks.split(",")]:
        if mode == "causal" and n > 16384 and args.iters > 3:
            iters = 3
        else:
            iters = args.iters
        idx = ops.build_block_index(cu, sl, cu_h, sl_h, si, sl_, last, bm, bn, ops.MODES[mode], chunk_tiles=chunk,
                                    block_table=bt if args.paged == 1 else None, page_size=PAGE)
        work = ops.alloc_workspace(idx, H, D, dev)
        def run(flags=TMX_ATTN_SKIP_COMBINE):
            if args.paged:
                ops.attention(q, kc, vc, out, idx, cu, sl, D ** -0.5, block_table=bt, workspace=work, flags=flags)
            else:
                ops.attention(q, k, v, out, idx, cu, sl, D ** -0.5, cu_seqlens_k=cu, workspace=work, flags=flags)
        med, mn = timed(run, iters)
        cmed, _ = timed(lambda: run(TMX_ATTN_ONLY_COMBINE), iters)
        print(f"{name:28s} chunk={chunk:4d} items={idx.sizes.num_items:6d} tiles={idx.nnz_tiles:7d} attn med {med*1e3:8.1f} us min {mn*1e3:8.1f} us "
              f"({algo/med/1e6:7.1f} GB/s algo)  combine {cmed*1e3:6.1f} us", flush=True)
