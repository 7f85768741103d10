"""The last 2048-token chunk of a 32768-token prompt (30720 tokens already in the paged cache), one layer, two ways:
(a) cache write, then attention over the paged cache, on one stream; (b) attention that reads the chunk's keys
from the step's own rows (k_new / v_new) while the cache write runs on a second stream; (c) the write fused
into the attention launch (cache_slots).  Also a decode step
(64 requests x 8192 cached tokens) the same two ways.
usage: python tools/chunk_probe.py [--seq-len N] [--chunk C]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vllm_trianglemix_amd import ops  # noqa: E402
from vllm_trianglemix_amd._lib import TMX_ATTN_SHORT_QUERIES  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seq-len", type=int, default=32768)
ap.add_argument("--chunk", type=int, default=2048)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
H, Hkv, D = 32, 8, 128
si, sl, last = bench.PATTERN
bm, bn = ops.attention_block_m(H, Hkv), ops.attention_block_n()
scale = D ** -0.5
side = torch.cuda.Stream(dev)


def timed(fn):
    t_end = time.perf_counter() + 0.15   # let the clocks settle
    while time.perf_counter() < t_end:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2] * 1e3


def case(name, q_lens, seq_lens, flags=0):
    B = len(seq_lens)
    g = torch.Generator().manual_seed(0)
    T = sum(q_lens)
    pages = [(s + bench.PAGE - 1) // bench.PAGE for s in seq_lens]
    nblk = sum(pages)
    perm = torch.randperm(nblk, generator=g).to(torch.int32)
    bt = torch.zeros((B, max(pages)), dtype=torch.int32)
    o = 0
    for b in range(B):
        bt[b, :pages[b]] = perm[o:o + pages[b]]
        o += pages[b]
    kc = torch.randn((nblk, bench.PAGE, Hkv, D), generator=g).to(dev).bfloat16()
    vc = torch.randn((nblk, bench.PAGE, Hkv, D), generator=g).to(dev).bfloat16()
    q = torch.randn((T, H, D), generator=g).to(dev).bfloat16()
    kn = torch.randn((T, Hkv, D), generator=g).to(dev).bfloat16()
    vn = torch.randn((T, Hkv, D), generator=g).to(dev).bfloat16()
    slots = torch.cat([bt[b].long()[torch.arange(s - ql, s) // bench.PAGE] * bench.PAGE + torch.arange(s - ql, s) % bench.PAGE
                       for b, (ql, s) in enumerate(zip(q_lens, seq_lens))]).to(dev)
    cu_h = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    sl_h = np.asarray(seq_lens, np.int32)
    cu, sld, btd = torch.from_numpy(cu_h).to(dev), torch.from_numpy(sl_h).to(dev), bt.to(dev)
    for mode in ("triangle", "causal"):
        pat = (si, sl, last) if mode == "triangle" else (0, 0, 0)
        chunk = ops.suggest_chunk_tiles(cu_h, sl_h, *pat, bm, bn, ops.MODES[mode], ops.units_per_item(H, Hkv))
        idx = ops.build_block_index(cu, sld, cu_h, sl_h, *pat, bm, bn, ops.MODES[mode], chunk_tiles=chunk,
                                    block_table=btd, page_size=bench.PAGE)
        work = ops.alloc_workspace(idx, H, D, dev)
        out_a, out_b = torch.empty_like(q), torch.empty_like(q)

        def serial():
            ops.reshape_and_cache(kn, vn, kc, vc, slots)
            ops.attention(q, kc, vc, out_a, idx, cu, sld, scale, block_table=btd, workspace=work, flags=flags)

        def overlapped():
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ops.reshape_and_cache(kn, vn, kc, vc, slots)
            ops.attention(q, kc, vc, out_b, idx, cu, sld, scale, block_table=btd, workspace=work, flags=flags,
                          k_new=kn, v_new=vn)
            main.wait_stream(side)

        def fused():
            ops.attention(q, kc, vc, out_b, idx, cu, sld, scale, block_table=btd, workspace=work, flags=flags,
                          k_new=kn, v_new=vn, cache_slots=slots)

        def attention_alone():
            ops.attention(q, kc, vc, out_b, idx, cu, sld, scale, block_table=btd, workspace=work, flags=flags,
                          k_new=kn, v_new=vn)

        ta, tb, tf, tc = timed(serial), timed(overlapped), timed(fused), timed(attention_alone)
        print(f"{name:28s} {mode:8s}: write then attention {ta:8.1f} us | write beside two-source attention {tb:8.1f} us "
              f"| write fused into the attention launch {tf:8.1f} us | attention alone {tc:8.1f} us | same output: "
              f"{bool(torch.equal(out_a, out_b))}")


case(f"whole {a.seq_len}-token prompt", [a.seq_len], [a.seq_len])
case(f"last {a.chunk} of {a.seq_len} tokens", [a.chunk], [a.seq_len])
case("decode 64 x 8192", [1] * 64, [8192] * 64, TMX_ATTN_SHORT_QUERIES)
