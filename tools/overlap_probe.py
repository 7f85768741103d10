"""A whole-prompt prefill layer two ways: (a) cache write, then attention over the paged cache, on one stream;
(b) attention straight over the step's K/V rows while the cache write runs on a second stream.
usage: python tools/overlap_probe.py [--seq-len N]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vllm_trianglemix_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seq-len", type=int, default=32768)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
H, Hkv, D, n = 32, 8, 128, a.seq_len
inp = bench.make_inputs(ops, dev, n, 1, H, Hkv, D, seed=0)
si, sl, last = bench.PATTERN
bm, bn = ops.attention_block_m(H, Hkv), ops.attention_block_n()
kc, vc = inp["kv_cache"][0], inp["kv_cache"][1]
pos = torch.arange(n, dtype=torch.int64)
slots = (inp["block_table"][0].cpu().to(torch.int64)[pos // bench.PAGE] * bench.PAGE + pos % bench.PAGE).to(dev)
chunk = ops.suggest_chunk_tiles(inp["cu_h"], inp["sl_h"], si, sl, last, bm, bn, ops.MODES["triangle"],
                                ops.units_per_item(H, Hkv))
idx = ops.build_block_index(inp["cu"], inp["sl"], inp["cu_h"], inp["sl_h"], si, sl, last, bm, bn,
                            ops.MODES["triangle"], chunk_tiles=chunk, block_table=inp["block_table"],
                            page_size=bench.PAGE)
work = ops.alloc_workspace(idx, H, D, dev)
out_a, out_b = torch.empty_like(inp["q"]), torch.empty_like(inp["q"])
side = torch.cuda.Stream(dev)
scale = D ** -0.5


def serial():
    ops.reshape_and_cache(inp["k"], inp["v"], kc, vc, slots)
    ops.attention(inp["q"], kc, vc, out_a, idx, inp["cu"], inp["sl"], scale, block_table=inp["block_table"],
                  workspace=work)


def overlapped():
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ops.reshape_and_cache(inp["k"], inp["v"], kc, vc, slots)
    ops.attention(inp["q"], inp["k"], inp["v"], out_b, idx, inp["cu"], inp["sl"], scale, cu_seqlens_k=inp["cu"],
                  workspace=work)
    main.wait_stream(side)


def rows_only():
    ops.attention(inp["q"], inp["k"], inp["v"], out_b, idx, inp["cu"], inp["sl"], scale, cu_seqlens_k=inp["cu"],
                  workspace=work)


def timed(fn):
    for _ in range(5):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2] * 1e3


for r in range(2):
    print(f"cache write then paged attention: {timed(serial):7.1f} us   attention over the rows + cache write on a "
          f"second stream: {timed(overlapped):7.1f} us   attention over the rows alone: {timed(rows_only):7.1f} us")
print("same output:", bool(torch.equal(out_a, out_b)))
