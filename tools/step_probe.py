"""Where one bench step's time goes: host enqueue time against device time, and the same step replayed from a
HIP graph (index build + attention + combine captured once).  usage: python tools/step_probe.py [--seq-len N]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vllm_trianglemix_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seq-len", type=int, default=32768)
ap.add_argument("--iters", type=int, default=100)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
H, Hkv, D = 32, 8, 128
inp = bench.make_inputs(ops, dev, a.seq_len, 1, H, Hkv, D, seed=0)
si, sl, last = bench.PATTERN
bm, bn = ops.attention_block_m(H, Hkv), ops.attention_block_n()
out = torch.empty_like(inp["q"])
kc, vc = inp["kv_cache"][0], inp["kv_cache"][1]
chunk = ops.suggest_chunk_tiles(inp["cu_h"], inp["sl_h"], si, sl, last, bm, bn, ops.MODES["triangle"],
                                ops.units_per_item(H, Hkv))


def build():
    return ops.build_block_index(inp["cu"], inp["sl"], inp["cu_h"], inp["sl_h"], si, sl, last, bm, bn,
                                 ops.MODES["triangle"], chunk_tiles=chunk, block_table=inp["block_table"],
                                 page_size=bench.PAGE)


work = ops.alloc_workspace(build(), H, D, dev)


def step():
    ops.attention(inp["q"], kc, vc, out, build(), inp["cu"], inp["sl"], D ** -0.5, block_table=inp["block_table"],
                  workspace=work)


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"eager: host enqueue {1e6 * (t1 - t0) / a.iters:7.1f} us/step, total {1e6 * (t2 - t0) / a.iters:7.1f} us/step")

t0 = time.perf_counter()
for _ in range(a.iters):
    build()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"build only: host {1e6 * (t1 - t0) / a.iters:7.1f} us")

ref = out.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
out.zero_()
g.replay()
torch.cuda.synchronize()
print("graph replay equals eager:", bool(torch.equal(out, ref)))
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    g.replay()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"graph: total {1e6 * (t2 - t0) / a.iters:7.1f} us/step")
