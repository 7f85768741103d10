#!/usr/bin/env python3
"""Diagnostic: per-segment cycle shares of the attention loop (needs tools/libtrianglemix_hip_diag.so,
built with -DTMX_STAMPS).  Usage: TMX_LIB_PATH=tools/libtrianglemix_hip_diag.so python tools/stamps.py MODE N"""
import os, sys
os.environ["TMX_STAMPS_BUFFER"] = "1"  # this script's (max, sum) buffer has room for the stamps
os.environ.setdefault("TRIANGLEMIX_BINDING", "ctypes")  # the library named by TMX_LIB_PATH, not the operator's
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_trianglemix_amd import ops
from vllm_trianglemix_amd._lib import TMX_ATTN_SKIP_COMBINE
mode = sys.argv[1]; n = int(sys.argv[2]); chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 64
H, Hkv, D = 32, 8, 128
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
q = torch.randn((n, H, D), generator=g).to(dev).bfloat16()
k = torch.randn((n, Hkv, D), generator=g).to(dev).bfloat16()
v = torch.randn((n, Hkv, D), generator=g).to(dev).bfloat16()
out = torch.empty_like(q)
cu_h, sl_h = np.asarray([0, n], np.int32), np.asarray([n], np.int32)
cu, sl = torch.from_numpy(cu_h).to(dev), torch.from_numpy(sl_h).to(dev)
pat = {"triangle": (4, 32, 64, "triangle"), "stream": (4, 32, 0, "triangle"), "window": (-1, 32, 0, "triangle"),
       "causal": (0, 0, 0, "causal")}[mode]
idx = ops.build_block_index(cu, sl, cu_h, sl_h, pat[0], pat[1], pat[2], ops.attention_block_m(H, Hkv), 32,
                            ops.MODES[pat[3]], chunk_tiles=chunk)
need = max(1, int(idx.sizes.num_partials) * H * idx.block_m)
po = torch.empty((need * D,), dtype=torch.float32, device=dev)
assert need * 2 <= 1 << 19
pml = torch.zeros((1 << 21,), dtype=torch.float32, device=dev)  # (max, sum) pairs, then from float 2^19 on the stamps
for _ in range(3):
    ops.attention(q, k, v, out, idx, cu, sl, D ** -0.5, cu_seqlens_k=cu, workspace=(po, pml), flags=TMX_ATTN_SKIP_COMBINE)
torch.cuda.synchronize()
role = sys.argv[4] if len(sys.argv) > 4 else "list"   # "list": work-list workgroups, "band": band workgroups
allst = pml[1 << 19:].view(torch.int64)[: 4096 * 8 * 8].view(-1, 8, 8).cpu().numpy()  # [workgroup, wave, segment]
n_list = min(512, 4 * max(int(idx.sizes.num_partials), 4) * 8 // 4)  # work-list workgroups come first in the grid
if role == "band":
    st = allst[256:]
    names = ["wait vmcnt", "barrier", "scores", "issue", "softmax", "PV + store", "TOTAL", "steps"]
else:
    st = allst[:256]
    names = ["wait", "barrier", "issue", "tile compute", "epilogue", "next unit", "TOTAL", "steps"]
st = st[(st[:, 0, 6] > 0) & (st[:, 0, 7] > 0)]
tot = st[:, :, 6].mean()
print(f"{mode} n={n}: {len(st)} workgroups, mean total {tot:.0f} cycles (100 MHz ticks? see below), steps/wave {st[:, :, 7].mean():.1f}")
for i, nm in enumerate(names[:6]):
    print(f"  {nm:14s} {st[:, :, i].mean():12.0f}  {100 * st[:, :, i].mean() / tot:5.1f} %   per step {st[:, :, i].mean() / st[:, :, 7].mean():8.1f}")
wt = st[:, :, 6].max(axis=1)  # a workgroup is done when its slowest wave is
print(f"  workgroup totals: min {wt.min():.0f}  p10 {np.percentile(wt, 10):.0f}  median {np.median(wt):.0f}  p90 {np.percentile(wt, 90):.0f}  max {wt.max():.0f}")
if role != "band":
    # waves 0..3 fetch the tiles and run the scalar walk, waves 4..7 only compute
    for nm, sl_ in (("issuing waves", slice(0, 4)), ("other waves", slice(4, 8))):
        sub = st[:, sl_, :]
        print(f"  {nm}: " + "  ".join(f"{n} {sub[:, :, i].mean() / sub[:, :, 7].mean():.0f}" for i, n in enumerate(names[:6])) + "  (cycles per step)")
acc = st[:, :, :6].sum(axis=2).mean()
print(f"  unaccounted    {tot - acc:12.0f}  {100 * (tot - acc) / tot:5.1f} %")
