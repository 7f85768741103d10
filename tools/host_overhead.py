#!/usr/bin/env python3
"""Host time to enqueue one step (index build + attention) through the two bindings of the library:
the ctypes path the backend uses (vllm_trianglemix_amd.ops) and the torch.ops._trianglemix_C operators.
The GPU queue is kept empty between calls (synchronize before every timed call), so the figures are the
host's own work, not back pressure.  Usage: python tools/host_overhead.py [N]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_trianglemix_amd import ops, torch_ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
H, Hkv, D, PAGE = 32, 8, 128, 16
si, sl, last = 4, 32, 64
dev = torch.device("cuda:0")
C = torch_ops.load()
g = torch.Generator().manual_seed(0)
q = torch.randn((n, H, D), generator=g).to(dev).bfloat16()
nblk = n // PAGE
kc = torch.randn((nblk, PAGE, Hkv, D), generator=g).to(dev).bfloat16()
vc = torch.randn((nblk, PAGE, Hkv, D), generator=g).to(dev).bfloat16()
bt = torch.randperm(nblk, generator=g).to(torch.int32).view(1, -1).to(dev)
out = torch.empty_like(q)
cu_h, sl_h = np.asarray([0, n], np.int32), np.asarray([n], np.int32)
cu, sld = torch.from_numpy(cu_h).to(dev), torch.from_numpy(sl_h).to(dev)
cu_t, sl_t = torch.from_numpy(cu_h), torch.from_numpy(sl_h)
bm, bn, mode = ops.attention_block_m(H, Hkv), ops.attention_block_n(), ops.MODES["triangle"]
chunk = ops.suggest_chunk_tiles(cu_h, sl_h, si, sl, last, bm, bn, mode, ops.units_per_item(H, Hkv))
scale = D ** -0.5


def host_us(fn, reps=200):
    for _ in range(10):
        fn()
    tot = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        tot += time.perf_counter() - t0
    torch.cuda.synchronize()
    return tot / reps * 1e6


# ---- ctypes path ----
idx = ops.build_block_index(cu, sld, cu_h, sl_h, si, sl, last, bm, bn, mode, chunk_tiles=chunk, block_table=bt, page_size=PAGE)
ws = ops.alloc_workspace(idx, H, D, dev)
t_idx_c = host_us(lambda: ops.build_block_index(cu, sld, cu_h, sl_h, si, sl, last, bm, bn, mode, chunk_tiles=chunk,
                                                  block_table=bt, page_size=PAGE))
ops._BINDING = "ctypes"
t_att_c = host_us(lambda: ops.attention(q, kc, vc, out, idx, cu, sld, scale, block_table=bt, workspace=ws))
ops._BINDING = "torch"
t_att_o = host_us(lambda: ops.attention(q, kc, vc, out, idx, cu, sld, scale, block_table=bt, workspace=ws))

# ---- torch.ops path ----
R, nnz, n_items, n_part, n_split, _ = C.trianglemix_index_sizes(cu_t, sl_t, si, sl, last, bm, bn, mode, 0, chunk)
i32 = lambda m: torch.empty((max(m, 1),), dtype=torch.int32, device=dev)  # noqa: E731
index = [i32(R + 1), i32(nnz), i32(2 * R), i32(R + 1), i32(8 * n_items), i32(n_split), i32(8), i32(2 + 2 * (R + 1) + 8 * (R // 1024 + 1))]
tile_pages = i32(2 * nnz)
part_o = torch.empty((max(1, n_part * H * bm * D),), dtype=torch.float32, device=dev)
part_ml = torch.empty((max(1, n_part * H * bm * 2),), dtype=torch.float32, device=dev)
t_sz_t = host_us(lambda: C.trianglemix_index_sizes(cu_t, sl_t, si, sl, last, bm, bn, mode, 0, chunk))
t_idx_t = host_us(lambda: C.trianglemix_build_block_index(index, tile_pages, cu, sld, cu_t, sl_t, bt, PAGE, si, sl, last, bm, bn,
                                                          mode, 0, chunk))
t_att_t = host_us(lambda: C.trianglemix_attention(out, q, kc, vc, bt, cu, sld, None, index, tile_pages, part_o, part_ml, None,
                                                  scale, si, sl, last, mode, 0, chunk, n_items, n_part, n_split, None, 0, nnz))
print(f"n={n}: host microseconds per call (GPU idle)")
print(f"  ops.build_block_index (sizes + one buffer + launch, ctypes) {t_idx_c:7.1f}")
print(f"  ops.attention through ctypes (C struct filled in Python)    {t_att_c:7.1f}")
print(f"  ops.attention through the registered operator (default)     {t_att_o:7.1f}")
print(f"  torch.ops._trianglemix_C called directly: index sizes {t_sz_t:5.1f}, index build (launch only) {t_idx_t:5.1f}, "
      f"attention {t_att_t:5.1f}")
