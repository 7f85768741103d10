"""Times the dense causal kernel alone (one prompt, paged 16-bit cache, Qwen3-8B heads) and checks it against torch's
fp32 SDPA on a few rows.  usage (GPU box): TRIANGLEMIX_BINDING=ctypes TMX_LIB_PATH=... python tools/dense_bench.py [n]
TMX_DENSE_STAGE=0: without the staging area (the register form of the tile fetch); 2: staged at any length."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vllm_trianglemix_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device("cuda", 0)
H, Hkv, D = 32, 8, 128
inp = bench.make_inputs(ops, dev, n, 1, H, Hkv, D, seed=0, fp8=False)
si, sl, last = bench.PATTERN
bm, bn = ops.attention_block_m(H, Hkv), ops.attention_block_n()
# (TMX_DENSE_FULL_INDEX=1: the index lists the prompt's tiles although the dense kernel does not read them)
leave = os.environ.get("TMX_DENSE_FULL_INDEX", "0") != "1"
idx = ops.build_block_index(inp["cu"], inp["sl"], inp["cu_h"], inp["sl_h"], 0, 0, 0, bm, bn, ops.MODES["causal"],
                            chunk_tiles=64, block_table=inp["block_table"], page_size=bench.PAGE, leave_dense=leave)
staged = os.environ.get("TMX_DENSE_STAGE", "1") != "0"
if os.environ.get("TMX_DENSE_STAGE") == "2":  # staged whatever the length (the library's default: from 4096 rows on)
    idx.stage_rows = ops.dense_stage_rows(inp["cu_h"], inp["sl_h"], 0)
work = ops.alloc_workspace(idx, H, D, dev, num_kv_heads=Hkv if staged else None)
out = torch.empty_like(inp["q"])
kc, vc = inp["kv_cache"][0], inp["kv_cache"][1]


def run():
    ops.attention(inp["q"], kc, vc, out, idx, inp["cu"], inp["sl"], D ** -0.5, block_table=inp["block_table"], workspace=work)


for _ in range(3):
    run()
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in evs:
    a.record()
    run()
    b.record()
torch.cuda.synchronize()
ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
fl = 4.0 * D * H * bench.allowed_pairs(n, si, sl, last, False)
# a few rows against fp32 arithmetic on the same keys
bt = inp["block_table"][0].long()
k = kc[bt].reshape(-1, Hkv, D)[:n].float()
v = vc[bt].reshape(-1, Hkv, D)[:n].float()
err = 0.0
for row in (0, 1, 63, 64, min(2047, n - 1), n // 2 + 17, n - 65, n - 1):
    q = inp["q"][row].float().view(Hkv, H // Hkv, D)
    sc = torch.einsum("kgd,nkd->kgn", q, k[: row + 1]) * D ** -0.5
    ref = torch.einsum("kgn,nkd->kgd", torch.softmax(sc, -1), v[: row + 1]).reshape(H, D)
    err = max(err, float((out[row].float() - ref).abs().max()))
print(f"dense causal n={n} ({'staged' if len(work) > 2 else 'register'} form): {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s  max err on sampled rows {err:.2e}", flush=True)
