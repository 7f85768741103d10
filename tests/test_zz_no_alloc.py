"""The entry point that allocates nothing (tmx_attention_static_ws): same results, bit for bit, as
tmx_attention with a caller-provided work area; two calls in a row on one GPU, the second smaller."""
import numpy as np
import pytest
import torch

from oracle import trianglemix_oracle as orc
from vllm_trianglemix_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(seed, q_lens, seq_lens, H, Hkv, D, chunk):
    rng = np.random.default_rng(seed)
    T, rows = int(sum(q_lens)), int(sum(seq_lens))
    q = torch.from_numpy(rng.standard_normal((T, H, D), dtype=np.float32)).to(DEV).bfloat16()
    k = torch.from_numpy(rng.standard_normal((rows, Hkv, D), dtype=np.float32)).to(DEV).bfloat16()
    v = torch.from_numpy(rng.standard_normal((rows, Hkv, D), dtype=np.float32)).to(DEV).bfloat16()
    cu_h = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    sl_h = np.asarray(seq_lens, np.int32)
    cuk_h = np.concatenate([[0], np.cumsum(seq_lens)]).astype(np.int32)
    cu, sld, cuk = (torch.from_numpy(x).to(DEV) for x in (cu_h, sl_h, cuk_h))
    bm, bn = ops.attention_block_m(H, Hkv), ops.attention_block_n()
    idx = ops.build_block_index(cu, sld, cu_h, sl_h, 4, 32, 64, bm, bn, ops.MODES["triangle"], chunk_tiles=chunk)
    return q, k, v, idx, cu, sld, cuk, cu_h, cuk_h


def _run(case, static_workspace, out=None):
    q, k, v, idx, cu, sld, cuk, _, _ = case
    if out is None:
        out = torch.empty(q.shape, dtype=torch.float32, device=DEV)
    out.fill_(float("nan"))
    ops.attention(q, k, v, out, idx, cu, sld, q.shape[-1] ** -0.5, cu_seqlens_k=cuk, static_workspace=static_workspace)
    return out


def test_static_workspace_entry_point_matches_the_allocating_one():
    assert _lib.load().tmx_static_workspace_bytes() == 32 << 20
    # the set-up call a capturing caller makes once per GPU (resolves the array's device address); idempotent
    assert _lib.load().tmx_static_workspace_warmup() == 0 and _lib.load().tmx_static_workspace_warmup() == 0
    H, Hkv, D = 8, 2, 128
    # first call: two long prompts + a decode row over a long context, many pieces (chunk 4);
    # second call, straight after on the same stream: a smaller step (fewer, other pieces)
    big = _case(1, [3000, 1, 2500], [3000, 2700, 2500], H, Hkv, D, chunk=4)
    small = _case(2, [700, 1], [700, 900], H, Hkv, D, chunk=8)
    assert big[3].sizes.num_partials > small[3].sizes.num_partials > 0
    got_big = torch.empty(big[0].shape, dtype=torch.float32, device=DEV)
    got_small = torch.empty(small[0].shape, dtype=torch.float32, device=DEV)
    torch.cuda.synchronize()
    stats = torch.cuda.memory_stats()
    before = (torch.cuda.memory_allocated(), stats["allocation.all.allocated"])
    _run(big, True, got_big)
    _run(small, True, got_small)
    # the calls took nothing from torch's allocator either (outputs are the test's own, made above)
    stats = torch.cuda.memory_stats()
    assert (torch.cuda.memory_allocated(), stats["allocation.all.allocated"]) == before
    want_big, want_small = _run(big, False), _run(small, False)
    torch.cuda.synchronize()
    assert torch.equal(got_big, want_big) and torch.equal(got_small, want_small)
    assert not torch.isnan(got_big).any() and not torch.isnan(got_small).any()
    # and against the oracle, to the bar of the other parity tests
    q, k, v, idx, cu, sld, cuk, cu_h, cuk_h = small
    kf, vf = k.float().cpu().numpy(), v.float().cpu().numpy()
    ref = orc.varlen_gqa_attention(q.float().cpu().numpy(), lambda b: kf[cuk_h[b]:cuk_h[b + 1]],
                                   lambda b: vf[cuk_h[b]:cuk_h[b + 1]], cu_h, [700, 900], D ** -0.5, "triangle",
                                   4, 32, 64)
    from parity_tolerance import assert_close
    assert_close(got_small.cpu().numpy(), ref, "tmx_attention_static_ws")


def test_static_workspace_rejects_steps_beyond_its_size():
    H, Hkv, D = 32, 8, 128
    # 40 pieces x 32 heads x 32 rows x 130 floats = 21 MB fits; chunk 1 on a 3000-token strip does not
    q_lens = seq_lens = [3000]
    case = _case(3, q_lens, seq_lens, H, Hkv, D, chunk=1)
    need = int(case[3].sizes.num_partials) * H * 32 * (D + 2) * 4
    assert need > (32 << 20)
    with pytest.raises(_lib.TmxError) as e:
        _run(case, True)
    assert e.value.code == _lib.TMX_ERR_CAPACITY and "work area" in str(e.value)
