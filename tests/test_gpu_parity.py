"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
golden fixtures generated from the reference.  Needs an MI355X."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import trianglemix_oracle as orc
from vllm_trianglemix_amd._lib import TMX_ATTN_NO_BAND, TMX_ATTN_SHORT_QUERIES

pytestmark = pytest.mark.gpu

DEV = "cuda:0"

# attention tolerance: fp32 kernel output vs the fp64 oracle on the same 16-bit-rounded inputs, the rule of
# tests/parity_tolerance.py (1e-3, relative above |ref| = 1; a 16-bit OUTPUT adds its own rounding, 2^-8 relative);
# every check goes through assert_close, which records the error it measured under the kernel's / route's name
from parity_tolerance import ATOL as ATOL_F32_OUT, assert_close  # noqa: E402


def _pack(m):
    return np.packbits(np.asarray(m) > 0, axis=None).tobytes()


def _ops():
    from vllm_trianglemix_amd import ops
    return ops


# ------------------------------------------------------------------ masks
def test_dense_masks_match_golden(golden_masks):
    ops = _ops()
    kinds = {"streaming": "streaming", "last_mask": "last", "middle": "middle", "triangle": "triangle",
             "dense": "dense"}
    for c in golden_masks:
        si, sl, last, n = c["si"], c["sl"], c["last"], c["n"]
        for key, kind in kinds.items():
            if key not in c:
                continue
            m = ops.dense_mask(n, si, sl, last, kind, DEV, torch.float32).cpu().numpy()
            assert set(np.unique(m)) <= {0.0, 1.0}
            assert int(m.sum()) == c[key]["nnz"], (c, key)
            assert hashlib.sha256(_pack(m)).hexdigest() == c[key]["sha256"], (c, key)
        for dt_name, dt in (("float32", torch.float32), ("bfloat16", torch.bfloat16)):
            for tri in (False, True):
                key = f"additive_{dt_name}_{'tri' if tri else 'dense'}"
                if key not in c:
                    continue
                m = ops.dense_mask(n, si, sl, last, "triangle" if tri else "dense", DEV, dt, additive=True)
                assert m.dtype == dt and tuple(m.shape) == (n, n)
                mf = m.float().cpu().numpy()
                assert np.all((mf == 0) | np.isneginf(mf))
                assert hashlib.sha256(_pack(mf == 0)).hexdigest() == c[key]["sha256"]


def test_dense_mask_dtypes_and_oracle_large():
    ops = _ops()
    for n, (si, sl, last) in ((3000, (4, 32, 64)), (777, (9, 100, 300)), (8, (0, 0, 0))):
        want = orc.triangle_mask(n, si, sl, last)
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            got = ops.dense_mask(n, si, sl, last, "triangle", DEV, dt).float().cpu().numpy()
            assert np.array_equal(got > 0, want) and set(np.unique(got)) <= {0.0, 1.0}
            add = ops.dense_mask(n, si, sl, last, "triangle", DEV, dt, additive=True).float().cpu().numpy()
            assert np.array_equal(add == 0, want) and np.all(np.isneginf(add[~want]))


def test_generator_api_and_helper(golden_selection):
    from vllm_trianglemix_amd import (AscendNPUTriangleMixOptimizer, TriangleMixAttention, TriangleMixConfig,
                                      TriangleMixMaskGenerator)
    gen = TriangleMixMaskGenerator(TriangleMixConfig(4, 8, 16))
    m = gen._create_streaming_mask(24, DEV, torch.float32).cpu().numpy()
    assert m[10, 3] == 1 and m[15, 8] == 1 and m[15, 1] == 1 and m[15, 5] == 0 and m.sum() == 245
    assert gen._create_last_mask(24, DEV, torch.float32).sum().item() == 55
    assert gen._create_middle_mask(24, DEV, torch.float32).sum().item() == 0
    assert gen._create_triangle_mask(24, DEV, torch.float32).sum().item() == 300
    a = gen.get_attention_mask(24, 0, DEV, torch.float32, use_triangle=True)
    assert a is gen.get_attention_mask(24, 0, DEV, torch.float32, use_triangle=True)
    gen.clear_cache()
    for rec in golden_selection["helper"]:
        cfg = TriangleMixConfig(4, 8, 16, num_triangle_layers=rec["num_triangle_layers"],
                                triangle_layer_indices=rec["triangle_layer_indices"])
        helper = TriangleMixAttention(cfg, rec["num_layers"])
        assert list(cfg.triangle_layer_indices) == rec["indices_after_init"]
        for d in rec["decisions"]:
            m = helper.get_attn_mask(d["n"], d["layer"], DEV)
            if d["kind"] == "none":
                assert m is None
                continue
            assert m.dtype == torch.float32
            allowed = (m == 0).cpu().numpy()
            if d["kind"] == "dense":
                assert np.array_equal(allowed, np.tril(np.ones((d["n"], d["n"]), bool)))
            else:
                assert hashlib.sha256(_pack(allowed)).hexdigest() == rec["triangle_sha256"][str(d["n"])]
    om = golden_selection["optimize_mask"]
    m = torch.full((4, 4), float("-inf"), device=DEV)
    m[0, 0] = 0.0
    m[1, 0:2] = 0.0
    m[2, 0:3] = 0.0
    m[3, :] = 0.0
    o = AscendNPUTriangleMixOptimizer.optimize_mask_for_npu(m)
    assert np.array_equal(o.cpu().numpy(), np.asarray(om["out"], np.float32))
    assert torch.isinf(m).any()  # the input is left alone
    ob = AscendNPUTriangleMixOptimizer.optimize_mask_for_npu(m.bfloat16())
    assert ob.dtype == torch.bfloat16 and not torch.isinf(ob).any()
    assert ob.float().min().item() == torch.tensor(-1e6).bfloat16().float().item()


def test_ascend_block_level_masks(golden_selection):
    """Block-level variant: block masks vs the reference's own output, value conventions, thresholds."""
    from vllm_trianglemix_amd.ascend_npu import (AscendBlockSparseMaskGenerator, AscendNPUTriangleMixAttention,
                                                  AscendNPUTriangleMixConfig)
    for rec in golden_selection["ascend_block"]:
        cfg = AscendNPUTriangleMixConfig(rec["si"], rec["sl"], rec["last"], npu_block_size=rec["block"])
        gen = AscendBlockSparseMaskGenerator(cfg)
        bm = gen._get_block_sparse_mask(rec["n"], DEV, torch.float32)
        got = bm.cpu().numpy()
        assert list(got.shape) == rec["shape"] and int(got.sum()) == rec["nnz"]
        assert hashlib.sha256(_pack(got)).hexdigest() == rec["sha256"]
        tok = gen._expand_block_mask_to_token_level(bm, rec["n"], DEV, torch.float32).cpu().numpy()
        want = np.kron(got, np.ones((rec["block"], rec["block"]), np.float32))[:rec["n"], :rec["n"]]
        assert np.array_equal(tok, want)
    cfg = AscendNPUTriangleMixConfig(num_triangle_layers=1)
    helper = AscendNPUTriangleMixAttention(cfg, num_layers=2)
    assert helper.get_attn_mask(512, 0, DEV) is None
    m = helper.get_attn_mask(600, 0, DEV, torch.float32)            # triangle layer, token level below 2048
    want = np.where(orc.triangle_mask(600, 4, 32, 64), 0.0, -1e9).astype(np.float32)
    assert np.array_equal(m.cpu().numpy(), want)
    mb = helper.get_attn_mask(4096, 0, DEV, torch.float32)           # triangle layer, block level above 2048
    assert tuple(mb.shape) == (64, 64)
    assert np.array_equal(mb.cpu().numpy(), np.where(orc.ascend_block_mask(4096, 4, 32, 64, 64) > 0, 0.0, -1e9)
                          .astype(np.float32))
    md = helper.get_attn_mask(5000, 1, DEV, torch.bfloat16)          # dense layer above 4096: 64-token blocks
    assert tuple(md.shape) == (79, 79) and md.dtype == torch.bfloat16
    assert np.array_equal(md.float().cpu().numpy() == 0, np.tril(np.ones((79, 79), bool)))


# ------------------------------------------------------------------ index
def _build(ops, q_lens, seq_lens, si, sl, last, bm, bn, mode, chunk=4, gate=0, block_table=None, page_size=0,
           full_lens=None):
    cu_h = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    sl_h = np.asarray(seq_lens, np.int32)
    cu = torch.from_numpy(cu_h).to(DEV)
    sld = torch.from_numpy(sl_h).to(DEV)
    fl_h = None if full_lens is None else np.asarray(full_lens, np.int32)
    fl = None if full_lens is None else torch.from_numpy(fl_h).to(DEV)
    idx = ops.build_block_index(cu, sld, cu_h, sl_h, si, sl, last, bm, bn, mode, chunk_tiles=chunk, tri_min_len=gate,
                                block_table=block_table, page_size=page_size, full_lens=fl, full_lens_cpu=fl_h)
    return idx, cu, sld


def _check_index(ops, q_lens, seq_lens, si, sl, last, bm, bn, mode, chunk, full_lens=None):
    idx, _, _ = _build(ops, q_lens, seq_lens, si, sl, last, bm, bn, mode, chunk, full_lens=full_lens)
    row_ptr, col_idx, qb_seq, qb_row0 = orc.block_index_closed_form(q_lens, seq_lens, si, sl, last, bm, bn, mode,
                                                                    full_lens)
    items, part_ptr = orc.work_items(row_ptr, chunk, qb_seq, qb_row0, q_lens, seq_lens)
    tot = idx.totals.cpu().numpy()
    assert tot[0] == len(row_ptr) - 1 == idx.sizes.num_qblocks
    assert tot[1] == len(col_idx) == idx.sizes.nnz_tiles
    assert tot[2] == len(items) == idx.sizes.num_items
    assert tot[3] == part_ptr[-1] == idx.sizes.num_partials
    assert np.array_equal(idx.row_ptr.cpu().numpy(), row_ptr)
    assert np.array_equal(idx.col_idx.cpu().numpy(), col_idx)
    meta = idx.qb_meta.cpu().numpy()
    assert np.array_equal(meta[:, 0], qb_seq) and np.array_equal(meta[:, 1], qb_row0)
    assert np.array_equal(idx.part_ptr.cpu().numpy(), part_ptr)
    assert np.array_equal(idx.items.cpu().numpy(), items)
    split = [qb for qb in range(len(row_ptr) - 1) if part_ptr[qb + 1] > part_ptr[qb]]
    assert np.array_equal(idx.split_qb.cpu().numpy(), np.asarray(split, np.int32))
    assert tot[4] == len(split)
    cnt = np.diff(row_ptr)
    assert tot[5] == (cnt.max() if len(cnt) else 0) == idx.sizes.max_tiles_per_qblock


def test_block_index_bit_exact_vs_oracle():
    ops = _ops()
    rng = np.random.default_rng(3)
    cases = [([24], [24], 4, 8, 16), ([70], [70], 4, 32, 64), ([300], [300], 4, 32, 64), ([1], [1], 0, 0, 0),
             ([5, 1, 40], [50, 90, 40], 4, 32, 64), ([33, 0, 17], [33, 12, 100], 2, 5, 9), ([0], [7], 4, 32, 64),
             ([1] * 9, [100, 2000, 1, 64, 65, 63, 33, 4097, 512], 4, 32, 64)]
    for _ in range(20):
        b = int(rng.integers(1, 6))
        seq = [int(rng.integers(1, 700)) for _ in range(b)]
        ql = [int(rng.integers(0, s + 1)) for s in seq]
        cases.append((ql, seq, int(rng.integers(0, 9)), int(rng.integers(0, 40)), int(rng.integers(0, 70))))
    for ql, seq, si, sl, last in cases:
        for bm, bn in ((16, 32), (64, 32), (32, 32), (64, 64)):
            for mode in (orc.MODE_TRIANGLE, orc.MODE_CAUSAL, orc.MODE_NONE):
                _check_index(ops, ql, seq, si, sl, last, bm, bn, mode, chunk=3)
    # chunked prefill: the pattern of the complete prompt (full length >= seq_len, or shorter: ignored)
    for ql, seq, si, sl, last in cases[:12]:
        full = [int(s + rng.integers(-5, 200)) for s in seq]
        for bm in (16, 32):
            _check_index(ops, ql, seq, si, sl, last, bm, 32, orc.MODE_TRIANGLE, chunk=3, full_lens=full)
        bf = orc.block_index_bruteforce(ql, seq, si, sl, last, 16, 32, orc.MODE_TRIANGLE, full)
        cf = orc.block_index_closed_form(ql, seq, si, sl, last, 16, 32, orc.MODE_TRIANGLE, full)
        assert np.array_equal(bf[0], cf[0]) and np.array_equal(bf[1], cf[1])
    # bruteforce cross-check from the token bitmap (the definition of the index)
    for ql, seq, si, sl, last in cases[:8]:
        idx, _, _ = _build(ops, ql, seq, si, sl, last, 16, 32, orc.MODE_TRIANGLE)
        bf = orc.block_index_bruteforce(ql, seq, si, sl, last, 16, 32, orc.MODE_TRIANGLE)
        assert np.array_equal(idx.row_ptr.cpu().numpy(), bf[0]) and np.array_equal(idx.col_idx.cpu().numpy(), bf[1])


def test_block_index_large_and_many_requests():
    ops = _ops()
    # more than 1024 requests and more than 1024 q-blocks: the scan runs several rounds
    rng = np.random.default_rng(5)
    seq = [int(x) for x in rng.integers(1, 300, size=2500)]
    ql = [int(rng.integers(0, min(s, 40) + 1)) for s in seq]
    _check_index(ops, ql, seq, 4, 32, 64, 16, 32, orc.MODE_TRIANGLE, chunk=2)
    for n in (8192, 32768, 65536):
        _check_index(ops, [n], [n], 4, 32, 64, 16, 32, orc.MODE_TRIANGLE, chunk=64)
        _check_index(ops, [n], [n], 4, 32, 64, 64, 64, orc.MODE_TRIANGLE, chunk=64)
    # several long prompts: split q-blocks (and their partial slots, pieces, positions in split_qb) in different
    # blocks of 1024 q-blocks, which the builder scans side by side and joins by the blocks' totals
    _check_index(ops, [9000, 5000, 7777, 3100, 1], [9000, 5000, 7777, 3100, 4000], 4, 32, 64, 16, 32, orc.MODE_TRIANGLE, chunk=8)
    _check_index(ops, [2000, 6000, 40, 8000], [9000, 6000, 7777, 8100], 4, 32, 64, 32, 32, orc.MODE_TRIANGLE, chunk=16,
                 full_lens=[9000, 6000, 9000, 8100])
    _check_index(ops, [20000, 20000], [20000, 20000], 0, 0, 0, 16, 32, orc.MODE_CAUSAL, chunk=100)
    idx, _, _ = _build(ops, [32768], [32768], 4, 32, 64, 64, 64, orc.MODE_TRIANGLE, chunk=64)
    assert idx.nnz_tiles == 2042
    _check_index(ops, [4096], [4096], 4, 32, 64, 16, 32, orc.MODE_CAUSAL, chunk=64)
    # the per-request length gate: short requests fall back to causal
    idx, _, _ = _build(ops, [100, 3000], [100, 3000], 4, 32, 64, 16, 32, orc.MODE_TRIANGLE, chunk=64, gate=2048)
    a = orc.block_index_closed_form([100], [100], 4, 32, 64, 16, 32, orc.MODE_CAUSAL)
    b = orc.block_index_closed_form([3000], [3000], 4, 32, 64, 16, 32, orc.MODE_TRIANGLE)
    assert np.array_equal(idx.col_idx.cpu().numpy(), np.concatenate([a[1], b[1]]))


def test_block_index_capacity_overflow_beyond_1024_qblocks():
    """The device counts more q-blocks than the caller's buffers were sized for (the host copies of the batch the
    sizes came from were stale), with more than 1024 q-blocks so that the build takes its two-launch form: nothing
    is written past the capacities, ``totals[0]`` reports the q-blocks the device counted so that the caller can
    see the overflow, and a build with the right sizes straight after is bit-exact."""
    ops = _ops()
    bm, bn = 32, 32
    small_q = [40000]            # what the host believed: 1250 q-blocks
    big_q = [70000]              # what the device holds: 2188 q-blocks (three blocks of 1024 instead of two)
    cu_small, sl_small = np.asarray([0, small_q[0]], np.int32), np.asarray(small_q, np.int32)
    cu_big, sl_big = np.asarray([0, big_q[0]], np.int32), np.asarray(big_q, np.int32)
    sz = ops.index_sizes(cu_small, sl_small, 4, 32, 64, bm, bn, ops.MODES["triangle"], 64)
    assert 1024 < sz.num_qblocks < 2048
    # device arrays of the BIG batch, host mirrors (and so every capacity) of the SMALL one
    idx = ops.build_block_index(torch.from_numpy(cu_big).to(DEV), torch.from_numpy(sl_big).to(DEV), cu_small, sl_small,
                                4, 32, 64, bm, bn, ops.MODES["triangle"], chunk_tiles=64)
    guard = torch.full((4096,), 0x5A5A5A5A, dtype=torch.int32, device=DEV)  # (a neighbour allocation, for good measure)
    torch.cuda.synchronize()
    assert int(idx.totals[0]) == -(-big_q[0] // bm) > idx.num_qblocks
    assert bool((guard == 0x5A5A5A5A).all())
    good = ops.build_block_index(torch.from_numpy(cu_big).to(DEV), torch.from_numpy(sl_big).to(DEV), cu_big, sl_big,
                                 4, 32, 64, bm, bn, ops.MODES["triangle"], chunk_tiles=64)
    want = orc.block_index_closed_form(big_q, big_q, 4, 32, 64, bm, bn, orc.MODE_TRIANGLE)
    assert np.array_equal(good.row_ptr.cpu().numpy(), want[0]) and np.array_equal(good.col_idx.cpu().numpy(), want[1])


# ------------------------------------------------------------------ attention
def _to_dev(x, dtype):
    return torch.from_numpy(x).to(DEV).to(dtype)


def _run_varlen(ops, q, k_list, v_list, q_lens, seq_lens, scale, mode, pattern, dtype=torch.bfloat16,
                out_dtype=torch.float32, paged=False, page_size=16, bias=None, chunk=64, gate=0, q_pad=0,
                resolve_pages=True, full_lens=None, flags=0, stage=False):
    """q: [T, H, D] float32 (already rounded); k_list/v_list: per-request [seq_len, Hkv, D].
    stage: a causal launch gets a staging area (poisoned with NaN) whatever the requests' lengths."""
    T, H, D = q.shape
    Hkv = k_list[0].shape[1]
    si, sl, last = pattern
    bm = ops.attention_block_m(H, Hkv)
    if not paged:
        idx, cu, sld = _build(ops, q_lens, seq_lens, si, sl, last, bm, ops.attention_block_n(), ops.MODES[mode],
                              chunk, gate, full_lens=full_lens)
    if q_pad:
        # rows of a wider buffer (as a slice of a fused qkv projection would be)
        buf = torch.zeros((T, H * D + q_pad), dtype=dtype, device=DEV)
        buf[:, :H * D] = _to_dev(q, dtype).view(T, H * D)
        qd = buf[:, :H * D].view(T, H, D)
    else:
        qd = _to_dev(q, dtype)
    out = torch.full((T, H, D), float("nan"), dtype=out_dtype, device=DEV)
    if paged:
        nblk = sum((s + page_size - 1) // page_size for s in seq_lens) + 3
        perm = np.random.default_rng(0).permutation(nblk).astype(np.int32)
        kc = torch.zeros((nblk, page_size, Hkv, D), dtype=dtype, device=DEV)
        vc = torch.zeros_like(kc)
        maxb = max((s + page_size - 1) // page_size for s in seq_lens)
        bt = np.zeros((len(seq_lens), maxb), np.int32)
        nxt = 0
        for b, s in enumerate(seq_lens):
            for pb in range((s + page_size - 1) // page_size):
                blk = int(perm[nxt])
                nxt += 1
                bt[b, pb] = blk
                lo, hi = pb * page_size, min((pb + 1) * page_size, s)
                kc[blk, :hi - lo] = _to_dev(k_list[b][lo:hi], dtype)
                vc[blk, :hi - lo] = _to_dev(v_list[b][lo:hi], dtype)
        btd = torch.from_numpy(bt).to(DEV)
        idx, cu, sld = _build(ops, q_lens, seq_lens, si, sl, last, bm, ops.attention_block_n(), ops.MODES[mode],
                              chunk, gate, block_table=btd if resolve_pages else None, page_size=page_size,
                              full_lens=full_lens)
        if resolve_pages:
            # the resolved pages are the block-table entries of each listed tile's two halves
            tp = idx.tile_pages.cpu().numpy()
            col = idx.col_idx.cpu().numpy()
            rp = idx.row_ptr.cpu().numpy()
            meta = idx.qb_meta.cpu().numpy()
            for r in range(len(rp) - 1):
                b_ = int(meta[r, 0])
                for e in range(rp[r], rp[r + 1]):
                    t0 = int(col[e]) * 32
                    assert tp[e, 0] == bt[b_, t0 // page_size]
                    assert tp[e, 1] == bt[b_, min(t0 + 16, seq_lens[b_] - 1) // page_size]
        ops.attention(qd, kc, vc, out, idx, cu, sld, scale, block_table=btd, bias=bias, flags=flags,
                      kv_stage=_stage_for(ops, idx, q_lens, seq_lens, Hkv) if stage else None)
    else:
        kd = _to_dev(np.concatenate(k_list, 0), dtype)
        vd = _to_dev(np.concatenate(v_list, 0), dtype)
        cu_k = torch.from_numpy(np.concatenate([[0], np.cumsum(seq_lens)]).astype(np.int32)).to(DEV)
        ops.attention(qd, kd, vd, out, idx, cu, sld, scale, cu_seqlens_k=cu_k, bias=bias, flags=flags,
                      kv_stage=_stage_for(ops, idx, q_lens, seq_lens, Hkv) if stage else None)
    torch.cuda.synchronize()
    return out.float().cpu().numpy()


def _stage_for(ops, idx, q_lens, seq_lens, Hkv, extra_rows=0):
    """A staging area for the batch whatever its lengths (the library's default stages long prompts only), full of
    NaN: every row the kernel reads must have been written by the gather of the same call."""
    cu_h = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    idx.stage_rows = ops.dense_stage_rows(cu_h, np.asarray(seq_lens, np.int32), 0)
    st = ops.alloc_kv_stage(idx.stage_rows + extra_rows, Hkv, DEV)
    if st is not None:
        st.fill_(float("nan"))
    return st


def _rand_case(seed, q_lens, seq_lens, H, Hkv, D):
    T = int(sum(q_lens))
    q = orc.seeded_normal(seed, (T, H, D))
    ks = [orc.seeded_normal(seed + 100 + b, (s, Hkv, D)) for b, s in enumerate(seq_lens)]
    vs = [orc.seeded_normal(seed + 200 + b, (s, Hkv, D)) for b, s in enumerate(seq_lens)]
    return q, ks, vs


def test_attention_matches_golden_eager(golden_attention):
    """AscendTriangleMixAttention mirror vs the reference's own outputs."""
    from vllm_trianglemix_amd import AscendTriangleMixAttention, TriangleMixConfig, TriangleMixMaskGenerator
    meta, arrays = golden_attention
    for c in meta["cases"]:
        if c.get("run_dtype"):
            continue
        shape = (c["B"], c["N"], c["H"], c["D"])
        q, k, v = (orc.seeded_normal(c["seed"] * 3 + t, shape) for t in range(3))
        assert hashlib.sha256(q.tobytes() + k.tobytes() + v.tobytes()).hexdigest() == c["inputs_sha256"]
        si, sl, last = c["cfg"]
        cfg = TriangleMixConfig(si, sl, last, num_triangle_layers=c.get("ntl"))
        qd, kd, vd = (_to_dev(x, torch.bfloat16) for x in (q, k, v))
        if c["mask"] == "helper":
            mod = AscendTriangleMixAttention(c["H"], c["D"], c["scale"], cfg, layer_idx=c["layer"],
                                             num_layers=c["num_layers"])
            out = mod(qd, kd, vd)
        else:
            gen = TriangleMixMaskGenerator(cfg)
            am = gen.get_attention_mask(c["N"], 0, DEV, torch.bfloat16, use_triangle=(c["mask"] == "triangle"))
            mod = AscendTriangleMixAttention(c["H"], c["D"], c["scale"])
            out = mod(qd, kd, vd, attn_mask=am)  # the explicit-mask route: additive bias, no index pattern
        assert out.shape == shape and out.is_contiguous() and out.dtype == torch.bfloat16
        ref = arrays[c["name"]]
        got = out.float().cpu().numpy()
        # bf16 output: 1e-3 plus half an output ulp (2^-8 relative)
        assert_close(got, ref, "eager mirror, bf16 inputs (reference's own outputs)", out16=True, info=c["name"])


def test_attention_matches_reference_config0():
    """The reference's own CPU-runnable configuration (2048 tokens, 16 heads, head dim 128, defaults
    4/32/64) against the sampled rows of the reference's fp32 output: once through the index path
    (pattern in the kernel) and once through the explicit-mask route of the eager mirror."""
    import json, os
    from vllm_trianglemix_amd import AscendTriangleMixAttention, TriangleMixConfig, TriangleMixMaskGenerator
    ops = _ops()
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(gold, "attention_config0.json")))
    z = np.load(os.path.join(gold, "attention_config0.npz"))
    rows, want = z["rows"], z["out_rows"]
    N, H, D, seed = meta["N"], meta["H"], meta["D"], meta["seed"]
    q, k, v = (orc.seeded_normal(seed * 3 + t, (1, N, H, D)) for t in range(3))
    assert hashlib.sha256(q.tobytes() + k.tobytes() + v.tobytes()).hexdigest() == meta["inputs_sha256"]
    got = _run_varlen(ops, q[0], [k[0]], [v[0]], [N], [N], meta["scale"], "triangle", tuple(meta["cfg"]))
    assert_close(got[rows], want, "reference config 0 (2048 tokens), pattern in the kernel")
    gen = TriangleMixMaskGenerator(TriangleMixConfig(*meta["cfg"]))
    am = gen.get_attention_mask(N, 0, DEV, torch.bfloat16, use_triangle=True)
    assert int((am == 0).sum().item()) == meta["mask_nnz"]
    mod = AscendTriangleMixAttention(H, D, meta["scale"])
    out = mod(*(_to_dev(x, torch.bfloat16) for x in (q, k, v)), attn_mask=am)
    got_b = out[0].float().cpu().numpy()
    assert_close(got_b[rows], want, "reference config 0 (2048 tokens), explicit mask", out16=True)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_varlen_gqa_vs_oracle(dtype):
    ops = _ops()
    cases = [
        # q_lens, seq_lens, H, Hkv, D, mode, pattern
        ([128], [128], 4, 4, 64, "triangle", (4, 8, 16)),
        ([257], [257], 8, 2, 128, "triangle", (4, 32, 64)),
        ([300, 1, 77], [300, 500, 130], 8, 2, 128, "triangle", (4, 32, 64)),
        ([64, 200], [64, 200], 4, 2, 128, "causal", (0, 0, 0)),
        ([50, 31], [50, 31], 2, 2, 64, "none", (0, 0, 0)),
        ([500], [500], 16, 8, 128, "triangle", (4, 32, 64)),
        ([100], [900], 8, 1, 128, "triangle", (4, 32, 64)),
        ([1, 1, 1], [700, 33, 1], 8, 2, 128, "triangle", (4, 32, 64)),
        ([40], [40], 3, 3, 64, "triangle", (0, 0, 0)),
        ([90], [90], 12, 2, 64, "triangle", (2, 5, 9)),
        # per-GPU head shards of the tensor-parallel configurations: 32B at TP=4 (16 / 2 heads),
        # 8B at TP=8 (4 / 1), 72B-shaped at TP=8 (8 / 1)
        ([700], [700], 16, 2, 128, "triangle", (4, 32, 64)),
        ([130, 40], [2300, 40], 4, 1, 128, "triangle", (4, 32, 64)),
        ([333], [333], 8, 1, 128, "triangle", (4, 32, 64)),
    ]
    for ci, (ql, sl, H, Hkv, D, mode, pat) in enumerate(cases):
        q, ks, vs = _rand_case(1000 + ci, ql, sl, H, Hkv, D)
        if dtype == torch.float16:
            q = q.astype(np.float16).astype(np.float32)
            ks = [x.astype(np.float16).astype(np.float32) for x in ks]
            vs = [x.astype(np.float16).astype(np.float32) for x in vs]
        scale = D ** -0.5
        ref = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], np.concatenate([[0], np.cumsum(ql)]),
                                       sl, scale, mode, *pat)
        for paged, page_size, resolve in ((False, 0, False), (True, 16, True), (True, 64, ci % 2 == 0)):
            got = _run_varlen(_ops(), q, ks, vs, ql, sl, scale, mode, pat, dtype=dtype, paged=paged,
                              page_size=page_size, chunk=(4 if ci % 2 else 64), q_pad=(64 if ci % 3 == 0 else 0),
                              resolve_pages=resolve)
            assert_close(got, ref, f"work list + band, small ragged batches, {str(dtype)[6:]}", info=(ci, paged, page_size))


def test_attention_decode_heavy_batch():
    """A serving-shaped step: 40 decode rows (one query each, contexts of 1 .. 3000 tokens, the
    dense-causal last row of the pattern), two chunked-prefill continuations and one fresh prefill,
    paged K/V with 16-token pages; every request is checked against the oracle."""
    ops = _ops()
    rng = np.random.default_rng(11)
    H, Hkv, D = 8, 2, 128
    seq_lens = [int(x) for x in rng.integers(1, 3000, size=40)] + [2600, 900, 350]
    q_lens = [1] * 40 + [37, 64, 350]
    q, ks, vs = _rand_case(9090, q_lens, seq_lens, H, Hkv, D)
    scale = D ** -0.5
    cu = np.concatenate([[0], np.cumsum(q_lens)])
    ref = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], cu, seq_lens, scale, "triangle", 4, 32, 64)
    for chunk, gate in ((32, 0), (8, 2048)):
        if gate:  # requests of at most 2048 tokens fall back to plain causal attention
            ref_g = ref.copy()
            for b, (ql, n) in enumerate(zip(q_lens, seq_lens)):
                if n <= gate:
                    ref_g[cu[b]:cu[b + 1]] = orc.varlen_gqa_attention(q[cu[b]:cu[b + 1]], lambda _: ks[b], lambda _: vs[b],
                                                                      [0, ql], [n], scale, "causal", 0, 0, 0)
        want = ref_g if gate else ref
        # the short-queries hint only moves the K/V fetch to other waves: same results with and without it,
        # also on this batch, which does not qualify for it
        for flags in (0, TMX_ATTN_SHORT_QUERIES):
            got = _run_varlen(ops, q, ks, vs, q_lens, seq_lens, scale, "triangle", (4, 32, 64), paged=True,
                              page_size=16, chunk=chunk, gate=gate, flags=flags)
            assert_close(got, want, "decode-heavy mixed step", info=(chunk, gate, flags))


def test_attention_split_and_combine_long():
    """One long prompt: trailing rows are split into many pieces and merged (130 tiles per row: 44 / 17 / 3 / 1 pieces;
    the merge pass takes 16 pieces per round, so the first two cases go through its running maximum)."""
    ops = _ops()
    n, H, Hkv, D = 4096 + 37, 8, 2, 128
    q, ks, vs = _rand_case(77, [n], [n], H, Hkv, D)
    scale = D ** -0.5
    ref = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], [0, n], [n], scale, "triangle", 4, 32, 64)
    for chunk in (3, 8, 64, 1000):
        got = _run_varlen(ops, q, ks, vs, [n], [n], scale, "triangle", (4, 32, 64), paged=True, chunk=chunk)
        assert_close(got, ref, "split q-blocks + merge pass", info=chunk)
    gb = _run_varlen(ops, q, ks, vs, [n], [n], scale, "triangle", (4, 32, 64), paged=True, out_dtype=torch.bfloat16)
    assert_close(gb, ref, "split q-blocks + merge pass", out16=True)
    refc = orc.varlen_gqa_attention(q[:, :2], lambda b: ks[b][:, :1], lambda b: vs[b][:, :1], [0, n], [n], scale,
                                    "causal", 0, 0, 0)
    gotc = _run_varlen(ops, q[:, :2].copy(), [ks[0][:, :1].copy()], [vs[0][:, :1].copy()], [n], [n], scale, "causal",
                       (0, 0, 0), chunk=16)
    assert_close(gotc, refc, "dense causal kernel (4k prompt)")


def test_chunked_prefill_equals_whole_prefill():
    """A prompt that arrives in steps attends exactly as in one step when the steps know the prompt's
    length: the trailing dense rows and the length gate refer to the whole prompt, not to the end of a
    chunk.  Also a mixed step: a middle chunk, a decode row past its prompt, a last chunk."""
    ops = _ops()
    P, H, Hkv, D = 2500, 8, 2, 128
    q, ks, vs = _rand_case(4242, [P], [P], H, Hkv, D)
    scale = D ** -0.5
    pat = (4, 32, 64)
    ref = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], [0, P], [P], scale, "triangle", *pat,
                                   tri_min_len=2048)
    whole = _run_varlen(ops, q, ks, vs, [P], [P], scale, "triangle", pat, paged=True, gate=2048)
    assert_close(whole, ref, "chunked prefill")
    bounds = [0, 1000, 2040, 2467, P]  # the last-but-one chunk ends inside the dense trailing rows
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        got = _run_varlen(ops, q[lo:hi].copy(), [ks[0][:hi]], [vs[0][:hi]], [hi - lo], [hi], scale, "triangle", pat,
                          paged=True, gate=2048, full_lens=[P])
        assert_close(got, ref[lo:hi], "chunked prefill", info=(lo, hi))
        want = orc.varlen_gqa_attention(q[lo:hi], lambda b: ks[0][:hi], lambda b: vs[0][:hi], [0, hi - lo], [hi], scale,
                                        "triangle", *pat, full_lens=[P], tri_min_len=2048)
        assert np.abs(want - ref[lo:hi]).max() < 1e-12  # the oracle itself: a chunk is a block of the whole
        if hi < P - 64:
            # without the prompt length the end of the chunk is taken for the end of the sequence
            # (rows near it attend densely), or the chunk falls under the length gate: a different result
            naive = _run_varlen(ops, q[lo:hi].copy(), [ks[0][:hi]], [vs[0][:hi]], [hi - lo], [hi], scale, "triangle",
                                pat, paged=True, gate=2048)
            assert np.abs(naive - ref[lo:hi]).max() > 10 * ATOL_F32_OUT
    # mixed step: request 0 = middle chunk of a 2500-token prompt, request 1 = decode row 40 tokens past a
    # 2100-token prompt (dense, as every decode row), request 2 = last chunk of a short prompt (gate: causal)
    q_lens, seq_lens, full = [300, 1, 200], [1500, 2140, 700], [2500, 2100, 700]
    q2, ks2, vs2 = _rand_case(777, q_lens, seq_lens, H, Hkv, D)
    cu = np.concatenate([[0], np.cumsum(q_lens)])
    want = orc.varlen_gqa_attention(q2, lambda b: ks2[b], lambda b: vs2[b], cu, seq_lens, scale, "triangle", *pat,
                                    full_lens=full, tri_min_len=2048)
    for paged in (True, False):
        got = _run_varlen(ops, q2, ks2, vs2, q_lens, seq_lens, scale, "triangle", pat, paged=paged, gate=2048,
                          full_lens=full)
        assert_close(got, want, "chunked prefill", info=paged)
    dense_row = orc.varlen_gqa_attention(q2[300:301], lambda b: ks2[1], lambda b: vs2[1], [0, 1], [2140], scale, "causal",
                                         0, 0, 0)
    assert np.abs(want[300:301] - dense_row).max() < 1e-12


def test_attention_bias_and_fully_masked_rows():
    ops = _ops()
    n, H, D = 96, 2, 64
    q, ks, vs = _rand_case(5, [n], [n], H, H, D)
    scale = D ** -0.5
    rng = np.random.default_rng(9)
    bias = orc.bf16_round(rng.standard_normal((n, n)).astype(np.float32))
    bias[rng.random((n, n)) < 0.3] = -np.inf
    bias[7, :] = -np.inf  # a fully masked row: softmax over all -inf is NaN in the reference
    ref = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], [0, n], [n], scale, "none", 0, 0, 0, bias=bias)
    for bdt in (torch.float32, torch.bfloat16):
        got = _run_varlen(ops, q, ks, vs, [n], [n], scale, "none", (0, 0, 0),
                          bias=torch.from_numpy(bias).to(DEV).to(bdt))
        assert np.isnan(got[7]).all() and np.isnan(ref[7]).all()
        ok = ~np.isnan(ref)
        assert_close(got[ok], ref[ok], "work list with additive bias")
    # bias on top of the triangle pattern (attn_mask + tri_mask in the reference)
    bias2 = orc.bf16_round(rng.standard_normal((n, n)).astype(np.float32))
    ref2 = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], [0, n], [n], scale, "triangle", 4, 8, 16,
                                    bias=bias2)
    got2 = _run_varlen(ops, q, ks, vs, [n], [n], scale, "triangle", (4, 8, 16), bias=torch.from_numpy(bias2).to(DEV))
    assert_close(got2, ref2, "work list with additive bias")


def test_empty_inputs():
    """A request without query rows between two others (its keys are never touched), and a step with no
    request at all: the index is empty and the attention call is a no-op that leaves the output alone."""
    ops = _ops()
    H, Hkv, D = 8, 2, 128
    q_lens, seq_lens = [40, 0, 70], [40, 25, 2100]
    q, ks, vs = _rand_case(99, q_lens, seq_lens, H, Hkv, D)
    scale = D ** -0.5
    for paged in (False, True):
        got = _run_varlen(ops, q, ks, vs, q_lens, seq_lens, scale, "triangle", (4, 32, 64), paged=paged, gate=2048)
        ref = np.zeros_like(got, dtype=np.float64)
        cu = np.concatenate([[0], np.cumsum(q_lens)])
        for b, (ql, s) in enumerate(zip(q_lens, seq_lens)):
            if ql:
                ref[cu[b]:cu[b + 1]] = orc.varlen_gqa_attention(q[cu[b]:cu[b + 1]], lambda _: ks[b], lambda _: vs[b], [0, ql], [s],
                                                                scale, "triangle" if s > 2048 else "causal", 4, 32, 64)
        assert_close(got, ref, "batch with an empty request")
    # no request at all
    cu_h, sl_h = np.zeros((1,), np.int32), np.zeros((0,), np.int32)
    cu, sld = torch.from_numpy(cu_h).to(DEV), torch.from_numpy(sl_h).to(DEV)
    idx = ops.build_block_index(cu, sld, cu_h, sl_h, 4, 32, 64, ops.attention_block_m(H, Hkv), 32, ops.MODES["triangle"])
    torch.cuda.synchronize()
    assert idx.num_qblocks == 0 and idx.nnz_tiles == 0 and int(idx.totals[0]) == 0
    qe = torch.zeros((0, H, D), dtype=torch.bfloat16, device=DEV)
    ke = torch.zeros((0, Hkv, D), dtype=torch.bfloat16, device=DEV)
    oe = torch.zeros((0, H, D), dtype=torch.bfloat16, device=DEV)
    ops.attention(qe, ke, ke, oe, idx, cu, sld, scale, cu_seqlens_k=cu)
    torch.cuda.synchronize()


@pytest.mark.parametrize("n,paged,H,Hkv,cache,mode", [
    (8192, False, 32, 8, "bf16", "triangle"),   # Qwen3-8B, 8k prompt
    (32768, True, 32, 8, "bf16", "triangle"),   # Qwen3-8B, 32k prompt (the benchmark's workload)
    (32768, True, 16, 2, "bf16", "triangle"),   # Qwen3-32B, one of 4 head shards
    (65536, True, 8, 1, "fp8", "triangle"),     # 64-head / 8-kv-head model, one of 8 head shards: fp8 cache READ IN PLACE
    (65536, True, 8, 1, "fp8-gather", "triangle"),  # the same through the gather + descale pass and contiguous rows
    (131072, True, 8, 2, "bf16", "triangle"),   # a 128k-token prompt (the index is then built by two launches)
    (32768, True, 32, 8, "bf16", "causal"),     # Qwen3-8B, 32k prompt on a layer outside the triangle set: dense kernel
    (8192, True, 8, 2, "fp8", "causal"),        # dense kernel out of an fp8 cache
    (131072, True, 8, 2, "bf16", "causal"),     # a 128k-token prompt on a causal layer: staged dense kernel, 2,048 tiles per row block
])
def test_attention_properties_at_full_size(n, paged, H, Hkv, cache, mode):
    """Size-independent properties at the sizes and head shards of the benchmark configurations (N = 8192 contiguous K/V; the others
    with the paged cache and shuffled block table of the benchmark): the output is linear in V, a constant V
    returns that constant, and sampled rows (first rows, a middle row, rows around the start of the dense trailing
    region, the last row) equal the fp64 oracle.  Triangle launches run on the band + work-list workgroups,
    causal launches on the dense kernel.  fp8 caches: "fp8" is the route the backend takes (the attention kernels
    read the fp8 bytes in place and descale inside, k_descale / v_descale of rocm_attn.py:356-365, 421-422),
    "fp8-gather" the gather + descale pass followed by the attention over contiguous rows; linearity in V is
    checked through the descale factor (doubling v_scale doubles the output exactly) and the oracle attends to the
    rows the gather returns, themselves checked against the oracle's fp8 conversion on sampled rows."""
    ops = _ops()
    D, page, G = 128, 16, H // Hkv
    fp8 = cache.startswith("fp8")
    in_place = cache == "fp8"
    g = torch.Generator(device="cpu").manual_seed(0)
    q = torch.randn((n, H, D), generator=g).to(DEV).bfloat16()
    k = torch.randn((n, Hkv, D), generator=g).to(DEV).bfloat16()
    v1 = torch.randn((n, Hkv, D), generator=g).to(DEV).bfloat16()
    v2 = torch.randn((n, Hkv, D), generator=g).to(DEV).bfloat16()
    cu_h, sl_h = np.asarray([0, n], np.int32), np.asarray([n], np.int32)
    cu, sld = torch.from_numpy(cu_h).to(DEV), torch.from_numpy(sl_h).to(DEV)
    bt = None
    if paged:
        nblk = n // page
        perm = torch.randperm(nblk, generator=g).to(torch.int32)
        bt = perm.view(1, -1).to(DEV)
        pos = torch.arange(n)
        slots = (perm.long()[pos // page] * page + pos % page).to(DEV)
    in_kernel_pages = paged and cache != "fp8-gather"
    pat = (4, 32, 64) if mode == "triangle" else (0, 0, 0)
    idx = ops.build_block_index(cu, sld, cu_h, sl_h, *pat, ops.attention_block_m(H, Hkv), 32, ops.MODES[mode],
                                chunk_tiles=32 if mode == "triangle" else 64,
                                block_table=bt if in_kernel_pages else None, page_size=page if in_kernel_pages else 0,
                                leave_dense=mode == "causal")  # (what the backend builds for a causal layer: no tile lists)
    # work areas of the calls below; a causal launch of this length gets the dense kernel's staging area (the library's
    # own rule) and runs the staged form: K/V rows gathered first, tiles by LDS-DMA
    ws = ops.alloc_workspace(idx, H, D, DEV, num_kv_heads=Hkv)
    assert len(ws) == (3 if mode == "causal" and cache != "fp8-gather" else 2)
    outs = []
    seen_k = seen_v = None
    if fp8:
        k_scale, v_scale = 0.75, 1.5
        kc = torch.zeros((n // page, page, Hkv, D), dtype=torch.uint8, device=DEV)
        vc = torch.zeros_like(kc)

        def attend(vs_):
            o = torch.full((n, H, D), float("nan"), dtype=torch.float32, device=DEV)
            if in_place:
                ops.attention(q, kc.view(torch.float8_e4m3fn), vc.view(torch.float8_e4m3fn), o, idx, cu, sld, D ** -0.5,
                              block_table=bt, k_scale=k_scale, v_scale=vs_, workspace=ws)
                return o, None
            rows = ops.gather_dequant_kv_fp8(kc, vc, bt, sld, cu, n, n, torch.bfloat16, k_scale, vs_)
            ops.attention(q, rows[0], rows[1], o, idx, cu, sld, D ** -0.5, cu_seqlens_k=cu)
            return o, rows

        ops.reshape_and_cache_fp8(k, v1, kc, vc, slots, k_scale, v_scale)
        # what the cache holds, as 16-bit rows: fp8 x scale rounded once (the gather pass; checked below)
        seen_k, seen_v = ops.gather_dequant_kv_fp8(kc, vc, bt, sld, cu, n, n, torch.bfloat16, k_scale, v_scale)
        for vs_ in (v_scale, 2 * v_scale):
            outs.append(attend(vs_)[0])
        assert torch.equal(outs[1], 2 * outs[0])
        if len(ws) == 3:
            # the register form of the dense kernel (no staging area): the same bits
            o = torch.full((n, H, D), float("nan"), dtype=torch.float32, device=DEV)
            ops.attention(q, kc.view(torch.float8_e4m3fn), vc.view(torch.float8_e4m3fn), o, idx, cu, sld, D ** -0.5,
                          block_table=bt, k_scale=k_scale, v_scale=v_scale, workspace=ws[:2])
            assert torch.equal(o, outs[0])
        ops.reshape_and_cache_fp8(k, torch.ones_like(v1), kc, vc, slots, k_scale, 1.0)
        o, _ = attend(1.0)
        assert (o - 1.0).abs().max().item() < (5e-4 if mode == "causal" else 1e-5)
        # the rows the attention saw are the oracle's conversion of the written rows
        for r in (0, 1, n // 3, n - 1):
            want = orc.bf16_round(orc.fp8_cache_roundtrip(k[r].float().cpu().numpy(), k_scale).astype(np.float32))
            assert np.array_equal(seen_k[r].float().cpu().numpy(), want)
            want = orc.bf16_round(orc.fp8_cache_roundtrip(v1[r].float().cpu().numpy(), v_scale).astype(np.float32))
            assert np.array_equal(seen_v[r].float().cpu().numpy(), want)
    else:
        for v in (v1, v2, (v1.float() + v2.float()).bfloat16(), torch.ones_like(v1)):
            o = torch.full((n, H, D), float("nan"), dtype=torch.float32, device=DEV)
            if paged:
                kc = torch.zeros((n // page, page, Hkv, D), dtype=torch.bfloat16, device=DEV)
                vc = torch.zeros_like(kc)
                ops.reshape_and_cache(k, v, kc, vc, slots)
                ops.attention(q, kc, vc, o, idx, cu, sld, D ** -0.5, block_table=bt, workspace=ws)
                if len(ws) == 3 and not outs:
                    # the register form of the dense kernel (no staging area): the same bits
                    o2 = torch.full_like(o, float("nan"))
                    ops.attention(q, kc, vc, o2, idx, cu, sld, D ** -0.5, block_table=bt, workspace=ws[:2])
                    assert torch.equal(o2, o)
            else:
                ops.attention(q, k, v, o, idx, cu, sld, D ** -0.5, cu_seqlens_k=cu, workspace=ws)
            outs.append(o)
        lin = (outs[0] + outs[1] - outs[2]).abs().max().item()
        assert lin < 0.05  # v1 + v2 is re-rounded to bf16: 2^-8 relative on values up to ~8
        # fp32 sums of up to n products against an fp32 sum of the weights: 4e-6 at 32k keys, 1.1e-5 at 128k
        # (the dense causal kernel adds up the unrounded probabilities on the vector units and multiplies fp16-rounded
        # ones with V: numerator and denominator differ by the rounding, 11 bits, averaged over the row)
        const_bar = 5e-4 if mode == "causal" else (1e-5 if n <= 65536 else 3e-5)
        const_err = (outs[3] - 1.0).abs().max().item()
        print(f"[parity] constant V through {n} keys ({mode}): max |out - 1| {const_err:.2e}")
        assert const_err < const_bar
        seen_k, seen_v = k, v1
    assert not torch.isnan(outs[0]).any()
    rows = [0, 5, 40, 1000, n // 2 + 17, n - 1 - 70, n - 64, n - 63, n - 10, n - 1]
    heads = sorted({0, (13 * H) // 32, H // 2 + 1, H - 1})  # (8 / 1 heads: both head chunks of the K/V head)
    qn, kn, vn = q.float().cpu().numpy(), seen_k.float().cpu().numpy(), seen_v.float().cpu().numpy()
    got = outs[0].cpu().numpy()
    want = np.zeros((len(rows), len(heads), D))
    for a, i in enumerate(rows):
        allow = np.zeros((n,), bool)
        if mode == "triangle":
            for lo, hi in orc.row_intervals(i, n, 4, 32, 64):
                allow[lo:hi + 1] = True
        else:
            allow[:i + 1] = True
        for b, h in enumerate(heads):
            s = (kn[:, h // G, :].astype(np.float64) @ qn[i, h].astype(np.float64)) * D ** -0.5
            s[~allow] = -np.inf
            pr = np.exp(s - s.max())
            pr /= pr.sum()
            want[a, b] = pr @ vn[:, h // G, :].astype(np.float64)
    kernel = "dense causal kernel" if mode == "causal" else "band + work list"
    assert_close(got[rows][:, heads], want, f"full size {n} x {H}/{Hkv} heads, {cache} cache: {kernel}")


@pytest.mark.parametrize("H,Hkv,world", [(16, 4, 2), (16, 2, 4)])  # K/V heads partitioned; K/V heads shared by 2 ranks
def test_head_shards_through_the_output_projection(H, Hkv, world):
    """What the ranks of a tensor-parallel group compute, on one GPU one rank after the other: every rank's
    head shard (qwen3.py:85-98) runs the HIP attention on its slice of Q and its K/V heads and multiplies with
    its slice of o_proj's input dimension (RowParallelOutputProjection); the sum over ranks -- what the
    all-reduce forms -- equals the unsharded attention through the whole projection, and the concatenated
    shard outputs equal the unsharded output bit for bit (heads are independent)."""
    from vllm_trianglemix_amd.parallel import RowParallelOutputProjection, shard_heads
    ops = _ops()
    D, hidden = 128, 256
    q_lens, seq_lens = [2300, 1, 40], [2300, 700, 2100]
    g = torch.Generator(device="cpu").manual_seed(5)
    T, S = sum(q_lens), sum(seq_lens)
    q = torch.randn((T, H, D), generator=g).to(DEV).bfloat16()
    k = torch.randn((S, Hkv, D), generator=g).to(DEV).bfloat16()
    v = torch.randn((S, Hkv, D), generator=g).to(DEV).bfloat16()
    w = (torch.randn((hidden, H * D), generator=g) * 0.05).to(DEV).float()
    cu_k = torch.from_numpy(np.concatenate([[0], np.cumsum(seq_lens)]).astype(np.int32)).to(DEV)

    def attend(qs, ks, vs):
        h, hkv = qs.shape[1], ks.shape[1]
        idx, cu, sld = _build(ops, q_lens, seq_lens, 4, 32, 64, ops.attention_block_m(h, hkv), 32, ops.MODES["triangle"], 16, 2048)
        out = torch.full(qs.shape, float("nan"), dtype=torch.float32, device=DEV)
        ops.attention(qs.contiguous(), ks.contiguous(), vs.contiguous(), out, idx, cu, sld, D ** -0.5, cu_seqlens_k=cu_k)
        return out

    whole = attend(q, k, v)
    assert not torch.isnan(whole).any()
    want = RowParallelOutputProjection(w)(whole)
    total = torch.zeros_like(want)
    for rank in range(world):
        sh = shard_heads(H, Hkv, rank, world)
        part = attend(q[:, sh.head_slice], k[:, sh.kv_head_slice], v[:, sh.kv_head_slice])
        # same keys, same order of summation per head: the shard's rows are the whole launch's rows
        assert torch.equal(part, whole[:, sh.head_slice]), rank
        w_shard = w[:, sh.head_start * D:(sh.head_start + sh.num_heads) * D]
        total += RowParallelOutputProjection(w_shard)(part)
    assert (total - want).abs().max().item() <= 1e-3 * max(1.0, want.abs().max().item())


def test_chunked_output_projection_on_the_gpu():
    """RowParallelOutputProjection with chunk_tokens > 0 on device tensors: chunks of the token dimension go through
    GEMM + all-reduce in turn, the all-reduce of one chunk on a second HIP stream beside the GEMM of the next
    (linear.py:1358-1359 is the reference's single all-reduce).  On one GPU the group has one rank (RCCL, world size
    1; ``always_reduce`` keeps the pipeline running although there is nothing to sum): the result must equal the
    unchunked product, and what follows on the caller's stream must see all of it."""
    import socket
    import torch.distributed as dist
    from vllm_trianglemix_amd.parallel import RowParallelOutputProjection
    T, H, D, hidden = 5000, 8, 128, 1024  # chunks of 1024 tokens: four whole ones and a ragged last one
    g = torch.Generator(device="cpu").manual_seed(12)
    x = torch.randn((T, H, D), generator=g).to(DEV).bfloat16()
    w = (torch.randn((hidden, H * D), generator=g) * 0.03).to(DEV).bfloat16()
    whole = RowParallelOutputProjection(w)(x)  # no group: the plain product
    started = False
    if not dist.is_initialized():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device(DEV))
        started = True
    try:
        proj = RowParallelOutputProjection(w, chunk_tokens=1024, always_reduce=True)
        got = proj(x)
        after = got.float().sum()  # queued on the caller's stream straight behind the call
        one = RowParallelOutputProjection(w, chunk_tokens=0, always_reduce=True)(x)  # one GEMM, one all-reduce
        torch.cuda.synchronize()
        assert proj._comm_stream is not None
        for name, y in (("chunked", got), ("single all-reduce", one)):
            err = (y.float() - whole.float()).abs().max().item()
            # (a GEMM over fewer rows may sum in another order: bf16 outputs one ulp apart)
            assert err <= 2.0 ** -7 * max(1.0, whole.float().abs().max().item()), (name, err)
        assert abs(after.item() - got.float().sum().item()) <= 1e-3 * abs(got.float().sum().item()) + 1e-3
    finally:
        if started:
            dist.destroy_process_group()


def test_reshape_and_cache_and_errors():
    ops = _ops()
    from vllm_trianglemix_amd._lib import TmxError
    T, Hkv, D, ps, nb = 37, 2, 128, 16, 9
    key = torch.randn((T, Hkv, D), device=DEV).bfloat16()
    val = torch.randn((T, Hkv, D), device=DEV).bfloat16()
    kc = torch.zeros((nb, ps, Hkv, D), dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    slots = torch.randperm(nb * ps)[:T].to(torch.int64)
    slots[3] = -1
    ops.reshape_and_cache(key, val, kc, vc, slots.to(DEV))
    kf, vf = kc.view(nb * ps, Hkv, D), vc.view(nb * ps, Hkv, D)
    for t in range(T):
        if slots[t] < 0:
            continue
        assert torch.equal(kf[slots[t]], key[t]) and torch.equal(vf[slots[t]], val[t])
    assert kc.float().abs().sum().item() > 0
    written = torch.zeros(nb * ps, dtype=torch.bool)
    written[slots[slots >= 0]] = True
    assert kf[~written.to(DEV)].abs().sum().item() == 0
    # loud failures: CPU tensors, wrong index geometry
    with pytest.raises(RuntimeError):
        ops.dense_mask(8, 4, 32, 64, "triangle", "cpu", torch.float32)
    idx, cu, sld = _build(ops, [64], [64], 4, 32, 64, 64, 64, 2)
    q = torch.zeros((64, 4, 128), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(TmxError):
        ops.attention(q, q, q, torch.empty_like(q), idx, cu, sld, 1.0, cu_seqlens_k=cu)


# ------------------------------------------------------------------ vLLM backend surface
def test_backend_forward_mixed_batch():
    """TriangleMixAttentionImpl.forward on a mixed step: a long prefill (triangle), a short prefill
    (plain causal below the 2048-token gate), a chunked-prefill continuation and a decode row; K/V
    of the step are written through slot_mapping, earlier context is already in the paged cache."""
    import types
    from vllm_trianglemix_amd import TriangleMixConfig
    from vllm_trianglemix_amd.backend import TriangleMixAttentionBackend, TriangleMixMetadataBuilder
    H, Hkv, D, page = 8, 2, 128, 16
    q_lens, seq_lens = [2300, 100, 64, 1], [2300, 100, 2500, 2100]
    q, ks, vs = _rand_case(4242, q_lens, seq_lens, H, Hkv, D)
    scale = D ** -0.5
    cu = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    T = int(cu[-1])
    nblk = sum((s + page - 1) // page for s in seq_lens) + 5
    perm = np.random.default_rng(3).permutation(nblk).astype(np.int32)
    shape = TriangleMixAttentionBackend.get_kv_cache_shape(nblk, page, Hkv, D)
    kv_cache = torch.zeros(shape, dtype=torch.bfloat16, device=DEV)
    maxb = max((s + page - 1) // page for s in seq_lens)
    bt = np.zeros((len(seq_lens), maxb), np.int32)
    slots = np.zeros((T,), np.int64)
    k_new, v_new = np.zeros((T, Hkv, D), np.float32), np.zeros((T, Hkv, D), np.float32)
    nxt = 0
    for b, (ql, s) in enumerate(zip(q_lens, seq_lens)):
        for pb in range((s + page - 1) // page):
            bt[b, pb] = perm[nxt]
            nxt += 1
        ctx = s - ql
        if ctx:  # context already cached
            jj = np.arange(ctx)
            blk = torch.from_numpy(bt[b, jj // page].astype(np.int64)).to(DEV)
            off = torch.from_numpy(jj % page).to(DEV)
            kv_cache[0, blk, off] = _to_dev(ks[b][:ctx], torch.bfloat16)
            kv_cache[1, blk, off] = _to_dev(vs[b][:ctx], torch.bfloat16)
        pos = np.arange(ctx, s)
        slots[cu[b]:cu[b + 1]] = bt[b, pos // page].astype(np.int64) * page + pos % page
        k_new[cu[b]:cu[b + 1]], v_new[cu[b]:cu[b + 1]] = ks[b][ctx:], vs[b][ctx:]
    pad = 5  # padded tokens beyond num_actual_tokens must be left alone
    common = types.SimpleNamespace(
        num_actual_tokens=T, max_query_len=max(q_lens), max_seq_len=max(seq_lens),
        query_start_loc=torch.from_numpy(cu).to(DEV), query_start_loc_cpu=torch.from_numpy(cu),
        seq_lens=torch.tensor(seq_lens, dtype=torch.int32, device=DEV),
        seq_lens_cpu=torch.tensor(seq_lens, dtype=torch.int32),
        block_table_tensor=torch.from_numpy(bt).to(DEV), slot_mapping=torch.from_numpy(slots).to(DEV))
    builder = TriangleMixAttentionBackend.get_builder_cls()(types.SimpleNamespace(block_size=page), ["l0", "l1"],
                                                            None, torch.device(DEV))
    assert isinstance(builder, TriangleMixMetadataBuilder)
    md = builder.build(0, common)
    cfg = TriangleMixConfig(num_triangle_layers=1)
    impl_cls = TriangleMixAttentionBackend.get_impl_cls()
    qd = torch.zeros((T + pad, H, D), dtype=torch.bfloat16, device=DEV)
    qd[:T] = _to_dev(q, torch.bfloat16)
    kd = torch.zeros((T + pad, Hkv, D), dtype=torch.bfloat16, device=DEV)
    vd = torch.zeros_like(kd)
    kd[:T], vd[:T] = _to_dev(k_new, torch.bfloat16), _to_dev(v_new, torch.bfloat16)
    for layer_idx, tri in ((0, True), (1, False)):
        impl = impl_cls(H, D, scale, Hkv, None, None, "auto", None, "decoder", None, trianglemix_config=cfg,
                        layer_idx=layer_idx)
        assert impl.use_triangle is tri
        out = torch.full((T + pad, H * D), 3.0, dtype=torch.bfloat16, device=DEV)
        ret = impl.forward(None, qd, kd, vd, kv_cache, md, output=out)
        assert ret is out
        torch.cuda.synchronize()
        got = out.float().cpu().numpy().reshape(T + pad, H, D)
        assert np.all(got[T:] == 3.0)
        ref = np.zeros((T, H, D))
        for b, (ql, s) in enumerate(zip(q_lens, seq_lens)):
            mode = "triangle" if (tri and s > 2048) else "causal"
            ref[cu[b]:cu[b + 1]] = orc.varlen_gqa_attention(q[cu[b]:cu[b + 1]], lambda _: ks[b], lambda _: vs[b],
                                                            [0, ql], [s], scale, mode, 4, 32, 64)
        assert_close(got[:T], ref, "backend forward, mixed step", out16=True, info=layer_idx)
    # both layers asked for different modes: two indices were built and cached on the step's metadata
    assert len(md._indices) == 2


def test_backend_whole_prompt_step_reads_rows_beside_cache_write(monkeypatch):
    """A step of whole prompts (no request has context in the cache): the backend attends to the step's own
    K/V rows while the cache write runs on a second stream.  Same output, bit for bit, as with the write first
    and the attention over the paged cache; the cache holds the rows afterwards; K/V come as strided slices of
    one fused projection output, as the model passes them."""
    import types
    from vllm_trianglemix_amd import TriangleMixConfig
    from vllm_trianglemix_amd import backend as be
    H, Hkv, D, page = 8, 2, 128, 16
    q_lens = seq_lens = [2300, 100, 33]
    q, ks, vs = _rand_case(2024, q_lens, seq_lens, H, Hkv, D)
    scale = D ** -0.5
    cu = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    T = int(cu[-1])
    nblk = sum((s + page - 1) // page for s in seq_lens) + 3
    perm = np.random.default_rng(11).permutation(nblk).astype(np.int32)
    bt = np.zeros((len(seq_lens), max((s + page - 1) // page for s in seq_lens)), np.int32)
    slots = np.zeros((T,), np.int64)
    nxt = 0
    for b, s in enumerate(seq_lens):
        for pb in range((s + page - 1) // page):
            bt[b, pb] = perm[nxt]
            nxt += 1
        pos = np.arange(s)
        slots[cu[b]:cu[b + 1]] = bt[b, pos // page].astype(np.int64) * page + pos % page
    qkv = torch.zeros((T, (H + 2 * Hkv) * D), dtype=torch.bfloat16, device=DEV)
    qkv[:, :H * D] = _to_dev(q, torch.bfloat16).view(T, H * D)
    qkv[:, H * D:(H + Hkv) * D] = _to_dev(np.concatenate(ks), torch.bfloat16).view(T, Hkv * D)
    qkv[:, (H + Hkv) * D:] = _to_dev(np.concatenate(vs), torch.bfloat16).view(T, Hkv * D)
    qd, kd, vd = qkv.split([H * D, Hkv * D, Hkv * D], dim=-1)
    qd, kd, vd = qd.view(T, H, D), kd.view(T, Hkv, D), vd.view(T, Hkv, D)
    results = []
    for write_mode in ("serial", "stream", "fused"):
        monkeypatch.setattr(be, "_CACHE_WRITE_MODE", write_mode)
        common = types.SimpleNamespace(
            num_actual_tokens=T, max_query_len=max(q_lens), max_seq_len=max(seq_lens),
            query_start_loc=torch.from_numpy(cu).to(DEV), query_start_loc_cpu=torch.from_numpy(cu),
            seq_lens=torch.tensor(seq_lens, dtype=torch.int32, device=DEV),
            seq_lens_cpu=torch.tensor(seq_lens, dtype=torch.int32),
            block_table_tensor=torch.from_numpy(bt).to(DEV), slot_mapping=torch.from_numpy(slots).to(DEV))
        md = be.TriangleMixAttentionBackend.get_builder_cls()(types.SimpleNamespace(block_size=page), ["l0"], None,
                                                              torch.device(DEV)).build(0, common)
        assert md.whole_prompts
        impl = be.TriangleMixAttentionBackend.get_impl_cls()(
            H, D, scale, Hkv, None, None, "auto", None, "decoder", None,
            trianglemix_config=TriangleMixConfig(num_triangle_layers=1), layer_idx=0)
        kv_cache = torch.zeros(be.TriangleMixAttentionBackend.get_kv_cache_shape(nblk, page, Hkv, D),
                               dtype=torch.bfloat16, device=DEV)
        out = torch.zeros((T, H * D), dtype=torch.bfloat16, device=DEV)
        impl.forward(None, qd, kd, vd, kv_cache, md, output=out)
        # what follows on the caller's stream sees the finished cache: read it on that stream, no device-wide wait
        flat = kv_cache.view(2, nblk * page, Hkv, D)[:, torch.from_numpy(slots).to(DEV)].clone()
        torch.cuda.current_stream().synchronize()
        assert torch.equal(flat[0], kd) and torch.equal(flat[1], vd)
        results.append((out.clone(), kv_cache.clone()))
    for other in results[1:]:
        assert torch.equal(results[0][0], other[0]) and torch.equal(results[0][1], other[1])
    got = results[0][0].float().cpu().numpy().reshape(T, H, D)
    ref = np.zeros((T, H, D))
    for b, s in enumerate(seq_lens):
        ref[cu[b]:cu[b + 1]] = orc.varlen_gqa_attention(q[cu[b]:cu[b + 1]], lambda _: ks[b], lambda _: vs[b], [0, s], [s],
                                                        scale, "triangle" if s > 2048 else "causal", 4, 32, 64)
    assert_close(got, ref, "backend forward, whole prompts (fused cache write)", out16=True)
    # the next step of the same requests, one new token each, launched straight after on the caller's stream:
    # it reads the cache the second stream wrote (the prompts of the last pass above, written beside the attention)
    B = len(seq_lens)
    q2, k2s, v2s = _rand_case(2025, [1] * B, [1] * B, H, Hkv, D)
    seq2 = [s + 1 for s in seq_lens]
    slots2 = np.asarray([int(bt[b, seq_lens[b] // page]) * page + seq_lens[b] % page for b in range(B)], np.int64)
    cu2 = np.arange(B + 1, dtype=np.int32)
    common2 = types.SimpleNamespace(
        num_actual_tokens=B, max_query_len=1, max_seq_len=max(seq2),
        query_start_loc=torch.from_numpy(cu2).to(DEV), query_start_loc_cpu=torch.from_numpy(cu2),
        seq_lens=torch.tensor(seq2, dtype=torch.int32, device=DEV), seq_lens_cpu=torch.tensor(seq2, dtype=torch.int32),
        block_table_tensor=torch.from_numpy(bt).to(DEV), slot_mapping=torch.from_numpy(slots2).to(DEV))
    md2 = be.TriangleMixAttentionBackend.get_builder_cls()(types.SimpleNamespace(block_size=page), ["l0"], None,
                                                           torch.device(DEV)).build(0, common2)
    assert not md2.whole_prompts
    out2 = torch.zeros((B, H * D), dtype=torch.bfloat16, device=DEV)
    impl.forward(None, qd, kd, vd, kv_cache, md, output=out)  # the prompts again, then the decode step behind it
    impl.forward(None, _to_dev(q2, torch.bfloat16), _to_dev(np.concatenate(k2s), torch.bfloat16),
                 _to_dev(np.concatenate(v2s), torch.bfloat16), kv_cache, md2, output=out2)
    torch.cuda.synchronize()
    ks2 = [np.concatenate([ks[b], k2s[b]]) for b in range(B)]
    vs2 = [np.concatenate([vs[b], v2s[b]]) for b in range(B)]
    ref2 = orc.varlen_gqa_attention(q2, lambda b: ks2[b], lambda b: vs2[b], cu2, seq2, scale, "causal", 0, 0, 0)
    assert_close(out2.float().cpu().numpy().reshape(B, H, D), ref2, "backend forward, decode step", out16=True)


@pytest.mark.parametrize("route", ["fixed launch shape", "block index"])
def test_backend_pure_decode_step(monkeypatch, route):
    """A step of decode rows only: every row is the dense last row of the pattern, contexts from one page to several
    pieces.  By default the layers take the launch with a fixed shape (no index is built); with
    TRIANGLEMIX_DECODE_FIXED=0 the step goes through the block index, cut into pieces that fill the chip."""
    import types
    from vllm_trianglemix_amd import TriangleMixConfig
    from vllm_trianglemix_amd import backend as be
    from vllm_trianglemix_amd.backend import TriangleMixAttentionBackend
    monkeypatch.setattr(be, "_DECODE_FIXED", route == "fixed launch shape")
    H, Hkv, D, page = 8, 2, 128, 16
    seq_lens = [9000, 4100, 1, 17, 2049, 600]
    q_lens = [1] * len(seq_lens)
    q, ks, vs = _rand_case(31337, q_lens, seq_lens, H, Hkv, D)
    scale = D ** -0.5
    B = len(seq_lens)
    nblk = sum((s + page - 1) // page for s in seq_lens) + 1
    perm = np.random.default_rng(9).permutation(nblk).astype(np.int32)
    kv_cache = torch.zeros(TriangleMixAttentionBackend.get_kv_cache_shape(nblk, page, Hkv, D), dtype=torch.bfloat16,
                           device=DEV)
    bt = np.zeros((B, max((s + page - 1) // page for s in seq_lens)), np.int32)
    slots = np.zeros((B,), np.int64)
    nxt = 0
    for b, s in enumerate(seq_lens):
        for pb in range((s + page - 1) // page):
            bt[b, pb] = perm[nxt]
            nxt += 1
        jj = np.arange(s - 1)  # the context; the last token is this step's
        if len(jj):
            blk = torch.from_numpy(bt[b, jj // page].astype(np.int64)).to(DEV)
            kv_cache[0, blk, torch.from_numpy(jj % page).to(DEV)] = _to_dev(ks[b][:s - 1], torch.bfloat16)
            kv_cache[1, blk, torch.from_numpy(jj % page).to(DEV)] = _to_dev(vs[b][:s - 1], torch.bfloat16)
        slots[b] = int(bt[b, (s - 1) // page]) * page + (s - 1) % page
    cu = np.arange(B + 1, dtype=np.int32)
    common = types.SimpleNamespace(
        num_actual_tokens=B, max_query_len=1, max_seq_len=max(seq_lens),
        query_start_loc=torch.from_numpy(cu).to(DEV), query_start_loc_cpu=torch.from_numpy(cu),
        seq_lens=torch.tensor(seq_lens, dtype=torch.int32, device=DEV),
        seq_lens_cpu=torch.tensor(seq_lens, dtype=torch.int32),
        block_table_tensor=torch.from_numpy(bt).to(DEV), slot_mapping=torch.from_numpy(slots).to(DEV))
    md = TriangleMixAttentionBackend.get_builder_cls()(types.SimpleNamespace(block_size=page), ["l0"], None,
                                                       torch.device(DEV)).build(0, common)
    impl = TriangleMixAttentionBackend.get_impl_cls()(H, D, scale, Hkv, None, None, "auto", None, "decoder", None,
                                                      trianglemix_config=TriangleMixConfig(num_triangle_layers=1),
                                                      layer_idx=0)
    k_new = np.stack([ks[b][-1] for b in range(B)])
    v_new = np.stack([vs[b][-1] for b in range(B)])
    out = torch.zeros((B, H * D), dtype=torch.bfloat16, device=DEV)
    impl.forward(None, _to_dev(q, torch.bfloat16), _to_dev(k_new, torch.bfloat16), _to_dev(v_new, torch.bfloat16),
                 kv_cache, md, output=out)
    torch.cuda.synchronize()
    if route == "block index":
        (idx,) = md._indices.values()
        # 497 tiles x 2 head groups for 512 workgroup slots: the shortest pieces the advice gives (8 tiles)
        tiles = [-(-s // 32) for s in seq_lens]
        assert idx.chunk_tiles == 8
        assert idx.sizes.num_partials == sum(-(-t // 8) for t in tiles if t > 8)
    else:
        assert not md._indices and md.uniform_decode
    got = out.float().cpu().numpy().reshape(B, H, D)
    ref = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], cu, seq_lens, scale, "causal", 0, 0, 0)
    assert_close(got, ref, f"backend forward, decode step ({route})", out16=True)
    # the step's rows are in the cache afterwards
    flat = kv_cache.view(2, -1, Hkv, D)
    sd = torch.from_numpy(slots).to(DEV)
    assert torch.equal(flat[0][sd], _to_dev(k_new, torch.bfloat16)) and torch.equal(flat[1][sd], _to_dev(v_new, torch.bfloat16))


def test_fp8_kv_cache_write_gather_and_backend():
    """kv_cache_dtype "fp8" (e4m3fn): the cache write is bit-exact against the oracle's conversion
    (itself pinned to torch's on the CPU), the gather returns fp8 * scale rounded once to bf16, and the
    backend attends to what the cache holds (oracle on the dequantised keys / values)."""
    import types
    from vllm_trianglemix_amd import TriangleMixConfig
    from vllm_trianglemix_amd.backend import TriangleMixAttentionBackend
    ops = _ops()
    H, Hkv, D, page = 8, 2, 128, 16
    k_scale, v_scale = 0.75, 1.5
    q_lens, seq_lens = [2200, 40, 1], [2200, 300, 700]
    q, ks, vs = _rand_case(808, q_lens, seq_lens, H, Hkv, D)
    ks[0][5, 0, :4] = [1000.0, -1000.0, 336.1, 1e-4]  # saturation at +-448 * scale, a tiny value
    scale = D ** -0.5
    cu = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    T = int(cu[-1])
    nblk = sum((s + page - 1) // page for s in seq_lens) + 2
    perm = np.random.default_rng(5).permutation(nblk).astype(np.int32)
    kv_cache = torch.zeros((2, nblk, page, Hkv, D), dtype=torch.uint8, device=DEV)
    maxb = max((s + page - 1) // page for s in seq_lens)
    bt = np.zeros((len(seq_lens), maxb), np.int32)
    nxt = 0
    # every token goes through the fp8 write: the context in a first call, the step's rows in forward()
    ctx_rows_k, ctx_rows_v, ctx_slots = [], [], []
    slots = np.zeros((T,), np.int64)
    k_new, v_new = np.zeros((T, Hkv, D), np.float32), np.zeros((T, Hkv, D), np.float32)
    for b, (ql, s) in enumerate(zip(q_lens, seq_lens)):
        for pb in range((s + page - 1) // page):
            bt[b, pb] = perm[nxt]
            nxt += 1
        ctx = s - ql
        jj = np.arange(ctx)
        ctx_rows_k.append(ks[b][:ctx]); ctx_rows_v.append(vs[b][:ctx])
        ctx_slots.append(bt[b, jj // page].astype(np.int64) * page + jj % page)
        pos = np.arange(ctx, s)
        slots[cu[b]:cu[b + 1]] = bt[b, pos // page].astype(np.int64) * page + pos % page
        k_new[cu[b]:cu[b + 1]], v_new[cu[b]:cu[b + 1]] = ks[b][ctx:], vs[b][ctx:]
    ck = _to_dev(np.concatenate(ctx_rows_k), torch.bfloat16)
    cv = _to_dev(np.concatenate(ctx_rows_v), torch.bfloat16)
    cs = np.concatenate(ctx_slots)
    cs_pad = np.concatenate([cs, [-1]])  # a padding slot is skipped
    ops.reshape_and_cache_fp8(torch.cat([ck, ck[:1]]), torch.cat([cv, cv[:1]]), kv_cache[0], kv_cache[1],
                              torch.from_numpy(cs_pad).to(DEV), k_scale, v_scale)
    torch.cuda.synchronize()
    flat_k = kv_cache[0].view(nblk * page, Hkv, D).cpu().numpy()
    flat_v = kv_cache[1].view(nblk * page, Hkv, D).cpu().numpy()
    # the write divides by the scale in fp32 (quant_utils.cuh:544,589), it does not multiply by a reciprocal
    want_k = orc.fp8_e4m3fn_encode(ck.float().cpu().numpy() / np.float32(k_scale))
    want_v = orc.fp8_e4m3fn_encode(cv.float().cpu().numpy() / np.float32(v_scale))
    assert np.array_equal(flat_k[cs], want_k) and np.array_equal(flat_v[cs], want_v)
    untouched = np.ones(nblk * page, bool)
    untouched[cs] = False
    assert not flat_k[untouched].any()
    # the step through the backend
    common = types.SimpleNamespace(
        num_actual_tokens=T, max_query_len=max(q_lens), max_seq_len=max(seq_lens),
        query_start_loc=torch.from_numpy(cu).to(DEV), query_start_loc_cpu=torch.from_numpy(cu),
        seq_lens=torch.tensor(seq_lens, dtype=torch.int32, device=DEV),
        seq_lens_cpu=torch.tensor(seq_lens, dtype=torch.int32),
        block_table_tensor=torch.from_numpy(bt).to(DEV), slot_mapping=torch.from_numpy(slots).to(DEV))
    md = TriangleMixAttentionBackend.get_builder_cls()(types.SimpleNamespace(block_size=page), ["l0"], None,
                                                       torch.device(DEV)).build(0, common)
    impl = TriangleMixAttentionBackend.get_impl_cls()(H, D, scale, Hkv, None, None, "fp8", None, "decoder", None,
                                                      trianglemix_config=TriangleMixConfig(num_triangle_layers=1),
                                                      layer_idx=0)
    layer = types.SimpleNamespace(_k_scale_float=k_scale, _v_scale_float=v_scale)
    out = torch.zeros((T, H * D), dtype=torch.bfloat16, device=DEV)
    impl.forward(layer, _to_dev(q, torch.bfloat16), _to_dev(k_new, torch.bfloat16), _to_dev(v_new, torch.bfloat16),
                 kv_cache.view(torch.float8_e4m3fn), md, output=out)
    torch.cuda.synchronize()
    got = out.float().cpu().numpy().reshape(T, H, D)
    # what the attention sees: bf16(fp8(bf16(x) / scale) * scale)
    def seen(x, sc):
        return orc.bf16_round(orc.fp8_cache_roundtrip(orc.bf16_round(x), sc).astype(np.float32))
    kd = [seen(k, k_scale) for k in ks]
    vd = [seen(v, v_scale) for v in vs]
    assert abs(kd[0][5, 0, 0] - 448 * k_scale) < 1e-6 and abs(kd[0][5, 0, 1] + 448 * k_scale) < 1e-6
    # the gather on its own: bit-exact rows
    cu_k = np.concatenate([[0], np.cumsum(seq_lens)]).astype(np.int32)
    k_rows, v_rows = ops.gather_dequant_kv_fp8(kv_cache[0], kv_cache[1], torch.from_numpy(bt).to(DEV),
                                               torch.tensor(seq_lens, dtype=torch.int32, device=DEV),
                                               torch.from_numpy(cu_k).to(DEV), int(cu_k[-1]), max(seq_lens),
                                               torch.bfloat16, k_scale, v_scale)
    torch.cuda.synchronize()
    assert np.array_equal(k_rows.float().cpu().numpy(), np.concatenate(kd).astype(np.float32))
    assert np.array_equal(v_rows.float().cpu().numpy(), np.concatenate(vd).astype(np.float32))
    ref = np.zeros((T, H, D))
    for b, (ql, s) in enumerate(zip(q_lens, seq_lens)):
        mode = "triangle" if s > 2048 else "causal"
        ref[cu[b]:cu[b + 1]] = orc.varlen_gqa_attention(q[cu[b]:cu[b + 1]], lambda _: kd[b], lambda _: vd[b], [0, ql],
                                                        [s], scale, mode, 4, 32, 64)
    assert_close(got, ref, "backend forward, fp8 cache", out16=True)


def test_torch_library_binding_equals_ctypes_path(monkeypatch):
    """torch.ops._trianglemix_C.* (TORCH_LIBRARY registration) drives the same entry points as the
    ctypes binding: identical index, bit-identical attention output -- called directly, through
    ops.attention's default route (the operator) and through its ctypes route."""
    from vllm_trianglemix_amd import torch_ops
    ops = _ops()
    C = torch_ops.load()
    H, Hkv, D, page = 8, 2, 128, 16
    q_lens, seq_lens = [300, 1, 77], [300, 500, 130]
    si, sl, last, chunk = 4, 32, 64, 4
    q, ks, vs = _rand_case(555, q_lens, seq_lens, H, Hkv, D)
    cu_h = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    sl_h = np.asarray(seq_lens, np.int32)
    cu, sld = torch.from_numpy(cu_h).to(DEV), torch.from_numpy(sl_h).to(DEV)
    nblk = sum((s + page - 1) // page for s in seq_lens)
    perm = np.random.default_rng(1).permutation(nblk).astype(np.int32)
    maxb = max((s + page - 1) // page for s in seq_lens)
    bt = np.zeros((len(seq_lens), maxb), np.int32)
    kc = torch.zeros((nblk, page, Hkv, D), dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    nxt = 0
    slot_list = []
    for b, s in enumerate(seq_lens):
        for pb in range((s + page - 1) // page):
            bt[b, pb] = perm[nxt]
            nxt += 1
        pos = np.arange(s)
        slot_list.append(bt[b, pos // page].astype(np.int64) * page + pos % page)
    slots = torch.from_numpy(np.concatenate(slot_list)).to(DEV)
    C.trianglemix_reshape_and_cache(_to_dev(np.concatenate(ks, 0), torch.bfloat16),
                                    _to_dev(np.concatenate(vs, 0), torch.bfloat16), kc, vc, slots)
    btd = torch.from_numpy(bt).to(DEV)
    bm, bn, mode = ops.attention_block_m(H, Hkv), ops.attention_block_n(), ops.MODES["triangle"]
    R, nnz, n_items, n_part, n_split, _ = C.trianglemix_index_sizes(torch.from_numpy(cu_h), torch.from_numpy(sl_h), si,
                                                                    sl, last, bm, bn, mode, 0, chunk)
    i32 = lambda n: torch.empty((max(n, 1),), dtype=torch.int32, device=DEV)  # noqa: E731
    index = [i32(R + 1), i32(nnz), i32(2 * R), i32(R + 1), i32(8 * n_items), i32(n_split), i32(8),
             i32(len(seq_lens) + 1 + 2 * (R + 1) + 8 * (R // 1024 + 1))]  # TMX_INDEX_SCRATCH_INTS
    tile_pages = i32(2 * nnz)
    C.trianglemix_build_block_index(index, tile_pages, cu, sld, torch.from_numpy(cu_h), torch.from_numpy(sl_h), btd,
                                    page, si, sl, last, bm, bn, mode, 0, chunk)
    ref_idx = ops.build_block_index(cu, sld, cu_h, sl_h, si, sl, last, bm, bn, mode, chunk_tiles=chunk,
                                    block_table=btd, page_size=page)
    assert torch.equal(index[0], ref_idx.row_ptr) and torch.equal(index[1][:nnz], ref_idx.col_idx)
    assert torch.equal(index[4][:8 * n_items].view(-1, 8), ref_idx.items)
    assert torch.equal(tile_pages[:2 * nnz].view(-1, 2), ref_idx.tile_pages)
    qd = _to_dev(q, torch.bfloat16)
    part_o = torch.empty((max(1, n_part * H * bm * D),), dtype=torch.float32, device=DEV)
    part_ml = torch.empty((max(1, n_part * H * bm * 2),), dtype=torch.float32, device=DEV)
    out_t = torch.full(qd.shape, float("nan"), dtype=torch.bfloat16, device=DEV)
    C.trianglemix_attention(out_t, qd, kc, vc, btd, cu, sld, None, index, tile_pages, part_o, part_ml, None,
                            D ** -0.5, si, sl, last, mode, 0, chunk, n_items, n_part, n_split)
    monkeypatch.setattr(ops, "_BINDING", "torch")  # the default route of ops.attention
    assert ops._op_binding() is not None
    out_d = torch.full(qd.shape, float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.attention(qd, kc, vc, out_d, ref_idx, cu, sld, D ** -0.5, block_table=btd)
    monkeypatch.setattr(ops, "_BINDING", "ctypes")
    out_c = torch.full(qd.shape, float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.attention(qd, kc, vc, out_c, ref_idx, cu, sld, D ** -0.5, block_table=btd)
    monkeypatch.undo()
    torch.cuda.synchronize()
    assert torch.equal(out_t.view(torch.int16), out_c.view(torch.int16))
    assert torch.equal(out_d.view(torch.int16), out_c.view(torch.int16))
    ref = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], cu_h, seq_lens, D ** -0.5, "triangle", si, sl,
                                   last)
    got = out_t.float().cpu().numpy()
    assert_close(got, ref, "torch.ops binding", out16=True)
    with pytest.raises(RuntimeError, match="trianglemix"):
        C.trianglemix_attention(out_t, qd, kc, vc, btd, cu, sld, None, index, tile_pages, part_o, part_ml, None,
                                -1.0, si, sl, last, mode, 0, chunk, n_items, n_part, n_split)


@pytest.mark.parametrize("binding", ["torch", "ctypes"])
def test_resolved_pages_are_used_only_with_the_block_table_they_came_from(monkeypatch, binding):
    """The index builder can resolve every listed tile's physical pages from a block table.  An attention call
    that passes ANOTHER block table (the same sequences moved to other pages, or a copy of the table elsewhere
    in memory) or another page size must not read those pages: it falls back to its own block-table lookups and
    still gives the oracle's rows, through both bindings of the library."""
    ops = _ops()
    monkeypatch.setattr(ops, "_BINDING", binding)
    H, Hkv, D, page = 8, 2, 128, 16
    q_lens, seq_lens = [2300, 1, 70], [2300, 900, 2500]
    q, ks, vs = _rand_case(6060, q_lens, seq_lens, H, Hkv, D)
    scale = D ** -0.5
    B = len(seq_lens)
    cu_h = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    ref = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], cu_h, seq_lens, scale, "triangle", 4, 32, 64,
                                   tri_min_len=2048)
    nblk = sum((s + page - 1) // page for s in seq_lens) + 4

    def layout(seed, page_):
        """A cache filled through a block table drawn from ``seed`` (pages of ``page_`` tokens)."""
        nb = sum((s + page_ - 1) // page_ for s in seq_lens) + 4
        perm = np.random.default_rng(seed).permutation(nb).astype(np.int32)
        bt = np.zeros((B, max((s + page_ - 1) // page_ for s in seq_lens)), np.int32)
        kc = torch.zeros((nb, page_, Hkv, D), dtype=torch.bfloat16, device=DEV)
        vc = torch.zeros_like(kc)
        nxt = 0
        for b, s in enumerate(seq_lens):
            for pb in range((s + page_ - 1) // page_):
                bt[b, pb] = perm[nxt]
                nxt += 1
            pos = np.arange(s)
            slots = torch.from_numpy(bt[b, pos // page_].astype(np.int64) * page_ + pos % page_).to(DEV)
            ops.reshape_and_cache(_to_dev(ks[b], torch.bfloat16), _to_dev(vs[b], torch.bfloat16), kc, vc, slots)
        return torch.from_numpy(bt).to(DEV), kc, vc

    bt1, kc1, vc1 = layout(1, page)
    bt2, kc2, vc2 = layout(2, page)       # the same sequences on other pages
    bt3, kc3, vc3 = layout(3, 4 * page)   # ... and in pages of another size
    idx, cu, sld = _build(ops, q_lens, seq_lens, 4, 32, 64, ops.attention_block_m(H, Hkv), 32, ops.MODES["triangle"], 16,
                          2048, block_table=bt1, page_size=page)
    assert idx.tile_pages is not None
    qd = _to_dev(q, torch.bfloat16)
    for name, bt, kc, vc in (("the table the pages came from", bt1, kc1, vc1), ("a copy of that table", bt1.clone(), kc1, vc1),
                             ("another table", bt2, kc2, vc2), ("another page size", bt3, kc3, vc3)):
        out = torch.full((sum(q_lens), H, D), float("nan"), dtype=torch.float32, device=DEV)
        ops.attention(qd, kc, vc, out, idx, cu, sld, scale, block_table=bt)
        torch.cuda.synchronize()
        assert_close(out.cpu().numpy(), ref, f"index with resolved pages, {binding} binding", info=name)


@pytest.mark.parametrize("binding", ["torch", "ctypes"])
@pytest.mark.parametrize("case", [
    # (q_lens, seq_lens, H, Hkv, dtype, paged, mode, chunk): one long prompt, the benchmark's shapes in small
    dict(q_lens=[8192], seq_lens=[8192], H=8, Hkv=2, dtype=torch.bfloat16, paged=True, mode="triangle", chunk=16),
    dict(q_lens=[32768], seq_lens=[32768], H=8, Hkv=2, dtype=torch.bfloat16, paged=True, mode="triangle", chunk=64),
    # a ragged batch below 1024 q-blocks, contiguous K/V, fp16, 8 : 1 heads (two head chunks per K/V head)
    dict(q_lens=[3000, 5000, 2049], seq_lens=[3000, 5000, 2049], H=8, Hkv=1, dtype=torch.float16, paged=False,
         mode="triangle", chunk=8),
    # chunked prefill: context in the cache, prompt lengths known
    dict(q_lens=[4096], seq_lens=[12288], H=8, Hkv=2, dtype=torch.bfloat16, paged=True, mode="triangle", chunk=16,
         full=[16384]),
    # steps whose build cannot ride in the attention launch get it as a launch of their own: a causal layer (dense
    # kernel first), decode rows beside a prompt (decode kernel first), more than 1024 q-blocks, head size 64
    dict(q_lens=[4096], seq_lens=[4096], H=8, Hkv=2, dtype=torch.bfloat16, paged=True, mode="causal", chunk=16),
    dict(q_lens=[4096, 1, 1], seq_lens=[4096, 700, 3000], H=8, Hkv=2, dtype=torch.bfloat16, paged=True, mode="triangle", chunk=16),
    dict(q_lens=[36000], seq_lens=[36000], H=4, Hkv=1, dtype=torch.bfloat16, paged=True, mode="triangle", chunk=64),
    dict(q_lens=[4096], seq_lens=[4096], H=8, Hkv=2, D=64, dtype=torch.bfloat16, paged=False, mode="triangle", chunk=16),
], ids=lambda c: f"{c['mode']}-{'+'.join(map(str, c['q_lens']))}-{c['H']}x{c['Hkv']}-{str(c['dtype'])[6:]}")
def test_index_built_by_the_attention_call(monkeypatch, binding, case):
    """``build_block_index(defer=True)`` + ``attention``: the call builds the index itself (TMX_ATTN_BUILD_INDEX),
    inside the attention launch where that launch has band workgroups to run meanwhile.  Every array of the index
    and the output must be bit-identical to the two-call form, repeatedly (the buffers of one round are the next
    round's: what an earlier build left in them must not read as "built"), and a later call reuses the index."""
    ops = _ops()
    monkeypatch.setattr(ops, "_BINDING", binding)
    H, Hkv, D, page = case["H"], case["Hkv"], case.get("D", 128), 16
    q_lens, seq_lens, dtype, paged = case["q_lens"], case["seq_lens"], case["dtype"], case["paged"]
    B, T = len(q_lens), sum(q_lens)
    g = torch.Generator().manual_seed(4242)
    q = torch.randn((T, H, D), generator=g).to(DEV).to(dtype)
    cu_h = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    sl_h = np.asarray(seq_lens, np.int32)
    cu, sld = torch.from_numpy(cu_h).to(DEV), torch.from_numpy(sl_h).to(DEV)
    kw = {}
    if paged:
        nblk = sum((s + page - 1) // page for s in seq_lens)
        kc = torch.randn((nblk, page, Hkv, D), generator=g).to(DEV).to(dtype)
        vc = torch.randn((nblk, page, Hkv, D), generator=g).to(DEV).to(dtype)
        perm = torch.randperm(nblk, generator=g).to(torch.int32)
        width = max((s + page - 1) // page for s in seq_lens)
        bt = torch.zeros((B, width), dtype=torch.int32)
        nxt = 0
        for b, sq in enumerate(seq_lens):
            nb = (sq + page - 1) // page
            bt[b, :nb] = perm[nxt:nxt + nb]
            nxt += nb
        bt = bt.to(DEV)
        kw = dict(block_table=bt)
        build_kw = dict(block_table=bt, page_size=page)
    else:
        kc = torch.randn((sum(seq_lens), Hkv, D), generator=g).to(DEV).to(dtype)
        vc = torch.randn((sum(seq_lens), Hkv, D), generator=g).to(DEV).to(dtype)
        cu_k = torch.from_numpy(np.concatenate([[0], np.cumsum(seq_lens)]).astype(np.int32)).to(DEV)
        kw = dict(cu_seqlens_k=cu_k)
        build_kw = {}
    if "full" in case:
        fl_h = np.asarray(case["full"], np.int32)
        build_kw.update(full_lens=torch.from_numpy(fl_h).to(DEV), full_lens_cpu=fl_h)
    bm = ops.attention_block_m(H, Hkv)
    mode = ops.MODES[case["mode"]]

    def make(defer):
        return ops.build_block_index(cu, sld, cu_h, sl_h, 4, 32, 64, bm, 32, mode, chunk_tiles=case["chunk"], tri_min_len=2048,
                                     defer=defer, **build_kw)

    ref_idx = make(False)
    ref_out = torch.full((T, H, D), float("nan"), dtype=dtype, device=DEV)
    ops.attention(q, kc, vc, ref_out, ref_idx, cu, sld, D ** -0.5, **kw)
    names = ["row_ptr", "col_idx", "qb_meta", "part_ptr", "items", "split_qb", "totals"] + (["tile_pages"] if paged else [])
    for rnd in range(3):
        idx = make(True)
        assert not idx.built
        out = torch.full((T, H, D), float("nan"), dtype=dtype, device=DEV)
        ops.attention(q, kc, vc, out, idx, cu, sld, D ** -0.5, **kw)
        assert idx.built
        torch.cuda.synchronize()
        for name in names:
            assert torch.equal(getattr(idx, name), getattr(ref_idx, name)), (name, rnd)
        assert torch.equal(out.view(torch.int16), ref_out.view(torch.int16)), rnd
        # the built index serves the step's other layers like any other
        out2 = torch.full((T, H, D), float("nan"), dtype=dtype, device=DEV)
        ops.attention(q, kc, vc, out2, idx, cu, sld, D ** -0.5, **kw)
        torch.cuda.synchronize()
        assert torch.equal(out2.view(torch.int16), ref_out.view(torch.int16)), rnd
        del idx  # (its buffer goes back to the allocator: the next round's index lands on it)
    # the batch description of the building call must be the one the index was sized for
    idx = make(True)
    with pytest.raises(RuntimeError, match="deferred index"):
        ops.attention(q, kc, vc, out, idx, cu.clone(), sld, D ** -0.5, **kw)
    idx.build_now()
    assert idx.built and torch.equal(idx.items, ref_idx.items)


def test_backend_length_gate_2048_vs_2049(golden_attention):
    """The reference gives no mask up to 2048 tokens and the triangle mask from 2049 on
    (trianglemix_attention.py:330-332).  Through the backend: a 2048-token prompt runs plain causal attention,
    a 2049-token prompt the triangle pattern; both in one step, checked against the oracle, and the 2049 case
    also against the reference's own output (golden ``helper_tri_2049``: one head, head size 64)."""
    import types
    from vllm_trianglemix_amd import TriangleMixConfig
    from vllm_trianglemix_amd.backend import TriangleMixAttentionBackend
    meta, arrays = golden_attention
    (g,) = [c for c in meta["cases"] if c["name"] == "helper_tri_2049"]
    H, Hkv, D, page = 1, 1, 64, 16
    qg, kg, vg = (orc.seeded_normal(g["seed"] * 3 + t, (1, 2049, 1, 64))[0] for t in range(3))
    q_lens = seq_lens = [2048, 2049]
    q, ks, vs = _rand_case(77, q_lens, seq_lens, H, Hkv, D)
    q[2048:], ks[1], vs[1] = qg, kg, vg
    scale = float(g["scale"])
    cu = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    T = int(cu[-1])
    nblk = sum((s + page - 1) // page for s in seq_lens) + 2
    perm = np.random.default_rng(5).permutation(nblk).astype(np.int32)
    bt = np.zeros((2, (2049 + page - 1) // page), np.int32)
    slots = np.zeros((T,), np.int64)
    nxt = 0
    for b, s in enumerate(seq_lens):
        for pb in range((s + page - 1) // page):
            bt[b, pb] = perm[nxt]
            nxt += 1
        pos = np.arange(s)
        slots[cu[b]:cu[b + 1]] = bt[b, pos // page].astype(np.int64) * page + pos % page
    common = types.SimpleNamespace(
        num_actual_tokens=T, max_query_len=2049, max_seq_len=2049,
        query_start_loc=torch.from_numpy(cu).to(DEV), query_start_loc_cpu=torch.from_numpy(cu),
        seq_lens=torch.tensor(seq_lens, dtype=torch.int32, device=DEV), seq_lens_cpu=torch.tensor(seq_lens, dtype=torch.int32),
        block_table_tensor=torch.from_numpy(bt).to(DEV), slot_mapping=torch.from_numpy(slots).to(DEV))
    md = TriangleMixAttentionBackend.get_builder_cls()(types.SimpleNamespace(block_size=page), ["l0"], None,
                                                       torch.device(DEV)).build(0, common)
    impl = TriangleMixAttentionBackend.get_impl_cls()(H, D, scale, Hkv, None, None, "auto", None, "decoder", None,
                                                      trianglemix_config=TriangleMixConfig(num_triangle_layers=1),
                                                      layer_idx=0)
    assert impl.tri_min_len == 2048
    kv_cache = torch.zeros(TriangleMixAttentionBackend.get_kv_cache_shape(nblk, page, Hkv, D), dtype=torch.bfloat16,
                           device=DEV)
    out = torch.zeros((T, H * D), dtype=torch.float32, device=DEV)
    impl.forward(None, _to_dev(q, torch.bfloat16), _to_dev(np.concatenate(ks), torch.bfloat16),
                 _to_dev(np.concatenate(vs), torch.bfloat16), kv_cache, md, output=out)
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(T, H, D)
    for b, (s, mode) in enumerate(zip(seq_lens, ("causal", "triangle"))):
        ref = orc.varlen_gqa_attention(q[cu[b]:cu[b + 1]], lambda _: ks[b], lambda _: vs[b], [0, s], [s], scale, mode,
                                       4, 32, 64)
        assert_close(got[cu[b]:cu[b + 1]], ref, "backend length gate 2048 / 2049", info=(s, mode))
    # 2049 tokens: causal attention would differ visibly from the triangle pattern on the middle rows
    dense = orc.varlen_gqa_attention(q[2048:], lambda _: ks[1], lambda _: vs[1], [0, 2049], [2049], scale, "causal",
                                     0, 0, 0)
    assert float(np.abs(got[2048:] - dense).max()) > 0.05
    # the reference's own fp32 run on the same (bf16-representable) inputs
    want = arrays["helper_tri_2049"][0]
    assert_close(got[2048:], want, "backend length gate 2048 / 2049 (reference's own output)")


@pytest.mark.parametrize("case", [
    # (q_lens, seq_lens, H, Hkv, D, pattern, dtype, paged, page, full_lens)
    dict(q=[3000], s=[3000], H=8, Hkv=2, D=128, pat=(4, 32, 64)),                                   # one prompt
    dict(q=[1024, 700, 1], s=[3072, 700, 900], H=8, Hkv=2, D=128, pat=(4, 32, 64), paged=True),     # context a multiple of 32
    dict(q=[1000, 2500], s=[3047, 2500], H=8, Hkv=2, D=128, pat=(4, 32, 64), paged=True),           # ... and not (no band)
    dict(q=[2600], s=[2600], H=16, Hkv=2, D=64, pat=(0, 17, 1), dtype=torch.float16),               # 8 heads per K/V head
    dict(q=[2400, 2400], s=[2400, 2400], H=4, Hkv=1, D=128, pat=(15, 31, 200), paged=True, page=64),
    dict(q=[2200], s=[2200], H=4, Hkv=1, D=64, pat=(-1, 32, 0), out=torch.bfloat16),                # no sink, no dense rows
    dict(q=[512, 512], s=[2560, 3072], H=8, Hkv=2, D=128, pat=(4, 32, 64), paged=True, full=[6000, 3072]),  # chunks
])
def test_band_path_matches_oracle_and_work_list(case):
    """Sparse q-blocks in canonical position (32 full rows at a multiple of 32, clear of the sink tile and of
    the trailing dense rows) run on the band workgroups.  Same results as the oracle, and as the same launch
    with every q-block on the work list (TMX_ATTN_NO_BAND), for patterns at the edges of what the band path
    takes (window 17..32, sink -1..15), grouped heads 4 and 8, both head sizes, paged and contiguous K/V,
    requests with context, chunked prompts."""
    from vllm_trianglemix_amd._lib import TMX_ATTN_NO_BAND
    ops = _ops()
    q_lens, seq_lens, H, Hkv, D = case["q"], case["s"], case["H"], case["Hkv"], case["D"]
    si, sl, last = case["pat"]
    dtype, out_dtype = case.get("dtype", torch.bfloat16), case.get("out", torch.float32)
    q, ks, vs = _rand_case(909 + H + D, q_lens, seq_lens, H, Hkv, D)
    if dtype == torch.float16:
        q, ks, vs = (np.float16(q).astype(np.float32), [np.float16(k).astype(np.float32) for k in ks],
                     [np.float16(v).astype(np.float32) for v in vs])
    scale = D ** -0.5
    kw = dict(dtype=dtype, out_dtype=out_dtype, paged=case.get("paged", False), page_size=case.get("page", 16),
              full_lens=case.get("full"), gate=2048)
    got = _run_varlen(ops, q, ks, vs, q_lens, seq_lens, scale, "triangle", (si, sl, last), **kw)
    plain = _run_varlen(ops, q, ks, vs, q_lens, seq_lens, scale, "triangle", (si, sl, last), flags=TMX_ATTN_NO_BAND, **kw)
    cu = np.concatenate([[0], np.cumsum(q_lens)])
    ref = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], cu, seq_lens, scale, "triangle", si, sl, last,
                                   full_lens=case.get("full"), tri_min_len=2048)
    o16 = out_dtype != torch.float32
    assert_close(got, ref, f"band workgroups, {str(dtype)[6:]}", out16=o16)
    assert_close(plain, ref, f"work list (no band), {str(dtype)[6:]}", out16=o16)


@pytest.mark.parametrize("case", [
    dict(q=[700], s=[700], H=8, Hkv=2),                                            # one prompt, ragged last block
    dict(q=[300, 1, 17, 64, 5], s=[300, 900, 217, 64, 2000], H=8, Hkv=2, paged=True),  # prompts, decode rows, context
    dict(q=[513, 130], s=[1999, 130], H=16, Hkv=2, paged=True, page=64, dtype=torch.float16),  # 8 heads per K/V head
    dict(q=[1100], s=[1100], H=4, Hkv=1, out=torch.bfloat16),
    dict(q=[200, 333], s=[200, 1333], H=4, Hkv=4, out=torch.float16, dtype=torch.float16),  # no grouping: work list only
    # 1 to 5 key tiles per 64-row block, odd and even counts (the tile loop is unrolled over its two LDS buffers)
    dict(q=[17, 64, 65, 128, 129, 192, 193], s=[67, 64, 128, 129, 129, 192, 300], H=8, Hkv=2, paged=True),
])
def test_dense_causal_kernel_matches_oracle_and_work_list(case):
    """Causal launches (the layers outside the triangle set): requests with 17 query rows or more run on the
    dense kernel (32-row waves, 64-key tiles, fp16 probabilities against V converted to fp16), the others on
    the work list; same results as the oracle and as the same launch kept on the work list (TMX_ATTN_NO_DENSE)."""
    from vllm_trianglemix_amd._lib import TMX_ATTN_NO_DENSE
    ops = _ops()
    q_lens, seq_lens, H, Hkv, D = case["q"], case["s"], case["H"], case["Hkv"], 128
    dtype, out_dtype = case.get("dtype", torch.bfloat16), case.get("out", torch.float32)
    q, ks, vs = _rand_case(4711 + H, q_lens, seq_lens, H, Hkv, D)
    if dtype == torch.float16:
        q, ks, vs = (np.float16(q).astype(np.float32), [np.float16(k).astype(np.float32) for k in ks],
                     [np.float16(v).astype(np.float32) for v in vs])
    vs[0][3, 0, :8] = [40000.0, -3.0e-6, 2.0 ** -15, -65504.0 if dtype == torch.float16 else -65280.0, 1e-30, 0.0, 255.0, -0.001]
    vs = [orc.bf16_round(v) if dtype == torch.bfloat16 else np.float16(v).astype(np.float32) for v in vs]
    scale = D ** -0.5
    kw = dict(dtype=dtype, out_dtype=out_dtype, paged=case.get("paged", False), page_size=case.get("page", 16), chunk=16)
    got = _run_varlen(ops, q, ks, vs, q_lens, seq_lens, scale, "causal", (0, 0, 0), **kw)
    plain = _run_varlen(ops, q, ks, vs, q_lens, seq_lens, scale, "causal", (0, 0, 0), flags=TMX_ATTN_NO_DENSE, **kw)
    cu = np.concatenate([[0], np.cumsum(q_lens)])
    ref = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], cu, seq_lens, scale, "causal", 0, 0, 0)
    # (the large V entries make |ref| large on a few rows: the bar is relative there)
    o16 = out_dtype != torch.float32
    assert_close(got, ref, f"dense causal kernel, {str(dtype)[6:]}", out16=o16)
    assert_close(plain, ref, f"work list (causal launch), {str(dtype)[6:]}", out16=o16)
    # the staged form (K/V rows gathered first, tiles by LDS-DMA) puts the same values into LDS: the same bits out
    staged = _run_varlen(ops, q, ks, vs, q_lens, seq_lens, scale, "causal", (0, 0, 0), stage=True, **kw)
    assert np.array_equal(staged, got), np.abs(staged - got).max()


@pytest.mark.parametrize("binding", ["torch", "ctypes"])
def test_causal_index_without_the_dense_kernels_requests(monkeypatch, binding):
    """leave_dense: the causal index of a step holds only the requests with fewer than 17 query rows (bit-exact
    against the oracle's index of the batch without the other requests' rows; request numbers and rows as in the
    whole batch), the attention call computes the others with the dense kernel and gives the same bits as with the
    full index; a call that cannot run the dense kernel refuses the index."""
    ops = _ops()
    monkeypatch.setattr(ops, "_BINDING", binding)
    H, Hkv, D, page = 8, 2, 128, 16
    q_lens, seq_lens = [300, 1, 16, 17, 0, 5, 129, 1, 70], [300, 2500, 67, 900, 0, 905, 129, 70, 2111]
    bm, bn = ops.attention_block_m(H, Hkv), 32
    cu_h = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    sl_h = np.asarray(seq_lens, np.int32)
    cu, sld = torch.from_numpy(cu_h).to(DEV), torch.from_numpy(sl_h).to(DEV)
    chunk = 8
    idx = ops.build_block_index(cu, sld, cu_h, sl_h, 0, 0, 0, bm, bn, ops.MODES["causal"], chunk_tiles=chunk, leave_dense=True)
    cu_f, sl_f = ops.without_dense_requests(cu_h, sl_h)
    qf, sf = np.diff(cu_f).tolist(), sl_f.tolist()
    row_ptr, col_idx, qb_seq, qb_row0 = orc.block_index_closed_form(qf, sf, 0, 0, 0, bm, bn, orc.MODE_CAUSAL, None)
    items, part_ptr = orc.work_items(row_ptr, chunk, qb_seq, qb_row0, qf, sf)
    # (an item carries the first query row of its q-block: a row of the WHOLE batch's q tensor)
    items = items.copy()
    items[:, 5] += (cu_h[:-1] - cu_f[:-1])[items[:, 4]]
    tot = idx.totals.cpu().numpy()
    assert tot[0] == len(row_ptr) - 1 == idx.sizes.num_qblocks and tot[1] == len(col_idx) == idx.sizes.nnz_tiles
    assert tot[2] == len(items) == idx.sizes.num_items and tot[3] == part_ptr[-1] == idx.sizes.num_partials
    assert np.array_equal(idx.row_ptr.cpu().numpy(), row_ptr) and np.array_equal(idx.col_idx.cpu().numpy(), col_idx)
    meta = idx.qb_meta.cpu().numpy()
    assert np.array_equal(meta[:, 0], qb_seq) and np.array_equal(meta[:, 1], qb_row0)
    assert set(meta[:, 0].tolist()) == {1, 2, 5, 7}   # the requests below 17 rows, numbered as in the whole batch
    assert np.array_equal(idx.items.cpu().numpy(), items) and np.array_equal(idx.part_ptr.cpu().numpy(), part_ptr)
    full = ops.build_block_index(cu, sld, cu_h, sl_h, 0, 0, 0, bm, bn, ops.MODES["causal"], chunk_tiles=chunk)
    assert full.sizes.num_partials > 3 * idx.sizes.num_partials and full.sizes.nnz_tiles > 3 * idx.sizes.nnz_tiles
    # attention: paged cache, both indexes
    q, ks, vs = _rand_case(321, q_lens, seq_lens, H, Hkv, D)
    nblk = sum((s + page - 1) // page for s in seq_lens) + 2
    perm = np.random.default_rng(3).permutation(nblk).astype(np.int32)
    kc = torch.zeros((nblk, page, Hkv, D), dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    bt = np.zeros((len(seq_lens), max((s + page - 1) // page for s in seq_lens)), np.int32)
    nxt = 0
    for b, s in enumerate(seq_lens):
        for pb in range((s + page - 1) // page):
            bt[b, pb] = perm[nxt]
            nxt += 1
            lo, hi = pb * page, min((pb + 1) * page, s)
            kc[bt[b, pb], :hi - lo] = _to_dev(ks[b][lo:hi], torch.bfloat16)
            vc[bt[b, pb], :hi - lo] = _to_dev(vs[b][lo:hi], torch.bfloat16)
    btd = torch.from_numpy(bt).to(DEV)
    qd = _to_dev(q, torch.bfloat16)
    scale = D ** -0.5
    outs = []
    for ix in (full, idx):
        out = torch.full((int(cu_h[-1]), H, D), float("nan"), dtype=torch.float32, device=DEV)
        ops.attention(qd, kc, vc, out, ix, cu, sld, scale, block_table=btd)
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    ref = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], cu_h, seq_lens, scale, "causal", 0, 0, 0)
    assert_close(outs[1].cpu().numpy(), ref, "causal launch, index without the dense kernel's requests")
    # calls that cannot run the dense kernel: a bias, the flag that keeps it off
    from vllm_trianglemix_amd._lib import TMX_ATTN_NO_DENSE
    with pytest.raises(RuntimeError):
        ops.attention(qd, kc, vc, out, idx, cu, sld, scale, block_table=btd, flags=TMX_ATTN_NO_DENSE)
    with pytest.raises(RuntimeError):
        ops.attention(qd, kc, vc, out, idx, cu, sld, scale, block_table=btd,
                      bias=torch.zeros((2200, 2600), dtype=torch.float32, device=DEV))
    with pytest.raises(RuntimeError):
        ops.build_block_index(cu, sld, cu_h, sl_h, 4, 32, 64, bm, bn, ops.MODES["triangle"], leave_dense=True)
    # a step of long prompts only: an empty index, the dense kernel alone
    q2, s2 = [700, 64], [700, 1000]
    cu2_h, sl2_h = np.asarray([0, 700, 764], np.int32), np.asarray(s2, np.int32)
    e = ops.build_block_index(torch.from_numpy(cu2_h).to(DEV), torch.from_numpy(sl2_h).to(DEV), cu2_h, sl2_h, 0, 0, 0, bm, bn,
                              ops.MODES["causal"], chunk_tiles=chunk, leave_dense=True)
    assert (e.sizes.num_qblocks, e.sizes.num_items, e.sizes.num_partials) == (0, 0, 0)
    qq, kk, vv = _rand_case(77, q2, s2, H, Hkv, D)
    got = _run_varlen(ops, qq, kk, vv, q2, s2, scale, "causal", (0, 0, 0), paged=True, page_size=16, chunk=chunk)
    kd, vd = _to_dev(np.concatenate(kk, 0), torch.bfloat16), _to_dev(np.concatenate(vv, 0), torch.bfloat16)
    cu_k = torch.from_numpy(np.concatenate([[0], np.cumsum(s2)]).astype(np.int32)).to(DEV)
    out = torch.full((764, H, D), float("nan"), dtype=torch.float32, device=DEV)
    ops.attention(_to_dev(qq, torch.bfloat16), kd, vd, out, e, torch.from_numpy(cu2_h).to(DEV), torch.from_numpy(sl2_h).to(DEV),
                  scale, cu_seqlens_k=cu_k)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), got)


@pytest.mark.parametrize("binding", ["torch", "ctypes"])
def test_dense_causal_staging_area(monkeypatch, binding):
    """The staging area of the dense causal kernel through both bindings: sized by dense_stage_rows (long prompts
    only by default; every request with 17 query rows or more when asked), handed over as kv_stage or as the third
    member of the workspace, larger than needed or exactly as needed; an area that is too small is refused; a
    batch whose index says "no staging" ignores the area.  Ragged batch with context, a request below 17 rows and
    a decode row between the staged ones."""
    ops = _ops()
    monkeypatch.setattr(ops, "_BINDING", binding)
    H, Hkv, D = 8, 2, 128
    q_lens, seq_lens = [300, 64, 17, 5, 129, 1, 70], [300, 500, 67, 900, 129, 70, 2111]
    cu_h = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    sl_h = np.asarray(seq_lens, np.int32)
    assert ops.dense_stage_rows(cu_h, sl_h) == 0                       # short requests: staging would not pay
    assert ops.dense_stage_rows(cu_h, sl_h, 0) == 320 + 512 + 128 + 192 + 2112
    assert ops.dense_stage_rows([0, 4096], [4096]) == 4096 and ops.dense_stage_rows([0, 4095], [4095]) == 0
    assert ops.dense_stage_rows([0, 4096, 4097], [5000, 77]) == 5056   # the decode row is not the kernel's
    q, ks, vs = _rand_case(99, q_lens, seq_lens, H, Hkv, D)
    scale = D ** -0.5
    kw = dict(paged=True, page_size=16, chunk=16)
    plain = _run_varlen(ops, q, ks, vs, q_lens, seq_lens, scale, "causal", (0, 0, 0), **kw)
    staged = _run_varlen(ops, q, ks, vs, q_lens, seq_lens, scale, "causal", (0, 0, 0), stage=True, **kw)
    assert np.array_equal(staged, plain)
    ref = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], cu_h, seq_lens, scale, "causal", 0, 0, 0)
    assert_close(staged, ref, "dense causal kernel, staged form")
    # contiguous K/V, the area as a member of the workspace, and more rows than needed
    bm = ops.attention_block_m(H, Hkv)
    idx, cu, sld = _build(ops, q_lens, seq_lens, 0, 0, 0, bm, 32, ops.MODES["causal"], 16, 0)
    assert idx.stage_rows == 0 and len(ops.alloc_workspace(idx, H, D, DEV, num_kv_heads=Hkv)) == 2
    idx.stage_rows = ops.dense_stage_rows(cu_h, sl_h, 0)
    ws = ops.alloc_workspace(idx, H, D, DEV, num_kv_heads=Hkv)
    assert len(ws) == 3 and tuple(ws[2].shape) == (2, idx.stage_rows, Hkv, 128)
    qd = _to_dev(q, torch.bfloat16)
    kd, vd = _to_dev(np.concatenate(ks, 0), torch.bfloat16), _to_dev(np.concatenate(vs, 0), torch.bfloat16)
    cu_k = torch.from_numpy(np.concatenate([[0], np.cumsum(seq_lens)]).astype(np.int32)).to(DEV)
    outs = []
    for work in (ws, ws[:2] + (ops.alloc_kv_stage(idx.stage_rows + 100, Hkv, DEV),), ws[:2]):
        out = torch.full((int(cu_h[-1]), H, D), float("nan"), dtype=torch.float32, device=DEV)
        ops.attention(qd, kd, vd, out, idx, cu, sld, scale, cu_seqlens_k=cu_k, workspace=work)
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[2])
    assert_close(outs[0].cpu().numpy(), ref, "dense causal kernel, staged form, contiguous K/V")
    with pytest.raises(RuntimeError):
        ops.attention(qd, kd, vd, out, idx, cu, sld, scale, cu_seqlens_k=cu_k,
                      kv_stage=ops.alloc_kv_stage(idx.stage_rows - 64, Hkv, DEV))
    with pytest.raises(RuntimeError):
        ops.attention(qd, kd, vd, out, idx, cu, sld, scale, cu_seqlens_k=cu_k,
                      kv_stage=ops.alloc_kv_stage(idx.stage_rows, Hkv + 1, DEV))


@pytest.mark.parametrize("case", [
    dict(H=8, Hkv=2, mode="triangle", paged=True, page=16, chunk=4),               # many pieces per row
    dict(H=32, Hkv=8, mode="triangle", paged=True, page=64, chunk=128),            # Qwen3-8B heads, one piece
    dict(H=16, Hkv=2, mode="causal", paged=False, chunk=8, dtype=torch.float16),   # two head chunks per K/V head
    dict(H=4, Hkv=1, mode="none", paged=True, page=32, chunk=16, out=torch.bfloat16),
    dict(H=8, Hkv=2, mode="triangle", paged=False, chunk=64, gate=2048, out=torch.float16, dtype=torch.float16),
    dict(H=8, Hkv=2, mode="triangle", paged=True, page=16, chunk=8, mixed=True),   # prompts in the same step
])
def test_decode_kernel_matches_oracle_and_work_list(case):
    """One-row requests whose row is dense (the decode rows of the pattern, every row of a causal or unmasked
    launch) run on the decode kernel: K/V straight to registers, 16 partial sums per workgroup merged at the
    end, pieces of long rows through the partial slots.  Same results as the oracle and as the same launch
    kept on the work list (TMX_ATTN_NO_DECODE); lengths around the 16-key groups, the 32-key tiles and the
    pieces' ends; with and without the caller's all-decode hint."""
    from vllm_trianglemix_amd._lib import TMX_ATTN_DECODE_ONLY, TMX_ATTN_DECODE_VALU, TMX_ATTN_NO_DECODE
    ops = _ops()
    H, Hkv, D = case["H"], case["Hkv"], 128
    seq_lens = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 700, 1023, 2049, 3000, 4100]
    q_lens = [1] * len(seq_lens)
    if case.get("mixed"):
        seq_lens += [300, 2500]
        q_lens += [300, 70]
    dtype, out_dtype = case.get("dtype", torch.bfloat16), case.get("out", torch.float32)
    q, ks, vs = _rand_case(8128 + H, q_lens, seq_lens, H, Hkv, D)
    if dtype == torch.float16:
        q, ks, vs = (np.float16(q).astype(np.float32), [np.float16(k).astype(np.float32) for k in ks],
                     [np.float16(v).astype(np.float32) for v in vs])
    # one key far above the others early in a long row: the running maximum must move, then stay
    ks[19][40] = 3.0 * q[19, ::H // Hkv]
    ks = [orc.bf16_round(k) if dtype == torch.bfloat16 else np.float16(k).astype(np.float32) for k in ks]
    scale = D ** -0.5
    mode, gate = case["mode"], case.get("gate", 0)
    pat = (4, 32, 64) if mode == "triangle" else (0, 0, 0)
    kw = dict(dtype=dtype, out_dtype=out_dtype, paged=case["paged"], page_size=case.get("page", 16), chunk=case["chunk"],
              gate=gate)
    cu = np.concatenate([[0], np.cumsum(q_lens)])
    ref = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], cu, seq_lens, scale, mode, *pat, tri_min_len=gate)
    variants = [("decode kernel", 0), ("decode kernel, sum on the vector units", TMX_ATTN_DECODE_VALU),
                ("work list", TMX_ATTN_NO_DECODE)]
    if not case.get("mixed"):
        variants.append(("decode kernel alone", TMX_ATTN_DECODE_ONLY))
    for name, flags in variants:
        got = _run_varlen(ops, q, ks, vs, q_lens, seq_lens, scale, mode, pat, flags=flags, **kw)
        assert_close(got, ref, f"decode rows: {name}, {str(dtype)[6:]}", out16=out_dtype != torch.float32)


def _paged_decode_batch(seq_lens, Hkv, D, page, dtype, seed, ks, vs, fp8_scales=None, extra_pages=0, pad_rows=0):
    """A paged cache holding every request's keys (shuffled pages), its block table, and the flat slot of each
    request's LAST key.  ``pad_rows``: block-table rows beyond the requests (padding of a captured batch)."""
    ops = _ops()
    B = len(seq_lens)
    nblk = sum((s + page - 1) // page for s in seq_lens) + 2 + extra_pages
    perm = np.random.default_rng(seed).permutation(nblk).astype(np.int32)
    cache_dtype = torch.uint8 if fp8_scales else dtype
    kc = torch.zeros((nblk, page, Hkv, D), dtype=cache_dtype, device=DEV)
    vc = torch.zeros_like(kc)
    width = max([(s + page - 1) // page for s in seq_lens] + [1]) + extra_pages
    bt = np.zeros((B + pad_rows, width), np.int32)
    last_slot = np.full((B + pad_rows,), -1, np.int64)
    nxt = 0
    for b, s in enumerate(seq_lens):
        for pb in range((s + page - 1) // page):
            bt[b, pb] = perm[nxt]
            nxt += 1
        if s == 0:
            continue
        pos = np.arange(s)
        slots = bt[b, pos // page].astype(np.int64) * page + pos % page
        last_slot[b] = slots[-1]
        sd = torch.from_numpy(slots).to(DEV)
        if fp8_scales:
            ops.reshape_and_cache_fp8(_to_dev(ks[b], dtype), _to_dev(vs[b], dtype), kc, vc, sd, *fp8_scales)
        else:
            ops.reshape_and_cache(_to_dev(ks[b], dtype), _to_dev(vs[b], dtype), kc, vc, sd)
    return kc, vc, torch.from_numpy(bt).to(DEV), last_slot


@pytest.mark.parametrize("case", [
    dict(H=8, Hkv=2),                                            # 32 tiles in all: pieces of 8 tiles, the merge launch
    dict(H=32, Hkv=8, page=64),                                  # Qwen3-8B heads
    dict(H=16, Hkv=2, dtype=torch.float16),                      # 8 query heads per workgroup
    dict(H=8, Hkv=1, out=torch.bfloat16, big=True),              # longer pieces (a larger step), bf16 output
    dict(H=16, Hkv=2, fp8=(0.75, 1.5)),                          # fp8 cache, descaled in registers
    dict(H=8, Hkv=1, fp8=(1.0, 1.0), dtype=torch.float16, out=torch.float16, big=True),
    dict(H=8, Hkv=2, many=True),                                 # more requests than threads of a workgroup
    dict(H=8, Hkv=2, binding="ctypes", fp8=(0.75, 1.5)),         # the ctypes binding of the same entry point
])
def test_decode_attention_with_a_fixed_launch_shape(monkeypatch, case):
    """tmx_decode_attention: a step of one-row requests with no index -- every workgroup lays the step's pieces out
    from seq_lens on the device and finds its own.  Lengths around the 16-key groups, the 32-key tiles and the pieces'
    ends; padding rows (length 0) are left alone; same rows as the oracle (every key of the sequence attended: what
    the pattern gives for a decode row)."""
    ops = _ops()
    monkeypatch.setattr(ops, "_BINDING", case.get("binding", "torch"))
    H, Hkv, D = case["H"], case["Hkv"], 128
    page = case.get("page", 16)
    dtype, out_dtype, fp8 = case.get("dtype", torch.bfloat16), case.get("out", torch.float32), case.get("fp8")
    seq_lens = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 0, 127, 128, 129, 255, 256, 257, 700, 1023, 2049, 3000, 4100, 0]
    if case.get("big"):
        seq_lens += [9000, 12000, 300, 8192, 8193]   # ~1500 tiles: pieces of 12 or 24 tiles, uneven ends
    if case.get("many"):
        seq_lens = seq_lens + [int(x) for x in np.random.default_rng(2).integers(0, 90, size=300)]  # 324 requests
    B = len(seq_lens)
    rnd = orc.bf16_round if dtype == torch.bfloat16 else (lambda x: np.float16(x).astype(np.float32))
    q, ks, vs = _rand_case(6400 + H, [1] * B, [max(s, 1) for s in seq_lens], H, Hkv, D)
    q, ks, vs = rnd(q), [rnd(k[:s]) for k, s in zip(ks, seq_lens)], [rnd(v[:s]) for v, s in zip(vs, seq_lens)]
    ks[20][40] = rnd(3.0 * q[20, ::H // Hkv])  # one key far above the others early in a long row
    kc, vc, bt, _ = _paged_decode_batch(seq_lens, Hkv, D, page, dtype, 5, ks, vs, fp8_scales=fp8)
    if fp8:
        seen = lambda x, sc: rnd(orc.fp8_cache_roundtrip(x, sc).astype(np.float32))  # noqa: E731
        ks, vs = [seen(k, fp8[0]) for k in ks], [seen(v, fp8[1]) for v in vs]
    ws = ops.decode_workspace(B, H, Hkv, D, DEV, bool(fp8))
    sld = torch.tensor(seq_lens, dtype=torch.int32, device=DEV)
    out = torch.full((B, H, D), 7.0, dtype=out_dtype, device=DEV)
    scale = D ** -0.5
    ops.decode_attention(_to_dev(q, dtype), kc, vc, out, bt, sld, scale, ws,
                         k_scale=fp8[0] if fp8 else 1.0, v_scale=fp8[1] if fp8 else 1.0)
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    live = [b for b, s in enumerate(seq_lens) if s > 0]
    pad = [b for b, s in enumerate(seq_lens) if s == 0]
    assert np.all(got[pad] == 7.0)
    ref = orc.varlen_gqa_attention(q[live], lambda i: ks[live[i]], lambda i: vs[live[i]], np.arange(len(live) + 1),
                                   [seq_lens[b] for b in live], scale, "causal", 0, 0, 0)
    what = f"decode rows, fixed launch shape{', fp8 cache' if fp8 else ''}, {str(dtype)[6:]}"
    assert_close(got[live], ref, what, out16=out_dtype != torch.float32, info=case)
    with pytest.raises(RuntimeError):  # the work area is the caller's to provide, at the size of the batch
        small = ops.decode_workspace(1, H, Hkv, D, DEV, bool(fp8))
        ops.decode_attention(_to_dev(q, dtype), kc, vc, out, bt, sld, scale, small)


@pytest.mark.parametrize("binding", ["torch", "ctypes"])
@pytest.mark.parametrize("dtype,scales,heads", [(torch.bfloat16, (0.75, 1.5), (8, 2)), (torch.float16, (1.0, 1.0), (16, 2)),
                                                (torch.bfloat16, (0.02, 0.04), (8, 1))])
def test_decode_attention_writes_an_fp8_cache_itself(monkeypatch, binding, dtype, scales, heads):
    """A decode step out of an fp8 cache with the step's rows handed to the launch (k_new / v_new / cache_slots): the
    launch quantises and stores them and attends to the row's own key as the cache holds it.  The cache bytes and the
    output must be IDENTICAL to tmx_reshape_and_cache_fp8 followed by the launch over the cache (the reference's
    order, rocm_attn.py:345-358), and the rows the oracle's.  Scales that saturate some values (0.02: |x| / scale
    beyond 448) are part of it; padding rows (slot -1, length 0) stay untouched."""
    ops = _ops()
    monkeypatch.setattr(ops, "_BINDING", binding)
    (H, Hkv), D, page = heads, 128, 16
    seq_lens = [1, 2, 16, 17, 32, 33, 64, 65, 0, 129, 700, 2049, 4100, 0]
    B = len(seq_lens)
    rnd = orc.bf16_round if dtype == torch.bfloat16 else (lambda x: np.float16(x).astype(np.float32))
    q, ks, vs = _rand_case(7100 + H, [1] * B, [max(s, 1) for s in seq_lens], H, Hkv, D)
    q, ks, vs = rnd(q), [rnd(k[:s]) for k, s in zip(ks, seq_lens)], [rnd(v[:s]) for v, s in zip(vs, seq_lens)]
    # the cache holds every sequence's context (all keys but the last); the last key is the step's own row
    ctx_k = [k[:max(s - 1, 0)] for k, s in zip(ks, seq_lens)]
    ctx_v = [v[:max(s - 1, 0)] for v, s in zip(vs, seq_lens)]
    kc, vc, bt, last_slot = _paged_decode_batch(seq_lens, Hkv, D, page, dtype, 9, ks, vs, fp8_scales=scales)
    # (the helper wrote all keys: wipe the step's own slots so that only the launch under test can fill them)
    k_rows = torch.zeros((B, Hkv, D), dtype=dtype, device=DEV)
    v_rows = torch.zeros((B, Hkv, D), dtype=dtype, device=DEV)
    for b, s_ in enumerate(seq_lens):
        if s_ > 0:
            k_rows[b] = _to_dev(ks[b][s_ - 1], dtype)
            v_rows[b] = _to_dev(vs[b][s_ - 1], dtype)
    slots = torch.from_numpy(last_slot).to(DEV)
    live_slots = slots[slots >= 0]
    kc_flat, vc_flat = kc.view(-1, Hkv, D), vc.view(-1, Hkv, D)
    want_k, want_v = kc_flat[live_slots].clone(), vc_flat[live_slots].clone()   # what reshape_and_cache_fp8 stored
    kc_flat[live_slots] = 0x55
    vc_flat[live_slots] = 0x55
    sld = torch.tensor(seq_lens, dtype=torch.int32, device=DEV)
    ws = ops.decode_workspace(B, H, Hkv, D, DEV, True)
    scale = D ** -0.5
    qd = _to_dev(q, dtype)
    # route A: the reference's order on a copy of the cache
    kc_a, vc_a = kc.clone(), vc.clone()
    ops.reshape_and_cache_fp8(k_rows, v_rows, kc_a, vc_a, slots, *scales)
    out_a = torch.full((B, H, D), 7.0, dtype=torch.float32, device=DEV)
    ops.decode_attention(qd, kc_a, vc_a, out_a, bt, sld, scale, ws, k_scale=scales[0], v_scale=scales[1])
    # route B: one launch
    out_b = torch.full((B, H, D), 7.0, dtype=torch.float32, device=DEV)
    ops.decode_attention(qd, kc, vc, out_b, bt, sld, scale, ws, k_scale=scales[0], v_scale=scales[1],
                         k_new=k_rows, v_new=v_rows, cache_slots=slots)
    torch.cuda.synchronize()
    assert torch.equal(kc.view(torch.uint8), kc_a.view(torch.uint8)) and torch.equal(vc.view(torch.uint8), vc_a.view(torch.uint8))
    assert torch.equal(kc_flat[live_slots].view(torch.uint8), want_k.view(torch.uint8))
    assert torch.equal(vc_flat[live_slots].view(torch.uint8), want_v.view(torch.uint8))
    assert torch.equal(out_a, out_b)
    live = [b for b, s_ in enumerate(seq_lens) if s_ > 0]
    pad = [b for b, s_ in enumerate(seq_lens) if s_ == 0]
    got = out_b.cpu().numpy()
    assert np.all(got[pad] == 7.0)
    seen = lambda x, sc: rnd(orc.fp8_cache_roundtrip(x, sc).astype(np.float32))  # noqa: E731
    ks8, vs8 = [seen(k, scales[0]) for k in ks], [seen(v, scales[1]) for v in vs]
    ref = orc.varlen_gqa_attention(q[live], lambda i: ks8[live[i]], lambda i: vs8[live[i]], np.arange(len(live) + 1),
                                   [seq_lens[b] for b in live], scale, "causal", 0, 0, 0)
    assert_close(got[live], ref, f"decode rows, fixed launch shape, fp8 cache written by the launch, {str(dtype)[6:]}",
                 info=(binding, scales, heads))
    del ctx_k, ctx_v


def test_decode_step_replays_under_a_graph():
    """A decode step through the backend, captured in a graph (torch.cuda.graph) the way the model runner does it:
    metadata from build_for_cudagraph_capture over persistent buffers, a warm-up call, the capture, then replays after
    seq_lens, the block table, the slot mapping, the cache and the step's q / k / v were changed IN PLACE (other
    lengths, other pages, fewer live requests: the rest are padding rows).  Every replay must give the oracle's
    rows and leave the step's K/V rows in the cache, for a triangle layer, a causal layer and an fp8-cache layer.
    (rocm_attn.py:63 declares ALWAYS for the reference's ROCm backend; this one replays decode steps.)"""
    import types
    from vllm_trianglemix_amd import TriangleMixConfig
    from vllm_trianglemix_amd.backend import AttentionCGSupport, TriangleMixAttentionBackend
    H, Hkv, D, page, cap, max_len = 8, 2, 128, 16, 6, 5000
    scale = D ** -0.5
    builder_cls = TriangleMixAttentionBackend.get_builder_cls()
    assert builder_cls.cudagraph_support is AttentionCGSupport.UNIFORM_SINGLE_TOKEN_DECODE
    vllm_config = types.SimpleNamespace(model_config=types.SimpleNamespace(max_model_len=max_len))
    builder = builder_cls(types.SimpleNamespace(block_size=page), ["l0", "l1"], vllm_config, torch.device(DEV))
    steps = [[4100, 1, 333, 2049, 64, 17], [900, 4999, 2, 1500], [31, 32, 33, 4097, 700, 5000]]
    width = (max_len + page - 1) // page + 1
    nblk_cap = cap * width + 4
    # the runner's persistent buffers
    seq_buf = torch.zeros((cap,), dtype=torch.int32, device=DEV)
    bt_buf = torch.zeros((cap, width), dtype=torch.int32, device=DEV)
    slot_buf = torch.full((cap,), -1, dtype=torch.int64, device=DEV)
    qsl = torch.arange(cap + 1, dtype=torch.int32, device=DEV)
    q_buf = torch.zeros((cap, H * D), dtype=torch.bfloat16, device=DEV)
    k_buf = torch.zeros((cap, Hkv * D), dtype=torch.bfloat16, device=DEV)
    v_buf = torch.zeros_like(k_buf)
    caches = {"bf16": torch.zeros((2, nblk_cap, page, Hkv, D), dtype=torch.bfloat16, device=DEV),
              "fp8": torch.zeros((2, nblk_cap, page, Hkv, D), dtype=torch.uint8, device=DEV)}
    k_scale, v_scale = 0.75, 1.5
    layer = types.SimpleNamespace(_k_scale_float=k_scale, _v_scale_float=v_scale)
    impl_cls = TriangleMixAttentionBackend.get_impl_cls()
    cfg = TriangleMixConfig(num_triangle_layers=1)
    layers = [("triangle layer", impl_cls(H, D, scale, Hkv, None, None, "auto", None, "decoder", None, trianglemix_config=cfg, layer_idx=0), "bf16"),
              ("causal layer", impl_cls(H, D, scale, Hkv, None, None, "auto", None, "decoder", None, trianglemix_config=cfg, layer_idx=1), "bf16"),
              ("fp8-cache layer", impl_cls(H, D, scale, Hkv, None, None, "fp8", None, "decoder", None, trianglemix_config=cfg, layer_idx=0), "fp8")]
    outs = [torch.zeros((cap, H * D), dtype=torch.bfloat16, device=DEV) for _ in layers]

    def common(max_seq):
        return types.SimpleNamespace(
            num_actual_tokens=cap, max_query_len=1, max_seq_len=max_seq, query_start_loc=qsl,
            query_start_loc_cpu=torch.arange(cap + 1, dtype=torch.int32), seq_lens=seq_buf,
            seq_lens_cpu=torch.ones((cap,), dtype=torch.int32), block_table_tensor=bt_buf, slot_mapping=slot_buf)

    def load_step(i, seq_lens):
        """Write step i's batch into the persistent buffers; returns what the oracle needs."""
        B = len(seq_lens)
        q, ks, vs = _rand_case(7000 + 10 * i, [1] * B, seq_lens, H, Hkv, D)
        q, ks, vs = orc.bf16_round(q), [orc.bf16_round(k) for k in ks], [orc.bf16_round(v) for v in vs]
        for name, cache in caches.items():
            # the context (all keys but the newest) is in the cache; the newest key / value is the step's row
            ctx_k, ctx_v = [k[:-1] for k in ks], [v[:-1] for v in vs]
            kc, vc, bt, _ = _paged_decode_batch([s - 1 for s in seq_lens], Hkv, D, page, torch.bfloat16, 40 + i, ctx_k, ctx_v,
                                                fp8_scales=(k_scale, v_scale) if name == "fp8" else None,
                                                extra_pages=1, pad_rows=cap - B)
            cache.zero_()
            cache[0, :kc.shape[0]], cache[1, :vc.shape[0]] = kc, vc
        bt_buf.zero_()
        bt_buf[:bt.shape[0], :bt.shape[1]] = bt
        btn = bt_buf.cpu().numpy()
        # (the page of every request's newest key: the layout above left one page to spare per request)
        for b, s in enumerate(seq_lens):
            if (s - 1) % page == 0:
                used = set(btn[:B, :].ravel().tolist()) | {0}
                btn[b, (s - 1) // page] = max(used) + 1 + b
        bt_buf.copy_(torch.from_numpy(btn).to(DEV))
        slots = np.full((cap,), -1, np.int64)
        for b, s in enumerate(seq_lens):
            slots[b] = int(btn[b, (s - 1) // page]) * page + (s - 1) % page
        slot_buf.copy_(torch.from_numpy(slots).to(DEV))
        seq_buf.zero_()
        seq_buf[:B] = torch.tensor(seq_lens, dtype=torch.int32, device=DEV)
        q_buf.zero_(); k_buf.zero_(); v_buf.zero_()
        q_buf[:B] = _to_dev(q, torch.bfloat16).view(B, H * D)
        k_buf[:B] = _to_dev(np.stack([k[-1] for k in ks]), torch.bfloat16).view(B, Hkv * D)
        v_buf[:B] = _to_dev(np.stack([v[-1] for v in vs]), torch.bfloat16).view(B, Hkv * D)
        return q, ks, vs, slots

    def run_layers(md):
        for (name, impl, cache), out in zip(layers, outs):
            kv = caches[cache]
            impl.forward(layer, q_buf, k_buf, v_buf, kv.view(torch.float8_e4m3fn) if cache == "fp8" else kv, md, output=out)

    def check(i, seq_lens, q, ks, vs, slots, how):
        B = len(seq_lens)
        torch.cuda.synchronize()
        ref16 = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], np.arange(B + 1), seq_lens, scale, "causal", 0, 0, 0)
        seen = lambda x, sc: orc.bf16_round(orc.fp8_cache_roundtrip(x, sc).astype(np.float32))  # noqa: E731
        ref8 = orc.varlen_gqa_attention(q, lambda b: seen(ks[b], k_scale), lambda b: seen(vs[b], v_scale), np.arange(B + 1),
                                        seq_lens, scale, "causal", 0, 0, 0)
        for (name, impl, cache), out in zip(layers, outs):
            got = out.float().cpu().numpy().reshape(cap, H, D)
            assert_close(got[:B], ref8 if cache == "fp8" else ref16, f"decode step {how}: {name}", out16=True, info=(i, seq_lens))
            assert np.all(got[B:] == 0.0), (name, "padding rows must be left alone")
        # the step's rows reached the cache through the same launches
        flat = caches["bf16"].view(2, nblk_cap * page, Hkv, D)
        sd = torch.from_numpy(slots[:B]).to(DEV)
        assert torch.equal(flat[0][sd], k_buf[:B].view(B, Hkv, D)) and torch.equal(flat[1][sd], v_buf[:B].view(B, Hkv, D))
        want8 = orc.fp8_e4m3fn_encode(k_buf[:B].float().cpu().numpy().reshape(B, Hkv, D) / np.float32(k_scale))
        assert np.array_equal(caches["fp8"].view(2, nblk_cap * page, Hkv, D)[0][sd].cpu().numpy(), want8)

    # eager first (also the warm-up the runner does ahead of a capture)
    data = load_step(0, steps[0])
    md_eager = builder.build(0, common(max(steps[0])))
    assert md_eager.uniform_decode and not md_eager.capturing
    run_layers(md_eager)
    check(0, steps[0], *data, "eager")
    for o in outs:
        o.zero_()
    md = builder.build_for_cudagraph_capture(common(max_len))
    assert md.capturing and md.uniform_decode
    run_layers(md)  # warm-up with the capture's own metadata (the work area of this batch size is made here)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run_layers(md)
    for i, seq_lens in enumerate(steps):
        for o in outs:
            o.zero_()
        data = load_step(i, seq_lens)
        graph.replay()
        check(i, seq_lens, *data, "replayed")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("scales", [(1.0, 1.0), (0.75, 1.5)])
def test_decode_kernel_fp8_cache(dtype, scales):
    """Decode rows out of an fp8 (e4m3fn) cache: the decode kernel converts in registers (unit scales: the
    exact conversion instruction; other scales: value = fp8 x scale rounded once to the 16-bit type).  Checked
    against the oracle on the dequantised values and against the work-list kernel's conversion stage."""
    from vllm_trianglemix_amd._lib import TMX_ATTN_DECODE_ONLY, TMX_ATTN_DECODE_VALU, TMX_ATTN_NO_DECODE
    ops = _ops()
    H, Hkv, D, page = 16, 2, 128, 16
    k_scale, v_scale = scales
    seq_lens = [1, 16, 17, 33, 500, 2049, 3333]
    q_lens = [1] * len(seq_lens)
    rnd = orc.bf16_round if dtype == torch.bfloat16 else (lambda x: np.float16(x).astype(np.float32))
    q, ks, vs = _rand_case(515, q_lens, seq_lens, H, Hkv, D)
    q = rnd(q)
    ks[4][7, 0, :4] = [400.0, -448.0, 0.0017, -0.002]   # the format's largest values and its subnormals
    vs[4][9, 1, :4] = [448.0, -300.0, 2.0 ** -9, -2.0 ** -8]
    B = len(seq_lens)
    nblk = sum((s + page - 1) // page for s in seq_lens) + 2
    perm = np.random.default_rng(3).permutation(nblk).astype(np.int32)
    kc = torch.zeros((nblk, page, Hkv, D), dtype=torch.uint8, device=DEV)
    vc = torch.zeros_like(kc)
    bt = np.zeros((B, max((s + page - 1) // page for s in seq_lens)), np.int32)
    nxt = 0
    for b, s in enumerate(seq_lens):
        for pb in range((s + page - 1) // page):
            bt[b, pb] = perm[nxt]
            nxt += 1
        pos = np.arange(s)
        slots = torch.from_numpy(bt[b, pos // page].astype(np.int64) * page + pos % page).to(DEV)
        ops.reshape_and_cache_fp8(_to_dev(ks[b], dtype), _to_dev(vs[b], dtype), kc, vc, slots, k_scale, v_scale)
    btd = torch.from_numpy(bt).to(DEV)
    seen = lambda x, sc: rnd(orc.fp8_cache_roundtrip(rnd(x), sc).astype(np.float32))
    kd, vd = [seen(k, k_scale) for k in ks], [seen(v, v_scale) for v in vs]
    scale = D ** -0.5
    ref = orc.varlen_gqa_attention(q, lambda b: kd[b], lambda b: vd[b], np.arange(B + 1), seq_lens, scale, "triangle",
                                   4, 32, 64)
    idx, cu, sld = _build(ops, q_lens, seq_lens, 4, 32, 64, ops.attention_block_m(H, Hkv), 32, ops.MODES["triangle"], 16,
                          block_table=btd, page_size=page)
    for name, flags in (("decode kernel", 0), ("decode kernel alone", TMX_ATTN_DECODE_ONLY),
                        ("decode kernel, sum on the vector units", TMX_ATTN_DECODE_VALU), ("work list", TMX_ATTN_NO_DECODE)):
        out = torch.full((B, H, D), float("nan"), dtype=torch.float32, device=DEV)
        ops.attention(_to_dev(q, dtype), kc.view(torch.float8_e4m3fn), vc.view(torch.float8_e4m3fn), out, idx, cu, sld,
                      scale, block_table=btd, flags=flags, k_scale=k_scale, v_scale=v_scale)
        torch.cuda.synchronize()
        assert_close(out.cpu().numpy(), ref, f"decode rows, fp8 cache: {name}, {str(dtype)[6:]}", info=scales)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("page,heads", [(16, (8, 2)), (64, (8, 2)), (16, (8, 1)), (64, (16, 2))])
def test_fp8_cache_read_in_place(dtype, page, heads):
    """fp8 (e4m3fn) paged K/V read by the attention kernel itself (head size 128): the kernel's conversion stage
    produces value = fp8 x scale rounded once to the 16-bit type, exactly what the gather pass produces, so the
    results must agree with the oracle on the dequantised values -- triangle prompts with dense trailing rows,
    a causal launch, decode rows with the short-queries hint, ragged lengths, non-power-of-two scales.  Head
    groupings: 8 / 2 (one head chunk per K/V head), 8 / 1 (the eighth of a 64 / 8-head model: two head chunks
    share one K/V head in the band and work-list roles) and 16 / 2."""
    ops = _ops()
    (H, Hkv), D = heads, 128
    k_scale, v_scale = 0.75, 1.5
    scale = D ** -0.5
    rnd = orc.bf16_round if dtype == torch.bfloat16 else (lambda x: np.float16(x).astype(np.float32))
    for mode, pat, q_lens, seq_lens, flags in (
            ("triangle", (4, 32, 64), [2300, 70, 1], [2300, 2500, 333], 0),
            ("causal", (0, 0, 0), [130, 1, 40], [130, 1000, 77], 0),
            ("triangle", (4, 32, 64), [1, 1, 1, 1], [5000, 31, 32, 2049], TMX_ATTN_SHORT_QUERIES)):
        q, ks, vs = _rand_case(1234 + page, q_lens, seq_lens, H, Hkv, D)
        q = rnd(q)
        ks[0][5, 0, :4] = [1000.0, -1000.0, 336.1, 1e-4]
        B = len(seq_lens)
        nblk = sum((s + page - 1) // page for s in seq_lens) + 2
        perm = np.random.default_rng(7).permutation(nblk).astype(np.int32)
        kc = torch.zeros((nblk, page, Hkv, D), dtype=torch.uint8, device=DEV)
        vc = torch.zeros_like(kc)
        bt = np.zeros((B, max((s + page - 1) // page for s in seq_lens)), np.int32)
        nxt = 0
        for b, s in enumerate(seq_lens):
            for pb in range((s + page - 1) // page):
                bt[b, pb] = perm[nxt]
                nxt += 1
            pos = np.arange(s)
            slots = torch.from_numpy(bt[b, pos // page].astype(np.int64) * page + pos % page).to(DEV)
            ops.reshape_and_cache_fp8(_to_dev(ks[b], dtype), _to_dev(vs[b], dtype), kc, vc, slots, k_scale, v_scale)
        btd = torch.from_numpy(bt).to(DEV)
        bm = ops.attention_block_m(H, Hkv)
        for resolve in (True, False):  # pages resolved by the index builder, or looked up by the kernel
            idx, cu, sld = _build(ops, q_lens, seq_lens, *pat, bm, 32, ops.MODES[mode], 8, 2048 if mode == "triangle" else 0,
                                  block_table=btd if resolve else None, page_size=page)
            out = torch.full((sum(q_lens), H, D), float("nan"), dtype=torch.float32, device=DEV)
            ops.attention(_to_dev(q, dtype), kc.view(torch.float8_e4m3fn), vc.view(torch.float8_e4m3fn), out, idx, cu, sld,
                          scale, block_table=btd, flags=flags, k_scale=k_scale, v_scale=v_scale)
            torch.cuda.synchronize()
            seen = lambda x, sc: rnd(orc.fp8_cache_roundtrip(rnd(x), sc).astype(np.float32))
            kd, vd = [seen(k, k_scale) for k in ks], [seen(v, v_scale) for v in vs]
            cu_h = np.concatenate([[0], np.cumsum(q_lens)])
            ref = orc.varlen_gqa_attention(q, lambda b: kd[b], lambda b: vd[b], cu_h, seq_lens, scale, mode, *pat,
                                           tri_min_len=2048 if mode == "triangle" else 0)
            got = out.cpu().numpy()
            if mode == "causal":
                # staged form of the dense kernel: the gather descales the fp8 rows (same rounding): the same bits
                out_s = torch.full_like(out, float("nan"))
                ops.attention(_to_dev(q, dtype), kc.view(torch.float8_e4m3fn), vc.view(torch.float8_e4m3fn), out_s, idx, cu,
                              sld, scale, block_table=btd, flags=flags, k_scale=k_scale, v_scale=v_scale,
                              kv_stage=_stage_for(ops, idx, q_lens, seq_lens, Hkv))
                assert torch.equal(out_s, out)
                idx.stage_rows = 0
            name = f"fp8 cache read in place, {H}/{Hkv} heads, {str(dtype)[6:]}"
            assert_close(got, ref, f"{name}: {'band + work list + decode' if mode == 'triangle' else 'dense causal + decode'}",
                         info=(mode, resolve))
            if mode == "triangle" and flags == 0:
                # the whole prompt's sparse q-blocks ran on the band workgroups (fp8 rows expanded in place, one
                # q-block ahead): the same launch with every q-block on the work list agrees with the oracle too,
                # and the two differ where they should -- the paths round the probabilities differently
                out2 = torch.full_like(out, float("nan"))
                ops.attention(_to_dev(q, dtype), kc.view(torch.float8_e4m3fn), vc.view(torch.float8_e4m3fn), out2, idx, cu,
                              sld, scale, block_table=btd, flags=TMX_ATTN_NO_BAND, k_scale=k_scale, v_scale=v_scale)
                got2 = out2.cpu().numpy()
                assert_close(got2, ref, f"{name}: work list only")
                assert np.abs(got2 - got).max() <= 2 * ATOL_F32_OUT
                if dtype == torch.bfloat16:
                    assert not np.array_equal(got2[:2048], got[:2048])
    with pytest.raises(RuntimeError):  # no in-place path for other head sizes
        q64 = torch.zeros((4, 2, 64), dtype=dtype, device=DEV)
        ops.attention(q64, torch.zeros((2, page, 2, 64), dtype=torch.uint8, device=DEV),
                      torch.zeros((2, page, 2, 64), dtype=torch.uint8, device=DEV), torch.empty_like(q64), idx, cu, sld,
                      1.0, block_table=btd)


def test_two_sources_equal_cache_only_and_backend_overlaps_every_step(monkeypatch):
    """k_new / v_new: a request's keys from position seq_len - q_len on are read from the step's own rows, the
    earlier ones from the paged cache.  Bit-identical to reading everything from a fully written cache, for a
    chunk continuation with a 32-aligned context (band path), an unaligned one (tiles that hold both kinds), decode
    rows and a whole prompt, with the triangle pattern and causally (dense kernel); the rows of the step need
    NOT be in the cache yet (the cache holds garbage there).  Then through the backend: with the write on the
    second stream the cache ends up bit-identical and the output too."""
    import types
    from vllm_trianglemix_amd import TriangleMixConfig
    from vllm_trianglemix_amd import backend as be
    ops = _ops()
    H, Hkv, D, page = 8, 2, 128, 16
    q_lens, seq_lens = [1024, 600, 1, 1, 300, 40], [3072, 2999, 2500, 17, 300, 2100]
    q, ks, vs = _rand_case(5150, q_lens, seq_lens, H, Hkv, D)
    scale = D ** -0.5
    B = len(seq_lens)
    cu = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    T = int(cu[-1])
    nblk = sum((s + page - 1) // page for s in seq_lens) + 2
    perm = np.random.default_rng(12).permutation(nblk).astype(np.int32)
    bt = np.zeros((B, max((s + page - 1) // page for s in seq_lens)), np.int32)
    full = torch.zeros((2, nblk, page, Hkv, D), dtype=torch.bfloat16, device=DEV)
    part = torch.full((2, nblk, page, Hkv, D), 7.0, dtype=torch.bfloat16, device=DEV)  # garbage where the step's rows go
    # the step's rows as strided slices of one fused projection output
    qkv = torch.zeros((T, (H + 2 * Hkv) * D), dtype=torch.bfloat16, device=DEV)
    slots = np.zeros((T,), np.int64)
    nxt = 0
    for b, (ql, s) in enumerate(zip(q_lens, seq_lens)):
        for pb in range((s + page - 1) // page):
            bt[b, pb] = perm[nxt]
            nxt += 1
        pos = np.arange(s)
        sl_all = torch.from_numpy(bt[b, pos // page].astype(np.int64) * page + pos % page).to(DEV)
        kd, vd = _to_dev(ks[b], torch.bfloat16), _to_dev(vs[b], torch.bfloat16)
        ops.reshape_and_cache(kd, vd, full[0], full[1], sl_all)
        ctx = s - ql
        if ctx:
            ops.reshape_and_cache(kd[:ctx], vd[:ctx], part[0], part[1], sl_all[:ctx])
        slots[cu[b]:cu[b + 1]] = sl_all[ctx:].cpu().numpy()
        qkv[cu[b]:cu[b + 1], H * D:(H + Hkv) * D] = kd[ctx:].reshape(ql, Hkv * D)
        qkv[cu[b]:cu[b + 1], (H + Hkv) * D:] = vd[ctx:].reshape(ql, Hkv * D)
    qkv[:, :H * D] = _to_dev(q, torch.bfloat16).view(T, H * D)
    qd, kn, vn = qkv.split([H * D, Hkv * D, Hkv * D], dim=-1)
    qd, kn, vn = qd.view(T, H, D), kn.view(T, Hkv, D), vn.view(T, Hkv, D)
    btd = torch.from_numpy(bt).to(DEV)
    bm = ops.attention_block_m(H, Hkv)
    for mode, pat, gate in (("triangle", (4, 32, 64), 2048), ("causal", (0, 0, 0), 0)):
        idx, cud, sld = _build(ops, q_lens, seq_lens, *pat, bm, 32, ops.MODES[mode], 16, gate, block_table=btd, page_size=page)
        want = torch.empty((T, H, D), dtype=torch.float32, device=DEV)
        got = torch.empty_like(want)
        ops.attention(qd, full[0], full[1], want, idx, cud, sld, scale, block_table=btd)
        ops.attention(qd, part[0], part[1], got, idx, cud, sld, scale, block_table=btd, k_new=kn, v_new=vn)
        torch.cuda.synchronize()
        assert torch.equal(got, want), (mode, (got - want).abs().max().item())
        ref = orc.varlen_gqa_attention(q, lambda b: ks[b], lambda b: vs[b], cu, seq_lens, scale, mode, *pat, tri_min_len=gate)
        assert_close(got.cpu().numpy(), ref, "two sources (k_new / v_new)")
        if mode == "causal":
            # staged form of the dense kernel: its gather reads both sources and makes the step's cache write
            cache = part.clone()
            got_s = torch.full_like(want, float("nan"))
            ops.attention(qd, cache[0], cache[1], got_s, idx, cud, sld, scale, block_table=btd, k_new=kn, v_new=vn,
                          cache_slots=torch.from_numpy(slots).to(DEV), kv_stage=_stage_for(ops, idx, q_lens, seq_lens, Hkv))
            torch.cuda.synchronize()
            assert torch.equal(got_s, want)
            sl_t = torch.from_numpy(slots).to(DEV)
            assert torch.equal(cache.view(2, nblk * page, Hkv, D)[:, sl_t], full.view(2, nblk * page, Hkv, D)[:, sl_t])
            idx.stage_rows = 0
    # through the backend, write beside the attention against write first
    results = []
    for write_mode in ("serial", "stream", "fused"):
        monkeypatch.setattr(be, "_CACHE_WRITE_MODE", write_mode)
        common = types.SimpleNamespace(
            num_actual_tokens=T, max_query_len=max(q_lens), max_seq_len=max(seq_lens),
            query_start_loc=torch.from_numpy(cu).to(DEV), query_start_loc_cpu=torch.from_numpy(cu),
            seq_lens=torch.tensor(seq_lens, dtype=torch.int32, device=DEV), seq_lens_cpu=torch.tensor(seq_lens, dtype=torch.int32),
            block_table_tensor=btd, slot_mapping=torch.from_numpy(slots).to(DEV))
        md = be.TriangleMixAttentionBackend.get_builder_cls()(types.SimpleNamespace(block_size=page), ["l0"], None,
                                                              torch.device(DEV)).build(0, common)
        assert not md.whole_prompts
        impl = be.TriangleMixAttentionBackend.get_impl_cls()(H, D, scale, Hkv, None, None, "auto", None, "decoder", None,
                                                             trianglemix_config=TriangleMixConfig(num_triangle_layers=1),
                                                             layer_idx=0)
        cache = part.clone()
        out = torch.zeros((T, H * D), dtype=torch.bfloat16, device=DEV)
        impl.forward(None, qd, kn, vn, cache, md, output=out)
        torch.cuda.current_stream().synchronize()
        results.append((out.clone(), cache.clone()))
    for other in results[1:]:
        assert torch.equal(results[0][0], other[0]) and torch.equal(results[0][1], other[1])
    used = torch.zeros(nblk * page, dtype=torch.bool, device=DEV)
    for b, s in enumerate(seq_lens):
        pos = np.arange(s)
        used[torch.from_numpy(bt[b, pos // page].astype(np.int64) * page + pos % page).to(DEV)] = True
    assert torch.equal(results[2][1].view(2, nblk * page, Hkv, D)[:, used], full.view(2, nblk * page, Hkv, D)[:, used])


def test_eager_module_fp32_inputs_match_the_reference_outputs(golden_attention):
    """fp32 inputs to the standalone module (the reference's own CPU configuration runs fp32): everything in fp32
    on the device, against the reference's fp32 outputs of the golden cases (explicit masks, the per-model helper
    above and below the 2048-token gate, no mask at all) and the sampled rows of its 2048-token configuration."""
    import json, os
    from vllm_trianglemix_amd import AscendTriangleMixAttention, TriangleMixConfig, TriangleMixMaskGenerator
    meta, arrays = golden_attention
    for c in meta["cases"]:
        if c.get("run_dtype"):
            continue
        shape = (c["B"], c["N"], c["H"], c["D"])
        q, k, v = (torch.from_numpy(orc.seeded_normal(c["seed"] * 3 + t, shape)).to(DEV) for t in range(3))
        cfg = TriangleMixConfig(*c["cfg"], num_triangle_layers=c.get("ntl"))
        if c["mask"] == "helper":
            out = AscendTriangleMixAttention(c["H"], c["D"], c["scale"], cfg, layer_idx=c["layer"],
                                             num_layers=c["num_layers"])(q, k, v)
        else:
            am = TriangleMixMaskGenerator(cfg).get_attention_mask(c["N"], 0, DEV, torch.float32,
                                                                  use_triangle=(c["mask"] == "triangle"))
            out = AscendTriangleMixAttention(c["H"], c["D"], c["scale"])(q, k, v, attn_mask=am)
        assert out.dtype == torch.float32 and tuple(out.shape) == shape and out.is_contiguous()
        err = np.abs(out.cpu().numpy() - arrays[c["name"]]).max()
        assert err <= 2e-6, (c["name"], err)  # fp32 against fp32: summation order only
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    m0 = json.load(open(os.path.join(gold, "attention_config0.json")))
    z = np.load(os.path.join(gold, "attention_config0.npz"))
    N, H, D = m0["N"], m0["H"], m0["D"]
    q, k, v = (torch.from_numpy(orc.seeded_normal(m0["seed"] * 3 + t, (1, N, H, D))).to(DEV) for t in range(3))
    am = TriangleMixMaskGenerator(TriangleMixConfig(*m0["cfg"])).get_attention_mask(N, 0, DEV, torch.float32, use_triangle=True)
    out = AscendTriangleMixAttention(H, D, m0["scale"])(q, k, v, attn_mask=am)
    assert np.abs(out[0].cpu().numpy()[z["rows"]] - z["out_rows"]).max() <= 2e-6
    # a fully masked row is NaN, as the reference's softmax over all -inf is
    am2 = am.clone()
    am2[7, :] = float("-inf")
    assert torch.isnan(AscendTriangleMixAttention(H, D, m0["scale"])(q, k, v, attn_mask=am2)[0, 7]).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_eager_module_broadcast_masks(dtype):
    """attn_mask of every shape that broadcasts to [batch, heads, N, N] (the reference simply adds it,
    ascend_trianglemix.py:83-84, 96-97): per request, per head, both, a single row vector, with and without
    the TriangleMix mask on top."""
    from vllm_trianglemix_amd import AscendTriangleMixAttention, TriangleMixConfig
    B, N, H, D = 2, 2100, 2, 64
    rng = np.random.default_rng(42)
    q, k, v = (orc.seeded_normal(900 + t, (B, N, H, D)) for t in range(3))
    scale = D ** -0.5
    cfg = TriangleMixConfig(4, 32, 64, num_triangle_layers=1)
    for shape in ((B, H, N, N), (B, 1, N, N), (1, H, N, N), (N, N), (1, 1, 1, N)):
        if dtype == torch.float32 and shape == (B, H, N, N):
            continue  # (kept small: the other shapes cover both strides)
        mask = orc.bf16_round(rng.standard_normal(shape).astype(np.float32))
        mask[rng.random(shape) < 0.2] = -np.inf
        full = np.broadcast_to(mask.reshape((1,) * (4 - len(shape)) + shape), (B, H, N, N))
        for with_tri in (False, True):
            mod = (AscendTriangleMixAttention(H, D, scale, cfg, layer_idx=0, num_layers=2) if with_tri
                   else AscendTriangleMixAttention(H, D, scale))
            out = mod(*(torch.from_numpy(x).to(DEV).to(dtype) for x in (q, k, v)), attn_mask=torch.from_numpy(mask).to(DEV))
            got = out.float().cpu().numpy()
            rows = [0, 5, 700, N - 70, N - 1]
            for b in range(B):
                for h in range(H):
                    ref = orc.varlen_gqa_attention(q[b][:, h:h + 1], lambda _: k[b][:, h:h + 1], lambda _: v[b][:, h:h + 1],
                                                   [0, N], [N], scale, "triangle" if with_tri else "none", 4, 32, 64,
                                                   bias=np.ascontiguousarray(full[b, h]))
                    e = np.abs(got[b, rows, h] - ref[rows, 0])
                    ok = np.isnan(ref[rows, 0]) | (e <= (1e-5 if dtype == torch.float32 else ATOL_F32_OUT + np.abs(ref[rows, 0]) * 2.0 ** -8))
                    assert ok.all(), (shape, with_tri, b, h, np.nanmax(e))
