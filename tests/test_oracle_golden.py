"""Pins the CPU oracle (numpy + C restatement) to fixtures generated from the
reference's own TriangleMix files (oracle/gen_golden.py).  CPU only."""
import ctypes
import hashlib
import os
import subprocess

import numpy as np
import pytest

from oracle import trianglemix_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pack(m):
    return np.packbits(np.asarray(m) > 0, axis=None).tobytes()


def _check(rec, mask):
    assert int((np.asarray(mask) > 0).sum()) == rec["nnz"]
    raw = _pack(mask)
    assert hashlib.sha256(raw).hexdigest() == rec["sha256"]
    if "hex" in rec:
        assert raw.hex() == rec["hex"]


def test_vectorised_masks_match_reference(golden_masks):
    assert len(golden_masks) >= 60
    for c in golden_masks:
        si, sl, last, n = c["si"], c["sl"], c["last"], c["n"]
        _check(c["triangle"], orc.triangle_mask(n, si, sl, last))
        if "streaming" in c:
            _check(c["streaming"], orc.streaming_mask(n, si, sl))
            _check(c["last_mask"], orc.last_mask(n, si, sl, last))
            _check(c["middle"], orc.middle_mask(n, si, sl, last))
            _check(c["dense"], orc.dense_mask(n))
        for key in c:
            if key.startswith("additive_"):
                tri = key.endswith("_tri")
                m = orc.triangle_mask(n, si, sl, last) if tri else orc.dense_mask(n)
                add = orc.additive_mask(m)
                assert np.all((add == 0) | np.isneginf(add))
                _check(c[key], add == 0)


def test_loop_masks_match_reference(golden_masks):
    """The literal double loops (small N only: pure-Python)."""
    for c in golden_masks:
        si, sl, last, n = c["si"], c["sl"], c["last"], c["n"]
        if n > 70 or "streaming" not in c:
            continue
        _check(c["streaming"], orc.streaming_mask_loops(n, si, sl))
        _check(c["last_mask"], orc.last_mask_loops(n, si, sl, last))
        _check(c["middle"], orc.middle_mask_loops(n, si, sl, last))
        _check(c["triangle"], orc.triangle_mask_loops(n, si, sl, last))


def test_partition_and_intervals(golden_masks):
    """streaming / last / middle are disjoint and cover tril; row intervals
    reproduce the bitmap."""
    for c in golden_masks:
        si, sl, last, n = c["si"], c["sl"], c["last"], c["n"]
        if n > 512:
            continue
        s, l, m = orc.streaming_mask(n, si, sl), orc.last_mask(n, si, sl, last), orc.middle_mask(n, si, sl, last)
        assert not (s & l).any() and not (s & m).any() and not (l & m).any()
        assert np.array_equal(s | l | m, orc.dense_mask(n))
        tri = orc.triangle_mask(n, si, sl, last)
        rebuilt = np.zeros_like(tri)
        for i in range(n):
            iv = orc.row_intervals(i, n, si, sl, last)
            for a, b in iv:
                rebuilt[i, a:b + 1] = True
            assert all(iv[x][1] + 1 < iv[x + 1][0] for x in range(len(iv) - 1))
        assert np.array_equal(tri, rebuilt)
        assert orc.triangle_nnz(n, si, sl, last) == c["triangle"]["nnz"]


def test_known_values_from_reference_test_file():
    """tests/trianglemix_attention_test.py:24-49 of the reference, with the
    values the code really yields (its printed expectation for [15,1] is off)."""
    m = orc.streaming_mask(24, 4, 8)
    assert m[10, 3] and m[15, 8] and m[15, 1] and m[15, 4] and not m[15, 5] and not m[15, 6] and m[15, 7]
    assert int(m.sum()) == 245
    assert int(orc.last_mask(24, 4, 8, 16).sum()) == 55
    assert int(orc.middle_mask(24, 4, 8, 16).sum()) == 0
    assert int(orc.triangle_mask(24, 4, 8, 16).sum()) == 300


def test_selection_tables(golden_selection):
    for rec in golden_selection["should_use_triangle"]:
        idx = rec["triangle_layer_indices"] or []
        assert rec["stored_indices"] == list(idx)
        got = [orc.should_use_triangle(i, rec["num_triangle_layers"], idx) for i in range(14)]
        assert got == rec["table"]
    for rec in golden_selection["helper"]:
        for d in rec["decisions"]:
            kind = orc.helper_mask_kind(d["n"], d["layer"], rec["num_triangle_layers"],
                                        rec["triangle_layer_indices"] or [])
            assert kind == d["kind"], (rec, d)
        for n, sha in rec.get("triangle_sha256", {}).items():
            assert hashlib.sha256(_pack(orc.triangle_mask(int(n), 4, 8, 16))).hexdigest() == sha
    for rec in golden_selection["analyzer"]:
        assert orc.analyzer_select(rec["grads"], rec["k"]) == rec["layers"]


def test_optimize_mask_and_ascend_block(golden_selection):
    om = golden_selection["optimize_mask"]
    m = np.full((4, 4), -np.inf, np.float32)
    m[0, 0] = 0
    m[1, 0:2] = 0
    m[2, 0:3] = 0
    m[3, :] = 0
    assert _pack(m == 0).hex() == om["in_is_zero_hex"]
    assert np.array_equal(orc.optimize_mask_for_npu(m), np.asarray(om["out"], np.float32))
    for rec in golden_selection["ascend_block"]:
        bm = orc.ascend_block_mask(rec["n"], rec["si"], rec["sl"], rec["last"], rec["block"])
        assert list(bm.shape) == rec["shape"]
        _check(rec, bm)


def _inputs(c):
    shape = (c["B"], c["N"], c["H"], c["D"])
    q, k, v = (orc.seeded_normal(c["seed"] * 3 + t, shape) for t in range(3))
    assert hashlib.sha256(q.tobytes() + k.tobytes() + v.tobytes()).hexdigest() == c["inputs_sha256"]
    return q, k, v


def _case_mask(c):
    si, sl, last = c["cfg"]
    n = c["N"]
    if c["mask"] == "triangle":
        return orc.additive_mask(orc.triangle_mask(n, si, sl, last))
    if c["mask"] == "dense":
        return orc.additive_mask(orc.dense_mask(n))
    kind = orc.helper_mask_kind(n, c["layer"], c.get("ntl"), [])
    if kind == "none":
        return None
    return orc.additive_mask(orc.triangle_mask(n, si, sl, last) if kind == "triangle" else orc.dense_mask(n))


def test_eager_attention_matches_reference(golden_attention):
    meta, arrays = golden_attention
    assert meta["gqa_in_eager_class"] == "RuntimeError"
    for c in meta["cases"]:
        q, k, v = _inputs(c)
        ref = arrays[c["name"]]
        got32 = orc.eager_attention(q, k, v, c["scale"], _case_mask(c), dtype=np.float32)
        got64 = orc.eager_attention(q, k, v, c["scale"], _case_mask(c), dtype=np.float64)
        assert not np.isnan(ref).any()
        if c.get("run_dtype") == "bfloat16":
            # the reference's own bf16 run: only its distance to fp32 is recorded
            assert np.abs(got64 - ref).max() < 5e-2
            continue
        assert np.abs(got32 - ref).max() < 2e-5, c["name"]
        assert np.abs(got64 - ref).max() < 2e-5, c["name"]


def test_varlen_reference_agrees_with_eager(golden_attention):
    """The ragged/grouped-head reference used for backend tests collapses to
    the eager oracle when every request is a whole prompt with Hkv == H."""
    meta, arrays = golden_attention
    for c in meta["cases"]:
        if c["mask"] == "helper" or c.get("run_dtype"):
            continue
        q, k, v = _inputs(c)
        b, n, h, d = q.shape
        si, sl, last = c["cfg"]
        mode = "triangle" if c["mask"] == "triangle" else "causal"
        out = orc.varlen_gqa_attention(q.reshape(b * n, h, d), lambda i: k[i], lambda i: v[i],
                                       [i * n for i in range(b + 1)], [n] * b, c["scale"], mode, si, sl, last)
        assert np.abs(out.reshape(b, n, h, d) - arrays[c["name"]]).max() < 2e-5


@pytest.fixture(scope="module")
def c_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libtmx_oracle.so"))
    return lib


def test_c_oracle_masks(c_oracle, golden_masks):
    fp = ctypes.POINTER(ctypes.c_float)
    for c in golden_masks:
        si, sl, last, n = c["si"], c["sl"], c["last"], c["n"]
        if n > 512:
            continue
        buf = np.empty((n, n), np.float32)
        p = buf.ctypes.data_as(fp)
        assert c_oracle.tmx_oracle_triangle_mask(p, n, si, sl, last) == 0
        _check(c["triangle"], buf)
        if "streaming" in c:
            c_oracle.tmx_oracle_streaming_mask(p, n, si, sl)
            _check(c["streaming"], buf)
            c_oracle.tmx_oracle_last_mask(p, n, si, sl, last)
            _check(c["last_mask"], buf)
            c_oracle.tmx_oracle_middle_mask(p, n, si, sl, last)
            _check(c["middle"], buf)
            c_oracle.tmx_oracle_dense_mask(p, n)
            _check(c["dense"], buf)
            c_oracle.tmx_oracle_triangle_mask(p, n, si, sl, last)
            c_oracle.tmx_oracle_to_additive(p, ctypes.c_size_t(n * n))
            assert np.all((buf == 0) | np.isneginf(buf))
            _check(c["triangle"], buf == 0)


def test_c_oracle_attention(c_oracle, golden_attention):
    meta, arrays = golden_attention
    fp = ctypes.POINTER(ctypes.c_float)
    for c in meta["cases"]:
        if c.get("run_dtype") or c["N"] > 600:
            continue
        q, k, v = _inputs(c)
        mask = _case_mask(c)
        out = np.empty_like(q)
        mp = None if mask is None else np.ascontiguousarray(mask, np.float32).ctypes.data_as(fp)
        rc = c_oracle.tmx_oracle_eager_attention(q.ctypes.data_as(fp), k.ctypes.data_as(fp), v.ctypes.data_as(fp),
                                                 mp, out.ctypes.data_as(fp), c["B"], c["N"], c["H"], c["D"],
                                                 ctypes.c_float(c["scale"]))
        assert rc == 0
        assert np.abs(out - arrays[c["name"]]).max() < 2e-5, c["name"]


def test_block_index_closed_form_equals_bruteforce():
    rng = np.random.default_rng(7)
    cases = [([24], [24], 4, 8, 16), ([70], [70], 4, 32, 64), ([300], [300], 4, 32, 64),
             ([5, 1, 40], [50, 90, 40], 4, 32, 64), ([1], [1], 0, 0, 0), ([33, 0, 17], [33, 12, 100], 2, 5, 9)]
    for _ in range(25):
        b = int(rng.integers(1, 4))
        seq = [int(rng.integers(1, 200)) for _ in range(b)]
        ql = [int(rng.integers(0, s + 1)) for s in seq]
        cases.append((ql, seq, int(rng.integers(0, 9)), int(rng.integers(0, 40)), int(rng.integers(0, 70))))
    for ql, seq, si, sl, last in cases:
        for bm, bn in ((16, 32), (64, 32), (32, 32), (64, 64), (16, 16)):
            for mode in (orc.MODE_TRIANGLE, orc.MODE_CAUSAL, orc.MODE_NONE):
                a = orc.block_index_bruteforce(ql, seq, si, sl, last, bm, bn, mode)
                b_ = orc.block_index_closed_form(ql, seq, si, sl, last, bm, bn, mode)
                for x, y in zip(a, b_):
                    assert np.array_equal(x, y), (ql, seq, si, sl, last, bm, bn, mode)
                masks = orc.expand_index_to_mask(*a, ql, seq, si, sl, last, bm, bn, mode)
                for bi, m in enumerate(masks):
                    assert np.array_equal(m, orc._mode_mask(mode, seq[bi], ql[bi], si, sl, last))
                items, part_ptr = orc.work_items(a[0], 4)
                cover = np.zeros(len(a[1]), np.int32)
                for qb, start, nt, slot in items:
                    cover[start:start + nt] += 1
                    assert a[0][qb] <= start and start + nt <= a[0][qb + 1]
                assert np.all(cover == 1)
                slots = sorted(int(s) for s in items[:, 3] if s >= 0)
                assert slots == list(range(len(slots))) and int(part_ptr[-1]) == len(slots)


def test_default_density_numbers():
    """Figures quoted for the default (4, 32, 64) pattern."""
    assert orc.triangle_nnz(32768, 4, 32, 64) == 3304518
    rp, ci, _, _ = orc.block_index_closed_form([32768], [32768], 4, 32, 64, 64, 64, orc.MODE_TRIANGLE)
    assert len(ci) == 2042
    rp, ci, _, _ = orc.block_index_closed_form([8192], [8192], 4, 32, 64, 64, 64, orc.MODE_TRIANGLE)
    assert len(ci) == 506


def test_oracle_matches_reference_config0():
    """The reference's own CPU-runnable configuration (2048 tokens, 16 heads, head dim 128, defaults):
    oracle rows vs the sampled rows of the reference's output."""
    import json, os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(gold, "attention_config0.json")))
    z = np.load(os.path.join(gold, "attention_config0.npz"))
    rows, want = z["rows"], z["out_rows"]
    N, H, D, seed = meta["N"], meta["H"], meta["D"], meta["seed"]
    q, k, v = (orc.seeded_normal(seed * 3 + t, (1, N, H, D)) for t in range(3))
    assert orc.triangle_nnz(N, *meta["cfg"]) == meta["mask_nnz"]
    for h in (0, 7, 15):
        for r_i in (0, 1, len(rows) // 2, len(rows) - 64, len(rows) - 63, len(rows) - 1):
            i = int(rows[r_i])
            allow = np.zeros((N,), bool)
            for lo, hi in orc.row_intervals(i, N, *meta["cfg"]):
                allow[lo:hi + 1] = True
            s_ = (k[0, :, h].astype(np.float64) @ q[0, i, h].astype(np.float64)) * meta["scale"]
            s_[~allow] = -np.inf
            pr = np.exp(s_ - s_.max())
            pr /= pr.sum()
            got = pr @ v[0, :, h].astype(np.float64)
            assert np.abs(got - want[r_i, h]).max() < 2e-5, (i, h)
