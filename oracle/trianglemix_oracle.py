"""CPU oracle for the TriangleMix hot path.  TEST INFRASTRUCTURE ONLY.

This module restates, on the CPU with numpy, what the reference's
TriangleMix code computes.  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it; the product package
never does (it fails loudly when the HIP library is missing).

Parity is pinned: ``oracle/gen_golden.py`` imports the reference's own four
TriangleMix source files in the build container and writes the fixtures in
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function here
against those fixtures.

Reference files restated (paths relative to the reference checkout):

* ``vllm/attention/trianglemix_attention.py``  (mask generator, helper)
* ``vllm/attention/ascend_trianglemix.py``     (eager masked attention)
* ``vllm/attention/ascend_npu_trianglemix.py`` (block-level mask, secondary)

The second half of the file derives, from the same token-level predicate,
the artefacts the GPU path produces (packed block-sparse index, work items)
so they can be compared bit-for-bit.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------
# Token-level predicate and masks
# --------------------------------------------------------------------------


def streaming_allowed(i: int, j: int, si: int, sl: int) -> bool:
    """trianglemix_attention.py:82-89 -- sink columns ``j <= si`` or window
    ``i - j <= sl``, both under causality ``i >= j``."""
    if i >= j and j <= si:
        return True
    if i >= j and (i - j) <= sl:
        return True
    return False


def last_allowed(i: int, j: int, n: int, si: int, sl: int, last: int) -> bool:
    """trianglemix_attention.py:126-129."""
    return i >= j and (n - i) < last and j > si and (i - j) > sl


def middle_allowed(i: int, j: int, n: int, si: int, sl: int, last: int) -> bool:
    """trianglemix_attention.py:166-169."""
    return i >= j and (n - i) >= last and j > si and (i - j) > sl


def triangle_allowed(i: int, j: int, n: int, si: int, sl: int, last: int) -> bool:
    """trianglemix_attention.py:192-196: clamp(streaming + last, 0, 1)."""
    return streaming_allowed(i, j, si, sl) or last_allowed(i, j, n, si, sl, last)


def streaming_mask_loops(n: int, si: int, sl: int) -> np.ndarray:
    """Faithful O(N^2) double loop, trianglemix_attention.py:53-91."""
    m = np.zeros((n, n), dtype=np.float32)
    for i in range(n):
        for j in range(n):
            if streaming_allowed(i, j, si, sl):
                m[i, j] = 1
    return m


def last_mask_loops(n: int, si: int, sl: int, last: int) -> np.ndarray:
    """trianglemix_attention.py:93-131."""
    m = np.zeros((n, n), dtype=np.float32)
    for i in range(n):
        for j in range(n):
            if last_allowed(i, j, n, si, sl, last):
                m[i, j] = 1
    return m


def middle_mask_loops(n: int, si: int, sl: int, last: int) -> np.ndarray:
    """trianglemix_attention.py:133-171."""
    m = np.zeros((n, n), dtype=np.float32)
    for i in range(n):
        for j in range(n):
            if middle_allowed(i, j, n, si, sl, last):
                m[i, j] = 1
    return m


def triangle_mask_loops(n: int, si: int, sl: int, last: int) -> np.ndarray:
    """trianglemix_attention.py:173-198."""
    t = streaming_mask_loops(n, si, sl) + last_mask_loops(n, si, sl, last)
    return np.clip(t, 0, 1)


def _ij(n: int, ctx: int = 0, n_keys: Optional[int] = None):
    i = (np.arange(n, dtype=np.int64) + ctx)[:, None]
    j = np.arange(n if n_keys is None else n_keys, dtype=np.int64)[None, :]
    return i, j


def streaming_mask(n: int, si: int, sl: int) -> np.ndarray:
    """Vectorised form of :func:`streaming_mask_loops` (bool)."""
    i, j = _ij(n)
    return (i >= j) & ((j <= si) | ((i - j) <= sl))


def last_mask(n: int, si: int, sl: int, last: int) -> np.ndarray:
    i, j = _ij(n)
    return (i >= j) & ((n - i) < last) & (j > si) & ((i - j) > sl)


def middle_mask(n: int, si: int, sl: int, last: int) -> np.ndarray:
    i, j = _ij(n)
    return (i >= j) & ((n - i) >= last) & (j > si) & ((i - j) > sl)


def triangle_mask(n: int, si: int, sl: int, last: int, q_len: Optional[int] = None,
                  n_full: Optional[int] = None) -> np.ndarray:
    """Closed form of the triangle mask (bool ``[q_len, n]``).

    ``allowed(i, j) = j <= i and (j <= si or i - j <= sl or n - i < last)``.
    With ``q_len < n`` the rows are the LAST ``q_len`` positions of the
    length-``n`` sequence (a query chunk that follows ``n - q_len`` cached
    tokens), which is how the attention backend indexes rows.  ``n_full > n``:
    the rows and keys are the first ``n`` of a sequence whose mask is the one of
    length ``n_full`` (a prompt still arriving in chunks): the result is the
    block ``[n - q_len:n, :n]`` of the length-``n_full`` mask.
    """
    q_len = n if q_len is None else q_len
    n_full = n if n_full is None else max(n, n_full)
    i, j = _ij(q_len, ctx=n - q_len, n_keys=n)
    return (j <= i) & ((j <= si) | ((i - j) <= sl) | ((n_full - i) < last))


def dense_mask(n: int, q_len: Optional[int] = None) -> np.ndarray:
    """trianglemix_attention.py:218-219 (tril of ones)."""
    q_len = n if q_len is None else q_len
    i, j = _ij(q_len, ctx=n - q_len, n_keys=n)
    return j <= i


def additive_mask(mask01: np.ndarray) -> np.ndarray:
    """trianglemix_attention.py:254-258: 0 where attended, -inf elsewhere."""
    return np.where(mask01 > 0, np.float32(0.0), np.float32(-np.inf)).astype(np.float32)


def optimize_mask_for_npu(mask: np.ndarray) -> np.ndarray:
    """ascend_trianglemix.py:131-141: -inf becomes -1e6."""
    if mask.size == 0:
        return mask
    return np.where(np.isinf(mask) & (mask < 0), np.asarray(-1e6, dtype=mask.dtype), mask)


def row_intervals(i: int, n: int, si: int, sl: int, last: int) -> List[Tuple[int, int]]:
    """Attended key set of row ``i`` as sorted, disjoint inclusive intervals."""
    if i < 0 or i >= n:
        return []
    if (n - i) < last:
        return [(0, i)]
    out: List[Tuple[int, int]] = []
    a_hi = min(i, si)
    b_lo = max(0, i - sl) if sl >= 0 else i + 1
    if a_hi >= 0:
        out.append((0, a_hi))
    if b_lo <= i:
        if out and b_lo <= out[-1][1] + 1:
            out[-1] = (out[-1][0], max(out[-1][1], i))
        else:
            out.append((b_lo, i))
    return out


# --------------------------------------------------------------------------
# Layer selection and the helper's length gate
# --------------------------------------------------------------------------


def should_use_triangle(layer_idx: int, num_triangle_layers: Optional[int],
                        triangle_layer_indices: Sequence[int]) -> bool:
    """trianglemix_attention.py:263-278."""
    if triangle_layer_indices:
        return layer_idx in triangle_layer_indices
    if num_triangle_layers is not None:
        return layer_idx < num_triangle_layers
    return False


def helper_mask_kind(seq_len: int, layer_idx: int, num_triangle_layers: Optional[int],
                     triangle_layer_indices: Sequence[int]) -> str:
    """trianglemix_attention.py:305-340 as a decision: which mask
    ``TriangleMixAttention.get_attn_mask`` hands back.

    ``"none"`` (``seq_len <= 2048`` -- the caller then applies NO mask, not
    even a causal one, ascend_trianglemix.py:76-97), ``"triangle"`` or
    ``"dense"``.
    """
    idx = list(triangle_layer_indices)
    if num_triangle_layers is not None and not idx:
        idx = list(range(num_triangle_layers))
    if seq_len <= 2048:
        return "none"
    return "triangle" if should_use_triangle(layer_idx, num_triangle_layers, idx) else "dense"


def analyzer_select(grads_per_layer: Sequence[Sequence[float]], k: int) -> List[int]:
    """trianglemix_attention.py:375-405: lowest-mean-gradient ``k`` layers,
    layers with no record count as +inf, ties keep layer order, result sorted."""
    avg = []
    for idx, g in enumerate(grads_per_layer):
        avg.append((idx, sum(g) / len(g) if len(g) else float("inf")))
    chosen = [idx for idx, _ in sorted(avg, key=lambda x: x[1])[:k]]
    return sorted(chosen)


def ascend_block_mask(seq_len: int, si_tok: int, sl_tok: int, last_tok: int, block: int = 64) -> np.ndarray:
    """ascend_npu_trianglemix.py:101-124 (block-level rules, NOT a cover of
    the token mask: parameters are divided by the block size, floor 1)."""
    nb = (seq_len + block - 1) // block
    si = max(1, si_tok // block)
    sl = max(1, sl_tok // block)
    last = max(1, last_tok // block)
    m = np.zeros((nb, nb), dtype=np.float32)
    for i in range(nb):
        for j in range(nb):
            if i >= j and (j <= si or (i - j) <= sl):
                m[i, j] = 1
            elif i >= j and (nb - i) < last and j > si:
                m[i, j] = 1
    return m


# --------------------------------------------------------------------------
# Eager attention (ascend_trianglemix.py:52-108)
# --------------------------------------------------------------------------


def eager_attention(q: np.ndarray, k: np.ndarray, v: np.ndarray, scale: float,
                    attn_mask: Optional[np.ndarray] = None, dtype=np.float32) -> np.ndarray:
    """``softmax(q k^T * scale + mask) v`` on ``[B, N, H, D]`` inputs.

    ascend_trianglemix.py:88-106: transpose to ``[B, H, N, D]``, matmul, scale,
    add the (broadcast) mask when there is one, softmax over keys, matmul with
    v, transpose back.  Without a mask the attention is bidirectional.  Done
    one head at a time so N = 8192 fits in memory; arithmetic in ``dtype``
    (float32 mirrors the reference run in fp32, float64 is the tighter check).
    """
    b, n, h, d = q.shape
    out = np.empty((b, n, h, v.shape[-1]), dtype=dtype)
    m = None if attn_mask is None else attn_mask.astype(dtype)
    for bi in range(b):
        for hi in range(h):
            s = (q[bi, :, hi, :].astype(dtype) @ k[bi, :, hi, :].astype(dtype).T) * dtype(scale)
            if m is not None:
                s = s + m
            with np.errstate(invalid="ignore"):
                s = s - s.max(axis=-1, keepdims=True)
                p = np.exp(s)
                p = p / p.sum(axis=-1, keepdims=True)
            out[bi, :, hi, :] = p @ v[bi, :, hi, :].astype(dtype)
    return out


def varlen_gqa_attention(q: np.ndarray, k_rows, v_rows, cu_q: Sequence[int], seq_lens: Sequence[int],
                         scale: float, mode: str, si: int, sl: int, last: int,
                         bias: Optional[np.ndarray] = None, full_lens: Optional[Sequence[int]] = None,
                         tri_min_len: int = 0) -> np.ndarray:
    """Reference for the backend entry point: ragged batch, grouped heads.

    ``q`` is ``[T, H, D]``; request ``b`` owns rows ``cu_q[b]:cu_q[b+1]`` which
    are the LAST ``q_len`` positions of a sequence of ``seq_lens[b]`` tokens
    (``i = seq_len - q_len + q_idx``, the convention of the reference's own
    backend tests, tests/v1/attention/test_attention_backends.py:376-380).
    ``k_rows(b)`` / ``v_rows(b)`` return ``[seq_len, Hkv, D]`` arrays (the
    caller resolves paging).  ``mode``: "triangle", "causal" or "none".
    ``full_lens[b]``: length of the whole sequence the triangle pattern is
    meant for (see :func:`triangle_mask`); ``tri_min_len``: sequences of at
    most that (full) length are attended causally (the helper's length gate,
    trianglemix_attention.py:330-332).
    Math is float64 (inputs are taken as given, i.e. already rounded).
    """
    t, h, d = q.shape
    out = np.zeros((t, h, d), dtype=np.float64)
    for b in range(len(seq_lens)):
        q0, q1 = int(cu_q[b]), int(cu_q[b + 1])
        ql, n = q1 - q0, int(seq_lens[b])
        if ql == 0:
            continue
        kb = np.asarray(k_rows(b), dtype=np.float64)
        vb = np.asarray(v_rows(b), dtype=np.float64)
        hkv = kb.shape[1]
        g = h // hkv
        nf = n if full_lens is None else max(n, int(full_lens[b]))
        if mode == "triangle" and nf > tri_min_len:
            allow = triangle_mask(n, si, sl, last, q_len=ql, n_full=nf)
        elif mode in ("causal", "triangle"):
            allow = dense_mask(n, q_len=ql)
        else:
            allow = np.ones((ql, n), dtype=bool)
        add = np.where(allow, 0.0, -np.inf)
        if bias is not None:
            add = add + bias[n - ql:n, :n].astype(np.float64)
        for hi in range(h):
            s = (q[q0:q1, hi, :].astype(np.float64) @ kb[:, hi // g, :].T) * scale + add
            with np.errstate(invalid="ignore"):
                s = s - s.max(axis=-1, keepdims=True)
                p = np.exp(s)
                p = p / p.sum(axis=-1, keepdims=True)
            out[q0:q1, hi, :] = p @ vb[:, hi // g, :]
    return out


# --------------------------------------------------------------------------
# Packed block-sparse index and work items (what the GPU builder writes)
# --------------------------------------------------------------------------

MODE_NONE, MODE_CAUSAL, MODE_TRIANGLE = 0, 1, 2


def _mode_mask(mode: int, n: int, ql: int, si: int, sl: int, last: int, n_full: Optional[int] = None) -> np.ndarray:
    if mode == MODE_TRIANGLE:
        return triangle_mask(n, si, sl, last, q_len=ql, n_full=n_full)
    if mode == MODE_CAUSAL:
        return dense_mask(n, q_len=ql)
    return np.ones((ql, n), dtype=bool)


def block_index_bruteforce(q_lens: Sequence[int], seq_lens: Sequence[int], si: int, sl: int, last: int,
                           bm: int, bn: int, mode: int, full_lens: Optional[Sequence[int]] = None):
    """CSR block index straight from the token mask.

    The batch is cut into q-blocks of ``bm`` rows, request by request (a
    q-block never spans two requests; the last one of a request may be
    ragged).  For q-block ``r`` the index lists, ascending, every key tile
    ``t`` (keys ``[t*bn, (t+1)*bn)``) holding at least one attended (row, key)
    pair.  Returns ``(row_ptr[int32, R+1], col_idx[int32, nnz], qb_seq[int32, R],
    qb_row0[int32, R])`` where ``qb_row0`` is the block's first row offset
    inside its request.
    """
    row_ptr, col_idx, qb_seq, qb_row0 = [0], [], [], []
    for b, (ql, n) in enumerate(zip(q_lens, seq_lens)):
        if ql == 0:
            continue
        allow = _mode_mask(mode, n, ql, si, sl, last, None if full_lens is None else int(full_lens[b]))
        nt = (n + bn - 1) // bn
        for r0 in range(0, ql, bm):
            rows = allow[r0:min(r0 + bm, ql)]
            for t in range(nt):
                if rows[:, t * bn:(t + 1) * bn].any():
                    col_idx.append(t)
            row_ptr.append(len(col_idx))
            qb_seq.append(b)
            qb_row0.append(r0)
    return (np.asarray(row_ptr, np.int32), np.asarray(col_idx, np.int32),
            np.asarray(qb_seq, np.int32), np.asarray(qb_row0, np.int32))


def qblock_tile_ranges(r0: int, r1: int, ql: int, n: int, si: int, sl: int, last: int, bn: int, mode: int,
                       n_full: Optional[int] = None):
    """Closed form of one q-block's key-tile set: at most two inclusive tile
    ranges ``[(a0, a1), (b0, b1)]`` (second may be empty), ascending and
    disjoint.  Rows ``r0 .. r1-1`` of the request, positions ``ctx + r``."""
    ctx = n - ql
    n_full = n if n_full is None else max(n, n_full)
    i_min, i_max = ctx + r0, ctx + r1 - 1
    nt = (n + bn - 1) // bn
    if mode == MODE_NONE:
        return [(0, nt - 1)] if nt > 0 else []
    t_diag = i_max // bn
    if mode == MODE_CAUSAL or (mode == MODE_TRIANGLE and (n_full - i_max) < last):
        return [(0, t_diag)]
    ranges = []
    a_hi = min(i_max, si)
    if a_hi >= 0:
        ranges.append((0, a_hi // bn))
    if sl >= 0:
        b_lo = max(0, i_min - sl) // bn
        if ranges and b_lo <= ranges[-1][1] + 1:
            ranges[-1] = (0, max(ranges[-1][1], t_diag))
        else:
            ranges.append((b_lo, t_diag))
    return ranges


def block_index_closed_form(q_lens, seq_lens, si, sl, last, bm, bn, mode, full_lens=None):
    """Same result as :func:`block_index_bruteforce`, O(active tiles)."""
    row_ptr, col_idx, qb_seq, qb_row0 = [0], [], [], []
    for b, (ql, n) in enumerate(zip(q_lens, seq_lens)):
        for r0 in range(0, ql, bm):
            r1 = min(r0 + bm, ql)
            for lo, hi in qblock_tile_ranges(r0, r1, ql, n, si, sl, last, bn, mode,
                                             None if full_lens is None else int(full_lens[b])):
                col_idx.extend(range(lo, hi + 1))
            row_ptr.append(len(col_idx))
            qb_seq.append(b)
            qb_row0.append(r0)
    return (np.asarray(row_ptr, np.int32), np.asarray(col_idx, np.int32),
            np.asarray(qb_seq, np.int32), np.asarray(qb_row0, np.int32))


def work_items(row_ptr: np.ndarray, chunk: int, qb_seq=None, qb_row0=None, q_lens=None, seq_lens=None):
    """Split every q-block's tile list into pieces of at most ``chunk`` tiles.

    The list starts with the pieces of all split q-blocks (the long jobs:
    q-blocks DESCENDING, pieces of one q-block ascending) and goes on with
    the unsplit q-blocks, DESCENDING: whatever the batch, a workgroup that
    gets a long piece gets it first.
    Each item is ``(qblock, first entry in col_idx, tile count, partial slot)``
    with ``partial slot = -1`` when the q-block has a single piece (the
    attention kernel then writes the final output itself) and otherwise a
    dense running number over all pieces that need the combine pass, counted
    in q-block ASCENDING order.  Also returns ``part_ptr[int32, R+1]``: the
    first partial slot of each q-block (equal consecutive values = no split).

    When the q-block geometry is given, four more columns are appended per
    item: ``(request, first row of the q-block in the packed q tensor, position
    of that row in its sequence, seq_len)``.
    """
    r = len(row_ptr) - 1
    cnt = np.diff(row_ptr.astype(np.int64))
    npieces = np.maximum(1, (cnt + chunk - 1) // chunk)
    nparts = np.where(npieces > 1, npieces, 0)
    part_ptr = np.zeros(r + 1, np.int64)
    part_ptr[1:] = np.cumsum(nparts)
    items = []
    order = [qb for qb in range(r - 1, -1, -1) if npieces[qb] > 1] + \
            [qb for qb in range(r - 1, -1, -1) if npieces[qb] <= 1]
    for qb in order:
        for p in range(int(npieces[qb])):
            start = int(row_ptr[qb]) + p * chunk
            ntile = int(min(chunk, int(row_ptr[qb + 1]) - start))
            slot = int(part_ptr[qb]) + p if npieces[qb] > 1 else -1
            rec = (qb, start, max(ntile, 0), slot)
            if qb_seq is not None:
                b = int(qb_seq[qb])
                cu = int(np.sum(q_lens[:b]))
                rec += (b, cu + int(qb_row0[qb]), int(seq_lens[b]) - int(q_lens[b]) + int(qb_row0[qb]),
                        int(seq_lens[b]))
            items.append(rec)
    width = 4 if qb_seq is None else 8
    return np.asarray(items, np.int32).reshape(-1, width), part_ptr.astype(np.int32)


def expand_index_to_mask(row_ptr, col_idx, qb_seq, qb_row0, q_lens, seq_lens, si, sl, last, bm, bn, mode):
    """Token mask implied by (index + in-tile predicate), per request -- must
    equal the oracle bitmap; a tile missing from the index shows as a hole."""
    out = [np.zeros((ql, n), dtype=bool) for ql, n in zip(q_lens, seq_lens)]
    for r in range(len(row_ptr) - 1):
        b, r0 = int(qb_seq[r]), int(qb_row0[r])
        ql, n = q_lens[b], seq_lens[b]
        allow = _mode_mask(mode, n, ql, si, sl, last)
        r1 = min(r0 + bm, ql)
        for t in col_idx[row_ptr[r]:row_ptr[r + 1]]:
            c0, c1 = int(t) * bn, min((int(t) + 1) * bn, n)
            out[b][r0:r1, c0:c1] = allow[r0:r1, c0:c1]
    return out


# --------------------------------------------------------------------------
# bf16 helpers (round-to-nearest-even, as torch's .bfloat16())
# --------------------------------------------------------------------------


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round float32 to the nearest bfloat16 value, returned as float32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    return r.view(np.float32).reshape(x.shape)


def bf16_bits(x: np.ndarray) -> np.ndarray:
    """bfloat16 bit pattern (uint16) of float32 values that are already
    representable (or to be truncated after :func:`bf16_round`)."""
    return (np.ascontiguousarray(bf16_round(x), dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


# --------------------------------------------------------------------------
# fp8 e4m3fn (the K/V cache format of kv_cache_dtype "fp8" on gfx950,
# vllm/platforms/rocm.py:402-406): 1 sign, 4 exponent (bias 7), 3 mantissa
# bits, no infinities, 0x7f / 0xff = NaN, largest finite value 448.
# --------------------------------------------------------------------------

def fp8_e4m3fn_decode(b: np.ndarray) -> np.ndarray:
    b = np.asarray(b, np.uint8).astype(np.int64)
    e, m = (b >> 3) & 15, b & 7
    mag = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1.0 + m / 8.0) * np.exp2((e - 7).astype(np.float64)))
    mag = np.where((e == 15) & (m == 7), np.nan, mag)
    return np.where(b >> 7, -mag, mag)


_FP8_POS = fp8_e4m3fn_decode(np.arange(127, dtype=np.uint8))  # codes 0 .. 126: 0 .. 448, ascending


def fp8_e4m3fn_encode(x: np.ndarray) -> np.ndarray:
    """Round to nearest, ties to even; finite values saturate at +-448 (the
    scaled conversion of csrc/quantization/fp8/amd/quant_utils.cuh); NaN -> 0x7f."""
    x = np.asarray(x, np.float64)
    ax = np.minimum(np.abs(x), 448.0)
    hi = np.clip(np.searchsorted(_FP8_POS, ax, side="left"), 0, 126)
    lo = np.maximum(hi - 1, 0)
    d_lo, d_hi = ax - _FP8_POS[lo], _FP8_POS[hi] - ax
    pick_hi = (d_hi < d_lo) | ((d_hi == d_lo) & (hi % 2 == 0))
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    code = np.where(np.isnan(x), np.uint8(0x7F), code)
    return (code | np.where(np.signbit(x) & ~np.isnan(x), 0x80, 0).astype(np.uint8)).astype(np.uint8)


def fp8_cache_roundtrip(x: np.ndarray, scale: float = 1.0) -> np.ndarray:
    """What the attention sees of a value written to an fp8 cache: fp8(x / scale) * scale (float64;
    the 16-bit rounding of the descaled value is the caller's, see :func:`bf16_round`)."""
    # the write divides in fp32, as the reference's scaled conversion does (quant_utils.cuh:544,589: a / scale)
    q = fp8_e4m3fn_encode((np.asarray(x, np.float32) / np.float32(scale)).astype(np.float32))
    return (fp8_e4m3fn_decode(q).astype(np.float32) * np.float32(scale)).astype(np.float64)


def seeded_normal(seed: int, shape) -> np.ndarray:
    """Deterministic N(0,1) float32 inputs (numpy PCG64), rounded to bf16."""
    return bf16_round(np.random.default_rng(seed).standard_normal(shape, dtype=np.float32))


def triangle_nnz(n: int, si: int, sl: int, last: int) -> int:
    """Number of attended pairs, by row intervals (O(N))."""
    tot = 0
    for i in range(n):
        tot += sum(hi - lo + 1 for lo, hi in row_intervals(i, n, si, sl, last))
    return tot
