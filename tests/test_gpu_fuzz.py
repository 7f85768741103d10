"""Seeded random batches through ``tmx_attention`` against the CPU oracle: every combination of request kinds
(whole prompts, chunks with context, short chunks, decode rows), head groupings, patterns, cache layouts, piece
lengths and kernel-selection flags must give the oracle's rows -- whichever of the library's kernels computes
them -- and, when the launch also stores the step's K/V rows, the cache the stand-alone write kernel gives.
Needs an MI355X."""
import os

import numpy as np
import pytest
import torch

from oracle import trianglemix_oracle as orc
from vllm_trianglemix_amd._lib import (TMX_ATTN_DECODE_VALU, TMX_ATTN_NO_BAND, TMX_ATTN_NO_DECODE, TMX_ATTN_NO_DENSE,
                                       TMX_ATTN_SHORT_QUERIES)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# fp32 output against the fp64 oracle on the same 16-bit-rounded inputs: the rule of tests/parity_tolerance.py
# (1e-3, absolute up to |ref| = 1 and relative above)
from parity_tolerance import assert_close  # noqa: E402


def _round(x, dtype):
    return orc.bf16_round(x) if dtype == torch.bfloat16 else np.float16(x).astype(np.float32)


def _random_batch(rng, big):
    reqs = []
    for _ in range(int(rng.integers(1, 7))):
        kind = rng.choice(["decode", "short", "whole", "chunk"], p=[0.35, 0.15, 0.3, 0.2])
        hi = 3000 if big else 400
        if kind == "decode":
            n = int(rng.integers(1, hi))
            q = 1
        elif kind == "short":
            n = int(rng.integers(2, hi))
            q = int(rng.integers(2, min(n, 40) + 1))
        elif kind == "whole":
            n = int(rng.choice([rng.integers(1, hi), rng.integers(2049, 2600) if big else rng.integers(100, 300)]))
            q = n
        else:
            n = int(rng.integers(40, hi))
            q = int(rng.integers(17, n))
        reqs.append((q, n))
    return [r[0] for r in reqs], [r[1] for r in reqs]


def _stage(ops, idx, cu_h, sl_h, Hkv):
    """A staging area for every request the dense causal kernel takes, full of NaN (None when it takes none)."""
    idx.stage_rows = ops.dense_stage_rows(cu_h, sl_h, 0)
    st = ops.alloc_kv_stage(idx.stage_rows, Hkv, DEV)
    if st is not None:
        st.fill_(float("nan"))
    return st


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("TMX_FUZZ_SEEDS", "24")))))  # more seeds: a longer hunt
def test_random_steps_match_the_oracle(seed):
    from vllm_trianglemix_amd import ops
    rng = np.random.default_rng(1000 + seed)
    big = seed % 3 != 0
    q_lens, seq_lens = _random_batch(rng, big)
    # (8, 1): the eighth of a 64 / 8-head model (two head chunks on one K/V head); (64, 8): that model whole
    head_list = [(8, 2), (4, 1), (16, 2), (32, 8), (4, 4), (6, 2), (8, 8), (8, 1), (64, 8)]
    H, Hkv = head_list[int(rng.integers(0, len(head_list)))]
    D = 64 if seed % 8 == 7 else 128
    dtype = torch.float16 if seed % 4 == 1 else torch.bfloat16
    mode = ["triangle", "triangle", "causal", "none"][int(rng.integers(0, 4))]
    pat = (4, 32, 64) if rng.random() < 0.6 else (int(rng.integers(-1, 9)), int(rng.integers(0, 41)), int(rng.integers(0, 81)))
    gate = 2048 if (mode == "triangle" and big and rng.random() < 0.7) else 0
    chunk = int(rng.choice([2, 4, 16, 64]))
    paged = rng.random() < 0.7
    page = int(rng.choice([16, 32, 64]))
    fused = paged and mode != "none" and rng.random() < 0.5  # the launch stores the step's rows itself
    defer = bool(np.random.default_rng(seed + 77777).random() < 0.5)  # (a stream of its own: the other draws stay as they were)
    fp8 = paged and not fused and D == 128 and rng.random() < 0.35  # an fp8 (e4m3fn) cache read in place
    k_scale, v_scale = ((1.0, 1.0) if rng.random() < 0.5 else (0.75, 1.5)) if fp8 else (1.0, 1.0)
    # chunked prefill: some requests are chunks of a longer prompt (the pattern then refers to the whole prompt)
    full_lens = None
    if mode == "triangle" and rng.random() < 0.4:
        full_lens = [n + (int(rng.integers(0, 500)) if q < n and rng.random() < 0.7 else 0) for q, n in zip(q_lens, seq_lens)]
    flags = 0
    if max(q_lens) <= 16 and rng.random() < 0.5:
        flags |= TMX_ATTN_SHORT_QUERIES
    for f in (TMX_ATTN_NO_BAND, TMX_ATTN_NO_DENSE, TMX_ATTN_NO_DECODE, TMX_ATTN_DECODE_VALU):
        if rng.random() < 0.25:
            flags |= f

    # causal launches (a stream of draws of its own): the index may leave the dense kernel's requests out where that
    # kernel can run, and the call may lend it a staging area (for every request it takes, whatever their lengths)
    rng2 = np.random.default_rng(seed + 424242)
    leave = bool(mode == "causal" and D == 128 and (H // Hkv) % 4 == 0 and not (flags & TMX_ATTN_NO_DENSE)
                 and rng2.random() < 0.5)
    staged = bool(mode == "causal" and D == 128 and rng2.random() < 0.5)

    T, B = sum(q_lens), len(seq_lens)
    q = _round(orc.seeded_normal(seed * 7 + 1, (T, H, D)), dtype)
    ks = [_round(orc.seeded_normal(seed * 7 + 100 + b, (n, Hkv, D)), dtype) for b, n in enumerate(seq_lens)]
    vs = [_round(orc.seeded_normal(seed * 7 + 200 + b, (n, Hkv, D)), dtype) for b, n in enumerate(seq_lens)]
    cu_h = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    sl_h = np.asarray(seq_lens, np.int32)
    cu, sld = torch.from_numpy(cu_h).to(DEV), torch.from_numpy(sl_h).to(DEV)
    to_dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV).to(dtype)  # noqa: E731
    qd = to_dev(q)
    out = torch.full((T, H, D), float("nan"), dtype=torch.float32, device=DEV)
    bm, bn = ops.attention_block_m(H, Hkv), ops.attention_block_n()
    scale = D ** -0.5
    desc = dict(seed=seed, defer=defer, leave=leave, staged=staged, q=q_lens, n=seq_lens, heads=(H, Hkv, D), dtype=str(dtype), mode=mode, pat=pat, gate=gate,
                chunk=chunk, paged=paged, page=page, fused=fused, flags=flags, fp8=fp8, scales=(k_scale, v_scale),
                full_lens=full_lens)
    fl_h = None if full_lens is None else np.asarray(full_lens, np.int32)
    fl_d = None if full_lens is None else torch.from_numpy(fl_h).to(DEV)
    k_ref, v_ref = ks, vs
    if paged:
        nblk = sum((n + page - 1) // page for n in seq_lens) + 2
        perm = rng.permutation(nblk).astype(np.int32)
        kc = torch.zeros((nblk, page, Hkv, D), dtype=torch.uint8 if fp8 else dtype, device=DEV)
        vc = torch.zeros_like(kc)
        bt = np.zeros((B, max((n + page - 1) // page for n in seq_lens)), np.int32)
        nxt, new_slots, k_new, v_new = 0, [], [], []
        for b, n in enumerate(seq_lens):
            for pb in range((n + page - 1) // page):
                bt[b, pb] = perm[nxt]
                nxt += 1
            pos = np.arange(n)
            slots = bt[b, pos // page].astype(np.int64) * page + pos % page
            ctx = n - q_lens[b]
            old = slice(0, ctx) if fused else slice(0, n)  # with the fused write only the context is in the cache yet
            if old.stop > 0 and fp8:
                ops.reshape_and_cache_fp8(to_dev(ks[b][old]), to_dev(vs[b][old]), kc, vc, torch.from_numpy(slots[old]).to(DEV),
                                          k_scale, v_scale)
            elif old.stop > 0:
                ops.reshape_and_cache(to_dev(ks[b][old]), to_dev(vs[b][old]), kc, vc, torch.from_numpy(slots[old]).to(DEV))
            new_slots.append(slots[ctx:])
            k_new.append(ks[b][ctx:])
            v_new.append(vs[b][ctx:])
        btd = torch.from_numpy(bt).to(DEV)
        resolve = rng.random() < 0.5
        # (half of the steps hand the attention call an index that is still to be built: TMX_ATTN_BUILD_INDEX)
        idx = ops.build_block_index(cu, sld, cu_h, sl_h, *pat, bm, bn, ops.MODES[mode], chunk_tiles=chunk, tri_min_len=gate,
                                    block_table=btd if resolve else None, page_size=page if resolve else 0,
                                    full_lens=fl_d, full_lens_cpu=fl_h, defer=defer, leave_dense=leave)
        stage = _stage(ops, idx, cu_h, sl_h, Hkv) if staged else None
        if fused:
            kn, vn = to_dev(np.concatenate(k_new, 0)), to_dev(np.concatenate(v_new, 0))
            slots_d = torch.from_numpy(np.concatenate(new_slots)).to(DEV)
            ops.attention(qd, kc, vc, out, idx, cu, sld, scale, block_table=btd, flags=flags, k_new=kn, v_new=vn,
                          cache_slots=slots_d, kv_stage=stage)
            # the cache now holds every row of every request
            kf, vf = kc.view(nblk * page, Hkv, D), vc.view(nblk * page, Hkv, D)
            assert torch.equal(kf[slots_d], kn) and torch.equal(vf[slots_d], vn), desc
        elif fp8:
            ops.attention(qd, kc.view(torch.float8_e4m3fn), vc.view(torch.float8_e4m3fn), out, idx, cu, sld, scale,
                          block_table=btd, flags=flags, k_scale=k_scale, v_scale=v_scale, kv_stage=stage)
            # the oracle attends to what the cache holds: fp8(x / scale) x scale, rounded once to the 16-bit type
            seen = lambda x, sc: _round(orc.fp8_cache_roundtrip(x, sc).astype(np.float32), dtype)  # noqa: E731
            k_ref, v_ref = [seen(k, k_scale) for k in ks], [seen(v, v_scale) for v in vs]
        else:
            ops.attention(qd, kc, vc, out, idx, cu, sld, scale, block_table=btd, flags=flags, kv_stage=stage)
    else:
        idx = ops.build_block_index(cu, sld, cu_h, sl_h, *pat, bm, bn, ops.MODES[mode], chunk_tiles=chunk, tri_min_len=gate,
                                    full_lens=fl_d, full_lens_cpu=fl_h, defer=defer, leave_dense=leave)
        stage = _stage(ops, idx, cu_h, sl_h, Hkv) if staged else None
        cu_k = torch.from_numpy(np.concatenate([[0], np.cumsum(seq_lens)]).astype(np.int32)).to(DEV)
        ops.attention(qd, to_dev(np.concatenate(ks, 0)), to_dev(np.concatenate(vs, 0)), out, idx, cu, sld, scale,
                      cu_seqlens_k=cu_k, flags=flags, kv_stage=stage)
    torch.cuda.synchronize()
    assert idx.built
    got = out.cpu().numpy()
    ref = orc.varlen_gqa_attention(q, lambda b: k_ref[b], lambda b: v_ref[b], cu_h, seq_lens, scale, mode, *pat,
                                   full_lens=full_lens, tri_min_len=gate)
    # rows the pattern leaves without any key (a window of 0 keys and no sink, say) are NaN in both
    assert_close(got, ref, f"random steps ({'fp8' if fp8 else str(dtype)[6:]}{', fused write' if fused else ''})",
                 info=desc, nan_equal=True)
