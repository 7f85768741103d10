#!/usr/bin/env python3
"""Decode steps of a whole model (36 attention layers, Qwen3-8B heads) three ways:

  index    the block index of the step is sized on the host and built on the device (once per step), every layer
           calls tmx_attention with it (the decode kernel runs alone: TMX_ATTN_DECODE_ONLY);
  fixed    every layer calls tmx_decode_attention: no index, nothing sized from the lengths (the pieces are laid out
           on the device from seq_lens);
  graph    the 36 fixed-shape calls captured once in a HIP graph (torch.cuda.graph), one replay per step.

Per route: host microseconds to enqueue one step with the GPU idle (synchronize before every timed step), and
wall-clock microseconds per step in steady state (back-to-back steps).  Usage: python tools/decode_overhead.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_trianglemix_amd import ops  # noqa: E402
from vllm_trianglemix_amd._lib import TMX_ATTN_DECODE_ONLY, TMX_ATTN_SHORT_QUERIES  # noqa: E402

H, Hkv, D, PAGE, LAYERS = 32, 8, 128, 16, 36
si, sl, last = 4, 32, 64
dev = torch.device("cuda:0")
scale = D ** -0.5


def run(B, N):
    g = torch.Generator().manual_seed(0)
    nblk = B * (N // PAGE)
    kc = torch.randn((nblk, PAGE, Hkv, D), generator=g).to(dev).bfloat16()
    vc = torch.randn((nblk, PAGE, Hkv, D), generator=g).to(dev).bfloat16()
    bt = torch.randperm(nblk, generator=g).to(torch.int32).view(B, -1).to(dev)
    q = torch.randn((B, H, D), generator=g).to(dev).bfloat16()
    out = torch.empty_like(q)
    cu_h, sl_h = np.arange(B + 1, dtype=np.int32), np.full((B,), N, np.int32)
    cu, sld = torch.from_numpy(cu_h).to(dev), torch.from_numpy(sl_h).to(dev)
    bm, bn, mode = ops.attention_block_m(H, Hkv), ops.attention_block_n(), ops.MODES["triangle"]
    flags = TMX_ATTN_SHORT_QUERIES | TMX_ATTN_DECODE_ONLY

    def step_index():
        chunk = ops.suggest_chunk_tiles(cu_h, sl_h, si, sl, last, bm, bn, mode, ops.units_per_item(H, Hkv))
        idx = ops.build_block_index(cu, sld, cu_h, sl_h, si, sl, last, bm, bn, mode, chunk_tiles=chunk, block_table=bt,
                                    page_size=PAGE)
        ws = ops.alloc_workspace(idx, H, D, dev)
        for _ in range(LAYERS):
            ops.attention(q, kc, vc, out, idx, cu, sld, scale, block_table=bt, workspace=ws, flags=flags)

    ws_f = ops.decode_workspace(B, H, Hkv, D, dev)

    def step_fixed():
        for _ in range(LAYERS):
            ops.decode_attention(q, kc, vc, out, bt, sld, scale, ws_f)

    step_fixed()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step_fixed()

    def step_graph():
        graph.replay()

    def host_us(fn, reps=30):
        for _ in range(3):
            fn()
        tot = 0.0
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            tot += time.perf_counter() - t0
        torch.cuda.synchronize()
        return tot / reps * 1e6

    def wall_us(fn, reps=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    print(f"decode step, {B} requests x {N} cached tokens, {LAYERS} layers, {H}/{Hkv} heads")
    for name, fn in (("index (tmx_attention)", step_index), ("fixed shape (tmx_decode_attention)", step_fixed),
                     ("graph replay of the fixed shape", step_graph)):
        h, w = host_us(fn), wall_us(fn)
        print(f"  {name:38s} host {h:8.1f} us/step ({h / LAYERS:5.1f} per layer)   wall {w:9.1f} us/step "
              f"({w / LAYERS:6.1f} per layer)")


def run_fp8_write(B, N):
    """Decode layers out of an fp8 cache, the step's K/V rows written by a launch of their own (the reference's order)
    or by the attention launch itself; eager and as one graph replay per step."""
    g = torch.Generator().manual_seed(0)
    nblk = B * (N // PAGE)
    kc = (torch.randn((nblk, PAGE, Hkv, D), generator=g) * 8).to(dev).to(torch.float8_e4m3fn)
    vc = (torch.randn((nblk, PAGE, Hkv, D), generator=g) * 8).to(dev).to(torch.float8_e4m3fn)
    bt = torch.randperm(nblk, generator=g).to(torch.int32).view(B, -1).to(dev)
    q = torch.randn((B, H, D), generator=g).to(dev).bfloat16()
    kn = torch.randn((B, Hkv, D), generator=g).to(dev).bfloat16()
    vn = torch.randn((B, Hkv, D), generator=g).to(dev).bfloat16()
    out = torch.empty_like(q)
    sld = torch.full((B,), N, dtype=torch.int32, device=dev)
    slots = (bt[:, (N - 1) // PAGE].to(torch.int64) * PAGE + (N - 1) % PAGE).contiguous()
    ws = ops.decode_workspace(B, H, Hkv, D, dev, True)

    def separate():
        for _ in range(LAYERS):
            ops.reshape_and_cache_fp8(kn, vn, kc, vc, slots, 1.0, 1.0)
            ops.decode_attention(q, kc, vc, out, bt, sld, scale, ws)

    def fused():
        for _ in range(LAYERS):
            ops.decode_attention(q, kc, vc, out, bt, sld, scale, ws, k_new=kn, v_new=vn, cache_slots=slots)

    def wall_us(fn, reps=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    print(f"decode step out of an fp8 cache, {B} requests x {N} cached tokens, {LAYERS} layers, {H}/{Hkv} heads")
    for name, fn in (("cache write + attention (two launches a layer)", separate), ("attention writes the cache (one launch)", fused)):
        fn()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fn()
        w, wg = wall_us(fn), wall_us(graph.replay)
        print(f"  {name:48s} eager {w:8.1f} us/step ({w / LAYERS:6.1f} per layer)   graph replay {wg:8.1f} us/step "
              f"({wg / LAYERS:6.1f} per layer)")


if __name__ == "__main__":
    # an idle GPU first: its clocks settle after tens of milliseconds of load
    x = torch.randn((4096, 4096), device=dev)
    for _ in range(50):
        x @ x
    torch.cuda.synchronize()
    run(64, 8192)
    run(8, 2048)
    run(1, 2048)
    run(256, 1024)
    run_fp8_write(8, 2048)
    run_fp8_write(64, 8192)
