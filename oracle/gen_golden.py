#!/usr/bin/env python3
"""Generate tests/golden/* from the reference's own TriangleMix source files.

Runs ONLY in the build container, where the reference checkout is mounted
(default /root/reference).  The four TriangleMix files depend on nothing but
torch, but the package ``vllm`` itself does not import offline (missing
third-party modules), so they are loaded by path under their real module
names after registering empty ``vllm`` / ``vllm.attention`` packages.

What is written (data only, no reference source text):
  tests/golden/masks.json        mask bitmaps (hex / sha256 / nnz) per case
  tests/golden/selection.json    layer selection, presets, helper decisions
  tests/golden/attention.npz     eager attention outputs (fp32) per case
  tests/golden/attention.json    the case list for attention.npz
  tests/golden/ref_timings.json  wall times of the reference loops here

Usage: python oracle/gen_golden.py [--reference /root/reference] [--quick]
"""
import argparse
import hashlib
import importlib.util
import json
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import trianglemix_oracle as orc  # noqa: E402  (bf16 helpers, seeded inputs only)


def load_reference(ref_root):
    base = os.path.join(ref_root, "vllm", "attention")
    for name in ("vllm", "vllm.attention"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    mods = {}
    for mod, fn in (("trianglemix_attention", "trianglemix_attention.py"),
                    ("ascend_trianglemix", "ascend_trianglemix.py"),
                    ("trianglemix_config", "trianglemix_config.py"),
                    ("ascend_npu_trianglemix", "ascend_npu_trianglemix.py")):
        full = "vllm.attention." + mod
        spec = importlib.util.spec_from_file_location(full, os.path.join(base, fn))
        m = importlib.util.module_from_spec(spec)
        sys.modules[full] = m
        spec.loader.exec_module(m)
        mods[mod] = m
    return mods


def pack(mask01):
    b = np.packbits(np.asarray(mask01) > 0, axis=None)
    return b.tobytes()


def mask_record(mask01, keep_hex):
    raw = pack(mask01)
    rec = {"nnz": int((np.asarray(mask01) > 0).sum()), "sha256": hashlib.sha256(raw).hexdigest()}
    if keep_hex:
        rec["hex"] = raw.hex()
    return rec


def mask_cases(quick):
    cases = [(4, 8, 16, 24), (2, 4, 6, 12), (3, 6, 8, 16)]
    big = [1, 33, 70, 512] if quick else [1, 33, 70, 512, 2048, 2049, 4096]
    cases += [(4, 32, 64, n) for n in big]
    # edge cases: zeros, N < last, N <= si, window wider than the sequence
    cases += [(0, 0, 0, 9), (0, 0, 1, 9), (0, 0, 2, 9), (0, 5, 0, 20), (5, 0, 0, 20), (0, 0, 64, 20),
              (4, 32, 64, 40), (4, 32, 64, 63), (4, 32, 64, 64), (4, 32, 64, 65), (40, 3, 5, 30),
              (4, 100, 8, 50), (1, 1, 1, 1), (1, 1, 1, 2), (3, 2, 200, 100), (7, 15, 17, 129)]
    rng = np.random.default_rng(20251212)
    for _ in range(40):
        cases.append((int(rng.integers(0, 13)), int(rng.integers(0, 49)), int(rng.integers(0, 81)),
                      int(rng.integers(1, 161))))
    return cases


def gen_masks(tm, quick):
    cpu = torch.device("cpu")
    out, timings = [], {}
    for (si, sl, last, n) in mask_cases(quick):
        gen = tm.TriangleMixMaskGenerator(tm.TriangleMixConfig(si, sl, last))
        rec = {"si": si, "sl": sl, "last": last, "n": n}
        keep_hex = n <= 64
        t0 = time.time()
        tri = gen._create_triangle_mask(n, cpu, torch.float32).numpy()
        dt = time.time() - t0
        if (si, sl, last) == (4, 32, 64) and n in (512, 2048, 4096):
            timings[f"triangle_mask_n{n}_s"] = dt
        rec["triangle"] = mask_record(tri, keep_hex)
        assert set(np.unique(tri)) <= {0.0, 1.0}
        if n <= 512:
            rec["streaming"] = mask_record(gen._create_streaming_mask(n, cpu, torch.float32).numpy(), keep_hex)
            rec["last_mask"] = mask_record(gen._create_last_mask(n, cpu, torch.float32).numpy(), keep_hex)
            rec["middle"] = mask_record(gen._create_middle_mask(n, cpu, torch.float32).numpy(), keep_hex)
            rec["dense"] = mask_record(gen._create_dense_mask(n, cpu, torch.float32).numpy(), keep_hex)
        if n <= 160:
            for dt_name, dtp in (("float32", torch.float32), ("bfloat16", torch.bfloat16)):
                for use_tri in (False, True):
                    g2 = tm.TriangleMixMaskGenerator(tm.TriangleMixConfig(si, sl, last))
                    am = g2.get_attention_mask(n, 0, cpu, dtp, use_triangle=use_tri)
                    assert am.dtype == dtp and tuple(am.shape) == (n, n)
                    amf = am.float().numpy()
                    assert np.all((amf == 0) | np.isneginf(amf))
                    rec[f"additive_{dt_name}_{'tri' if use_tri else 'dense'}"] = mask_record(amf == 0, keep_hex)
        out.append(rec)
        print(f"mask si={si} sl={sl} last={last} n={n}: nnz={rec['triangle']['nnz']} ({dt:.2f}s)", flush=True)
    return out, timings


def gen_selection(mods, quick):
    tm, at, cf, nb = (mods["trianglemix_attention"], mods["ascend_trianglemix"],
                      mods["trianglemix_config"], mods["ascend_npu_trianglemix"])
    cpu = torch.device("cpu")
    sel = {"should_use_triangle": [], "helper": [], "presets": {}, "from_hf": [], "analyzer": [],
           "optimize_mask": {}, "ascend_block": []}
    for ntl, idx in ((None, None), (4, None), (0, None), (None, [1, 5, 7]), (3, [2, 9]), (12, []), (2, [0])):
        cfg = tm.TriangleMixConfig(num_triangle_layers=ntl, triangle_layer_indices=idx)
        gen = tm.TriangleMixMaskGenerator(cfg)
        sel["should_use_triangle"].append({
            "num_triangle_layers": ntl, "triangle_layer_indices": idx,
            "stored_indices": list(cfg.triangle_layer_indices),
            "table": [bool(gen.should_use_triangle(i)) for i in range(14)]})
    # helper: mutation of the shared config + the <= 2048 gate + mask kind
    for ntl, idx, nl in ((4, None, 8), (None, [1, 3], 4), (None, None, 4), (2, None, 4)):
        cfg = tm.TriangleMixConfig(4, 8, 16, num_triangle_layers=ntl, triangle_layer_indices=idx)
        helper = tm.TriangleMixAttention(cfg, nl)
        rec = {"num_triangle_layers": ntl, "triangle_layer_indices": idx, "num_layers": nl,
               "indices_after_init": list(cfg.triangle_layer_indices), "decisions": []}
        lens = [1, 512, 2048] if quick else [1, 512, 2048, 2049]
        for layer in range(nl):
            for n in lens:
                if n > 2048 and not (ntl == 2 and idx is None):
                    continue  # the O(N^2) loops: once is enough
                m = helper.get_attn_mask(n, layer, cpu)
                if m is None:
                    kind = "none"
                else:
                    assert m.dtype == torch.float32
                    allowed = (m == 0).numpy()
                    dense = np.tril(np.ones((n, n), bool))
                    kind = "dense" if np.array_equal(allowed, dense) else "triangle"
                    if kind == "triangle":
                        rec.setdefault("triangle_sha256", {})[str(n)] = hashlib.sha256(pack(allowed)).hexdigest()
                rec["decisions"].append({"layer": layer, "n": n, "kind": kind})
        sel["helper"].append(rec)
        print("helper", ntl, idx, "done", flush=True)
    for size in ("14B", "32B", "8B", "0.5B", "72B"):
        c = cf.TriangleMixInferenceConfig.for_qwen3(size)
        t = c.to_trianglemix_config()
        sel["presets"][f"for_qwen3_{size}"] = {
            "enable_trianglemix": c.enable_trianglemix, "num_triangle_layers": c.num_triangle_layers,
            "num_sink_tokens": c.num_sink_tokens, "sliding_window_size": c.sliding_window_size,
            "num_last_tokens": c.num_last_tokens, "use_npu_optimization": c.use_npu_optimization,
            "npu_block_size": c.npu_block_size, "use_gradient_analysis": c.use_gradient_analysis,
            "analysis_warmup_steps": c.analysis_warmup_steps,
            "tm_cfg": [t.num_sink_tokens, t.sliding_window_size, t.num_last_tokens, t.num_triangle_layers,
                       list(t.triangle_layer_indices)]}
    c = cf.TriangleMixInferenceConfig.for_qwen3("14B", num_triangle_layers=3)
    sel["presets"]["for_qwen3_14B_override3"] = {"num_triangle_layers": c.num_triangle_layers}
    c = cf.TriangleMixInferenceConfig.for_ascend_npu()
    sel["presets"]["for_ascend_npu"] = {"enable_trianglemix": c.enable_trianglemix,
                                        "num_triangle_layers": c.num_triangle_layers,
                                        "use_gradient_analysis": c.use_gradient_analysis,
                                        "use_npu_optimization": c.use_npu_optimization}
    c = cf.TriangleMixInferenceConfig()
    sel["presets"]["default"] = {"enable_trianglemix": c.enable_trianglemix,
                                 "to_trianglemix_config_is_none": c.to_trianglemix_config() is None,
                                 "triangle_layer_indices": list(c.triangle_layer_indices)}
    c = cf.TriangleMixInferenceConfig(enable_trianglemix=True, triangle_layer_indices=[0, 1, 2])
    t = c.to_trianglemix_config()
    sel["presets"]["custom_indices"] = {"tm_cfg": [t.num_sink_tokens, t.sliding_window_size, t.num_last_tokens,
                                                   t.num_triangle_layers, list(t.triangle_layer_indices)]}
    for size in ("14B", "32B", "7B"):
        a = nb.create_ascend_trianglemix_config(size)
        sel["presets"][f"ascend_{size}"] = {"num_triangle_layers": a.num_triangle_layers,
                                            "npu_block_size": a.npu_block_size,
                                            "use_block_sparse": a.use_block_sparse, "use_acl_ops": a.use_acl_ops}
    for attrs, use, ntl in (({}, True, 6), ({"num_sink_tokens": 8, "sliding_window": 128}, True, None),
                            ({"sliding_window": 16}, True, 2), ({}, False, 4)):
        hf = types.SimpleNamespace(**attrs)
        r = at.create_trianglemix_config_from_hf(hf, use_trianglemix=use, num_triangle_layers=ntl)
        sel["from_hf"].append({"attrs": attrs, "use": use, "ntl": ntl,
                               "result": None if r is None else [r.num_sink_tokens, r.sliding_window_size,
                                                                 r.num_last_tokens, r.num_triangle_layers,
                                                                 list(r.triangle_layer_indices)]})
    # analyzer: the reference test's scenario + ties / missing layers
    scen = [[[1.0 - (i / 8) * 0.5] * 3 for i in range(8)],
            [[0.5], [], [0.25, 0.75], [0.1], [], [0.5], [0.05, 0.05]],
            [[0.3], [0.3], [0.3], [0.3]]]
    for grads in scen:
        for k in (0, 1, 2, 4, len(grads) + 2):
            an = tm.TriangleMixAnalyzer(len(grads))
            for li, gl in enumerate(grads):
                for gval in gl:
                    an.record_middle_qk_gradient(li, torch.tensor(gval))
            an.record_middle_qk_gradient(len(grads) + 3, torch.tensor(9.0))  # out of range: ignored
            sel["analyzer"].append({"grads": grads, "k": k, "layers": an.get_triangle_layers(k)})
    m = torch.full((4, 4), float("-inf"))
    m[0, 0] = 0.0
    m[1, 0:2] = 0.0
    m[2, 0:3] = 0.0
    m[3, :] = 0.0
    o = at.AscendNPUTriangleMixOptimizer.optimize_mask_for_npu(m)
    sel["optimize_mask"] = {"in_is_zero_hex": pack((m == 0).numpy()).hex(), "out": o.numpy().tolist()}
    for n, (si, sl, last), blk in ((4096, (4, 32, 64), 64), (1000, (4, 32, 64), 64), (3000, (130, 200, 400), 64),
                                   (777, (4, 32, 64), 32)):
        cfg = nb.AscendNPUTriangleMixConfig(si, sl, last, npu_block_size=blk)
        g = nb.AscendBlockSparseMaskGenerator(cfg)
        bm = g._get_block_sparse_mask(n, cpu, torch.float32).numpy()
        sel["ascend_block"].append({"n": n, "si": si, "sl": sl, "last": last, "block": blk,
                                    "shape": list(bm.shape), **mask_record(bm, True)})
    return sel


def attention_cases(quick):
    cases = [
        dict(name="tri_128", B=1, N=128, H=4, D=64, cfg=(4, 8, 16), mask="triangle", seed=11),
        dict(name="tri_257", B=2, N=257, H=4, D=128, cfg=(4, 32, 64), mask="triangle", seed=12),
        dict(name="tri_512", B=1, N=512, H=2, D=128, cfg=(4, 32, 64), mask="triangle", seed=13),
        dict(name="dense_96", B=1, N=96, H=2, D=64, cfg=(4, 32, 64), mask="dense", seed=14),
        dict(name="nomask_32", B=2, N=32, H=8, D=64, cfg=(4, 32, 64), mask="helper", layer=0, num_layers=8,
             ntl=4, seed=15),
        dict(name="tri_70_small_last", B=1, N=70, H=2, D=128, cfg=(2, 5, 9), mask="triangle", seed=16),
        dict(name="tri_128_bf16run", B=1, N=128, H=4, D=64, cfg=(4, 8, 16), mask="triangle", seed=11,
             run_dtype="bfloat16"),
    ]
    if not quick:
        cases += [
            dict(name="helper_tri_2049", B=1, N=2049, H=1, D=64, cfg=(4, 32, 64), mask="helper", layer=0,
                 num_layers=4, ntl=2, seed=17),
            dict(name="helper_dense_2049", B=1, N=2049, H=1, D=64, cfg=(4, 32, 64), mask="helper", layer=3,
                 num_layers=4, ntl=2, seed=18),
        ]
    return cases


def gen_attention(mods, quick):
    tm, at = mods["trianglemix_attention"], mods["ascend_trianglemix"]
    cpu = torch.device("cpu")
    arrays, meta, timings = {}, [], {}
    for c in attention_cases(quick):
        B, N, H, D = c["B"], c["N"], c["H"], c["D"]
        q = orc.seeded_normal(c["seed"] * 3 + 0, (B, N, H, D))
        k = orc.seeded_normal(c["seed"] * 3 + 1, (B, N, H, D))
        v = orc.seeded_normal(c["seed"] * 3 + 2, (B, N, H, D))
        run_dtype = getattr(torch, c.get("run_dtype", "float32"))
        tq, tk, tv = (torch.from_numpy(x).to(run_dtype) for x in (q, k, v))
        scale = D ** -0.5
        si, sl, last = c["cfg"]
        cfg = tm.TriangleMixConfig(si, sl, last, num_triangle_layers=c.get("ntl"))
        if c["mask"] == "helper":
            mod = at.AscendTriangleMixAttention(H, D, scale, cfg, layer_idx=c["layer"], num_layers=c["num_layers"])
            t0 = time.time()
            out = mod(tq, tk, tv)
            timings[c["name"] + "_s"] = time.time() - t0
        else:
            gen = tm.TriangleMixMaskGenerator(cfg)
            am = gen.get_attention_mask(N, 0, cpu, run_dtype, use_triangle=(c["mask"] == "triangle"))
            mod = at.AscendTriangleMixAttention(H, D, scale)
            t0 = time.time()
            out = mod(tq, tk, tv, attn_mask=am)
            timings[c["name"] + "_s"] = time.time() - t0
        assert out.shape == (B, N, H, D) and out.is_contiguous()
        arrays[c["name"]] = out.float().numpy()
        h = hashlib.sha256(q.tobytes() + k.tobytes() + v.tobytes()).hexdigest()
        meta.append({**c, "scale": scale, "inputs_sha256": h, "out_dtype": str(out.dtype)})
        print("attention", c["name"], "done", flush=True)
    # Hkv != H is rejected by the eager class (no grouped heads there)
    try:
        at.AscendTriangleMixAttention(4, 16, 0.25)(torch.zeros(1, 8, 4, 16), torch.zeros(1, 8, 2, 16),
                                                  torch.zeros(1, 8, 2, 16))
        gqa = "ok"
    except RuntimeError:
        gqa = "RuntimeError"
    return arrays, {"cases": meta, "gqa_in_eager_class": gqa}, timings


def gen_config0(mods, gold):
    """The reference's own CPU-runnable configuration (Qwen3-0.6B attention shape at 2048 tokens:
    16 heads, head dim 128, defaults 4/32/64, fp32 run on bf16-rounded inputs) with the explicit
    triangle mask.  Only sampled rows are kept (every 32nd row and the last 70)."""
    tm, at = mods["trianglemix_attention"], mods["ascend_trianglemix"]
    B, N, H, D, seed = 1, 2048, 16, 128, 21
    q = orc.seeded_normal(seed * 3 + 0, (B, N, H, D))
    k = orc.seeded_normal(seed * 3 + 1, (B, N, H, D))
    v = orc.seeded_normal(seed * 3 + 2, (B, N, H, D))
    cfg = tm.TriangleMixConfig(4, 32, 64)
    gen = tm.TriangleMixMaskGenerator(cfg)
    t0 = time.time()
    am = gen.get_attention_mask(N, 0, torch.device("cpu"), torch.float32, use_triangle=True)
    t_mask = time.time() - t0
    mod = at.AscendTriangleMixAttention(H, D, D ** -0.5)
    t0 = time.time()
    out = mod(*(torch.from_numpy(x) for x in (q, k, v)), attn_mask=am)
    t_attn = time.time() - t0
    rows = sorted(set(range(0, N, 32)) | set(range(N - 70, N)))
    np.savez_compressed(os.path.join(gold, "attention_config0.npz"), rows=np.asarray(rows, np.int32),
                        out_rows=out[0, rows].numpy())
    meta = {"B": B, "N": N, "H": H, "D": D, "seed": seed, "cfg": [4, 32, 64], "scale": D ** -0.5,
            "inputs_sha256": hashlib.sha256(q.tobytes() + k.tobytes() + v.tobytes()).hexdigest(),
            "mask_nnz": int((am == 0).sum()), "reference_mask_seconds": t_mask, "reference_attention_seconds": t_attn}
    with open(os.path.join(gold, "attention_config0.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("config0 done", meta, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--quick", action="store_true", help="skip the N > 512 cases (minutes of Python loops)")
    ap.add_argument("--config0-only", action="store_true", help="only (re)write attention_config0.*")
    args = ap.parse_args()
    mods = load_reference(args.reference)
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    torch.set_num_threads(8)
    if args.config0_only:
        gen_config0(mods, gold)
        return

    masks, t_mask = gen_masks(mods["trianglemix_attention"], args.quick)
    with open(os.path.join(gold, "masks.json"), "w") as f:
        json.dump({"cases": masks}, f, indent=0)
    gen_config0(mods, gold)
    sel = gen_selection(mods, args.quick)
    with open(os.path.join(gold, "selection.json"), "w") as f:
        json.dump(sel, f, indent=0)
    arrays, meta, t_attn = gen_attention(mods, args.quick)
    np.savez_compressed(os.path.join(gold, "attention.npz"), **arrays)
    with open(os.path.join(gold, "attention.json"), "w") as f:
        json.dump(meta, f, indent=0)

    # timing of the reference's own CPU path in this container (baseline only)
    tm, at = mods["trianglemix_attention"], mods["ascend_trianglemix"]
    cpu = torch.device("cpu")
    q = torch.randn(1, 512, 16, 128)
    am = tm.TriangleMixMaskGenerator(tm.TriangleMixConfig()).get_attention_mask(512, 0, cpu, torch.float32, True)
    mod = at.AscendTriangleMixAttention(16, 128, 128 ** -0.5)
    t0 = time.time()
    mod(q, q, q, attn_mask=am)
    t_attn["eager_fp32_n512_h16_d128_s"] = time.time() - t0
    with open(os.path.join(gold, "ref_timings.json"), "w") as f:
        json.dump({"host_cpus": os.cpu_count(), "torch_threads": torch.get_num_threads(),
                   "torch": torch.__version__, **t_mask, **t_attn}, f, indent=0)
    print("golden fixtures written to", gold)


if __name__ == "__main__":
    main()
