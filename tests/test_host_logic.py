"""CPU tests (no GPU): the package's host logic against the golden fixtures,
the C-ABI library's symbol table against the header, the host-side index sizing
against the oracle, and the loud failure of the product path without a device.
No compute entry point of the library is called here."""
import ctypes
import os
import re
import types

import numpy as np
import pytest
import torch

from oracle import trianglemix_oracle as orc
from vllm_trianglemix_amd import (AscendTriangleMixAttention, TriangleMixAnalyzer, TriangleMixAttention,
                                  TriangleMixConfig, TriangleMixInferenceConfig, TriangleMixMaskGenerator,
                                  create_trianglemix_config_from_hf, _lib, ops)
from vllm_trianglemix_amd.backend import (TriangleMixAttentionBackend, TriangleMixAttentionImpl,
                                          TriangleMixMetadataBuilder)
from vllm_trianglemix_amd.parallel import shard_heads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "trianglemix_hip.h")


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(tmx_[a-z0-9_]+)\s*\(", text)))
    assert declared, "no declarations found in the header"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared
    lib = _lib.load()  # raises if the library is missing: there is no fallback
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared but not exported"
    assert lib.tmx_abi_version() == _lib.TMX_ABI_VERSION
    assert lib.tmx_attention_block_n() == 32


def test_struct_layouts_match_the_header():
    """The ctypes mirrors must have the sizes the C compiler gives the header's structs."""
    import subprocess, tempfile
    src = ('#include <stdio.h>\n#include "trianglemix_hip.h"\n'
           'int main(void){printf("%zu %zu %zu\\n", sizeof(tmx_index_sizes_t), sizeof(tmx_index_t), '
           'sizeof(tmx_attention_args_t));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "s.c"), os.path.join(d, "s")
        open(c, "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    assert sizes == [ctypes.sizeof(_lib.IndexSizes), ctypes.sizeof(_lib.Index), ctypes.sizeof(_lib.AttentionArgs)]


def test_torch_library_binding_registers_ops():
    from vllm_trianglemix_amd import torch_ops
    C = torch_ops.load()
    for name in ("trianglemix_index_sizes", "trianglemix_build_block_index", "trianglemix_attention",
                 "trianglemix_reshape_and_cache", "trianglemix_reshape_and_cache_fp8",
                 "trianglemix_gather_dequant_kv_fp8"):
        assert hasattr(C, name)
    cu, sl = torch.tensor([0, 100, 164], dtype=torch.int32), torch.tensor([100, 2500], dtype=torch.int32)
    got = C.trianglemix_index_sizes(cu, sl, 4, 32, 64, 32, 32, ops.MODES["triangle"], 0, 32)
    sz = ops.index_sizes(cu.numpy(), sl.numpy(), 4, 32, 64, 32, 32, ops.MODES["triangle"], chunk_tiles=32)
    assert list(got) == [sz.num_qblocks, sz.nnz_tiles, sz.num_items, sz.num_partials, sz.num_split_qblocks,
                         sz.max_tiles_per_qblock]
    assert C.trianglemix_suggest_chunk_tiles(torch.tensor([0, 32768], dtype=torch.int32),
                                             torch.tensor([32768], dtype=torch.int32), 4, 32, 64, 32, 32,
                                             ops.MODES["triangle"], 0, 8, 256) == 64
    full = torch.tensor([100, 4000], dtype=torch.int32)
    got = C.trianglemix_index_sizes(cu, sl, 4, 32, 64, 32, 32, ops.MODES["triangle"], 0, 32, full)
    sz = ops.index_sizes(cu.numpy(), sl.numpy(), 4, 32, 64, 32, 32, ops.MODES["triangle"], chunk_tiles=32,
                         full_lens_cpu=full.numpy())
    assert list(got)[:3] == [sz.num_qblocks, sz.nnz_tiles, sz.num_items] and sz.num_partials == 0


@pytest.mark.parametrize("heads,kv,bm", [(32, 8, 32), (16, 8, 64), (8, 8, 128), (64, 8, 32), (4, 1, 32), (6, 2, 128)])
def test_block_m_rule(heads, kv, bm):
    assert ops.attention_block_m(heads, kv) == bm
    with pytest.raises(RuntimeError):
        ops.attention_block_m(7, 2)


# ------------------------------------------------------------------ index sizing (host-only entry point)
@pytest.mark.parametrize("mode", ["none", "causal", "triangle"])
@pytest.mark.parametrize("chunk", [2, 64])
def test_index_sizes_match_oracle(mode, chunk):
    rng = np.random.default_rng(7)
    for _ in range(6):
        batch = int(rng.integers(1, 5))
        seq = rng.integers(1, 700, size=batch)
        qlen = np.asarray([int(rng.integers(1, s + 1)) for s in seq])
        si, sl, last = int(rng.integers(0, 9)), int(rng.integers(0, 70)), int(rng.integers(0, 90))
        bm, bn, m = 32, 32, ops.MODES[mode]
        cu = np.concatenate([[0], np.cumsum(qlen)]).astype(np.int32)
        sz = ops.index_sizes(cu, seq.astype(np.int32), si, sl, last, bm, bn, m, chunk_tiles=chunk)
        row_ptr, col_idx, qb_seq, qb_row0 = orc.block_index_closed_form(qlen, seq, si, sl, last, bm, bn, m)
        items, part_ptr = orc.work_items(row_ptr, chunk)
        assert sz.num_qblocks == len(row_ptr) - 1
        assert sz.nnz_tiles == len(col_idx)
        assert sz.num_items == len(items)
        assert sz.num_partials == int(part_ptr[-1])
        assert sz.num_split_qblocks == int(np.count_nonzero(np.diff(part_ptr)))
        assert sz.max_tiles_per_qblock == int(np.max(np.diff(row_ptr)))
        # chunked prefill: the pattern of the complete prompt
        full = (seq + rng.integers(-3, 300, size=batch)).astype(np.int32)
        szf = ops.index_sizes(cu, seq.astype(np.int32), si, sl, last, bm, bn, m, chunk_tiles=chunk, full_lens_cpu=full)
        rp, ci, _, _ = orc.block_index_closed_form(qlen, seq, si, sl, last, bm, bn, m, full)
        bf = orc.block_index_bruteforce(qlen, seq, si, sl, last, bm, bn, m, full)
        assert np.array_equal(rp, bf[0]) and np.array_equal(ci, bf[1])
        assert (szf.num_qblocks, szf.nnz_tiles) == (len(rp) - 1, len(ci))
        assert szf.num_items == len(orc.work_items(rp, chunk)[0])


def test_chunk_masks_are_blocks_of_the_whole_mask():
    """Oracle: with the prompt's length known, the mask of a chunk (rows lo..hi-1, keys 0..hi-1) is the
    corresponding block of the reference's mask of the whole prompt (trianglemix_attention.py:173-198)."""
    for P, (si, sl, last) in ((300, (4, 32, 64)), (97, (0, 3, 10)), (130, (2, 0, 1))):
        whole = orc.triangle_mask_loops(P, si, sl, last).astype(bool)
        for lo, hi in ((0, P // 3), (P // 3, P - 20), (P - 20, P)):
            assert np.array_equal(orc.triangle_mask(hi, si, sl, last, q_len=hi - lo, n_full=P), whole[lo:hi, :hi])
        # a decode row past the prompt is dense-causal whatever the prompt length was
        assert orc.triangle_mask(P + 9, si, sl, last, q_len=1, n_full=P).all() or last <= 1  # N - i = 1 < last


def test_fp8_e4m3fn_oracle_matches_torch():
    """The oracle's fp8 conversions are pinned to torch's float8_e4m3fn (CPU) on every finite code, on
    midpoints between neighbouring codes (ties to even) and on random values; beyond +-448 the oracle
    saturates, as the cache write of the reference does (quant_utils.cuh), where torch gives NaN."""
    allb = np.arange(256, dtype=np.uint8)
    want = torch.from_numpy(allb).view(torch.float8_e4m3fn).float().numpy()
    got = orc.fp8_e4m3fn_decode(allb)
    assert np.array_equal(np.isnan(want), np.isnan(got)) and np.array_equal(want[~np.isnan(want)], got[~np.isnan(got)])
    pos = got[:127]
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(size=50000) * rng.choice([1e-3, 0.1, 1, 30, 200], 50000), pos, -pos,
                        (pos[1:] + pos[:-1]) / 2, [0.0, -0.0, 447.9, 2.0 ** -10, 2.0 ** -9 * 0.5]]).astype(np.float32)
    x = x[np.abs(x) <= 448]
    t = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(t, orc.fp8_e4m3fn_encode(x))
    assert orc.fp8_e4m3fn_encode(np.asarray([1e9, -500.0, np.nan])).tolist() == [126, 254, 127]
    assert orc.fp8_cache_roundtrip(np.asarray([3.0, 1000.0], np.float32), 2.0).tolist() == [3.0, 896.0]


def test_inference_config_from_env():
    from vllm_trianglemix_amd.config import TriangleMixInferenceConfig
    off = TriangleMixInferenceConfig.from_env({})
    assert not off.enable_trianglemix and off.to_trianglemix_config() is None
    on = TriangleMixInferenceConfig.from_env({"VLLM_TRIANGLEMIX_ENABLED": "1", "VLLM_TRIANGLEMIX_LAYERS": "12",
                                              "VLLM_NPU_OPTIMIZATION": "0"})
    cfg = on.to_trianglemix_config()
    assert on.enable_trianglemix and not on.use_npu_optimization
    assert (cfg.num_sink_tokens, cfg.sliding_window_size, cfg.num_last_tokens, cfg.num_triangle_layers) == (4, 32, 64, 12)
    assert cfg.uses_triangle(11) and not cfg.uses_triangle(12)
    assert not TriangleMixInferenceConfig.from_env({"VLLM_TRIANGLEMIX_ENABLED": "false"}).enable_trianglemix


def test_chunk_tiles_advice():
    """One piece of the dense rows per CU for prompts with sparse and dense rows (the measured optimum on a
    256-CU MI355X: 16 / 32 / 64 / 128 tiles at 8k / 16k / 32k / 64k tokens for Qwen3-8B's heads), fixed
    lengths for batches of dense rows only."""
    u = ops.units_per_item(32, 8)
    assert u == 8 and ops.units_per_item(16, 8) == 8 and ops.units_per_item(4, 1) == 1 and ops.units_per_item(6, 1) == 3
    for n, want in ((8192, 16), (16384, 32), (32768, 64), (65536, 128)):
        assert ops.suggest_chunk_tiles([0, n], [n], 4, 32, 64, 32, 32, ops.MODES["triangle"], u, num_cus=256) == want
    # a rank's share of an 8-GPU job: 8 prompts, 4 query heads on 1 K/V head
    assert ops.suggest_chunk_tiles(np.arange(9) * 32768, [32768] * 8, 4, 32, 64, 32, 32, ops.MODES["triangle"], 1,
                                   num_cus=256) == 64
    # decode steps: one piece per workgroup slot (2 x 256), between 8 and 512 tiles
    for b, n, want in ((64, 8192, 256), (8, 32768, 128), (4, 16384, 32), (1, 8192, 8), (1, 131072, 64), (256, 2048, 256)):
        assert ops.suggest_chunk_tiles(np.arange(b + 1), [n] * b, 4, 32, 64, 32, 32, ops.MODES["triangle"], u,
                                       num_cus=256) == want, (b, n)
    # a prompt with sparse rows and decode rows in one step: the prompt's dense rows make one piece per CU, the
    # decode rows ask for pieces of up to 32 tiles
    def mixed(p, b, n):
        q = [p] + [1] * b
        return ops.suggest_chunk_tiles(np.concatenate([[0], np.cumsum(q)]), [p] + [n] * b, 4, 32, 64, 32, 32,
                                       ops.MODES["triangle"], u, tri_min_len=2048, num_cus=256)
    assert mixed(8192, 32, 4096) == 32 and mixed(32768, 8, 8192) == 64 and mixed(2100, 64, 8192) == 32
    assert mixed(2048, 64, 8192) == 256  # a prompt below the gate is attended causally: dense rows only
    # causal prompts: between 64 and 256 tiles
    for n, want in ((32768, 256), (8192, 256), (2048, 64)):
        assert ops.suggest_chunk_tiles([0, n], [n], 4, 32, 64, 32, 32, ops.MODES["causal"], u, num_cus=256) == want
    # chunked prefill: a middle chunk of a long prompt has sparse rows only (no piece to balance): the default
    assert ops.suggest_chunk_tiles([0, 4096], [8192], 4, 32, 64, 32, 32, ops.MODES["triangle"], u, num_cus=256,
                                   full_lens_cpu=[32768]) == 64
    # a wide window: the sparse rows' own lists are never cut (1024-key window = up to 34 tiles per q-block)
    assert ops.suggest_chunk_tiles([0, 4096], [4096], 4, 1024, 64, 32, 32, ops.MODES["triangle"], u, num_cus=256) == 34
    with pytest.raises(RuntimeError):
        ops.suggest_chunk_tiles([0, 9], [5], 4, 32, 64, 32, 32, ops.MODES["triangle"], u, num_cus=256)


def test_index_sizes_rejects_bad_input():
    with pytest.raises(RuntimeError):
        ops.index_sizes([0, 5, 9], [5], 4, 32, 64, 32, 32, ops.MODES["triangle"])
    with pytest.raises(_lib.TmxError):  # query longer than its sequence
        ops.index_sizes([0, 9], [5], 4, 32, 64, 32, 32, ops.MODES["triangle"])


# ------------------------------------------------------------------ configuration and layer selection
def test_should_use_triangle_tables(golden_selection):
    for case in golden_selection["should_use_triangle"]:
        cfg = TriangleMixConfig(num_triangle_layers=case["num_triangle_layers"],
                                triangle_layer_indices=case["triangle_layer_indices"])
        gen = TriangleMixMaskGenerator(cfg)
        assert [gen.should_use_triangle(i) for i in range(len(case["table"]))] == case["table"]
        assert [cfg.uses_triangle(i) for i in range(len(case["table"]))] == case["table"]


def test_helper_decisions(golden_selection):
    for case in golden_selection["helper"]:
        cfg = TriangleMixConfig(num_triangle_layers=case["num_triangle_layers"],
                                triangle_layer_indices=case["triangle_layer_indices"])
        helper = TriangleMixAttention(cfg, num_layers=case["num_layers"])
        assert list(cfg.triangle_layer_indices) == case["indices_after_init"]  # the shared config is filled in place
        for d in case["decisions"]:
            assert helper.mask_kind(d["n"], d["layer"]) == d["kind"], d


def test_presets(golden_selection):
    p = golden_selection["presets"]

    def tm(cfg):
        t = cfg.to_trianglemix_config()
        return [t.num_sink_tokens, t.sliding_window_size, t.num_last_tokens, t.num_triangle_layers,
                list(t.triangle_layer_indices)]

    for size in ("14B", "32B", "8B"):
        want = p[f"for_qwen3_{size}"]
        cfg = TriangleMixInferenceConfig.for_qwen3(size)
        for k, v in want.items():
            assert (tm(cfg) if k == "tm_cfg" else getattr(cfg, k)) == v, (size, k)
    assert TriangleMixInferenceConfig.for_qwen3("14B", 3).num_triangle_layers == \
        p["for_qwen3_14B_override3"]["num_triangle_layers"]
    a = TriangleMixInferenceConfig.for_ascend_npu()
    for k, v in p["for_ascend_npu"].items():
        assert getattr(a, k) == v
    d = TriangleMixInferenceConfig()
    assert d.enable_trianglemix is False and d.to_trianglemix_config() is None and d.triangle_layer_indices == []
    c = TriangleMixInferenceConfig(enable_trianglemix=True, triangle_layer_indices=[0, 1, 2])
    assert tm(c) == p["custom_indices"]["tm_cfg"]


def test_ascend_presets(golden_selection):
    from vllm_trianglemix_amd.ascend_npu import (AscendBlockSparseMaskGenerator, AscendNPUTriangleMixAttention,
                                                  create_ascend_trianglemix_config)
    p = golden_selection["presets"]
    for size in ("14B", "32B", "7B"):
        cfg = create_ascend_trianglemix_config(size)
        for k, v in p[f"ascend_{size}"].items():
            assert getattr(cfg, k) == v, (size, k)
    gen = AscendBlockSparseMaskGenerator(create_ascend_trianglemix_config("14B"))
    assert gen.block_pattern() == (1, 1, 1)  # defaults at 64-token blocks: every parameter floors to 1
    helper = AscendNPUTriangleMixAttention(create_ascend_trianglemix_config("14B"), num_layers=40)
    assert helper.config.triangle_layer_indices == list(range(12))
    assert helper.get_attn_mask(512, 0, torch.device("cpu")) is None  # no mask up to 512 tokens
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        helper.get_attn_mask(513, 0, torch.device("cpu"))


def test_config_from_hf(golden_selection):
    for case in golden_selection["from_hf"]:
        hf = types.SimpleNamespace(**case["attrs"])
        cfg = create_trianglemix_config_from_hf(hf, use_trianglemix=case["use"], num_triangle_layers=case["ntl"])
        if case["result"] is None:
            assert cfg is None
        else:
            assert [cfg.num_sink_tokens, cfg.sliding_window_size, cfg.num_last_tokens, cfg.num_triangle_layers,
                    list(cfg.triangle_layer_indices)] == case["result"]
    with pytest.raises(TypeError):  # a config whose sliding_window is None fails as in the reference
        create_trianglemix_config_from_hf(types.SimpleNamespace(sliding_window=None), use_trianglemix=True)


def test_analyzer(golden_selection):
    for case in golden_selection["analyzer"]:
        an = TriangleMixAnalyzer(num_layers=len(case["grads"]))
        for layer, grads in enumerate(case["grads"]):
            for gval in grads:
                an.record_middle_qk_gradient(layer, torch.tensor([gval, gval]))
        assert an.get_triangle_layers(case["k"]) == case["layers"]


# ------------------------------------------------------------------ no CPU fallback
def test_product_path_refuses_cpu_tensors():
    gen = TriangleMixMaskGenerator(TriangleMixConfig())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gen._create_streaming_mask(8, torch.device("cpu"), torch.float32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gen.get_attention_mask(8, 0, torch.device("cpu"), torch.float32)
    attn = AscendTriangleMixAttention(num_heads=2, head_dim=64, scale=0.125)
    x = torch.zeros((1, 8, 2, 64), dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        attn(x, x, x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.reshape_and_cache(torch.zeros(1, 1, 64), torch.zeros(1, 1, 64), torch.zeros(1, 16, 1, 64),
                              torch.zeros(1, 16, 1, 64), torch.zeros(1, dtype=torch.int64))


def test_missing_library_is_an_error(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libtrianglemix_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


# ------------------------------------------------------------------ backend surface
def test_backend_surface():
    b = TriangleMixAttentionBackend
    assert b.accept_output_buffer is True
    assert b.get_name() == "TRIANGLEMIX_ROCM"
    assert b.get_impl_cls() is TriangleMixAttentionImpl and b.get_builder_cls() is TriangleMixMetadataBuilder
    assert b.get_kv_cache_shape(10, 16, 8, 128) == (2, 10, 16, 8, 128)
    with pytest.raises(ValueError):
        b.get_kv_cache_shape(10, 24, 8, 128)
    assert set(b.get_supported_dtypes()) == {torch.float16, torch.bfloat16}
    b.validate_head_size(128)
    with pytest.raises(ValueError):
        b.validate_head_size(96)


def test_impl_constructor_rules():
    cfg = TriangleMixConfig(num_triangle_layers=2)
    on = TriangleMixAttentionImpl(32, 128, 0.088, 8, None, None, "auto", None, "decoder", None,
                                  trianglemix_config=cfg, layer_idx=1)
    off = TriangleMixAttentionImpl(32, 128, 0.088, 8, None, None, "auto", trianglemix_config=cfg, layer_idx=2)
    assert on.use_triangle and not off.use_triangle and on.pattern == (4, 32, 64) and on.tri_min_len == 2048
    flag = TriangleMixAttentionImpl(16, 64, 0.125, 8, None, None, "auto", use_triangle=True, num_sink_tokens=2,
                                    sliding_window_size=8, num_last_tokens=16)
    assert flag.use_triangle and flag.pattern == (2, 8, 16)
    for bad in (dict(alibi_slopes=[0.5]), dict(sliding_window=128), dict(logits_soft_cap=30.0),
                dict(kv_cache_dtype="fp8_e5m2"), dict(attn_type="encoder")):
        kw = dict(num_heads=32, head_size=128, scale=0.088, num_kv_heads=8)
        kw.update(bad)
        with pytest.raises(NotImplementedError):
            TriangleMixAttentionImpl(**kw)
    with pytest.raises(ValueError):
        TriangleMixAttentionImpl(32, 96, 0.1, 8)
    # profiling run: no metadata -> output untouched
    out = torch.full((4, 32 * 128), 7.0)
    assert on.forward(None, None, None, None, None, None, output=out) is out and bool((out == 7.0).all())


def test_metadata_builder_copies_host_description():
    spec = types.SimpleNamespace(block_size=16)
    builder = TriangleMixMetadataBuilder(spec, ["layers.0.attn"], None, torch.device("cpu"))
    common = types.SimpleNamespace(
        num_actual_tokens=7, max_query_len=4, max_seq_len=9,
        query_start_loc=torch.tensor([0, 3, 7], dtype=torch.int32), query_start_loc_cpu=torch.tensor([0, 3, 7], dtype=torch.int32),
        seq_lens=torch.tensor([9, 4], dtype=torch.int32), seq_lens_cpu=torch.tensor([9, 4], dtype=torch.int32),
        block_table_tensor=torch.zeros((2, 4), dtype=torch.int32), slot_mapping=torch.arange(7, dtype=torch.int64))
    md = builder.build(0, common)
    assert md.page_size == 16 and md.num_actual_tokens == 7
    assert md.query_start_loc_cpu.dtype == np.int32 and md.query_start_loc_cpu.tolist() == [0, 3, 7]
    assert md.seq_lens_cpu.tolist() == [9, 4]
    assert md.prompt_lens is None and md.prompt_lens_cpu is None
    with pytest.raises(NotImplementedError):
        builder.build(5, common)
    # chunked prefill: a runner that passes the prompt lengths along gets the whole-prompt pattern
    common.prompt_lens_cpu = torch.tensor([3000, 4], dtype=torch.int32)
    md = builder.build(0, common)
    assert md.prompt_lens_cpu.tolist() == [3000, 4] and md.prompt_lens.dtype == torch.int32
    assert md.prompt_lens.tolist() == [3000, 4]


# ------------------------------------------------------------------ head sharding rule
def test_shard_heads_rule():
    # K/V heads partitioned while there are enough of them, replicated beyond (one K/V head per rank)
    for world in (1, 2, 4, 8):
        shards = [shard_heads(32, 8, r, world) for r in range(world)]
        assert sum(s.num_heads for s in shards) == 32
        assert [s.head_start for s in shards] == [r * 32 // world for r in range(world)]
        assert all(s.num_kv_heads == 8 // world and not s.kv_replicated for s in shards)
        for s in shards:  # every local query head finds its K/V head inside the local K/V slice
            for h in range(s.head_start, s.head_start + s.num_heads):
                assert s.kv_head_start <= h // 4 < s.kv_head_start + s.num_kv_heads
    s = [shard_heads(64, 8, r, 16) for r in range(16)]
    assert all(x.num_kv_heads == 1 and x.kv_replicated for x in s)
    assert [x.kv_head_start for x in s] == [r // 2 for r in range(16)]
    with pytest.raises(ValueError):
        shard_heads(30, 8, 0, 4)
    with pytest.raises(ValueError):
        shard_heads(32, 8, 0, 3)


def test_bench_flop_count_equals_mask_population():
    """bench.py prices the matrix-core side with the number of kept (query, key) pairs: its closed form
    equals the population of the oracle's masks."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tmx_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for n, (si, sl, last) in ((1, (4, 32, 64)), (70, (4, 32, 64)), (300, (4, 32, 64)), (50, (0, 0, 0)),
                              (200, (3, 6, 8)), (129, (2, 4, 6))):
        assert bench.allowed_pairs(n, si, sl, last, True) == int(orc.triangle_mask(n, si, sl, last).sum())
        assert bench.allowed_pairs(n, si, sl, last, False) == n * (n + 1) // 2


def test_index_sizes_run_lengths_on_long_requests():
    """The host-side sizing answers for runs of q-blocks at once inside long triangle requests (every full
    q-block between the first few and the trailing dense rows lists the same number of tiles); the totals must
    equal the oracle's q-block by q-block count for every block height, with context, with the prompt's full
    length known, with wide windows and sinks, and without a sink or a window at all."""
    rng = np.random.default_rng(11)
    m = ops.MODES["triangle"]
    for case in range(60):
        bm = int(rng.choice([32, 64, 128]))
        batch = int(rng.integers(1, 4))
        seq = rng.integers(1, 5000, size=batch)
        qlen = np.asarray([int(rng.integers(1, s + 1)) if rng.random() < 0.5 else int(s) for s in seq])
        si = int(rng.choice([-1, 0, 4, 31, 32, 33, 100]))
        sl = int(rng.choice([0, 1, 31, 32, 33, 64, 200, 1024]))
        last = int(rng.choice([0, 1, 64, 65, 300]))
        chunk = int(rng.choice([2, 16, 64]))
        full = None if case % 3 else (seq + rng.integers(0, 3000, size=batch)).astype(np.int32)
        cu = np.concatenate([[0], np.cumsum(qlen)]).astype(np.int32)
        sz = ops.index_sizes(cu, seq.astype(np.int32), si, sl, last, bm, 32, m, chunk_tiles=chunk, full_lens_cpu=full)
        rp, ci, _, _ = orc.block_index_closed_form(qlen, seq, si, sl, last, bm, 32, m, full)
        items, part_ptr = orc.work_items(rp, chunk)
        got = (sz.num_qblocks, sz.nnz_tiles, sz.num_items, sz.num_partials, sz.num_split_qblocks,
               sz.max_tiles_per_qblock)
        want = (len(rp) - 1, len(ci), len(items), int(part_ptr[-1]), int(np.count_nonzero(np.diff(part_ptr))),
                int(np.max(np.diff(rp))))
        assert got == want, (case, bm, seq, qlen, si, sl, last, chunk, full)


def test_dense_stage_rows_host_rule():
    """Rows of the dense causal kernel's staging area: the sequence lengths of the requests with 17 query rows or
    more, each rounded up to 64; 0 when those requests average fewer query rows than the threshold (default 4096)."""
    assert ops.dense_stage_rows([0, 4096], [4096]) == 4096
    assert ops.dense_stage_rows([0, 4095], [4095]) == 0
    assert ops.dense_stage_rows([0, 4095], [4095], 0) == 4096
    assert ops.dense_stage_rows([0, 2048], [100000], 2048) == 100032            # a chunk behind a long context
    assert ops.dense_stage_rows([0, 8192, 8193, 8209], [8192, 500, 16]) == 8192  # decode row and a 16-row request: not staged
    assert ops.dense_stage_rows([0, 8192, 8209], [8192, 17]) == 8192 + 64        # 17 rows: the kernel's, average still 4104
    assert ops.dense_stage_rows([0, 8192, 8209, 8226], [8192, 17, 17]) == 0      # average below the threshold
    assert ops.dense_stage_rows([0], []) == 0
    with pytest.raises(RuntimeError):
        ops.dense_stage_rows([0, 5], [5, 6])


def test_index_sizes_without_the_dense_kernels_requests():
    """An index that leaves the requests of 17 query rows or more to the dense causal kernel is sized like the
    index of the batch without those requests' rows."""
    rng = np.random.default_rng(5)
    for _ in range(20):
        B = int(rng.integers(1, 12))
        q = rng.choice([0, 1, 1, 3, 16, 17, 18, 64, 700, 5000], size=B)
        ctx = rng.choice([0, 0, 5, 100, 4000], size=B)
        cu = np.concatenate([[0], np.cumsum(q)]).astype(np.int32)
        sl = (q + ctx).astype(np.int32)
        cu_f, sl_f = ops.without_dense_requests(cu, sl)
        assert np.array_equal(np.diff(cu_f), np.where(q >= 17, 0, q)) and np.array_equal(sl_f, np.where(q >= 17, ctx, q + ctx))
        for chunk in (8, 64):
            a = ops.index_sizes(cu, sl, 0, 0, 0, 32, 32, ops.MODES["causal"], chunk_tiles=chunk, dense_rows_from=17)
            b = ops.index_sizes(cu_f, sl_f, 0, 0, 0, 32, 32, ops.MODES["causal"], chunk_tiles=chunk)
            for f in ("num_qblocks", "nnz_tiles", "num_items", "num_partials", "num_split_qblocks"):
                assert getattr(a, f) == getattr(b, f), (f, q, ctx)
    # one 32k prompt of a layer outside the triangle set: nothing left to list
    sz = ops.index_sizes([0, 32768], [32768], 0, 0, 0, 32, 32, ops.MODES["causal"], chunk_tiles=64, dense_rows_from=17)
    assert (sz.num_qblocks, sz.nnz_tiles, sz.num_items, sz.num_partials) == (0, 0, 0, 0)
    assert ops.index_sizes([0, 32768], [32768], 0, 0, 0, 32, 32, ops.MODES["causal"], chunk_tiles=64).num_partials == 8640
