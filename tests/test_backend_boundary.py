"""The vLLM-facing surface, replayed the way vLLM's own code consumes it (no GPU, no vLLM needed):
the expressions below are the consumer lines of the reference, applied to this package's classes."""
import enum
import types

import numpy as np
import pytest
import torch

from vllm_trianglemix_amd import (TriangleMixAttentionBackend, TriangleMixAttentionImpl, TriangleMixConfig,
                                  TriangleMixInferenceConfig, TriangleMixMetadataBuilder)
from vllm_trianglemix_amd import backend as be
from vllm_trianglemix_amd.qwen3_hook import (Qwen3TriangleMixAttention, model_trianglemix_config,
                                             trianglemix_attention_kwargs)


def test_cudagraph_support_is_consumed_like_the_model_runner_does():
    """vllm/v1/worker/gpu_model_runner.py:3612-3619: min over builders of ``cudagraph_support.value``."""
    AttentionCGSupport = be.AttentionCGSupport
    assert [m.name for m in AttentionCGSupport] == ["ALWAYS", "UNIFORM_BATCH", "UNIFORM_SINGLE_TOKEN_DECODE", "NEVER"]
    assert [m.value for m in AttentionCGSupport] == [3, 2, 1, 0]
    builder = TriangleMixAttentionBackend.get_builder_cls()(types.SimpleNamespace(block_size=16), ["l0"], None,
                                                            torch.device("cpu"))
    min_cg_support, min_cg_builder_name = AttentionCGSupport.ALWAYS, None
    for b in (builder,):
        if b.cudagraph_support.value < min_cg_support.value:
            min_cg_support = b.cudagraph_support
            min_cg_builder_name = b.__class__.__name__
    # decode steps replay under a graph; steps with prompts do not (FULL mixed mode resolves to
    # FULL_DECODE_ONLY / FULL_AND_PIECEWISE at gpu_model_runner.py:3620-3643, and the check at :3645-3647 that
    # refuses decode graphs for NEVER does not fire)
    assert min_cg_support is AttentionCGSupport.UNIFORM_SINGLE_TOKEN_DECODE
    assert min_cg_support != AttentionCGSupport.ALWAYS and min_cg_support != AttentionCGSupport.NEVER
    assert min_cg_builder_name == "TriangleMixMetadataBuilder"
    assert isinstance(TriangleMixMetadataBuilder.cudagraph_support, enum.Enum)
    # the base class's members (vllm/v1/attention/backends/utils.py:232-341)
    assert issubclass(TriangleMixMetadataBuilder, be._BuilderBase)
    assert builder.reorder_batch_threshold is None
    assert builder.use_cascade_attention(0, np.zeros(1), 32, 8, False, False, False, 256) is False
    for name in ("build", "build_for_cudagraph_capture", "build_for_drafting", "reorder_batch"):
        assert callable(getattr(builder, name))
    assert (builder.kv_cache_spec.block_size, builder.layer_names, builder.vllm_config) == (16, ["l0"], None)


def test_builder_entry_points_of_the_base_class():
    builder = TriangleMixMetadataBuilder(types.SimpleNamespace(block_size=16), ["l0"], None, torch.device("cpu"))
    common = types.SimpleNamespace(
        num_actual_tokens=3, max_query_len=3, max_seq_len=3,
        query_start_loc=torch.tensor([0, 3], dtype=torch.int32), query_start_loc_cpu=torch.tensor([0, 3], dtype=torch.int32),
        seq_lens=torch.tensor([3], dtype=torch.int32), seq_lens_cpu=torch.tensor([3], dtype=torch.int32),
        block_table_tensor=torch.zeros((1, 1), dtype=torch.int32), slot_mapping=torch.arange(3, dtype=torch.int64))
    for md in (builder.build(common_prefix_len=0, common_attn_metadata=common, fast_build=True),
               builder.build_for_drafting(common, draft_index=0)):
        assert isinstance(md, TriangleMixAttentionBackend.get_metadata_cls()) and md.num_actual_tokens == 3
        assert not md.uniform_decode and not md.capturing
    # a graph is captured from uniform single-token decode steps only; a step with a prompt is refused
    with pytest.raises(NotImplementedError):
        builder.build_for_cudagraph_capture(common)
    # the capture batch the runner builds (gpu_model_runner.py:3612-3619 has read cudagraph_support): one token per
    # request; the layers' launches then depend on the batch size alone, whatever lengths the capture batch carries
    decode = types.SimpleNamespace(
        num_actual_tokens=4, max_query_len=1, max_seq_len=9,
        query_start_loc=torch.arange(5, dtype=torch.int32), query_start_loc_cpu=torch.arange(5, dtype=torch.int32),
        seq_lens=torch.tensor([9, 1, 1, 1], dtype=torch.int32), seq_lens_cpu=torch.tensor([9, 1, 1, 1], dtype=torch.int32),
        block_table_tensor=torch.zeros((4, 1), dtype=torch.int32), slot_mapping=torch.arange(4, dtype=torch.int64))
    md = builder.build_for_cudagraph_capture(decode)
    assert md.capturing and md.uniform_decode and md.builder is builder
    md = builder.build(0, decode)  # the same step outside a capture
    assert not md.capturing and md.uniform_decode
    # which layers can take the fixed-shape decode launch
    mk = lambda **kw: TriangleMixAttentionImpl(kw.pop("H", 32), kw.pop("D", 128), 0.1, kw.pop("Hkv", 8), None, None, "auto",  # noqa: E731
                                               None, "decoder", None, **kw)
    assert mk(use_triangle=True)._decode_fixed_ok() and mk(use_triangle=False)._decode_fixed_ok()
    assert not mk(use_triangle=True, num_last_tokens=1)._decode_fixed_ok()   # the decode row would be sparse
    assert mk(use_triangle=False, num_last_tokens=1)._decode_fixed_ok()
    assert not mk(D=64)._decode_fixed_ok() and not mk(H=6, Hkv=2)._decode_fixed_ok()


def test_backend_name_is_tolerated_by_the_selector_lookup():
    """vllm/attention/selector.py:21-32: an out-of-tree name maps to None instead of raising."""
    class _Backend(enum.Enum):  # a few of the in-tree members (vllm/platforms/interface.py)
        FLASH_ATTN = enum.auto()
        TRITON_ATTN = enum.auto()
        ROCM_ATTN = enum.auto()

    def backend_name_to_enum(backend_name):
        assert backend_name is not None
        return _Backend[backend_name] if backend_name in _Backend.__members__ else None

    name = TriangleMixAttentionBackend.get_name()
    assert isinstance(name, str) and backend_name_to_enum(name) is None


class FakeAttention:
    """The constructor tail of vllm/attention/layer.py:186-201, line for line in behaviour: pick the backend
    class, build the impl with TEN positional arguments and ``**extra_impl_args``."""

    def __init__(self, num_heads, head_size, scale, num_kv_heads=None, alibi_slopes=None, cache_config=None,
                 quant_config=None, logits_soft_cap=None, per_layer_sliding_window=None, use_mla=False,
                 use_sparse=False, prefix="", attn_type="decoder", kv_sharing_target_layer_name=None,
                 attn_backend=None, **extra_impl_args):
        kv_cache_dtype = "auto" if cache_config is None else cache_config.cache_dtype
        sliding_window = per_layer_sliding_window
        self.default_backend_used = attn_backend is None
        self.attn_backend = attn_backend if attn_backend is not None else types.SimpleNamespace(
            get_impl_cls=lambda: (lambda *a, **k: ("default-impl", a, k)), get_name=lambda: "ROCM_ATTN",
            accept_output_buffer=True)
        impl_cls = self.attn_backend.get_impl_cls()
        self.impl = impl_cls(num_heads, head_size, scale, num_kv_heads, alibi_slopes, sliding_window, kv_cache_dtype,
                             logits_soft_cap, attn_type, kv_sharing_target_layer_name, **extra_impl_args)
        self.use_output = self.attn_backend.accept_output_buffer
        self.prefix = prefix


def test_qwen3_call_site_builds_the_trianglemix_impl():
    """qwen3.py:79-114, 143-154 made live: 36 layers of Qwen3-8B sharing one config; layers below
    num_triangle_layers run the triangle pattern, the rest dense causal attention, on this backend."""
    cfg = model_trianglemix_config(TriangleMixInferenceConfig.for_qwen3("8B"))
    assert cfg.num_triangle_layers == 8 and cfg.triangle_layer_indices == []
    layers = [Qwen3TriangleMixAttention(FakeAttention, 32, 128, 128 ** -0.5, 8, prefix=f"model.layers.{i}.self_attn",
                                        trianglemix_config=cfg, layer_idx=i, num_layers=36) for i in range(36)]
    # the helper filled the shared config in place, as the reference's does (trianglemix_attention.py:305-307)
    assert cfg.triangle_layer_indices == list(range(8))
    for i, layer in enumerate(layers):
        impl = layer.attn.impl
        assert isinstance(impl, TriangleMixAttentionImpl) and layer.attn.attn_backend is TriangleMixAttentionBackend
        assert layer.attn.use_output is True and layer.attn.prefix == f"model.layers.{i}.self_attn.attn"
        assert impl.use_triangle is (i < 8) and impl.layer_idx == i
        assert (impl.num_heads, impl.num_kv_heads, impl.head_size) == (32, 8, 128)
        assert impl.pattern == (4, 32, 64) and impl.tri_min_len == 2048
    # the fp8 cache dtype reaches the impl through cache_config (layer.py:118-127)
    fp8 = Qwen3TriangleMixAttention(FakeAttention, 8, 128, 0.1, 1, cache_config=types.SimpleNamespace(cache_dtype="fp8"),
                                    trianglemix_config=cfg, layer_idx=0)
    assert fp8.attn.impl.fp8_cache is True


def test_qwen3_call_site_without_config_keeps_the_default_backend():
    assert trianglemix_attention_kwargs(None, 3, 36) == {}
    layer = Qwen3TriangleMixAttention(FakeAttention, 32, 128, 0.1, 8, prefix="model.layers.0.self_attn")
    assert layer.attn.default_backend_used and layer.attn.impl[0] == "default-impl"
    # the positional tail the default impl received is exactly what ours accepts
    TriangleMixAttentionImpl(*layer.attn.impl[1], **layer.attn.impl[2])
    assert model_trianglemix_config(TriangleMixInferenceConfig()) is None
    assert model_trianglemix_config(TriangleMixInferenceConfig.from_env({"VLLM_TRIANGLEMIX_ENABLED": "1",
                                                                         "VLLM_TRIANGLEMIX_LAYERS": "5"})
                                    ).num_triangle_layers == 5


def test_extra_impl_args_unknown_keys_are_accepted():
    """Qwen3 already forwards other per-layer settings through **extra_impl_args (qwen3.py:150-153)."""
    impl = TriangleMixAttentionImpl(32, 128, 0.088, 8, None, None, "auto", None, "decoder", None,
                                    trianglemix_config=TriangleMixConfig(num_triangle_layers=1), layer_idx=0,
                                    dual_chunk_attention_config=None)
    assert impl.use_triangle
    with pytest.raises(TypeError):  # an eleventh positional argument is a caller bug, as for every in-tree impl
        TriangleMixAttentionImpl(32, 128, 0.088, 8, None, None, "auto", None, "decoder", None, "extra")


def test_plain_import_of_the_extension_registers_the_ops():
    """The reference loads its native ops by importing the extension module (vllm/platforms/rocm.py:32-40;
    PyInit_ via REGISTER_EXTENSION, csrc/core/registration.h:13-27).  In a fresh interpreter, importing
    the shared object alone must make torch.ops._trianglemix_C.* available."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import torch\n"
            "import vllm_trianglemix_amd._trianglemix_C as ext\n"
            "assert ext.__name__.endswith('_trianglemix_C')\n"
            "op = torch.ops._trianglemix_C.trianglemix_index_sizes\n"
            "cu = torch.tensor([0, 64], dtype=torch.int32); sl = torch.tensor([64], dtype=torch.int32)\n"
            "print(list(op(cu, sl, 4, 32, 64, 32, 32, 2, 0, 64)))\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip().startswith("[2, ")


def test_stand_in_bases_carry_the_concrete_members_of_the_abstract_classes():
    """abstract.py:27-83 and :114-182: members callers use without the backend defining them."""
    assert TriangleMixAttentionBackend.full_cls_name() == ("vllm_trianglemix_amd.backend", "TriangleMixAttentionBackend")
    assert TriangleMixAttentionBackend.supports_quant_query_input is False
    md = TriangleMixAttentionBackend.make_metadata(
        num_actual_tokens=1, max_query_len=1, query_start_loc=None, max_seq_len=1, seq_lens=None, block_table=None,
        slot_mapping=None, query_start_loc_cpu=np.array([0, 1], np.int32), seq_lens_cpu=np.array([1], np.int32), page_size=16)
    assert isinstance(md, TriangleMixAttentionBackend.get_metadata_cls())
    impl = TriangleMixAttentionImpl(32, 128, 0.1, 8, None, None, "auto", None, "decoder", None, use_triangle=True)
    assert impl.can_return_lse_for_decode is False and impl.need_to_return_lse_for_decode is False
    assert (impl.dcp_world_size, impl.dcp_rank) == (1, 0)
    assert impl.fused_output_quant_supported(object()) is False


def _common_metadata(**extra):
    return types.SimpleNamespace(
        num_actual_tokens=4096, max_query_len=4096, max_seq_len=4096,
        query_start_loc=torch.tensor([0, 4096], dtype=torch.int32), query_start_loc_cpu=torch.tensor([0, 4096], dtype=torch.int32),
        seq_lens=torch.tensor([4096], dtype=torch.int32), seq_lens_cpu=torch.tensor([4096], dtype=torch.int32),
        block_table_tensor=torch.zeros((1, 512), dtype=torch.int32), slot_mapping=torch.arange(4096, dtype=torch.int64), **extra)


def test_prompt_lengths_from_the_runner_are_optional():
    """CommonAttentionMetadata has no prompt-length field (utils.py:44-84); a runner that carries one
    (``prompt_lens_cpu``, filled where gpu_model_runner.py:1189-1205 builds the batch) gets the chunked-prefill
    semantics of a one-step prefill, one that does not gets per-chunk semantics."""
    from vllm_trianglemix_amd import ops
    builder = TriangleMixMetadataBuilder(types.SimpleNamespace(block_size=16), ["l0"], None, torch.device("cpu"))
    absent = builder.build(0, _common_metadata())
    assert absent.prompt_lens is None and absent.prompt_lens_cpu is None
    for carrier in (np.array([8192], np.int64), torch.tensor([8192]), [8192]):
        present = builder.build(0, _common_metadata(prompt_lens_cpu=carrier))
        assert present.prompt_lens_cpu.dtype == np.int32 and present.prompt_lens_cpu.tolist() == [8192]
        assert present.prompt_lens.dtype == torch.int32 and present.prompt_lens.tolist() == [8192]
    # what the field changes: the first 4096-token chunk of an 8192-token prompt has NO dense trailing rows (they
    # are the prompt's last 64, in the second chunk); without the field the chunk's own last 64 rows are dense
    cu, sl = np.array([0, 4096], np.int32), np.array([4096], np.int32)
    chunk = ops.index_sizes(cu, sl, 4, 32, 64, 32, 32, ops.MODES["triangle"], chunk_tiles=64)
    whole = ops.index_sizes(cu, sl, 4, 32, 64, 32, 32, ops.MODES["triangle"], chunk_tiles=64,
                            full_lens_cpu=present.prompt_lens_cpu)
    assert 0 < whole.nnz_tiles < chunk.nnz_tiles   # fewer attended tiles: no dense tail in this chunk


def test_plugin_entry_point_is_declared_and_resolves():
    """vllm/plugins/__init__.py:11-72 loads ``group='vllm.general_plugins'`` entry points and calls them."""
    import importlib
    import os
    import tomli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "pyproject.toml"), "rb") as f:
        eps = tomli.load(f)["project"]["entry-points"]["vllm.general_plugins"]
    (name, target), = eps.items()
    module, func = target.split(":")
    assert callable(getattr(importlib.import_module(module), func)) and name == "trianglemix_amd"


def test_plugin_wraps_the_attention_layer_constructor():
    from vllm_trianglemix_amd import plugin

    class Attention(FakeAttention):  # a class of its own: the wrapper replaces __init__ on it
        pass

    cfg = model_trianglemix_config(TriangleMixInferenceConfig.for_qwen3("8B"))
    original = plugin.install(Attention, cfg, num_layers=36)
    wrapped = Attention.__init__
    assert plugin.install(Attention, cfg) is wrapped and Attention.__init__ is wrapped   # a second call wraps nothing new
    try:
        a = Attention(32, 128, 0.1, 8, prefix="model.layers.3.self_attn.attn")
        assert isinstance(a.impl, TriangleMixAttentionImpl) and a.impl.layer_idx == 3 and a.impl.use_triangle
        b = Attention(32, 128, 0.1, num_kv_heads=8, prefix="model.layers.30.self_attn.attn")
        assert isinstance(b.impl, TriangleMixAttentionImpl) and not b.impl.use_triangle
        # layers outside the path keep the platform's backend
        for kw in (dict(prefix="vision.blocks.0.attn"), dict(prefix="model.layers.1.attn", use_mla=True),
                   dict(prefix="model.layers.1.attn", attn_type="encoder_only"),
                   dict(prefix="model.layers.1.attn", per_layer_sliding_window=4096),
                   dict(prefix="model.layers.1.attn", sinks=torch.zeros(32)),
                   dict(prefix="model.layers.1.attn", cache_config=types.SimpleNamespace(cache_dtype="auto", sliding_window=128))):
            assert Attention(32, 128, 0.1, 8, **kw).default_backend_used, kw
        assert plugin.layer_index_of("model.layers.17.self_attn.attn") == 17 and plugin.layer_index_of("lm_head") is None
    finally:
        Attention.__init__ = original


def test_plugin_register_in_a_process_with_and_without_the_switch():
    """``register`` in a fresh interpreter with a stand-in ``vllm.attention.layer``: the operators are registered
    either way; the constructor is wrapped only with VLLM_TRIANGLEMIX_ENABLED=1."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, types, torch\n"
            "class Attention:\n"
            "    def __init__(self, *a, attn_backend=None, **k): self.b = attn_backend; self.k = k\n"
            "for n in ('vllm', 'vllm.attention', 'vllm.attention.layer'): sys.modules[n] = types.ModuleType(n)\n"
            "sys.modules['vllm.attention.layer'].Attention = Attention\n"
            "from vllm_trianglemix_amd.plugin import register\n"
            "register(); register()\n"
            "assert hasattr(torch.ops._trianglemix_C, 'trianglemix_attention')\n"
            "a = Attention(32, 128, 0.1, 8, prefix='model.layers.0.self_attn.attn')\n"
            "print(getattr(a.b, '__name__', None), sorted(a.k))\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("VLLM_TRIANGLEMIX")}
    off = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert off.returncode == 0, off.stderr
    assert off.stdout.split()[0] == "None"
    on = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300,
                        env={**env, "VLLM_TRIANGLEMIX_ENABLED": "1", "VLLM_TRIANGLEMIX_LAYERS": "8", "VLLM_TRIANGLEMIX_NUM_LAYERS": "36"})
    assert on.returncode == 0, on.stderr
    assert on.stdout.startswith("TriangleMixAttentionBackend ['layer_idx', 'prefix', 'trianglemix_config']"), on.stdout
