"""vLLM v1 attention-backend surface for the TriangleMix path on MI355X.

Mirrors the three classes a vLLM backend provides -- the reference's own ROCm
backend is the model (vllm/v1/attention/backends/rocm_attn.py:31-426, abstract
interfaces vllm/attention/backends/abstract.py:27-182, builder interface
vllm/v1/attention/backends/utils.py:232-341):

* ``TriangleMixAttentionBackend``  -- names, KV-cache shape, supported dtypes / head sizes;
* ``TriangleMixMetadataBuilder``   -- turns the per-step ``CommonAttentionMetadata`` into the
  per-step metadata; the block-sparse index is built once per step and shared by all layers;
* ``TriangleMixAttentionImpl``     -- ``forward(layer, query, key, value, kv_cache,
  attn_metadata, output)``: KV-cache write, then masked attention, in place into ``output``.

When vLLM is importable the classes derive from its abstract bases so that
``Attention(..., attn_backend=TriangleMixAttentionBackend)`` (vllm/attention/layer.py:102,
186-201) accepts them; otherwise small stand-ins with the same members are used.
Everything here is host logic; the compute is the HIP library (no CPU fallback).
"""
from __future__ import annotations

import enum
import os
from dataclasses import dataclass, field
from typing import ClassVar, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops
from .config import SHORT_SEQUENCE_GATE, TriangleMixConfig

try:  # pragma: no cover - depends on the host installation
    from vllm.attention.backends.abstract import AttentionBackend as _BackendBase
    from vllm.attention.backends.abstract import AttentionImpl as _ImplBase
    from vllm.attention.backends.abstract import AttentionType as _AttentionType
    _DECODER = _AttentionType.DECODER
except Exception:  # vLLM (or one of its dependencies) is not installed
    class _BackendBase:
        """Stand-in with the concrete members of ``AttentionBackend`` (vllm/attention/backends/abstract.py:27-83);
        the abstract ones are defined by the subclass below."""
        accept_output_buffer: bool = False
        supports_quant_query_input: bool = False

        @classmethod
        def make_metadata(cls, *args, **kwargs):
            return cls.get_metadata_cls()(*args, **kwargs)

        @staticmethod
        def get_kv_cache_stride_order() -> Tuple[int, ...]:
            raise NotImplementedError

        @classmethod
        def full_cls_name(cls) -> Tuple[str, str]:
            return (cls.__module__, cls.__qualname__)

    class _ImplBase:
        """Stand-in with the concrete members of ``AttentionImpl`` (abstract.py:114-182): no decode context
        parallelism outside a vLLM process, so the group size is 1."""
        can_return_lse_for_decode: bool = False
        need_to_return_lse_for_decode: bool = False
        dcp_world_size: int = 1
        dcp_rank: int = 0

        def fused_output_quant_supported(self, quant_key) -> bool:
            return False

    _DECODER = "decoder"  # vllm/attention/backends/abstract.py:17-24

try:  # pragma: no cover - depends on the host installation
    from vllm.v1.attention.backends.utils import AttentionCGSupport
    from vllm.v1.attention.backends.utils import AttentionMetadataBuilder as _BuilderBase
except Exception:
    class AttentionCGSupport(enum.Enum):
        """Stand-in with the member names and values of vllm/v1/attention/backends/utils.py:215-229: the
        model runner compares ``builder.cudagraph_support.value`` across builders
        (vllm/v1/worker/gpu_model_runner.py:3612-3619)."""
        ALWAYS = 3
        UNIFORM_BATCH = 2
        UNIFORM_SINGLE_TOKEN_DECODE = 1
        NEVER = 0

    class _BuilderBase:  # the members vllm/v1/attention/backends/utils.py:232-341 gives every builder
        cudagraph_support: ClassVar["AttentionCGSupport"] = AttentionCGSupport.NEVER
        reorder_batch_threshold: Optional[int] = None

        def reorder_batch(self, input_batch, scheduler_output) -> bool:
            raise NotImplementedError

        def build_for_cudagraph_capture(self, common_attn_metadata):
            return self.build(common_prefix_len=0, common_attn_metadata=common_attn_metadata)

        def build_for_drafting(self, common_attn_metadata, draft_index: int):
            return self.build(common_prefix_len=0, common_attn_metadata=common_attn_metadata, fast_build=True)

        def use_cascade_attention(self, *args, **kwargs) -> bool:
            return False


@dataclass
class TriangleMixAttentionMetadata:
    """Per-step metadata (the fields rocm_attn.py:31-57 carries, plus the host copies the
    index sizing needs and the per-step index cache)."""
    num_actual_tokens: int
    max_query_len: int
    query_start_loc: torch.Tensor      # int32 [B + 1], device
    max_seq_len: int
    seq_lens: torch.Tensor             # int32 [B], device
    block_table: torch.Tensor          # int32 [B, max_blocks], device
    slot_mapping: torch.Tensor         # int64 [T], device
    query_start_loc_cpu: np.ndarray    # int32 [B + 1]
    seq_lens_cpu: np.ndarray           # int32 [B]
    page_size: int
    use_cascade: bool = False
    common_prefix_len: int = 0
    # Chunked prefill: the length every request's sequence has once its prompt is complete (the runner
    # knows it: gpu_model_runner.py:1189-1205 builds the batch from the requests' prompt lengths).  With it
    # the trailing dense rows and the length gate refer to the whole prompt, as in a one-step prefill.
    prompt_lens: Optional[torch.Tensor] = None       # int32 [B], device
    prompt_lens_cpu: Optional[np.ndarray] = None     # int32 [B]
    # A step of one-row requests (uniform single-token decode): the layers may take the launch whose shape depends on
    # the batch size only and that reads seq_lens / block table / slot mapping on the device.
    uniform_decode: bool = False
    capturing: bool = False
    builder: Optional["TriangleMixMetadataBuilder"] = field(default=None, repr=False)
    _indices: Dict[Tuple, "ops.BlockIndex"] = field(default_factory=dict, repr=False)
    _workspaces: Dict[Tuple, tuple] = field(default_factory=dict, repr=False)

    def block_index(self, mode: str, pattern: Tuple[int, int, int], num_heads: int, num_kv_heads: int,
                    tri_min_len: int, fp8_cache: bool = False, dense_kernel: bool = False) -> "ops.BlockIndex":
        """The step's index for (mode, pattern, head grouping): built on first use, then
        shared by every layer that asks for the same one (the mask does not depend on the layer
        beyond the triangle/dense choice).  ``dense_kernel``: the layer's causal attention call can run the dense
        causal kernel; the causal index then leaves the requests of 17 query rows or more out (no tile lists, work
        items or partial-result slots for prompts that kernel computes without an index)."""
        block_m = ops.attention_block_m(num_heads, num_kv_heads)
        # a step of one-row requests out of an fp8 cache is cut for the decode kernel's workgroups (8 query heads
        # each when the group size allows): half the bytes per key, so twice the pieces keep the chip busy
        units = ops.units_per_item(num_heads, num_kv_heads)
        if fp8_cache and self.max_query_len == 1:
            units = ops.decode_units_per_item(num_heads, num_kv_heads)
        leave_dense = bool(dense_kernel and mode == "causal" and len(self.seq_lens_cpu) <= 512)
        key = (mode, pattern, block_m, tri_min_len, units, leave_dense)
        idx = self._indices.get(key)
        if idx is None:
            # piece length for this batch and head grouping: one piece of the dense rows per CU when the step
            # also holds sparse rows, 128-tile pieces for pure decode steps (tmx_suggest_chunk_tiles); an index
            # without the dense kernel's requests is cut for the requests that remain
            cu_s, sl_s = ((self.query_start_loc_cpu, self.seq_lens_cpu) if not leave_dense else
                          ops.without_dense_requests(self.query_start_loc_cpu, self.seq_lens_cpu))
            chunk = ops.suggest_chunk_tiles(cu_s, sl_s, pattern[0], pattern[1],
                                            pattern[2], block_m, ops.attention_block_n(), ops.MODES[mode],
                                            units, tri_min_len=tri_min_len,
                                            full_lens_cpu=None if leave_dense else self.prompt_lens_cpu)
            idx = ops.build_block_index(self.query_start_loc, self.seq_lens, self.query_start_loc_cpu,
                                        self.seq_lens_cpu, pattern[0], pattern[1], pattern[2], block_m,
                                        ops.attention_block_n(), ops.MODES[mode], chunk_tiles=chunk,
                                        tri_min_len=tri_min_len,
                                        block_table=self.block_table, page_size=self.page_size,
                                        full_lens=self.prompt_lens, full_lens_cpu=self.prompt_lens_cpu,
                                        leave_dense=leave_dense)
            self._indices[key] = idx
        return idx

    @property
    def whole_prompts(self) -> bool:
        """True when no request of the step has earlier context in the cache (every request brings all of its
        keys with it: the one-step prefill of one or more prompts)."""
        w = self.__dict__.get("_whole_prompts")
        if w is None:
            w = bool(np.array_equal(np.diff(self.query_start_loc_cpu), self.seq_lens_cpu))
            self.__dict__["_whole_prompts"] = w
        return w

    def fp8_row_offsets(self, device):
        """Row offsets of every request's keys in a contiguous copy of the step's K/V (``cu_seqlens_k``), on the
        device, and their total: made once per step, shared by the layers."""
        hit = self.__dict__.get("_fp8_rows")
        if hit is None:
            cu_k_cpu = np.concatenate([[0], np.cumsum(self.seq_lens_cpu, dtype=np.int64)]).astype(np.int32)
            hit = (torch.from_numpy(cu_k_cpu).to(device, non_blocking=True), int(cu_k_cpu[-1]))
            self.__dict__["_fp8_rows"] = hit
        return hit

    def workspace(self, idx: "ops.BlockIndex", num_heads: int, head_size: int, device,
                  num_kv_heads: Optional[int] = None) -> tuple:
        """Work areas of the step's attention calls with `idx`, made once and shared by the layers (they run one
        after the other): the partial results of split q-blocks and, for the causal index of a step with long
        prompts, the staging area of the dense causal kernel."""
        key = (id(idx), num_heads, head_size, num_kv_heads)
        ws = self._workspaces.get(key)
        if ws is None:
            ws = ops.alloc_workspace(idx, num_heads, head_size, device, num_kv_heads=num_kv_heads)
            self._workspaces[key] = ws
        return ws


# How the step's K/V rows reach the 16-bit paged cache (TRIANGLEMIX_CACHE_WRITE):
#   "fused"  (default) the attention launch stores them itself (cache_slots) and reads the step's keys from the
#            rows, not from the cache: one launch per layer, no second stream;
#   "stream" the write kernel runs beside the attention on a second stream (joined before forward returns);
#   "serial" write kernel, then attention over the cache, on the caller's stream (what the reference does,
#            rocm_attn.py:345-354).
_CACHE_WRITE_MODE = os.environ.get("TRIANGLEMIX_CACHE_WRITE", "fused")
_CACHE_WRITE_STREAM = os.environ.get("TRIANGLEMIX_CACHE_WRITE_STREAM", "1") != "0"  # "stream" mode's switch
# Steps of one-row requests take the launch with a fixed shape (no index, nothing sized on the host from the
# lengths; the only form that replays under a graph).  "0": they go through the block index like every other step.
_DECODE_FIXED = os.environ.get("TRIANGLEMIX_DECODE_FIXED", "1") != "0"
_side_streams: Dict[int, "torch.cuda.Stream"] = {}


def _side_stream(device: torch.device) -> "torch.cuda.Stream":
    i = device.index if device.index is not None else torch.cuda.current_device()
    s = _side_streams.get(i)
    if s is None:
        s = _side_streams[i] = torch.cuda.Stream(device=i)
    return s


def _step_flags(m: "TriangleMixAttentionMetadata") -> int:
    """Hints for the attention kernel: a step whose requests all have at most 16 query rows (decode), and a
    step whose requests all have exactly one (then the decode kernel runs alone)."""
    from ._lib import TMX_ATTN_DECODE_ONLY, TMX_ATTN_SHORT_QUERIES
    f = m.__dict__.get("_flags")
    if f is None:
        f = TMX_ATTN_SHORT_QUERIES if m.max_query_len <= 16 else 0
        if m.max_query_len == 1 and bool(np.all(np.diff(m.query_start_loc_cpu) == 1)):
            f |= TMX_ATTN_DECODE_ONLY
        m.__dict__["_flags"] = f
    return f


class TriangleMixMetadataBuilder(_BuilderBase):
    """``AttentionMetadataBuilder`` subclass (utils.py:232-341; rocm_attn.py:60-135)."""
    # Steps of one-row requests replay under a graph: their launch (tmx_decode_attention) depends on the batch size
    # only, and reads seq_lens / block table / slot mapping from the runner's persistent buffers.  Steps with prompts need the block index, whose size follows the batch: built every step
    # on the host side, so they stay outside graphs (utils.py:215-229; read as ``.value`` at
    # gpu_model_runner.py:3612-3619; the reference's own ROCm backend declares ALWAYS, rocm_attn.py:63).
    cudagraph_support: ClassVar[AttentionCGSupport] = AttentionCGSupport.UNIFORM_SINGLE_TOKEN_DECODE
    reorder_batch_threshold: Optional[int] = None

    def __init__(self, kv_cache_spec, layer_names: List[str], vllm_config, device: torch.device):
        self.kv_cache_spec = kv_cache_spec
        self.layer_names = layer_names
        self.vllm_config = vllm_config
        self.device = device
        self.block_size = int(kv_cache_spec.block_size)
        self._decode_ws: Dict[Tuple, tuple] = {}

    def decode_workspace(self, batch: int, num_heads: int, num_kv_heads: int, head_size: int, fp8: bool) -> tuple:
        """Work area of the fixed-shape decode launch for a batch size, held for the builder's life: a captured
        launch replays with these pointers (a few MB per batch size: (512 / K/V head chunks + batch) x heads rows)."""
        key = (batch, num_heads, num_kv_heads, head_size, fp8)
        ws = self._decode_ws.get(key)
        if ws is None:
            ws = self._decode_ws[key] = ops.decode_workspace(batch, num_heads, num_kv_heads, head_size, self.device, fp8)
        return ws

    def build(self, common_prefix_len: int, common_attn_metadata, fast_build: bool = False
              ) -> TriangleMixAttentionMetadata:
        m = common_attn_metadata
        if common_prefix_len > 0:
            raise NotImplementedError("cascade attention (a shared prefix pass) is not part of the TriangleMix path")

        def host_i32(t):
            return np.ascontiguousarray(t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t), dtype=np.int32)

        # optional extension of CommonAttentionMetadata (utils.py:44-84 has no such field): the prompt
        # lengths of the scheduled requests, on the host
        prompt_cpu = getattr(m, "prompt_lens_cpu", None)
        prompt_dev = None
        if prompt_cpu is not None:
            prompt_cpu = host_i32(prompt_cpu)
            prompt_dev = torch.from_numpy(prompt_cpu).to(m.seq_lens.device, non_blocking=True)
        md = TriangleMixAttentionMetadata(
            num_actual_tokens=int(m.num_actual_tokens), max_query_len=int(m.max_query_len),
            query_start_loc=m.query_start_loc, max_seq_len=int(m.max_seq_len), seq_lens=m.seq_lens,
            block_table=m.block_table_tensor, slot_mapping=m.slot_mapping,
            query_start_loc_cpu=host_i32(m.query_start_loc_cpu), seq_lens_cpu=host_i32(m.seq_lens_cpu),
            page_size=self.block_size, prompt_lens=prompt_dev, prompt_lens_cpu=prompt_cpu, builder=self)
        # uniform single-token decode: one query row per request, row b = request b
        if (_DECODE_FIXED and md.max_query_len == 1 and md.num_actual_tokens == md.seq_lens_cpu.shape[0] > 0
                and md.query_start_loc_cpu.shape[0] == md.num_actual_tokens + 1
                and int(md.query_start_loc_cpu[-1]) == md.num_actual_tokens):
            md.uniform_decode = True
        return md

    def build_for_cudagraph_capture(self, common_attn_metadata) -> TriangleMixAttentionMetadata:
        """Metadata of the step a graph is captured from (utils.py:296-303; the runner calls it with a uniform
        decode batch, gpu_model_runner.py:3612-3619 having checked ``cudagraph_support``).  The layers' launches then
        depend on the batch size alone, so replays serve any later lengths."""
        md = self.build(common_prefix_len=0, common_attn_metadata=common_attn_metadata)
        if not md.uniform_decode:
            raise NotImplementedError("TriangleMix steps replay under a graph only as uniform single-token decode "
                                      "(AttentionCGSupport.UNIFORM_SINGLE_TOKEN_DECODE)")
        md.capturing = True
        return md

    def use_cascade_attention(self, *args, **kwargs) -> bool:
        return False


class TriangleMixAttentionBackend(_BackendBase):
    accept_output_buffer: bool = True  # abstract.py:32, rocm_attn.py:139

    @classmethod
    def get_supported_dtypes(cls) -> List[torch.dtype]:
        return [torch.float16, torch.bfloat16]

    @classmethod
    def get_supported_head_sizes(cls) -> List[int]:
        return [64, 128]

    @classmethod
    def validate_head_size(cls, head_size: int) -> None:
        supported = cls.get_supported_head_sizes()
        if head_size not in supported:
            raise ValueError(f"Head size {head_size} is not supported by TriangleMixAttention. "
                             f"Supported head sizes are: {supported}.")

    @staticmethod
    def get_name() -> str:
        return "TRIANGLEMIX_ROCM"

    @staticmethod
    def get_impl_cls():
        return TriangleMixAttentionImpl

    @staticmethod
    def get_metadata_cls():
        return TriangleMixAttentionMetadata

    @staticmethod
    def get_builder_cls():
        return TriangleMixMetadataBuilder

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int, head_size: int,
                           cache_dtype_str: str = "auto") -> Tuple[int, ...]:
        # same layout as rocm_attn.py:173-181; pages must be a power of two of at least 16 tokens
        if block_size < 16 or block_size & (block_size - 1):
            raise ValueError("Block size must be a power of two of at least 16.")
        return (2, num_blocks, block_size, num_kv_heads, head_size)

    @staticmethod
    def use_cascade_attention(*args, **kwargs) -> bool:
        return False


class TriangleMixAttentionImpl(_ImplBase):
    """Positional arguments as vllm/attention/layer.py:198-201 passes them; the TriangleMix
    settings arrive through ``**extra_impl_args`` (layer.py:103): either a ``trianglemix_config``
    + ``layer_idx`` (the layer decides with ``should_use_triangle``,
    trianglemix_attention.py:263-278) or an explicit ``use_triangle`` flag + pattern."""

    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: Optional[int] = None,
                 alibi_slopes: Optional[List[float]] = None, sliding_window: Optional[int] = None,
                 kv_cache_dtype: str = "auto", logits_soft_cap: Optional[float] = None,
                 attn_type=_DECODER, kv_sharing_target_layer_name: Optional[str] = None, *,
                 trianglemix_config: Optional[TriangleMixConfig] = None, layer_idx: Optional[int] = None,
                 use_triangle: Optional[bool] = None, num_sink_tokens: Optional[int] = None,
                 sliding_window_size: Optional[int] = None, num_last_tokens: Optional[int] = None,
                 short_sequence_gate: int = SHORT_SEQUENCE_GATE, **unused) -> None:
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = float(scale)
        self.num_kv_heads = num_heads if num_kv_heads is None else num_kv_heads
        self.kv_cache_dtype = kv_cache_dtype
        self.kv_sharing_target_layer_name = kv_sharing_target_layer_name
        TriangleMixAttentionBackend.validate_head_size(head_size)
        if self.num_heads % self.num_kv_heads != 0:
            raise ValueError(f"num_heads {num_heads} must be a multiple of num_kv_heads {self.num_kv_heads}")
        if str(getattr(attn_type, "value", attn_type)) != str(getattr(_DECODER, "value", _DECODER)):
            raise NotImplementedError("Encoder self-attention and encoder/decoder cross-attention are not "
                                      "implemented for TriangleMixAttentionImpl")
        if alibi_slopes is not None:
            raise NotImplementedError("ALiBi slopes are not supported by TriangleMixAttentionImpl")
        if sliding_window is not None:
            raise NotImplementedError("a model-level sliding window is not supported by TriangleMixAttentionImpl "
                                      "(the TriangleMix pattern carries its own window)")
        if logits_soft_cap:
            raise NotImplementedError("logits soft cap is not supported by TriangleMixAttentionImpl")
        if kv_cache_dtype not in ("auto", "bfloat16", "float16", "half", "fp8", "fp8_e4m3"):
            raise NotImplementedError(f"kv_cache_dtype '{kv_cache_dtype}' is not supported "
                                      "(16-bit caches and fp8 e4m3 are)")
        self.fp8_cache = kv_cache_dtype.startswith("fp8")

        cfg = trianglemix_config
        if cfg is None:
            cfg = TriangleMixConfig()
            if num_sink_tokens is not None:
                cfg.num_sink_tokens = num_sink_tokens
            if sliding_window_size is not None:
                cfg.sliding_window_size = sliding_window_size
            if num_last_tokens is not None:
                cfg.num_last_tokens = num_last_tokens
        self.pattern = cfg.pattern()
        self.layer_idx = layer_idx or 0
        if use_triangle is None:
            use_triangle = trianglemix_config is not None and trianglemix_config.uses_triangle(self.layer_idx)
        self.use_triangle = bool(use_triangle)
        # requests of at most this many tokens keep plain causal attention: the reference applies
        # its masks only above the gate (``seq_len <= 2048`` gives no mask, a 2049-token request gets
        # the triangle: trianglemix_attention.py:330-332); the library's rule is ``n <= tri_min_len``
        self.tri_min_len = int(short_sequence_gate)

    def forward(self, layer, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, kv_cache: torch.Tensor,
                attn_metadata: Optional[TriangleMixAttentionMetadata], output: Optional[torch.Tensor] = None,
                output_scale: Optional[torch.Tensor] = None, output_block_scale: Optional[torch.Tensor] = None
                ) -> torch.Tensor:
        """query ``[T, H, D]``, key/value ``[T, Hkv, D]``, kv_cache ``[2, num_blocks, block_size, Hkv, D]``;
        the result is written in place into ``output[:num_actual_tokens]`` (rocm_attn.py:275-426)."""
        assert output is not None, "Output tensor must be provided."
        if output_scale is not None or output_block_scale is not None:
            raise NotImplementedError("fused output quantization is not supported for TriangleMixAttentionImpl")
        if attn_metadata is None:  # profiling run
            return output
        m = attn_metadata
        n_tok = m.num_actual_tokens
        key_cache, value_cache = kv_cache.unbind(0)
        q = query.view(-1, self.num_heads, self.head_size)[:n_tok]
        if self.fp8_cache:
            return self._forward_fp8_cache(layer, q, key, value, key_cache, value_cache, m, output)
        writes_cache = self.kv_sharing_target_layer_name is None and key is not None and value is not None
        out = output.view(-1, self.num_heads, self.head_size)[:n_tok]
        if m.uniform_decode and self._decode_fixed_ok():
            return self._forward_decode_fixed(q, key, value, key_cache, value_cache, m, out, output, writes_cache)
        if m.capturing:
            raise NotImplementedError("this layer's shape has no decode launch that replays under a graph (head size "
                                      "128, query-head groups that are multiples of 4, num_last_tokens >= 2)")
        mode = "triangle" if self.use_triangle else "causal"
        pattern = self.pattern if self.use_triangle else (0, 0, 0)
        tri_min_len = self.tri_min_len if self.use_triangle else 0
        idx = m.block_index(mode, pattern, self.num_heads, self.num_kv_heads, tri_min_len,
                            dense_kernel=self._dense_kernel_ok())
        ws = m.workspace(idx, self.num_heads, self.head_size, q.device, self.num_kv_heads)
        k_rows = v_rows = None
        if writes_cache and n_tok > 0:
            k_rows = key[:n_tok].view(n_tok, self.num_kv_heads, self.head_size)
            v_rows = value[:n_tok].view(n_tok, self.num_kv_heads, self.head_size)
        if k_rows is not None and _CACHE_WRITE_MODE == "fused":
            # One launch: every workgroup first stores its share of the step's rows to their cache slots, and the
            # attention reads the step's keys from the rows themselves (earlier context from the paged cache).
            ops.attention(q, key_cache, value_cache, out, idx, m.query_start_loc, m.seq_lens, self.scale,
                          block_table=m.block_table, workspace=ws, flags=_step_flags(m), k_new=k_rows, v_new=v_rows,
                          cache_slots=m.slot_mapping[:n_tok])
            return output
        if k_rows is not None and _CACHE_WRITE_MODE == "stream" and _CACHE_WRITE_STREAM:
            # The write kernel beside the attention on a second stream, joined before anything else follows on
            # this one.  Whole prompts attend to the rows directly (request b's key j is row query_start_loc[b] + j);
            # steps with context read the earlier keys from the paged cache and the step's from the rows.
            main = torch.cuda.current_stream(q.device)
            side = _side_stream(q.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ops.reshape_and_cache(key[:n_tok], value[:n_tok], key_cache, value_cache, m.slot_mapping[:n_tok])
            if m.whole_prompts:
                ops.attention(q, k_rows, v_rows, out, idx, m.query_start_loc, m.seq_lens, self.scale,
                              cu_seqlens_k=m.query_start_loc, workspace=ws, flags=_step_flags(m))
            else:
                ops.attention(q, key_cache, value_cache, out, idx, m.query_start_loc, m.seq_lens, self.scale,
                              block_table=m.block_table, workspace=ws, flags=_step_flags(m), k_new=k_rows, v_new=v_rows)
            main.wait_stream(side)
            return output
        if k_rows is not None:
            ops.reshape_and_cache(key[:n_tok], value[:n_tok], key_cache, value_cache, m.slot_mapping[:n_tok])
        ops.attention(q, key_cache, value_cache, out, idx, m.query_start_loc, m.seq_lens, self.scale,
                      block_table=m.block_table, workspace=ws, flags=_step_flags(m))
        return output

    def _dense_kernel_ok(self) -> bool:
        """A causal attention call of this layer can run the dense causal kernel (head size 128, query-head groups
        that are multiples of 4; the backend passes no bias): its index then leaves the long requests out."""
        return self.head_size == 128 and (self.num_heads // self.num_kv_heads) % 4 == 0

    def _decode_fixed_ok(self) -> bool:
        """The fixed-shape decode launch covers head size 128 and query-head groups that are multiples of 4; a decode
        row attends to all of its keys in a causal layer, and in a triangle layer when it lies among the trailing
        dense rows, i.e. with ``num_last_tokens >= 2`` (N - i = 1 < last, trianglemix_attention.py:126-129)."""
        return (self.head_size == 128 and (self.num_heads // self.num_kv_heads) % 4 == 0
                and (not self.use_triangle or self.pattern[2] >= 2))

    def _forward_decode_fixed(self, q, key, value, key_cache, value_cache, m, out, output, writes_cache,
                              k_scale: float = 1.0, v_scale: float = 1.0):
        """A step of one-row requests through ``tmx_decode_attention``: no index, nothing derived on the host from the
        requests' lengths -- the call replays under a graph with whatever ``seq_lens`` / ``block_table`` /
        ``slot_mapping`` hold by then.  The cache gets the step's rows from the same launch (an fp8 cache: quantised
        on the way, and the row's own key attended to as the cache holds it)."""
        n_tok = m.num_actual_tokens
        fp8 = self.fp8_cache
        if m.builder is not None:
            ws = m.builder.decode_workspace(n_tok, self.num_heads, self.num_kv_heads, self.head_size, fp8)
        else:
            ws = ops.decode_workspace(n_tok, self.num_heads, self.num_kv_heads, self.head_size, q.device, fp8)
        k_rows = v_rows = slots = None
        if writes_cache:
            k_rows = key[:n_tok].view(n_tok, self.num_kv_heads, self.head_size)
            v_rows = value[:n_tok].view(n_tok, self.num_kv_heads, self.head_size)
            slots = m.slot_mapping[:n_tok]
        ops.decode_attention(q, key_cache, value_cache, out, m.block_table, m.seq_lens, self.scale, ws,
                             k_scale=k_scale, v_scale=v_scale, k_new=k_rows, v_new=v_rows, cache_slots=slots)
        return output

    def _forward_fp8_cache(self, layer, q, key, value, key_cache, value_cache, m, output):
        """kv_cache_dtype "fp8" (rocm_attn.py:356-358, 421-422): the step's rows are written as
        fp8(x / scale); the attention then reads what the cache holds, times the scale.  The query
        stays 16-bit, as on ROCm in the reference (rocm_attn.py:362-365).  With head size 128 the attention
        kernel descales inside its tile loop; other head sizes go through a descale pass into 16-bit rows."""
        n_tok = m.num_actual_tokens
        k_scale = float(getattr(layer, "_k_scale_float", 1.0))
        v_scale = float(getattr(layer, "_v_scale_float", 1.0))
        writes_cache = self.kv_sharing_target_layer_name is None and key is not None and value is not None
        out = output.view(-1, self.num_heads, self.head_size)[:n_tok]
        if m.uniform_decode and self._decode_fixed_ok() and _CACHE_WRITE_MODE == "fused":
            # decode steps: ONE launch quantises and stores the step's rows and attends
            return self._forward_decode_fixed(q, key, value, key_cache, value_cache, m, out, output, writes_cache,
                                              k_scale=k_scale, v_scale=v_scale)
        if writes_cache:
            ops.reshape_and_cache_fp8(key[:n_tok].view(n_tok, self.num_kv_heads, self.head_size),
                                      value[:n_tok].view(n_tok, self.num_kv_heads, self.head_size), key_cache, value_cache,
                                      m.slot_mapping[:n_tok], k_scale, v_scale)
        if m.uniform_decode and self._decode_fixed_ok():
            return self._forward_decode_fixed(q, key, value, key_cache, value_cache, m, out, output, False,
                                              k_scale=k_scale, v_scale=v_scale)
        if m.capturing:
            raise NotImplementedError("this layer's shape has no decode launch that replays under a graph")
        mode = "triangle" if self.use_triangle else "causal"
        pattern = self.pattern if self.use_triangle else (0, 0, 0)
        tri_min_len = self.tri_min_len if self.use_triangle else 0
        idx = m.block_index(mode, pattern, self.num_heads, self.num_kv_heads, tri_min_len, fp8_cache=self.head_size == 128,
                            dense_kernel=self._dense_kernel_ok())
        ws = m.workspace(idx, self.num_heads, self.head_size, q.device, self.num_kv_heads)
        if self.head_size == 128:
            # the attention kernel reads the fp8 pages in place and descales inside its tile loop
            ops.attention(q, key_cache, value_cache, out, idx, m.query_start_loc, m.seq_lens, self.scale,
                          block_table=m.block_table, workspace=ws, flags=_step_flags(m), k_scale=k_scale,
                          v_scale=v_scale)
            return output
        # other head sizes: descale ahead of the kernel into contiguous 16-bit rows kept for the step
        cu_k, total = m.fp8_row_offsets(q.device)
        k_rows, v_rows = ops.gather_dequant_kv_fp8(key_cache, value_cache, m.block_table, m.seq_lens, cu_k,
                                                   total, int(m.max_seq_len), q.dtype, k_scale, v_scale)
        ops.attention(q, k_rows, v_rows, out, idx, m.query_start_loc, m.seq_lens, self.scale, cu_seqlens_k=cu_k,
                      workspace=ws, flags=_step_flags(m))
        return output
