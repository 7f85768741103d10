"""Standalone attention module with the interface of the reference's
``AscendTriangleMixAttention`` (vllm/attention/ascend_trianglemix.py:16-108),
running on the MI355X kernels instead of dense matmul / softmax / matmul.

Behaviour kept from the reference:
* inputs ``[batch, seq_len, num_heads, head_dim]`` for query, key AND value
  (no grouped heads in this class: a key with fewer heads is an error);
* with a TriangleMix config and ``num_layers`` the per-layer helper decides:
  up to 2048 tokens there is NO mask at all -- attention is bidirectional
  (ascend_trianglemix.py:76-97) -- above that the layer's triangle or dense
  causal mask applies;
* an explicit ``attn_mask`` is ADDED to the scores, and to the TriangleMix mask
  when both exist (ascend_trianglemix.py:83-84, 96-97);
* ``layer_idx or 0`` (ascend_trianglemix.py:43);
* the output is a new contiguous ``[batch, seq_len, num_heads, head_dim]`` tensor.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
from torch import nn

from . import ops
from .config import TriangleMixConfig
from .mask_generator import TriangleMixAttention


class AscendTriangleMixAttention(nn.Module):

    def __init__(self, num_heads: int, head_dim: int, scale: float,
                 trianglemix_config: Optional[TriangleMixConfig] = None, layer_idx: Optional[int] = None,
                 num_layers: Optional[int] = None):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.scale = scale
        self.layer_idx = layer_idx or 0
        self.trianglemix_attn: Optional[TriangleMixAttention] = None
        if trianglemix_config is not None and num_layers is not None:
            self.trianglemix_attn = TriangleMixAttention(config=trianglemix_config, num_layers=num_layers)
        self._index_cache = {}

    def _index(self, mode: str, batch: int, seq_len: int, device, pattern):
        key = (mode, batch, seq_len, str(device), pattern)
        hit = self._index_cache.get(key)
        if hit is None:
            cu_h = np.arange(batch + 1, dtype=np.int32) * seq_len
            sl_h = np.full((batch,), seq_len, dtype=np.int32)
            cu = torch.from_numpy(cu_h).to(device)
            sl = torch.from_numpy(sl_h).to(device)
            bm = ops.attention_block_m(self.num_heads, self.num_heads)
            idx = ops.build_block_index(cu, sl, cu_h, sl_h, pattern[0], pattern[1], pattern[2], bm,
                                        ops.attention_block_n(), ops.MODES[mode])
            hit = (idx, cu, sl)
            self._index_cache[key] = hit
        return hit

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                attn_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        if query.dim() != 4:
            raise RuntimeError("query must be [batch, seq_len, num_heads, head_dim]")
        if key.shape != query.shape or value.shape != query.shape:
            # the reference's matmul raises on a head-count mismatch as well
            raise RuntimeError(f"key/value shape {tuple(key.shape)}/{tuple(value.shape)} must equal the query shape "
                               f"{tuple(query.shape)} (this class has no grouped heads)")
        if not query.is_cuda:
            raise RuntimeError("AscendTriangleMixAttention runs on HIP devices only (no CPU fallback)")
        if query.dtype not in (torch.bfloat16, torch.float16, torch.float32):
            raise NotImplementedError(f"inputs must be float32, bfloat16 or float16, got {query.dtype}")
        if key.dtype != query.dtype or value.dtype != query.dtype:
            raise RuntimeError("query, key and value must share a dtype")
        batch, seq_len, heads, dim = query.shape

        mode, pattern = "none", (0, 0, 0)
        if self.trianglemix_attn is not None:
            kind = self.trianglemix_attn.mask_kind(seq_len, self.layer_idx)
            if kind == "triangle":
                mode, pattern = "triangle", self.trianglemix_attn.config.pattern()
            elif kind == "dense":
                mode = "causal"
        if query.dtype == torch.float32:
            # fp32 inputs, as the reference's own CPU runs: everything in fp32 (no index, one wave per row)
            return ops.eager_attention_f32(query, key, value, self.scale, ops.MODES[mode], *pattern, bias=attn_mask)
        bias = None
        if attn_mask is not None:
            # anything that broadcasts to [batch, heads, seq_len, seq_len] (ascend_trianglemix.py:83-84, 96-97)
            bias = attn_mask
            if bias.dim() > 4 or bias.dim() < 2:
                raise RuntimeError("attn_mask must broadcast to [batch, heads, seq_len, seq_len]")
            while bias.dim() < 4:
                bias = bias[None]
            if (bias.shape[0] not in (1, batch) or bias.shape[1] not in (1, heads) or bias.shape[2] not in (1, seq_len)
                    or bias.shape[3] not in (1, seq_len)):
                raise RuntimeError(f"attn_mask shape {tuple(attn_mask.shape)} does not broadcast to "
                                   f"[{batch}, {heads}, {seq_len}, {seq_len}]")
            if bias.dtype not in (torch.float32, torch.bfloat16, torch.float16):
                bias = bias.float()
            if bias.shape[2] != seq_len or bias.shape[3] != seq_len or bias.stride(3) != 1:
                bias = bias.expand(bias.shape[0], bias.shape[1], seq_len, seq_len).contiguous()

        idx, cu, sl = self._index(mode, batch, seq_len, query.device, pattern)
        q = query.reshape(batch * seq_len, heads, dim)
        k = key.reshape(batch * seq_len, heads, dim)
        v = value.reshape(batch * seq_len, heads, dim)
        if q.stride(2) != 1:
            q = q.contiguous()
        if k.stride(2) != 1 or k.stride() != v.stride():
            k, v = k.contiguous(), v.contiguous()
        out = torch.empty((batch * seq_len, heads, dim), dtype=query.dtype, device=query.device)
        ops.attention(q, k, v, out, idx, cu, sl, self.scale, cu_seqlens_k=cu, bias=bias)
        return out.view(batch, seq_len, heads, dim)


class AscendNPUTriangleMixOptimizer:
    """``optimize_mask_for_npu`` of the reference (ascend_trianglemix.py:114-141):
    -inf entries become -1e6, ``block_size`` is accepted and unused.  The
    reference's ``sparse_attention_kernel`` raises a shape error for every
    input with seq_len != num_heads and is not mirrored."""

    @staticmethod
    def optimize_mask_for_npu(mask: torch.Tensor, block_size: int = 64) -> torch.Tensor:
        if mask.numel() == 0:
            return mask
        out = mask.clone(memory_format=torch.contiguous_format)
        return ops.replace_neg_inf_(out, -1e6)
