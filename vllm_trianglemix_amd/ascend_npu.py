"""Compatibility mirror of the reference's block-level variant
(vllm/attention/ascend_npu_trianglemix.py:30-375): same class and function
names, thresholds and value conventions, on the MI355X kernels.

The block-level mask is NOT a cover of the token mask: the pattern parameters
are divided by the block size with a floor of 1 (``si = max(1, sink // block)``
...), so with the defaults and 64-token blocks it keeps 2 sink block columns, a
2-block band and no dense trailing block rows (ascend_npu_trianglemix.py:101-124).
On block indices that is the triangle predicate itself, so the block mask comes
from the same mask kernel at ``ceil(N / block)`` resolution.  Additive masks of
this variant use -1e9 instead of -inf (ascend_npu_trianglemix.py:243-247, 266-270).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops
from .config import TriangleMixConfig
from .mask_generator import TriangleMixMaskGenerator

BLOCK_SPARSE_MIN_LEN = 2048   # triangle layers switch to the block-level mask above this (:197)
DENSE_BLOCK_MIN_LEN = 4096    # dense layers switch to a 64-token block mask above this (:206)
NO_MASK_MAX_LEN = 512         # the helper returns no mask up to this (:325-326)
MASKED_VALUE = -1e9


class AscendNPUTriangleMixConfig(TriangleMixConfig):
    """TriangleMixConfig + the three block-level switches (ascend_npu_trianglemix.py:30-62)."""

    def __init__(self, num_sink_tokens: int = 4, sliding_window_size: int = 32, num_last_tokens: int = 64,
                 num_triangle_layers: Optional[int] = None, triangle_layer_indices: Optional[List[int]] = None,
                 npu_block_size: int = 64, use_block_sparse: bool = True, use_acl_ops: bool = False):
        super().__init__(num_sink_tokens, sliding_window_size, num_last_tokens, num_triangle_layers,
                         triangle_layer_indices)
        self.npu_block_size = npu_block_size
        self.use_block_sparse = use_block_sparse
        self.use_acl_ops = use_acl_ops


class AscendBlockSparseMaskGenerator(TriangleMixMaskGenerator):

    def __init__(self, config: AscendNPUTriangleMixConfig):
        super().__init__(config)
        self.npu_block_size = config.npu_block_size
        self.use_block_sparse = config.use_block_sparse
        self._ascend_cache = {}

    def block_pattern(self):
        """(si, sl, last) in blocks: each divided by the block size, floor 1 (:101-104)."""
        b = self.npu_block_size
        return (max(1, self.config.num_sink_tokens // b), max(1, self.config.sliding_window_size // b),
                max(1, self.config.num_last_tokens // b))

    def _get_block_sparse_mask(self, seq_len: int, device, dtype: torch.dtype) -> torch.Tensor:
        """``[num_blocks, num_blocks]`` 0/1 mask (ascend_npu_trianglemix.py:80-124)."""
        nb = (seq_len + self.npu_block_size - 1) // self.npu_block_size
        si, sl, last = self.block_pattern()
        return ops.dense_mask(nb, si, sl, last, "triangle", device, dtype, additive=False)

    def _expand_block_mask_to_token_level(self, block_mask: torch.Tensor, seq_len: int, device,
                                          dtype: torch.dtype) -> torch.Tensor:
        """Every kept block becomes a block of ones, clipped to ``seq_len`` (:126-167)."""
        b = self.npu_block_size
        keep = (block_mask > 0).repeat_interleave(b, dim=0).repeat_interleave(b, dim=1)
        return keep[:seq_len, :seq_len].to(device=device, dtype=dtype).contiguous()

    def get_attention_mask(self, seq_len: int, layer_idx: int, device, dtype: torch.dtype,
                           use_triangle: bool = False) -> Optional[torch.Tensor]:
        """ascend_npu_trianglemix.py:169-222: block-level additive masks for long sequences, token-level
        ones otherwise, 0 where attended and -1e9 elsewhere."""
        key = (seq_len, str(torch.device(device)), str(dtype), bool(use_triangle))
        hit = self._ascend_cache.get(key)
        if hit is not None:
            return hit
        if use_triangle:
            if self.use_block_sparse and seq_len > BLOCK_SPARSE_MIN_LEN:
                nb = (seq_len + self.npu_block_size - 1) // self.npu_block_size
                si, sl, last = self.block_pattern()
                mask = ops.dense_mask(nb, si, sl, last, "triangle", device, dtype, additive=True)
            else:
                si, sl, last = self.config.pattern()
                mask = ops.dense_mask(seq_len, si, sl, last, "triangle", device, dtype, additive=True)
        else:
            n = (seq_len + 63) // 64 if seq_len > DENSE_BLOCK_MIN_LEN else seq_len
            mask = ops.dense_mask(n, 0, 0, 0, "dense", device, dtype, additive=True)
        mask = ops.replace_neg_inf_(mask, MASKED_VALUE)
        self._ascend_cache[key] = mask
        return mask

    @staticmethod
    def _mask_to_attn_mask(mask: torch.Tensor, device, dtype: torch.dtype) -> torch.Tensor:
        """0/1 -> 0/-1e9 (:224-247)."""
        zero = torch.zeros((), device=mask.device, dtype=dtype)
        return torch.where(mask > 0, zero, torch.full((), MASKED_VALUE, device=mask.device, dtype=dtype))

    _convert_block_mask_for_cann = _mask_to_attn_mask  # same values on this platform (:249-279)

    def clear_cache(self) -> None:
        super().clear_cache()
        self._ascend_cache.clear()


class AscendNPUTriangleMixAttention:
    """Per-model helper of the block-level variant (ascend_npu_trianglemix.py:282-345)."""

    def __init__(self, config: AscendNPUTriangleMixConfig, num_layers: int):
        self.config = config
        self.num_layers = num_layers
        self.mask_generator = AscendBlockSparseMaskGenerator(config)
        if config.num_triangle_layers is not None and not config.triangle_layer_indices:
            self.config.triangle_layer_indices = list(range(config.num_triangle_layers))

    def get_attn_mask(self, seq_len: int, layer_idx: int, device, dtype: torch.dtype = torch.float32
                      ) -> Optional[torch.Tensor]:
        use_triangle = self.mask_generator.should_use_triangle(layer_idx)
        if seq_len <= NO_MASK_MAX_LEN:
            return None
        return self.mask_generator.get_attention_mask(seq_len=seq_len, layer_idx=layer_idx, device=device,
                                                      dtype=dtype, use_triangle=use_triangle)

    def clear_cache(self) -> None:
        self.mask_generator.clear_cache()


def create_ascend_trianglemix_config(model_size: str = "14B", num_triangle_layers: Optional[int] = None,
                                     use_block_sparse: bool = True) -> AscendNPUTriangleMixConfig:
    """14B -> 12 triangle layers, 32B -> 16, anything else -> 8 (ascend_npu_trianglemix.py:347-375)."""
    if num_triangle_layers is None:
        num_triangle_layers = {"14B": 12, "32B": 16}.get(model_size, 8)
    return AscendNPUTriangleMixConfig(num_sink_tokens=4, sliding_window_size=32, num_last_tokens=64,
                                      num_triangle_layers=num_triangle_layers, npu_block_size=64,
                                      use_block_sparse=use_block_sparse, use_acl_ops=False)
