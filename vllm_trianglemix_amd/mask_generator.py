"""Mask generator, per-layer helper and layer analyzer with the reference's
interface (vllm/attention/trianglemix_attention.py:39-409), computed by the
HIP library: the dense ``[N, N]`` masks the reference builds with Python
double loops come from one mask kernel launch, and the attention kernels do
not need them at all -- they take the packed block-sparse index
(:meth:`TriangleMixMaskGenerator.get_block_index`)."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import ops
from .config import SHORT_SEQUENCE_GATE, TriangleMixConfig


class TriangleMixMaskGenerator(nn.Module):
    """Same methods as the reference generator; ``device`` must be a HIP
    device (no CPU path)."""

    def __init__(self, config: TriangleMixConfig):
        super().__init__()
        self.config = config
        self._mask_cache: Dict[tuple, torch.Tensor] = {}
        self._index_cache: Dict[tuple, ops.BlockIndex] = {}

    # -- dense 0/1 masks (trianglemix_attention.py:53-219) -----------------
    def _dense(self, kind: str, seq_len: int, device, dtype: torch.dtype) -> torch.Tensor:
        si, sl, last = self.config.pattern()
        return ops.dense_mask(seq_len, si, sl, last, kind, device, dtype, additive=False)

    def _create_streaming_mask(self, seq_len: int, device, dtype: torch.dtype) -> torch.Tensor:
        return self._dense("streaming", seq_len, device, dtype)

    def _create_last_mask(self, seq_len: int, device, dtype: torch.dtype) -> torch.Tensor:
        return self._dense("last", seq_len, device, dtype)

    def _create_middle_mask(self, seq_len: int, device, dtype: torch.dtype) -> torch.Tensor:
        return self._dense("middle", seq_len, device, dtype)

    def _create_triangle_mask(self, seq_len: int, device, dtype: torch.dtype) -> torch.Tensor:
        return self._dense("triangle", seq_len, device, dtype)

    def _create_dense_mask(self, seq_len: int, device, dtype: torch.dtype) -> torch.Tensor:
        return self._dense("dense", seq_len, device, dtype)

    # -- additive mask (trianglemix_attention.py:221-261) ------------------
    def get_attention_mask(self, seq_len: int, layer_idx: int, device, dtype: torch.dtype,
                           use_triangle: bool = False) -> torch.Tensor:
        """Additive ``[N, N]`` mask: 0 where attended, -inf elsewhere.

        The reference caches one copy per layer (its key holds ``layer_idx``);
        the values do not depend on the layer, so one tensor is shared by all
        layers here (2 GiB per copy at N = 32768 in bf16)."""
        key = (seq_len, str(torch.device(device)), str(dtype), bool(use_triangle))
        hit = self._mask_cache.get(key)
        if hit is not None:
            return hit
        si, sl, last = self.config.pattern()
        mask = ops.dense_mask(seq_len, si, sl, last, "triangle" if use_triangle else "dense", device, dtype,
                              additive=True)
        self._mask_cache[key] = mask
        return mask

    def should_use_triangle(self, layer_idx: int) -> bool:
        return self.config.uses_triangle(layer_idx)

    def clear_cache(self) -> None:
        self._mask_cache.clear()
        self._index_cache.clear()

    # -- what the attention kernel consumes --------------------------------
    def get_block_index(self, seq_len: int, device, num_heads: int, num_kv_heads: int, mode: str = "triangle",
                        batch: int = 1) -> Tuple[ops.BlockIndex, torch.Tensor, torch.Tensor]:
        """Packed block-sparse index for ``batch`` whole prompts of ``seq_len``
        tokens.  Returns ``(index, cu_seqlens, seq_lens)`` (device int32)."""
        device = torch.device(device)
        bm = ops.attention_block_m(num_heads, num_kv_heads)
        key = (seq_len, str(device), bm, mode, batch)
        hit = self._index_cache.get(key)
        if hit is not None:
            return hit
        cu_h = np.arange(batch + 1, dtype=np.int32) * seq_len
        sl_h = np.full((batch,), seq_len, dtype=np.int32)
        cu = torch.from_numpy(cu_h).to(device)
        sl = torch.from_numpy(sl_h).to(device)
        si, sl_w, last = self.config.pattern()
        idx = ops.build_block_index(cu, sl, cu_h, sl_h, si, sl_w, last, bm, ops.attention_block_n(),
                                    ops.MODES[mode])
        self._index_cache[key] = (idx, cu, sl)
        return self._index_cache[key]


class TriangleMixAttention:
    """Per-model helper (trianglemix_attention.py:285-340)."""

    def __init__(self, config: TriangleMixConfig, num_layers: int):
        self.config = config
        self.num_layers = num_layers
        self.mask_generator = TriangleMixMaskGenerator(config)
        # as the reference: the shared config object is filled in place
        if config.num_triangle_layers is not None and not config.triangle_layer_indices:
            self.config.triangle_layer_indices = list(range(config.num_triangle_layers))

    def mask_kind(self, seq_len: int, layer_idx: int) -> str:
        """``"none"`` up to 2048 tokens (the caller then applies no mask at
        all), else ``"triangle"`` or ``"dense"`` by layer."""
        if seq_len <= SHORT_SEQUENCE_GATE:
            return "none"
        return "triangle" if self.mask_generator.should_use_triangle(layer_idx) else "dense"

    def get_attn_mask(self, seq_len: int, layer_idx: int, device, dtype: torch.dtype = torch.float32
                      ) -> Optional[torch.Tensor]:
        kind = self.mask_kind(seq_len, layer_idx)
        if kind == "none":
            return None
        return self.mask_generator.get_attention_mask(seq_len=seq_len, layer_idx=layer_idx, device=device,
                                                      dtype=dtype, use_triangle=(kind == "triangle"))


class TriangleMixAnalyzer(nn.Module):
    """Offline layer selection (trianglemix_attention.py:344-409): mean of the
    recorded Middle Q-K gradients per layer, lowest ``k`` layers win."""

    def __init__(self, num_layers: int):
        super().__init__()
        self.num_layers = num_layers
        self.middle_qk_gradients: List[List[float]] = [[] for _ in range(num_layers)]

    def record_middle_qk_gradient(self, layer_idx: int, gradient: torch.Tensor) -> None:
        if layer_idx < self.num_layers:
            self.middle_qk_gradients[layer_idx].append(gradient.detach().cpu().mean().item())

    def get_triangle_layers(self, num_triangle_layers: int) -> List[int]:
        means = []
        for layer_idx, g in enumerate(self.middle_qk_gradients):
            means.append((layer_idx, sum(g) / len(g) if g else float("inf")))
        ranked = sorted(means, key=lambda item: item[1])  # stable: ties keep layer order
        return sorted(idx for idx, _ in ranked[:num_triangle_layers])

    def reset(self) -> None:
        self.middle_qk_gradients = [[] for _ in range(self.num_layers)]
