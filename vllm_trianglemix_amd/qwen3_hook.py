"""Model-side wiring: what ``Qwen3Attention`` has to hand to vLLM's ``Attention`` layer so that the
TriangleMix backend runs behind it.

The reference's constructor takes ``trianglemix_config / layer_idx / num_layers``
(vllm/model_executor/models/qwen3.py:79-81), builds the per-model helper (:105-114) and then creates
its ``Attention`` layer WITHOUT passing any of it on (:143-154); ``forward`` computes a mask and drops
it (:176-189).  The live form of that call site is three keyword arguments:

    self.attn = Attention(self.num_heads, self.head_dim, self.scaling, num_kv_heads=self.num_kv_heads,
                          cache_config=cache_config, quant_config=quant_config, prefix=f"{prefix}.attn",
                          attn_type=attn_type,
                          **trianglemix_attention_kwargs(trianglemix_config, self.layer_idx, num_layers))

``attn_backend=`` selects the backend class (vllm/attention/layer.py:102, 186-195); everything else
travels through ``**extra_impl_args`` into ``TriangleMixAttentionImpl.__init__`` (layer.py:103,
198-201).  Nothing here touches a device.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

from .backend import TriangleMixAttentionBackend
from .config import TriangleMixConfig, TriangleMixInferenceConfig
from .mask_generator import TriangleMixAttention


def trianglemix_attention_kwargs(trianglemix_config: Optional[TriangleMixConfig], layer_idx: Optional[int],
                                 num_layers: Optional[int] = None) -> Dict[str, Any]:
    """Keyword arguments for ``Attention(...)``; empty when TriangleMix is off, so the layer keeps the
    platform's default backend (vllm/attention/layer.py:186-193), exactly as the reference does with
    ``trianglemix_config=None``.

    With ``num_layers`` given the per-model helper is constructed first, as qwen3.py:108-114 does: it
    fills ``triangle_layer_indices`` of the SHARED config object from ``num_triangle_layers``
    (trianglemix_attention.py:305-307), which later layers and callers then see."""
    if trianglemix_config is None:
        return {}
    if num_layers is not None:
        TriangleMixAttention(config=trianglemix_config, num_layers=num_layers)
    return {"attn_backend": TriangleMixAttentionBackend, "trianglemix_config": trianglemix_config,
            "layer_idx": layer_idx or 0}


class Qwen3TriangleMixAttention:
    """The ``self.attn`` member of ``Qwen3Attention`` with TriangleMix wired in: builds the attention
    layer the way qwen3.py:143-154 does, plus the three TriangleMix arguments.  ``attention_cls`` is
    ``vllm.attention.Attention`` in a vLLM process (it is a parameter so that this module imports
    without vLLM)."""

    def __init__(self, attention_cls, num_heads: int, head_dim: int, scaling: float, num_kv_heads: int,
                 cache_config=None, quant_config=None, prefix: str = "", attn_type: str = "decoder",
                 trianglemix_config: Optional[TriangleMixConfig] = None, layer_idx: Optional[int] = None,
                 num_layers: Optional[int] = None, **attention_kwargs) -> None:
        self.trianglemix_config = trianglemix_config
        self.layer_idx = layer_idx
        kw = trianglemix_attention_kwargs(trianglemix_config, layer_idx, num_layers)
        self.attn = attention_cls(num_heads, head_dim, scaling, num_kv_heads=num_kv_heads, cache_config=cache_config,
                                  quant_config=quant_config, prefix=f"{prefix}.attn", attn_type=attn_type,
                                  **kw, **attention_kwargs)

    def __call__(self, q, k, v):
        return self.attn(q, k, v)


def model_trianglemix_config(inference_config: Optional[TriangleMixInferenceConfig] = None
                             ) -> Optional[TriangleMixConfig]:
    """What ``Qwen2Model`` would pass to every decoder layer (vllm/model_executor/models/qwen2.py:319-324
    passes nothing today): ONE shared ``TriangleMixConfig`` from the deployment switches, or ``None``.
    Without an argument the documented environment variables decide
    (vllm/attention/trianglemix_config.py:106-110)."""
    cfg = TriangleMixInferenceConfig.from_env() if inference_config is None else inference_config
    return cfg.to_trianglemix_config()
