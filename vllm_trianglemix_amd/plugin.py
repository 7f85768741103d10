"""vLLM general plugin: makes the TriangleMix backend available in every vLLM process without editing the tree.

vLLM calls the functions registered under the ``vllm.general_plugins`` entry-point group once per process --
engine, workers, API server -- before any model is built (vllm/plugins/__init__.py:11-72).  ``register``:

1. loads the operator library, so that ``torch.ops._trianglemix_C.*`` exist in that process (the reference's own ops
   are loaded by the platform module the same way, vllm/platforms/rocm.py:32-40);
2. when ``VLLM_TRIANGLEMIX_ENABLED=1`` (the switch vllm/attention/trianglemix_config.py:106-110 documents), wraps
   ``vllm.attention.layer.Attention.__init__`` so that decoder self-attention layers are built with
   ``attn_backend=TriangleMixAttentionBackend`` and the shared TriangleMix settings (layer.py:102-103, 186-201) --
   the three keyword arguments the reference's ``Qwen3Attention`` never passes on (qwen3.py:143-154).

The function may run several times per process and in several processes; it does its work once per process.
"""
from __future__ import annotations

import functools
import os
import re
from typing import Optional

_LAYER_IN_PREFIX = re.compile(r"(?:^|\.)layers\.(\d+)\.")
_installed = False


def layer_index_of(prefix: str) -> Optional[int]:
    """The decoder layer a module prefix belongs to (``model.layers.17.self_attn.attn`` -> 17), or None."""
    m = _LAYER_IN_PREFIX.search(prefix or "")
    return int(m.group(1)) if m else None


def _wants_default_backend(kwargs) -> bool:
    """Layers the TriangleMix path does not cover keep the platform's backend: MLA, sparse-index and non-decoder
    attention, layers with attention sinks, and layers whose model already chose a backend."""
    if kwargs.get("attn_backend") is not None or kwargs.get("use_mla") or kwargs.get("use_sparse"):
        return True
    if kwargs.get("sinks") is not None or kwargs.get("alibi_slopes") is not None:
        return True
    if kwargs.get("per_layer_sliding_window") is not None or kwargs.get("logits_soft_cap") is not None:
        return True
    attn_type = kwargs.get("attn_type", "decoder")
    return str(getattr(attn_type, "value", attn_type)) != "decoder"


def install(attention_cls, trianglemix_config, num_layers: Optional[int] = None):
    """Wrap ``attention_cls.__init__`` (vLLM's ``Attention``, layer.py:86-104): decoder layers whose prefix names a layer
    index are constructed with the TriangleMix backend and settings.  Returns the original ``__init__`` (for undoing)."""
    from .qwen3_hook import trianglemix_attention_kwargs

    original = attention_cls.__init__
    if getattr(original, "_trianglemix_wrapped", False):
        return original

    # positional parameters of Attention.__init__ after self (layer.py:86-102)
    names = ("num_heads", "head_size", "scale", "num_kv_heads", "alibi_slopes", "cache_config", "quant_config",
             "logits_soft_cap", "per_layer_sliding_window", "use_mla", "use_sparse", "prefix", "attn_type",
             "kv_sharing_target_layer_name", "attn_backend")

    @functools.wraps(original)
    def init(self, *args, **kwargs):
        seen = dict(zip(names, args))
        seen.update(kwargs)
        layer = layer_index_of(seen.get("prefix", ""))
        cache_config = seen.get("cache_config")
        sliding = getattr(cache_config, "sliding_window", None) if cache_config is not None else None
        if layer is None or sliding is not None or _wants_default_backend(seen):
            return original(self, *args, **kwargs)
        extra = trianglemix_attention_kwargs(trianglemix_config, layer, num_layers)
        return original(self, *args, **{**kwargs, **extra})

    init._trianglemix_wrapped = True
    attention_cls.__init__ = init
    return original


def register() -> None:
    """Entry point of the ``vllm.general_plugins`` group (see the module docstring)."""
    global _installed
    if _installed:
        return
    _installed = True
    from . import torch_ops
    torch_ops.load()   # raises with the build command if the operator library is not built
    from .qwen3_hook import model_trianglemix_config
    cfg = model_trianglemix_config()   # None unless VLLM_TRIANGLEMIX_ENABLED=1
    if cfg is None:
        return
    try:
        from vllm.attention.layer import Attention
    except Exception:  # not a vLLM process: the operators are registered, nothing to wrap
        return
    num_layers = os.environ.get("VLLM_TRIANGLEMIX_NUM_LAYERS")
    install(Attention, cfg, int(num_layers) if num_layers else None)
