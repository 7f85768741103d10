"""TriangleMix configuration objects (host logic only).

Same names, defaults and quirks as the reference:
``TriangleMixConfig``            vllm/attention/trianglemix_attention.py:11-36
``TriangleMixInferenceConfig``   vllm/attention/trianglemix_config.py:11-81
``create_trianglemix_config_from_hf``  vllm/attention/ascend_trianglemix.py:193-224
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

# Default pattern of the reference: 5 sink columns (j <= 4), a 33-wide window
# (i - j <= 32), 63 fully causal trailing rows (N - i < 64).
DEFAULT_SINK, DEFAULT_WINDOW, DEFAULT_LAST = 4, 32, 64

# TriangleMixAttention.get_attn_mask returns no mask up to this length
# (trianglemix_attention.py:330-332).
SHORT_SEQUENCE_GATE = 2048


class TriangleMixConfig:
    """The five pattern scalars.  ``triangle_layer_indices=None`` is stored as
    ``[]`` (trianglemix_attention.py:36)."""

    def __init__(self, num_sink_tokens: int = DEFAULT_SINK, sliding_window_size: int = DEFAULT_WINDOW,
                 num_last_tokens: int = DEFAULT_LAST, num_triangle_layers: Optional[int] = None,
                 triangle_layer_indices: Optional[List[int]] = None):
        self.num_sink_tokens = num_sink_tokens
        self.sliding_window_size = sliding_window_size
        self.num_last_tokens = num_last_tokens
        self.num_triangle_layers = num_triangle_layers
        self.triangle_layer_indices = triangle_layer_indices or []

    def pattern(self):
        return int(self.num_sink_tokens), int(self.sliding_window_size), int(self.num_last_tokens)

    def uses_triangle(self, layer_idx: int) -> bool:
        """Layer selection rule of trianglemix_attention.py:263-278."""
        if self.triangle_layer_indices:
            return layer_idx in self.triangle_layer_indices
        if self.num_triangle_layers is not None:
            return layer_idx < self.num_triangle_layers
        return False

    def __repr__(self) -> str:
        return (f"TriangleMixConfig(num_sink_tokens={self.num_sink_tokens}, "
                f"sliding_window_size={self.sliding_window_size}, num_last_tokens={self.num_last_tokens}, "
                f"num_triangle_layers={self.num_triangle_layers}, "
                f"triangle_layer_indices={self.triangle_layer_indices})")


@dataclass
class TriangleMixInferenceConfig:
    """Deployment-level switches with the reference's presets
    (trianglemix_config.py:11-81).  ``use_npu_optimization`` /
    ``npu_block_size`` are kept for drop-in compatibility; on MI355X the
    block sizes are fixed by the attention kernel."""
    enable_trianglemix: bool = False
    num_triangle_layers: Optional[int] = None
    triangle_layer_indices: List[int] = field(default_factory=list)
    num_sink_tokens: int = DEFAULT_SINK
    sliding_window_size: int = DEFAULT_WINDOW
    num_last_tokens: int = DEFAULT_LAST
    use_npu_optimization: bool = True
    npu_block_size: int = 64
    use_gradient_analysis: bool = False
    analysis_warmup_steps: int = 100

    def to_trianglemix_config(self) -> Optional[TriangleMixConfig]:
        if not self.enable_trianglemix:
            return None
        return TriangleMixConfig(self.num_sink_tokens, self.sliding_window_size, self.num_last_tokens,
                                 self.num_triangle_layers, self.triangle_layer_indices or None)

    @staticmethod
    def for_ascend_npu(num_triangle_layers: int = 12,
                       enable_gradient_analysis: bool = True) -> "TriangleMixInferenceConfig":
        return TriangleMixInferenceConfig(enable_trianglemix=True, num_triangle_layers=num_triangle_layers,
                                          use_npu_optimization=True,
                                          use_gradient_analysis=enable_gradient_analysis)

    @staticmethod
    def for_qwen3(model_size: str = "14B", num_triangle_layers: Optional[int] = None) -> "TriangleMixInferenceConfig":
        """14B -> 10 triangle layers, 32B -> 12, anything else -> 8
        (trianglemix_config.py:65-72)."""
        if num_triangle_layers is None:
            num_triangle_layers = {"14B": 10, "32B": 12}.get(model_size, 8)
        return TriangleMixInferenceConfig(enable_trianglemix=True, num_triangle_layers=num_triangle_layers,
                                          num_sink_tokens=DEFAULT_SINK, sliding_window_size=DEFAULT_WINDOW,
                                          num_last_tokens=DEFAULT_LAST, use_npu_optimization=True)

    # MI355X deployments use the same pattern defaults
    for_mi355x = for_qwen3

    @staticmethod
    def from_env(environ=None) -> "TriangleMixInferenceConfig":
        """The run-time switches the reference documents but never reads
        (trianglemix_config.py:106-110): ``VLLM_TRIANGLEMIX_ENABLED`` (0/1),
        ``VLLM_TRIANGLEMIX_LAYERS`` (number of triangle layers),
        ``VLLM_NPU_OPTIMIZATION`` (0/1, kept for compatibility).  Unset variables leave the
        dataclass defaults (TriangleMix off)."""
        import os
        env = os.environ if environ is None else environ

        def flag(name: str, default: bool) -> bool:
            v = env.get(name)
            return default if v is None else v.strip().lower() not in ("", "0", "false", "no", "off")

        layers = env.get("VLLM_TRIANGLEMIX_LAYERS")
        return TriangleMixInferenceConfig(enable_trianglemix=flag("VLLM_TRIANGLEMIX_ENABLED", False),
                                          num_triangle_layers=None if layers in (None, "") else int(layers),
                                          use_npu_optimization=flag("VLLM_NPU_OPTIMIZATION", True))


def create_trianglemix_config_from_hf(hf_config, use_trianglemix: bool = False,
                                      num_triangle_layers: Optional[int] = None) -> Optional[TriangleMixConfig]:
    """ascend_trianglemix.py:193-224: sink / window from the HF config when
    present, ``last = max(64, 2 * window)``.  A config whose
    ``sliding_window`` attribute is ``None`` fails here exactly as it does in
    the reference (``None * 2``)."""
    if not use_trianglemix:
        return None
    sink = getattr(hf_config, "num_sink_tokens", DEFAULT_SINK)
    window = getattr(hf_config, "sliding_window", DEFAULT_WINDOW)
    return TriangleMixConfig(num_sink_tokens=sink, sliding_window_size=window,
                             num_last_tokens=max(64, window * 2), num_triangle_layers=num_triangle_layers)
