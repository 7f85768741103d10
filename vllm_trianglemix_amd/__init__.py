"""MI355X-native TriangleMix hot path: packed block-sparse index builder and
masked prefill/decode attention (HIP, gfx950) behind the reference's
TriangleMix interface and vLLM's attention-backend interface.

Importing the package does not load the native library; the first kernel call
does, and raises if ``libtrianglemix_hip.so`` has not been built."""
from .config import (TriangleMixConfig, TriangleMixInferenceConfig,  # noqa: F401
                     create_trianglemix_config_from_hf)
from .mask_generator import (TriangleMixAnalyzer, TriangleMixAttention,  # noqa: F401
                             TriangleMixMaskGenerator)
from .eager_attention import AscendNPUTriangleMixOptimizer, AscendTriangleMixAttention  # noqa: F401
from .backend import (TriangleMixAttentionBackend, TriangleMixAttentionImpl,  # noqa: F401
                      TriangleMixAttentionMetadata, TriangleMixMetadataBuilder)

__all__ = [
    "TriangleMixConfig", "TriangleMixInferenceConfig", "create_trianglemix_config_from_hf",
    "TriangleMixMaskGenerator", "TriangleMixAttention", "TriangleMixAnalyzer",
    "AscendTriangleMixAttention", "AscendNPUTriangleMixOptimizer",
    "TriangleMixAttentionBackend", "TriangleMixAttentionImpl", "TriangleMixAttentionMetadata",
    "TriangleMixMetadataBuilder",
]
