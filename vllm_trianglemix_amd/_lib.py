"""ctypes binding of ``libtrianglemix_hip.so`` (declared in
``include/trianglemix_hip.h``).  There is no fallback: when the library is
missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TMX_LIB_PATH") or os.path.join(_HERE, "libtrianglemix_hip.so")

TMX_OK, TMX_ERR_INVALID_ARG, TMX_ERR_UNSUPPORTED, TMX_ERR_HIP, TMX_ERR_CAPACITY = range(5)
TMX_F32, TMX_BF16, TMX_F16, TMX_FP8_E4M3 = range(4)
TMX_MODE_NONE, TMX_MODE_CAUSAL, TMX_MODE_TRIANGLE = range(3)
TMX_MASK_STREAMING, TMX_MASK_LAST, TMX_MASK_MIDDLE, TMX_MASK_TRIANGLE, TMX_MASK_DENSE = range(5)
TMX_ABI_VERSION = 6

# every symbol include/trianglemix_hip.h declares
EXPORTED_SYMBOLS = (
    "tmx_abi_version", "tmx_last_error", "tmx_attention_block_m", "tmx_attention_block_n", "tmx_dense_mask",
    "tmx_replace_neg_inf", "tmx_index_sizes", "tmx_build_block_index", "tmx_attention", "tmx_reshape_and_cache",
    "tmx_reshape_and_cache_fp8", "tmx_gather_dequant_kv_fp8", "tmx_suggest_chunk_tiles",
    "tmx_attention_static_ws", "tmx_static_workspace_bytes", "tmx_static_workspace_warmup", "tmx_eager_attention_f32",
    "tmx_decode_attention", "tmx_decode_workspace_rows", "tmx_dense_stage_rows", "tmx_index_sizes_without_dense",
)


class TmxError(RuntimeError):
    """A call into libtrianglemix_hip.so returned an error code."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libtrianglemix_hip error {code}: {message}")
        self.code = code


class IndexSizes(Structure):
    _fields_ = [("num_qblocks", c_int64), ("nnz_tiles", c_int64), ("num_items", c_int64),
                ("num_partials", c_int64), ("num_split_qblocks", c_int64), ("max_tiles_per_qblock", c_int64)]


class Index(Structure):
    _fields_ = [("row_ptr", c_void_p), ("col_idx", c_void_p), ("qb_meta", c_void_p), ("part_ptr", c_void_p),
                ("items", c_void_p), ("split_qb", c_void_p), ("totals", c_void_p), ("scratch", c_void_p),
                ("cap_qblocks", c_int64), ("cap_nnz", c_int64), ("cap_items", c_int64), ("cap_split", c_int64),
                ("block_m", c_int32), ("block_n", c_int32), ("mode", c_int32), ("chunk_tiles", c_int32),
                ("si", c_int32), ("sl", c_int32), ("last", c_int32), ("batch", c_int32),
                ("tri_min_len", c_int32), ("pages_page_size", c_int32),
                ("tile_pages", c_void_p), ("pages_block_table", c_void_p), ("pages_block_table_stride", c_int64),
                ("full_lens", c_void_p), ("full_lens_host", c_void_p),
                ("dense_rows_from", c_int32), ("reserved", c_int32)]


class AttentionArgs(Structure):
    _fields_ = [
        ("struct_size", c_uint32), ("io_dtype", c_int32), ("out_dtype", c_int32), ("mode", c_int32),
        ("si", c_int32), ("sl", c_int32), ("last", c_int32), ("scale", c_float),
        ("batch", c_int32), ("num_heads", c_int32), ("num_kv_heads", c_int32), ("head_dim", c_int32),
        ("total_q", c_int32),
        ("out", c_void_p), ("out_stride_t", c_int64), ("out_stride_h", c_int64),
        ("q", c_void_p), ("q_stride_t", c_int64), ("q_stride_h", c_int64),
        ("k", c_void_p), ("v", c_void_p),
        ("kv_stride_page", c_int64), ("kv_stride_t", c_int64), ("kv_stride_h", c_int64),
        ("block_table", c_void_p), ("block_table_stride", c_int64), ("page_size", c_int32),
        ("cu_seqlens_k", c_void_p), ("kv_dtype", c_int32), ("k_scale", c_float), ("v_scale", c_float),
        ("k_new", c_void_p), ("v_new", c_void_p), ("new_stride_t", c_int64), ("new_stride_h", c_int64),
        ("cache_slots", c_void_p),
        ("cu_seqlens_q", c_void_p), ("seq_lens", c_void_p),
        ("bias", c_void_p), ("bias_dtype", c_int32), ("bias_stride", c_int64),
        ("bias_rows", c_int32), ("bias_cols", c_int32), ("bias_stride_b", c_int64), ("bias_stride_h", c_int64),
        ("index", POINTER(Index)),
        ("num_items", c_int64), ("num_partials", c_int64), ("num_split_qblocks", c_int64),
        ("nnz_tiles", c_int64),
        ("partial_o", c_void_p), ("partial_ml", c_void_p),
        ("flags", c_uint32), ("reserved", c_uint32),
        ("kv_stage", c_void_p), ("kv_stage_rows", c_int64),
    ]

class DecodeArgs(Structure):
    _fields_ = [
        ("struct_size", c_uint32), ("io_dtype", c_int32), ("out_dtype", c_int32), ("scale", c_float),
        ("batch", c_int32), ("num_heads", c_int32), ("num_kv_heads", c_int32), ("head_dim", c_int32),
        ("out", c_void_p), ("out_stride_t", c_int64), ("out_stride_h", c_int64),
        ("q", c_void_p), ("q_stride_t", c_int64), ("q_stride_h", c_int64),
        ("k", c_void_p), ("v", c_void_p),
        ("kv_stride_page", c_int64), ("kv_stride_t", c_int64), ("kv_stride_h", c_int64),
        ("block_table", c_void_p), ("block_table_stride", c_int64), ("page_size", c_int32),
        ("kv_dtype", c_int32), ("k_scale", c_float), ("v_scale", c_float),
        ("k_new", c_void_p), ("v_new", c_void_p), ("new_stride_t", c_int64), ("new_stride_h", c_int64),
        ("cache_slots", c_void_p), ("seq_lens", c_void_p),
        ("partial_o", c_void_p), ("partial_ml", c_void_p), ("workspace_rows", c_int64),
        ("flags", c_uint32), ("reserved", c_uint32),
    ]


TMX_ATTN_SKIP_COMBINE, TMX_ATTN_ONLY_COMBINE, TMX_ATTN_SHORT_QUERIES, TMX_ATTN_NO_BAND, TMX_ATTN_NO_DENSE = 1, 2, 4, 8, 16
TMX_ATTN_NO_DECODE, TMX_ATTN_DECODE_ONLY, TMX_ATTN_DECODE_VALU = 32, 64, 128
TMX_DENSE_MIN_ROWS = 17  # query rows from which a request of a causal launch goes to the dense causal kernel
TMX_ATTN_BUILD_INDEX = 256  # the index is still to be built: this call builds it (inside the attention launch where it can)


_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP library (once).  Raises ``ImportError`` with the build
    command when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    lib.tmx_abi_version.restype = c_int
    lib.tmx_last_error.restype = c_char_p
    lib.tmx_attention_block_m.argtypes = [c_int, c_int]
    lib.tmx_attention_block_n.argtypes = []
    lib.tmx_dense_mask.argtypes = [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.tmx_replace_neg_inf.argtypes = [c_void_p, c_int64, c_int, c_float, c_void_p]
    lib.tmx_index_sizes.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    POINTER(IndexSizes)]
    lib.tmx_index_sizes_without_dense.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                  c_int, c_int, c_int, POINTER(IndexSizes)]
    lib.tmx_suggest_chunk_tiles.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                            c_int, c_int, c_int, c_int, POINTER(c_int)]
    lib.tmx_build_block_index.argtypes = [POINTER(Index), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.tmx_attention.argtypes = [POINTER(AttentionArgs), c_void_p]
    lib.tmx_attention_static_ws.argtypes = [POINTER(AttentionArgs), c_void_p]
    lib.tmx_eager_attention_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                            c_float, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int64,
                                            c_int64, c_int64, c_void_p]
    lib.tmx_static_workspace_bytes.argtypes = []
    lib.tmx_static_workspace_bytes.restype = ctypes.c_size_t
    lib.tmx_static_workspace_warmup.argtypes = []
    lib.tmx_decode_attention.argtypes = [POINTER(DecodeArgs), c_void_p]
    lib.tmx_decode_workspace_rows.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.tmx_decode_workspace_rows.restype = c_int64
    lib.tmx_dense_stage_rows.argtypes = [c_void_p, c_void_p, c_int, c_int]
    lib.tmx_dense_stage_rows.restype = c_int64
    lib.tmx_reshape_and_cache.argtypes = [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64,
                                          c_int64, c_int64, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                          c_void_p]
    lib.tmx_reshape_and_cache_fp8.argtypes = [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64,
                                              c_int64, c_int64, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                              c_float, c_float, c_void_p]
    lib.tmx_gather_dequant_kv_fp8.argtypes = [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int32,
                                              c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                              c_int32, c_float, c_float, c_void_p]
    for name in EXPORTED_SYMBOLS:
        getattr(lib, name)  # AttributeError if a declared symbol is not exported
    if lib.tmx_abi_version() != TMX_ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.tmx_abi_version()} != {TMX_ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != TMX_OK:
        raise TmxError(code, load().tmx_last_error().decode("utf-8", "replace"))
