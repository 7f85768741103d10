"""Tensor-level wrappers over the C ABI (the counterpart of the reference's
``vllm/_custom_ops.py:153-196``: outputs are allocated here, the native code
only fills them).  Every tensor must live on a HIP device; torch is used for
device memory and the current stream only."""
from __future__ import annotations

import ctypes
import math
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import (AttentionArgs, Index, IndexSizes, TMX_BF16, TMX_F16, TMX_F32, TMX_MODE_CAUSAL, TMX_MODE_NONE,
                   TMX_MODE_TRIANGLE, check)

DEFAULT_CHUNK_TILES = 64  # key tiles per work item before a q-block is split (64 x 32 = 2048 keys)

_DTYPE = {torch.float32: TMX_F32, torch.bfloat16: TMX_BF16, torch.float16: TMX_F16}
FP8_DTYPE = torch.float8_e4m3fn  # the K/V cache format vllm/platforms/rocm.py:402-406 selects on gfx950
MODES = {"none": TMX_MODE_NONE, "causal": TMX_MODE_CAUSAL, "triangle": TMX_MODE_TRIANGLE}
MASK_KINDS = {"streaming": 0, "last": 1, "middle": 2, "triangle": 3, "dense": 4}


def _require_gpu(t: torch.Tensor, name: str) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a tensor on a HIP device (the TriangleMix path has no CPU fallback)")


def _device_of(device) -> torch.device:
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"TriangleMix kernels run on HIP devices only, got device '{device}' "
                           "(there is no CPU fallback)")
    return device


def _stream(device: torch.device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _np_ptr(a: np.ndarray) -> ctypes.c_void_p:
    """Address of a (contiguous) numpy array; ``a.ctypes`` builds a helper object per access, ten times as slow."""
    return ctypes.c_void_p(a.__array_interface__["data"][0])


def _ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def dtype_code(dtype: torch.dtype) -> int:
    if dtype not in _DTYPE:
        raise RuntimeError(f"unsupported dtype {dtype} (float32, bfloat16, float16)")
    return _DTYPE[dtype]


# --------------------------------------------------------------------------- masks
def dense_mask(n: int, si: int, sl: int, last: int, kind: str, device, dtype: torch.dtype,
               additive: bool = False) -> torch.Tensor:
    """``[n, n]`` mask of ``dtype`` on ``device``: 1/0 values, or 0/-inf when
    ``additive`` (reference trianglemix_attention.py:53-261)."""
    device = _device_of(device)
    out = torch.empty((n, n), dtype=dtype, device=device)
    with torch.cuda.device(device):
        check(_lib.load().tmx_dense_mask(_ptr(out), n, n, si, sl, last, MASK_KINDS[kind], dtype_code(dtype),
                                         int(additive), _stream(device)))
    return out


def replace_neg_inf_(mask: torch.Tensor, value: float) -> torch.Tensor:
    """In place: -inf -> ``value`` (reference ascend_trianglemix.py:131-141)."""
    _require_gpu(mask, "mask")
    if not mask.is_contiguous():
        raise RuntimeError("mask must be contiguous")
    with torch.cuda.device(mask.device):
        check(_lib.load().tmx_replace_neg_inf(_ptr(mask), mask.numel(), dtype_code(mask.dtype), float(value),
                                              _stream(mask.device)))
    return mask


# --------------------------------------------------------------------------- index
def attention_block_m(num_heads: int, num_kv_heads: int) -> int:
    bm = _lib.load().tmx_attention_block_m(num_heads, num_kv_heads)
    if bm <= 0:
        raise RuntimeError(f"num_heads {num_heads} must be a positive multiple of num_kv_heads {num_kv_heads}")
    return bm


def attention_block_n() -> int:
    return _lib.load().tmx_attention_block_n()


def index_sizes(cu_seqlens_q_cpu, seq_lens_cpu, si: int, sl: int, last: int, block_m: int, block_n: int,
                mode: int, chunk_tiles: int = DEFAULT_CHUNK_TILES, tri_min_len: int = 0,
                full_lens_cpu=None, dense_rows_from: int = 0) -> IndexSizes:
    """Exact buffer sizes of the index, from the host copies of the batch.  ``full_lens_cpu`` (optional):
    the length every sequence will have once its prompt is complete (chunked prefill).  ``dense_rows_from``
    (0 or ``TMX_DENSE_MIN_ROWS``): requests with that many query rows or more count nothing (an index that leaves
    them to the dense causal kernel)."""
    cu = np.ascontiguousarray(np.asarray(cu_seqlens_q_cpu), dtype=np.int32)
    sl_ = np.ascontiguousarray(np.asarray(seq_lens_cpu), dtype=np.int32)
    if cu.ndim != 1 or sl_.ndim != 1 or cu.shape[0] != sl_.shape[0] + 1:
        raise RuntimeError("cu_seqlens_q must have one more entry than seq_lens")
    fl = None
    if full_lens_cpu is not None:
        fl = np.ascontiguousarray(np.asarray(full_lens_cpu), dtype=np.int32)
        if fl.shape != sl_.shape:
            raise RuntimeError("full_lens must have one entry per request")
    sz = IndexSizes()
    check(_lib.load().tmx_index_sizes_without_dense(_np_ptr(cu), _np_ptr(sl_), None if fl is None else _np_ptr(fl),
                                                    int(sl_.shape[0]), si, sl, last, block_m, block_n, mode, tri_min_len,
                                                    chunk_tiles, int(dense_rows_from), ctypes.byref(sz)))
    return sz


def units_per_item(num_heads: int, num_kv_heads: int) -> int:
    """Workgroups that serve one work item: K/V heads x head chunks of a K/V group (4 query heads per
    workgroup when the group size allows)."""
    group = num_heads // num_kv_heads
    return num_kv_heads * (group // math.gcd(group, 4))


def decode_units_per_item(num_heads: int, num_kv_heads: int) -> int:
    """Workgroups of the decode kernel per work item: it takes 8 query heads of a K/V head per workgroup when the
    group size is a multiple of 8.  The count to give :func:`suggest_chunk_tiles` for a step of one-row requests
    out of an fp8 cache (measured, 64 x 8192: 16 / 2 heads 54 us in 64-tile pieces against 78 us in 128-tile ones,
    8 / 1 heads 33 us in 32-tile pieces against 42 us; a 16-bit cache is fastest with :func:`units_per_item`)."""
    group = num_heads // num_kv_heads
    if group % 8 == 0:
        return num_kv_heads * (group // 8)
    return units_per_item(num_heads, num_kv_heads)


def suggest_chunk_tiles(cu_seqlens_q_cpu, seq_lens_cpu, si: int, sl: int, last: int, block_m: int, block_n: int,
                        mode: int, num_units: int, tri_min_len: int = 0, full_lens_cpu=None, num_cus: int = 0) -> int:
    """``chunk_tiles`` to build the index with (``tmx_suggest_chunk_tiles``): one piece of the dense rows
    per CU when the batch also holds sparse rows, 64 / 128 (decode) otherwise.  ``num_units`` =
    :func:`units_per_item`; ``num_cus`` 0 asks the current device."""
    cu = np.ascontiguousarray(np.asarray(cu_seqlens_q_cpu), dtype=np.int32)
    sl_ = np.ascontiguousarray(np.asarray(seq_lens_cpu), dtype=np.int32)
    fl = None if full_lens_cpu is None else np.ascontiguousarray(np.asarray(full_lens_cpu), dtype=np.int32)
    out = ctypes.c_int(0)
    check(_lib.load().tmx_suggest_chunk_tiles(_np_ptr(cu), _np_ptr(sl_),
                                              None if fl is None else _np_ptr(fl),
                                              int(sl_.shape[0]), si, sl, last, block_m, block_n, mode, tri_min_len,
                                              int(num_units), int(num_cus), ctypes.byref(out)))
    return int(out.value)


class BlockIndex:
    """Device-resident packed block-sparse index + work list (all int32), in ONE device allocation: the
    arrays are slices of ``buf`` (16-byte aligned starts), made on demand, so a step pays for one
    allocation and one launch sequence."""
    __slots__ = ("buf", "_off", "sizes", "c", "batch", "block_m", "block_n", "mode", "si", "sl", "last",
                 "chunk_tiles", "tri_min_len", "block_table", "full_lens", "full_lens_cpu", "_op_args", "built",
                 "_batch_tensors", "stage_rows")

    def __init__(self, buf, off, sizes, c, batch, block_m, block_n, mode, si, sl, last, chunk_tiles, tri_min_len,
                 block_table, full_lens, full_lens_cpu):
        self.buf, self._off, self.sizes, self.c = buf, off, sizes, c
        self.batch, self.block_m, self.block_n, self.mode = batch, block_m, block_n, mode
        self.si, self.sl, self.last, self.chunk_tiles, self.tri_min_len = si, sl, last, chunk_tiles, tri_min_len
        # kept alive: the C struct points at them
        self.block_table, self.full_lens, self.full_lens_cpu = block_table, full_lens, full_lens_cpu
        self._op_args = None
        # False: the buffers are allocated and described, nothing is in them yet (build_block_index(defer=True));
        # the first attention call with the index builds it (TMX_ATTN_BUILD_INDEX), or build_now()
        self.built = True
        self._batch_tensors = None
        # rows of the dense causal kernel's staging area for this batch (0: none is used): dense_stage_rows()
        self.stage_rows = 0

    def build_now(self) -> None:
        """Builds a deferred index with a launch of its own (what the first attention call would otherwise do)."""
        if self.built:
            return
        cu, sl = self._batch_tensors
        with torch.cuda.device(cu.device):
            check(_lib.load().tmx_build_block_index(ctypes.byref(self.c), _ptr(cu), _ptr(sl), None, None, _stream(cu.device)))
        self.built = True

    def op_args(self):
        """(the eight index arrays as the list ``torch.ops._trianglemix_C`` takes, tile_pages or None, the
        counts num_items / num_partials / num_split_qblocks / nnz_tiles): made once per index."""
        if self._op_args is None:
            def arr(name):
                o, n = self._off[name]
                return self.buf[o:o + max(n, 1)]  # never empty: the library wants a pointer for every array
            sz = self.sizes
            self._op_args = ([arr(n) for n in ("row_ptr", "col_idx", "qb_meta", "part_ptr", "items", "split_qb", "totals",
                                               "scratch")],
                             arr("tile_pages") if "tile_pages" in self._off else None,
                             (int(sz.num_items), int(sz.num_partials), int(sz.num_split_qblocks), int(sz.nnz_tiles)))
        return self._op_args

    def _arr(self, name: str) -> torch.Tensor:
        o, n = self._off[name]
        return self.buf[o:o + n]

    row_ptr = property(lambda self: self._arr("row_ptr"))          # [R + 1]
    col_idx = property(lambda self: self._arr("col_idx"))          # [nnz]
    qb_meta = property(lambda self: self._arr("qb_meta").view(-1, 2))   # [R, 2]
    part_ptr = property(lambda self: self._arr("part_ptr"))        # [R + 1]
    items = property(lambda self: self._arr("items").view(-1, 8))  # [items, 8]
    split_qb = property(lambda self: self._arr("split_qb"))        # [split q-blocks]
    totals = property(lambda self: self._arr("totals"))            # [8]
    scratch = property(lambda self: self._arr("scratch"))

    @property
    def tile_pages(self) -> Optional[torch.Tensor]:                # [nnz, 2] or None
        return self._arr("tile_pages").view(-1, 2) if "tile_pages" in self._off else None

    def pages_resolved_from(self, block_table: Optional[torch.Tensor], page_size: int) -> bool:
        """True when ``tile_pages`` holds the physical pages of exactly this block table (same memory, same row
        stride) at this page size: only then may an attention call read them instead of the block table."""
        bt = self.block_table
        return (bt is not None and block_table is not None and "tile_pages" in self._off
                and bt.data_ptr() == block_table.data_ptr() and bt.stride(0) == block_table.stride(0)
                and int(self.c.pages_page_size) == int(page_size))

    @property
    def num_qblocks(self) -> int:
        return int(self.sizes.num_qblocks)

    @property
    def nnz_tiles(self) -> int:
        return int(self.sizes.nnz_tiles)


def build_block_index(cu_seqlens_q: torch.Tensor, seq_lens: torch.Tensor, cu_seqlens_q_cpu, seq_lens_cpu,
                      si: int, sl: int, last: int, block_m: int, block_n: int, mode: int,
                      chunk_tiles: int = DEFAULT_CHUNK_TILES, tri_min_len: int = 0,
                      block_table: Optional[torch.Tensor] = None, page_size: int = 0,
                      full_lens: Optional[torch.Tensor] = None, full_lens_cpu=None, defer: bool = False,
                      leave_dense: bool = False) -> BlockIndex:
    """Build the packed block-sparse index on the device (one allocation, at most two launches, no
    host synchronisation; buffer sizes come from the host copies).  ``defer``: only size, allocate and describe
    the index; the first ``attention`` call that gets it builds it as part of its own work
    (``TMX_ATTN_BUILD_INDEX``: inside the attention launch where that launch has workgroups that do not need the
    index, else as a launch of its own first) -- same index, same results.  With a
    ``block_table`` (int32 ``[B, max_blocks]``) and ``page_size`` the physical
    pages of every listed tile are resolved as well (paged K/V, ``block_n`` 32).
    ``full_lens`` (int32 ``[B]`` on the device, with its host copy): the length each
    sequence will have once its prompt is complete, so that the steps of a chunked
    prefill attend exactly as the one-step prefill of the whole prompt does.
    ``leave_dense`` (causal mode): requests with ``TMX_DENSE_MIN_ROWS`` (17) query rows or more get no q-blocks at
    all -- the attention call computes them with its dense causal kernel, which reads no index.  For callers that
    know the call can run that kernel (head size 128, query heads per K/V head a multiple of 4, no bias, at most 512
    requests; ``attention`` refuses the index otherwise): a long prompt of a layer outside the triangle set then
    costs neither the list of its half a million tiles nor gigabytes of partial-result slots nobody reads."""
    _require_gpu(cu_seqlens_q, "cu_seqlens_q")
    _require_gpu(seq_lens, "seq_lens")
    if cu_seqlens_q.dtype != torch.int32 or seq_lens.dtype != torch.int32:
        raise RuntimeError("cu_seqlens_q and seq_lens must be int32")
    dev = cu_seqlens_q.device
    cu_h = np.ascontiguousarray(np.asarray(cu_seqlens_q_cpu), dtype=np.int32)
    sl_h = np.ascontiguousarray(np.asarray(seq_lens_cpu), dtype=np.int32)
    batch = int(sl_h.shape[0])
    if seq_lens.numel() != batch or cu_seqlens_q.numel() != batch + 1:
        raise RuntimeError("device and host batch descriptions differ in length")
    fl_h = None
    if (full_lens is None) != (full_lens_cpu is None):
        raise RuntimeError("full_lens and full_lens_cpu go together")
    if full_lens is not None:
        _require_gpu(full_lens, "full_lens")
        fl_h = np.ascontiguousarray(np.asarray(full_lens_cpu), dtype=np.int32)
        if full_lens.dtype != torch.int32 or full_lens.numel() != batch or fl_h.shape[0] != batch:
            raise RuntimeError("full_lens must be int32 with one entry per request")
    if leave_dense and mode != MODES["causal"]:
        raise RuntimeError("leave_dense goes with the causal mode")
    dense_from = _lib.TMX_DENSE_MIN_ROWS if leave_dense else 0
    sz = index_sizes(cu_h, sl_h, si, sl, last, block_m, block_n, mode, chunk_tiles, tri_min_len, fl_h, dense_from)
    R, nnz, n_items, n_split = (int(sz.num_qblocks), int(sz.nnz_tiles), int(sz.num_items),
                                int(sz.num_split_qblocks))
    if block_table is not None:
        _require_gpu(block_table, "block_table")
        if block_table.dtype != torch.int32 or block_table.dim() != 2 or block_table.stride(1) != 1:
            raise RuntimeError("block_table must be int32 [B, max_blocks] with contiguous rows")
    # one allocation; every array starts on a multiple of 4 ints (16 bytes: the alignment the work list needs)
    off, total = {}, 0
    for name, n in (("items", 8 * n_items), ("tile_pages", 2 * nnz if block_table is not None else -1),
                    ("row_ptr", R + 1), ("col_idx", nnz), ("qb_meta", 2 * R), ("part_ptr", R + 1),
                    ("split_qb", n_split), ("totals", 8), ("scratch", batch + 1 + 2 * (R + 1) + 8 * (R // 1024 + 1))):
        if n < 0:
            continue
        off[name] = (total, n)
        total += (max(n, 1) + 3) & ~3
    buf = torch.empty((total,), dtype=torch.int32, device=dev)
    base = buf.data_ptr()

    def at(name):
        return base + 4 * off[name][0]

    c = Index()
    c.row_ptr, c.col_idx, c.qb_meta, c.part_ptr = at("row_ptr"), at("col_idx"), at("qb_meta"), at("part_ptr")
    c.items, c.split_qb, c.totals, c.scratch = at("items"), at("split_qb"), at("totals"), at("scratch")
    c.cap_qblocks, c.cap_nnz, c.cap_items, c.cap_split = R, nnz, n_items, n_split
    c.block_m, c.block_n, c.mode, c.chunk_tiles = block_m, block_n, mode, chunk_tiles
    c.si, c.sl, c.last, c.batch = si, sl, last, batch
    c.tri_min_len = tri_min_len
    if block_table is not None:
        c.tile_pages, c.pages_block_table = at("tile_pages"), block_table.data_ptr()
        c.pages_block_table_stride, c.pages_page_size = block_table.stride(0), int(page_size)
    else:
        c.tile_pages, c.pages_block_table, c.pages_block_table_stride, c.pages_page_size = None, None, 0, 0
    c.full_lens = None if full_lens is None else full_lens.data_ptr()
    c.full_lens_host = None if fl_h is None else _np_ptr(fl_h)
    c.dense_rows_from, c.reserved = dense_from, 0
    idx = BlockIndex(buf, off, sz, c, batch, block_m, block_n, mode, si, sl, last, chunk_tiles, tri_min_len,
                     block_table, full_lens, fl_h)
    if mode == MODES["causal"]:
        idx.stage_rows = dense_stage_rows(cu_h, sl_h)
    if defer:
        idx.built = False
        idx._batch_tensors = (cu_seqlens_q, seq_lens)
        return idx
    with torch.cuda.device(dev):
        check(_lib.load().tmx_build_block_index(ctypes.byref(c), _ptr(cu_seqlens_q), _ptr(seq_lens),
                                                _np_ptr(cu_h),
                                                _np_ptr(sl_h), _stream(dev)))
    return idx


def dense_stage_rows(cu_seqlens_q_cpu, seq_lens_cpu, min_rows_per_request: int = -1) -> int:
    """Rows of the dense causal kernel's staging area for a batch (``tmx_dense_stage_rows``): the sequence lengths of
    its long requests, each rounded up to 64 -- or 0 where staging would not pay (fewer than
    ``min_rows_per_request`` query rows per such request on average; negative: the library's default; 0: always)."""
    cu = np.ascontiguousarray(np.asarray(cu_seqlens_q_cpu), dtype=np.int32)
    sl_ = np.ascontiguousarray(np.asarray(seq_lens_cpu), dtype=np.int32)
    if cu.shape[0] != sl_.shape[0] + 1:
        raise RuntimeError("cu_seqlens_q must have one entry more than seq_lens")
    return int(_lib.load().tmx_dense_stage_rows(_np_ptr(cu), _np_ptr(sl_), int(sl_.shape[0]), int(min_rows_per_request)))


def without_dense_requests(cu_seqlens_q_cpu, seq_lens_cpu):
    """The host batch description as an index built with ``leave_dense`` sees it: requests with
    ``TMX_DENSE_MIN_ROWS`` query rows or more have no rows (their sequence shrinks to its context).  For the
    scheduling advice (``suggest_chunk_tiles``) of such an index."""
    cu = np.asarray(cu_seqlens_q_cpu, dtype=np.int64)
    sl_ = np.asarray(seq_lens_cpu, dtype=np.int64)
    q = np.diff(cu)
    keep = np.where(q >= _lib.TMX_DENSE_MIN_ROWS, 0, q)
    return (np.concatenate([[0], np.cumsum(keep)]).astype(np.int32), (sl_ - (q - keep)).astype(np.int32))


def alloc_kv_stage(rows: int, num_kv_heads: int, device) -> Optional[torch.Tensor]:
    """The staging area for ``rows`` key rows: ``[2, rows, Hkv, 128]`` 16-bit values (``None`` for 0 rows)."""
    if rows <= 0:
        return None
    return torch.empty((2, int(rows), int(num_kv_heads), 128), dtype=torch.float16, device=device)


# --------------------------------------------------------------------------- attention
# Which binding of the library `attention` goes through (TRIANGLEMIX_BINDING): "torch" (default) =
# torch.ops._trianglemix_C.trianglemix_attention, the way the reference calls its own native ops
# (vllm/_custom_ops.py:153-196); "ctypes" = the C struct filled here.  Same library entry point either way.
_BINDING = os.environ.get("TRIANGLEMIX_BINDING", "torch")
_op_ns = None


def _op_binding():
    global _op_ns, _BINDING
    if _BINDING != "torch":
        return None
    if _op_ns is None:
        from . import torch_ops
        try:
            _op_ns = torch_ops.load()
        except ImportError:
            _BINDING = "ctypes"  # the operator library is not built: the ctypes binding serves
            return None
    return _op_ns


def _check_batch_of(index: BlockIndex, cu_seqlens_q: torch.Tensor, seq_lens: torch.Tensor) -> None:
    """A deferred index is built from the batch description of the attention call that builds it: it must be the one
    the index was sized for."""
    cu, sl = index._batch_tensors
    if cu.data_ptr() != cu_seqlens_q.data_ptr() or sl.data_ptr() != seq_lens.data_ptr():
        raise RuntimeError("a deferred index is built by the first attention call: pass the cu_seqlens_q / seq_lens "
                           "tensors it was sized with (or call index.build_now() first)")


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, index: BlockIndex,
              cu_seqlens_q: torch.Tensor, seq_lens: torch.Tensor, scale: float,
              cu_seqlens_k: Optional[torch.Tensor] = None, block_table: Optional[torch.Tensor] = None,
              bias: Optional[torch.Tensor] = None, workspace: Optional[Sequence[torch.Tensor]] = None,
              flags: int = 0, static_workspace: bool = False, k_scale: float = 1.0, v_scale: float = 1.0,
              kv_stage: Optional[torch.Tensor] = None,
              k_new: Optional[torch.Tensor] = None, v_new: Optional[torch.Tensor] = None,
              cache_slots: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Masked attention over the tiles ``index`` lists.

    ``k_new`` / ``v_new`` (paged 16-bit cache): the step's own key / value rows ``[T, Hkv, D]``; every request's
    keys from position ``seq_len - q_len`` on are read from them instead of the cache, so the call does not
    depend on this step's cache write.  ``cache_slots`` (int64 ``[T]``, the step's ``slot_mapping``): the
    same call also stores those rows into the cache (``reshape_and_cache`` fused into the attention launch).

    An fp8 (e4m3fn) paged cache (``k``/``v`` of dtype uint8 or float8_e4m3fn, as ``reshape_and_cache_fp8`` writes
    it) is read in place: the kernel descales inside its tile loop with ``k_scale`` / ``v_scale`` (head size 128).

    ``kv_stage`` (or a third member of ``workspace``, as :func:`alloc_workspace` makes it): the staging area of
    the dense causal kernel, :func:`alloc_kv_stage` of at least ``index.stage_rows`` rows.  A causal launch whose
    index says that staging pays (``index.stage_rows > 0``) gathers its K/V rows into it first and streams its
    tiles from there: same results, less time on long prompts.

    ``static_workspace``: call ``tmx_attention_static_ws`` -- the partial results of split q-blocks go
    to the library's fixed device array instead of ``workspace`` (nothing is allocated; steps that need
    more than ``tmx_static_workspace_bytes()`` raise).

    ``q``/``out``: ``[T, H, D]`` (last dim contiguous).  Contiguous K/V:
    ``k``/``v`` ``[rows, Hkv, D]`` + ``cu_seqlens_k``.  Paged K/V: ``k``/``v``
    ``[num_blocks, page_size, Hkv, D]`` + ``block_table`` ``[B, max_blocks]`` int32.
    ``bias``: optional additive ``[rows >= max position + 1, cols >= max seq_len]``, or 4-D
    ``[1|batch, 1|heads, rows, cols]`` (per request and / or per head).
    """
    if workspace is not None and len(workspace) > 2:
        kv_stage = workspace[2] if kv_stage is None else kv_stage
        workspace = workspace[:2]
    if kv_stage is not None:
        if index.stage_rows <= 0:
            kv_stage = None  # (this batch does not stage)
        elif (kv_stage.dim() != 4 or kv_stage.shape[0] != 2 or kv_stage.shape[3] != 128 or kv_stage.element_size() != 2
              or not kv_stage.is_contiguous() or kv_stage.shape[1] < index.stage_rows or kv_stage.device != q.device
              or kv_stage.shape[2] != (k.shape[2] if k.dim() == 4 else k.shape[1])):
            raise RuntimeError("kv_stage must be a contiguous 16-bit [2, rows >= index.stage_rows, Hkv, 128] tensor on "
                               "the device of q (ops.alloc_kv_stage)")
    C = None if static_workspace else _op_binding()
    if C is not None:
        # the registered operator (csrc/torch_bindings.cpp): the same library call, argument checks in C++,
        # about half the host time of filling the C struct from Python
        index_list, tile_pages, (n_items, n_part, n_split, nnz) = index.op_args()
        part_o = part_ml = None
        if workspace is not None:
            part_o, part_ml = workspace
        elif n_part > 0:
            part_o, part_ml = alloc_workspace(index, q.shape[1], q.shape[2], q.device)
        # the operator stamps the call's own block table on the index it hands to the library, so the pages the
        # builder resolved go along only when they were resolved from that very table and page size (an index
        # used with another table falls back to the kernel's own block-table lookups, as the ctypes route does)
        if tile_pages is not None and not index.pages_resolved_from(block_table, k.shape[1] if k.dim() == 4 else 0):
            tile_pages = None
            index.build_now()  # (a deferred index is built with the table its pages belong to, not with this call's)
        if not index.built:
            _check_batch_of(index, cu_seqlens_q, seq_lens)
            flags = int(flags) | _lib.TMX_ATTN_BUILD_INDEX
        try:
            C.trianglemix_attention(out, q, k, v, block_table, cu_seqlens_q, seq_lens, cu_seqlens_k, index_list,
                                    tile_pages, part_o, part_ml, bias, float(scale),
                                    index.si, index.sl, index.last, index.mode, index.tri_min_len, index.chunk_tiles,
                                    n_items, n_part, n_split, index.full_lens, int(flags), nnz, k_new, v_new, cache_slots,
                                    float(k_scale), float(v_scale), index.block_m, index.block_n, kv_stage,
                                    int(index.c.dense_rows_from))
        except RuntimeError as e:
            raise _lib.TmxError(-1, str(e)) from None
        index.built = True
        return out
    if not index.built:
        _check_batch_of(index, cu_seqlens_q, seq_lens)
        flags = int(flags) | _lib.TMX_ATTN_BUILD_INDEX
    for t, name in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (cu_seqlens_q, "cu_seqlens_q"),
                    (seq_lens, "seq_lens")):
        _require_gpu(t, name)
    if q.dim() != 3 or out.dim() != 3 or q.shape != out.shape:
        raise RuntimeError("q and out must both be [T, H, D]")
    fp8 = k.dtype in (torch.uint8, FP8_DTYPE)
    if fp8:
        if block_table is None or bias is not None or q.shape[-1] != 128:
            raise RuntimeError("an fp8 K/V cache is read in place only in the paged layout, with head size 128 and "
                               "no bias (otherwise: gather_dequant_kv_fp8 first)")
        k, v = _as_bytes(k), _as_bytes(v)
    if q.dtype not in (torch.bfloat16, torch.float16) or (not fp8 and (k.dtype != q.dtype or v.dtype != q.dtype)):
        raise RuntimeError("q, k, v must share a dtype of bfloat16 or float16 (or K/V be an fp8 cache)")
    if q.stride(2) != 1 or out.stride(2) != 1 or k.stride(-1) != 1 or v.stride(-1) != 1:
        raise RuntimeError("the head dimension must be contiguous")
    if k.shape != v.shape or k.stride() != v.stride():
        raise RuntimeError("k and v must have the same shape and strides")
    T, H, D = q.shape
    paged = block_table is not None
    a = AttentionArgs()
    a.struct_size = ctypes.sizeof(AttentionArgs)
    a.io_dtype, a.out_dtype = dtype_code(q.dtype), dtype_code(out.dtype)
    a.mode, a.si, a.sl, a.last, a.scale = index.mode, index.si, index.sl, index.last, float(scale)
    a.batch, a.num_heads, a.head_dim, a.total_q = index.batch, H, D, T
    a.out, a.out_stride_t, a.out_stride_h = out.data_ptr(), out.stride(0), out.stride(1)
    a.q, a.q_stride_t, a.q_stride_h = q.data_ptr(), q.stride(0), q.stride(1)
    a.k, a.v = k.data_ptr(), v.data_ptr()
    if paged:
        _require_gpu(block_table, "block_table")
        if k.dim() != 4 or block_table.dim() != 2 or block_table.dtype != torch.int32:
            raise RuntimeError("paged K/V: cache [num_blocks, page_size, Hkv, D], int32 block_table [B, max_blocks]")
        a.num_kv_heads = k.shape[2]
        a.kv_stride_page, a.kv_stride_t, a.kv_stride_h = k.stride(0), k.stride(1), k.stride(2)
        a.block_table, a.block_table_stride, a.page_size = block_table.data_ptr(), block_table.stride(0), k.shape[1]
        if block_table.stride(1) != 1 or block_table.shape[0] < index.batch:
            raise RuntimeError("block_table rows must be contiguous and cover the batch")
        a.cu_seqlens_k = None
    else:
        if cu_seqlens_k is None or k.dim() != 3:
            raise RuntimeError("contiguous K/V: k/v [rows, Hkv, D] and cu_seqlens_k are required")
        _require_gpu(cu_seqlens_k, "cu_seqlens_k")
        a.num_kv_heads = k.shape[1]
        a.kv_stride_page, a.kv_stride_t, a.kv_stride_h = 0, k.stride(0), k.stride(1)
        a.block_table, a.block_table_stride, a.page_size = None, 0, 0
        a.cu_seqlens_k = cu_seqlens_k.data_ptr()
    if k.shape[-1] != D:
        raise RuntimeError("k/v head size differs from q")
    a.kv_dtype, a.k_scale, a.v_scale = (_lib.TMX_FP8_E4M3, float(k_scale), float(v_scale)) if fp8 else (0, 1.0, 1.0)
    if k_new is not None or v_new is not None:
        if k_new is None or v_new is None or not paged or fp8:
            raise RuntimeError("k_new and v_new go together and need a 16-bit paged cache")
        _require_gpu(k_new, "k_new")
        _require_gpu(v_new, "v_new")
        if (k_new.dim() != 3 or k_new.shape != v_new.shape or k_new.stride() != v_new.stride() or k_new.stride(2) != 1
                or k_new.dtype != q.dtype or v_new.dtype != q.dtype or k_new.shape[0] < T
                or tuple(k_new.shape[1:]) != (a.num_kv_heads, D)):
            raise RuntimeError("k_new / v_new must be [T, Hkv, D] rows of the q dtype with equal strides")
        a.k_new, a.v_new, a.new_stride_t, a.new_stride_h = k_new.data_ptr(), v_new.data_ptr(), k_new.stride(0), k_new.stride(1)
    else:
        a.k_new, a.v_new, a.new_stride_t, a.new_stride_h = None, None, 0, 0
    a.cache_slots = None
    if cache_slots is not None:
        _require_gpu(cache_slots, "cache_slots")
        if k_new is None or cache_slots.dtype != torch.int64 or not cache_slots.is_contiguous() or cache_slots.numel() < T:
            raise RuntimeError("cache_slots must be contiguous int64 [T] and comes with k_new / v_new")
        a.cache_slots = cache_slots.data_ptr()
    a.cu_seqlens_q, a.seq_lens = cu_seqlens_q.data_ptr(), seq_lens.data_ptr()
    if bias is not None:
        _require_gpu(bias, "bias")
        if bias.dim() == 2:
            bias = bias[None, None]
        if bias.dim() != 4 or bias.stride(3) != 1 or bias.shape[0] not in (1, index.batch) or bias.shape[1] not in (1, H):
            raise RuntimeError("bias must be [rows, cols] or [1|batch, 1|heads, rows, cols] with contiguous rows")
        a.bias, a.bias_dtype, a.bias_stride = bias.data_ptr(), dtype_code(bias.dtype), bias.stride(2)
        a.bias_rows, a.bias_cols = bias.shape[2], bias.shape[3]
        a.bias_stride_b = bias.stride(0) if bias.shape[0] > 1 else 0
        a.bias_stride_h = bias.stride(1) if bias.shape[1] > 1 else 0
    else:
        a.bias, a.bias_dtype, a.bias_stride, a.bias_rows, a.bias_cols = None, 0, 0, 0, 0
        a.bias_stride_b = a.bias_stride_h = 0
    a.index = ctypes.pointer(index.c)
    sz = index.sizes
    a.num_items, a.num_partials, a.num_split_qblocks = sz.num_items, sz.num_partials, sz.num_split_qblocks
    a.nnz_tiles = sz.nnz_tiles
    part_o = part_ml = None
    if static_workspace:
        pass
    elif workspace is not None:
        part_o, part_ml = workspace
    elif sz.num_partials > 0:
        part_o, part_ml = alloc_workspace(index, H, D, q.device)
    if sz.num_partials > 0 and not static_workspace:
        need = int(sz.num_partials) * H * index.block_m
        if part_o.numel() < need * D or part_ml.numel() < need * 2 or part_o.dtype != torch.float32:
            raise RuntimeError("attention workspace too small for this index")
    a.partial_o, a.partial_ml = _ptr(part_o), _ptr(part_ml)
    a.flags, a.reserved = int(flags), 0
    a.kv_stage, a.kv_stage_rows = (kv_stage.data_ptr(), int(kv_stage.shape[1])) if kv_stage is not None else (None, 0)
    with torch.cuda.device(q.device):
        lib = _lib.load()
        entry = lib.tmx_attention_static_ws if static_workspace else lib.tmx_attention
        check(entry(ctypes.byref(a), _stream(q.device)))
    index.built = True
    return out


# --------------------------------------------------------------------------- decode steps with a fixed launch shape
def decode_workspace(batch: int, num_heads: int, num_kv_heads: int, head_dim: int, device, fp8: bool = False):
    """Work area of :func:`decode_attention` for a batch size (``tmx_decode_workspace_rows``).  Make it once and keep
    it: a captured step replays with the same pointers."""
    with torch.cuda.device(device):
        rows = int(_lib.load().tmx_decode_workspace_rows(int(batch), int(num_heads), int(num_kv_heads),
                                                         _lib.TMX_FP8_E4M3 if fp8 else 0, 0))
    if rows < 0:
        raise RuntimeError("tmx_decode_workspace_rows failed: " + _lib.load().tmx_last_error().decode("utf-8", "replace"))
    rows = max(rows, 1)
    return (torch.empty((rows * head_dim,), dtype=torch.float32, device=device),
            torch.empty((rows * 2,), dtype=torch.float32, device=device))


def decode_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, block_table: torch.Tensor,
                     seq_lens: torch.Tensor, scale: float, workspace, k_scale: float = 1.0, v_scale: float = 1.0,
                     k_new: Optional[torch.Tensor] = None, v_new: Optional[torch.Tensor] = None,
                     cache_slots: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A step of one-row requests (``tmx_decode_attention``): row ``b`` of ``q`` ``[B, H, D]`` is request ``b``'s
    query at position ``seq_lens[b] - 1`` and attends to every key of its sequence in the paged cache ``k`` / ``v``
    (16-bit or fp8).  No index and nothing sized from the requests' lengths: the launch depends on ``B`` only (its
    pieces are laid out on the device from ``seq_lens``), so the call can be captured in a HIP graph
    (``torch.cuda.graph``) and replayed after ``seq_lens`` / ``block_table`` / the caches changed in place.
    ``seq_lens[b] <= 0`` marks a padding row.  ``workspace``: :func:`decode_workspace` for this batch size.
    ``k_new`` / ``v_new`` / ``cache_slots``: the step's own rows, stored to their slots by the same launch -- into
    an fp8 cache as ``fp8(x / scale)``, the bytes ``reshape_and_cache_fp8`` stores, and the row's own key is then
    attended to as the cache holds it."""
    if workspace is None or workspace[0] is None or workspace[1] is None:
        raise RuntimeError("decode_attention needs its work area (ops.decode_workspace for the batch size)")
    C = _op_binding()
    if C is not None:
        # the registered operator: the same library call, argument checks in C++
        try:
            C.trianglemix_decode_attention(out, q, k, v, block_table, seq_lens, workspace[0], workspace[1], float(scale),
                                           k_new, v_new, cache_slots, float(k_scale), float(v_scale))
        except RuntimeError as e:
            raise _lib.TmxError(-1, str(e)) from None
        return out
    for t, name in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (block_table, "block_table"), (seq_lens, "seq_lens")):
        _require_gpu(t, name)
    if q.dim() != 3 or out.shape != q.shape or q.stride(2) != 1 or out.stride(2) != 1:
        raise RuntimeError("q and out must both be [B, H, D] with a contiguous head dimension")
    if seq_lens.dtype != torch.int32 or seq_lens.numel() < q.shape[0] or not seq_lens.is_contiguous():
        raise RuntimeError("seq_lens must be contiguous int32 [B]")
    if block_table.dtype != torch.int32 or block_table.dim() != 2 or block_table.stride(1) != 1 or block_table.shape[0] < q.shape[0]:
        raise RuntimeError("block_table must be int32 [B, max_blocks] with contiguous rows")
    fp8 = k.dtype in (torch.uint8, FP8_DTYPE)
    if fp8:
        k, v = _as_bytes(k), _as_bytes(v)
    elif k.dtype != q.dtype or v.dtype != q.dtype:
        raise RuntimeError("q, k, v must share a dtype of bfloat16 or float16 (or K/V be an fp8 cache)")
    if k.dim() != 4 or k.shape != v.shape or k.stride() != v.stride() or k.stride(-1) != 1 or k.shape[-1] != q.shape[-1]:
        raise RuntimeError("paged K/V: caches [num_blocks, page_size, Hkv, D] with equal strides")
    B, H, D = q.shape
    a = _lib.DecodeArgs()
    a.struct_size = ctypes.sizeof(_lib.DecodeArgs)
    a.io_dtype, a.out_dtype, a.scale = dtype_code(q.dtype), dtype_code(out.dtype), float(scale)
    a.batch, a.num_heads, a.num_kv_heads, a.head_dim = B, H, k.shape[2], D
    a.out, a.out_stride_t, a.out_stride_h = out.data_ptr(), out.stride(0), out.stride(1)
    a.q, a.q_stride_t, a.q_stride_h = q.data_ptr(), q.stride(0), q.stride(1)
    a.k, a.v = k.data_ptr(), v.data_ptr()
    a.kv_stride_page, a.kv_stride_t, a.kv_stride_h = k.stride(0), k.stride(1), k.stride(2)
    a.block_table, a.block_table_stride, a.page_size = block_table.data_ptr(), block_table.stride(0), k.shape[1]
    a.kv_dtype, a.k_scale, a.v_scale = (_lib.TMX_FP8_E4M3, float(k_scale), float(v_scale)) if fp8 else (0, 1.0, 1.0)
    if (k_new is None) != (v_new is None):
        raise RuntimeError("k_new and v_new go together")
    if k_new is not None:
        _require_gpu(k_new, "k_new")
        _require_gpu(v_new, "v_new")
        if (k_new.dim() != 3 or k_new.shape != v_new.shape or k_new.stride() != v_new.stride() or k_new.stride(2) != 1
                or k_new.dtype != q.dtype or v_new.dtype != q.dtype or k_new.shape[0] < B
                or tuple(k_new.shape[1:]) != (a.num_kv_heads, D)):
            raise RuntimeError("k_new / v_new must be [B, Hkv, D] rows of the q dtype with equal strides")
        a.k_new, a.v_new, a.new_stride_t, a.new_stride_h = k_new.data_ptr(), v_new.data_ptr(), k_new.stride(0), k_new.stride(1)
    a.cache_slots = None
    if cache_slots is not None:
        _require_gpu(cache_slots, "cache_slots")
        if k_new is None or cache_slots.dtype != torch.int64 or not cache_slots.is_contiguous() or cache_slots.numel() < B:
            raise RuntimeError("cache_slots must be contiguous int64 [B] and comes with k_new / v_new")
        a.cache_slots = cache_slots.data_ptr()
    a.seq_lens = seq_lens.data_ptr()
    if workspace is None or workspace[0] is None or workspace[1] is None or workspace[0].dtype != torch.float32:
        raise RuntimeError("decode_attention needs its work area (ops.decode_workspace for the batch size)")
    part_o, part_ml = workspace
    a.partial_o, a.partial_ml = part_o.data_ptr(), part_ml.data_ptr()
    a.workspace_rows = min(part_o.numel() // D, part_ml.numel() // 2)
    a.flags = a.reserved = 0
    with torch.cuda.device(q.device):
        check(_lib.load().tmx_decode_attention(ctypes.byref(a), _stream(q.device)))
    return out


def eager_attention_f32(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, mode: int, si: int = 0,
                        sl: int = 0, last: int = 0, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``softmax(q k^T * scale + mask) v`` in fp32 on contiguous ``[B, N, H, D]`` float32 tensors (the reference's
    eager attention run with fp32 inputs, ascend_trianglemix.py:88-106).  ``bias``: anything that broadcasts to
    ``[B, H, N, N]``."""
    for t, name in ((q, "q"), (k, "k"), (v, "v")):
        _require_gpu(t, name)
    if q.dim() != 4 or q.dtype != torch.float32 or k.shape != q.shape or v.shape != q.shape or k.dtype != q.dtype or v.dtype != q.dtype:
        raise RuntimeError("q, k, v must be float32 [B, N, H, D] of one shape")
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    B, N, H, D = q.shape
    out = torch.empty_like(q)
    bptr, bdt, sb, sh, si_ = None, 0, 0, 0, 0
    if bias is not None:
        _require_gpu(bias, "bias")
        if bias.dtype not in _DTYPE:
            bias = bias.float()
        bias = torch.broadcast_to(bias, (B, H, N, N)) if tuple(bias.shape) != (B, H, N, N) else bias
        if bias.stride(3) != 1 and N > 1:
            bias = bias.contiguous()
        bptr, bdt, sb, sh, si_ = bias.data_ptr(), dtype_code(bias.dtype), bias.stride(0), bias.stride(1), bias.stride(2)
    with torch.cuda.device(q.device):
        check(_lib.load().tmx_eager_attention_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, N, H, D, float(scale), int(mode),
                                                  si, sl, last, bptr, bdt, sb, sh, si_, _stream(q.device)))
    return out


def alloc_workspace(index: BlockIndex, num_heads: int, head_dim: int, device, num_kv_heads: Optional[int] = None):
    """Partial-result buffers for the q-blocks the index splits -- and, with ``num_kv_heads``, the dense causal
    kernel's staging area as a third member where the index says that staging pays (``index.stage_rows``)."""
    n = max(1, int(index.sizes.num_partials) * num_heads * index.block_m)
    ws = (torch.empty((n * head_dim,), dtype=torch.float32, device=device),
          torch.empty((n * 2,), dtype=torch.float32, device=device))
    if num_kv_heads is not None and head_dim == 128 and index.stage_rows > 0:
        ws = ws + (alloc_kv_stage(index.stage_rows, num_kv_heads, device),)
    return ws


# --------------------------------------------------------------------------- KV cache
def reshape_and_cache(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                      slot_mapping: torch.Tensor) -> None:
    """Write this step's K/V rows into the paged cache (contract of the
    reference's reshape_and_cache_flash, rocm_attn.py:345-354)."""
    for t, name in ((key, "key"), (value, "value"), (key_cache, "key_cache"), (value_cache, "value_cache"),
                    (slot_mapping, "slot_mapping")):
        _require_gpu(t, name)
    if slot_mapping.dtype != torch.int64 or not slot_mapping.is_contiguous():
        raise RuntimeError("slot_mapping must be contiguous int64")
    if key.shape != value.shape or key.stride() != value.stride() or key.dim() != 3:
        raise RuntimeError("key and value must be [T, Hkv, D] with equal strides")
    if key_cache.shape != value_cache.shape or key_cache.stride() != value_cache.stride() or key_cache.dim() != 4:
        raise RuntimeError("caches must be [num_blocks, page_size, Hkv, D] with equal strides")
    if key_cache.dtype != key.dtype:
        raise RuntimeError("cache dtype must equal the key dtype (16-bit caches)")
    n_tok = min(int(slot_mapping.numel()), key.shape[0])
    with torch.cuda.device(key.device):
        check(_lib.load().tmx_reshape_and_cache(
            _ptr(key), _ptr(value), key.stride(0), key.stride(1), _ptr(key_cache), _ptr(value_cache),
            key_cache.stride(0), key_cache.stride(1), key_cache.stride(2), _ptr(slot_mapping), n_tok,
            key.shape[1], key.shape[2], key_cache.shape[1], dtype_code(key.dtype), _stream(key.device)))


# --------------------------------------------------------------------------- fp8 K/V cache

def _as_bytes(cache: torch.Tensor) -> torch.Tensor:
    if cache.dtype not in (torch.uint8, FP8_DTYPE):
        raise RuntimeError(f"an fp8 cache is uint8 or {FP8_DTYPE}, not {cache.dtype}")
    return cache.view(torch.uint8)


def reshape_and_cache_fp8(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                          slot_mapping: torch.Tensor, k_scale: float = 1.0, v_scale: float = 1.0) -> None:
    """Write this step's K/V rows into an fp8 (e4m3fn) paged cache: byte = fp8(x / scale)
    (reshape_and_cache_flash with kv_cache_dtype "fp8", rocm_attn.py:345-354)."""
    for t, name in ((key, "key"), (value, "value"), (key_cache, "key_cache"), (value_cache, "value_cache"),
                    (slot_mapping, "slot_mapping")):
        _require_gpu(t, name)
    if slot_mapping.dtype != torch.int64 or not slot_mapping.is_contiguous():
        raise RuntimeError("slot_mapping must be contiguous int64")
    if key.shape != value.shape or key.stride() != value.stride() or key.dim() != 3:
        raise RuntimeError("key and value must be [T, Hkv, D] with equal strides")
    kc, vc = _as_bytes(key_cache), _as_bytes(value_cache)
    if kc.shape != vc.shape or kc.stride() != vc.stride() or kc.dim() != 4:
        raise RuntimeError("caches must be [num_blocks, page_size, Hkv, D] with equal strides")
    n_tok = min(int(slot_mapping.numel()), key.shape[0])
    with torch.cuda.device(key.device):
        check(_lib.load().tmx_reshape_and_cache_fp8(
            _ptr(key), _ptr(value), key.stride(0), key.stride(1), _ptr(kc), _ptr(vc), kc.stride(0), kc.stride(1),
            kc.stride(2), _ptr(slot_mapping), n_tok, key.shape[1], key.shape[2], kc.shape[1], dtype_code(key.dtype),
            float(k_scale), float(v_scale), _stream(key.device)))


def gather_dequant_kv_fp8(key_cache: torch.Tensor, value_cache: torch.Tensor, block_table: torch.Tensor,
                          seq_lens: torch.Tensor, cu_seqlens_k: torch.Tensor, total_rows: int, max_seq_len: int,
                          out_dtype: torch.dtype, k_scale: float = 1.0, v_scale: float = 1.0,
                          out: Optional[Sequence[torch.Tensor]] = None):
    """The keys / values of every request of the step, out of an fp8 cache, as contiguous 16-bit rows
    ``[total_rows, Hkv, D]`` (request ``b``'s key ``j`` is row ``cu_seqlens_k[b] + j``): the descale of
    rocm_attn.py:421-422 done ahead of the attention kernel.  Returns ``(k_rows, v_rows)``."""
    for t, name in ((key_cache, "key_cache"), (value_cache, "value_cache"), (block_table, "block_table"),
                    (seq_lens, "seq_lens"), (cu_seqlens_k, "cu_seqlens_k")):
        _require_gpu(t, name)
    kc, vc = _as_bytes(key_cache), _as_bytes(value_cache)
    if kc.shape != vc.shape or kc.stride() != vc.stride() or kc.dim() != 4:
        raise RuntimeError("caches must be [num_blocks, page_size, Hkv, D] with equal strides")
    if block_table.dtype != torch.int32 or block_table.dim() != 2 or block_table.stride(1) != 1:
        raise RuntimeError("block_table must be int32 [B, max_blocks] with contiguous rows")
    hkv, d = kc.shape[2], kc.shape[3]
    if out is None:
        k_rows = torch.empty((max(total_rows, 1), hkv, d), dtype=out_dtype, device=kc.device)
        v_rows = torch.empty_like(k_rows)
    else:
        k_rows, v_rows = out
        if k_rows.shape[0] < total_rows or not k_rows.is_contiguous() or not v_rows.is_contiguous():
            raise RuntimeError("output rows too small or not contiguous")
    with torch.cuda.device(kc.device):
        check(_lib.load().tmx_gather_dequant_kv_fp8(
            _ptr(kc), _ptr(vc), kc.stride(0), kc.stride(1), kc.stride(2), _ptr(block_table), block_table.stride(0),
            kc.shape[1], _ptr(seq_lens), _ptr(cu_seqlens_k), int(seq_lens.numel()), int(max_seq_len), hkv, d,
            _ptr(k_rows), _ptr(v_rows), dtype_code(out_dtype), float(k_scale), float(v_scale), _stream(kc.device)))
    return k_rows, v_rows
