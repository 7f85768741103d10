/*
 * trianglemix_hip.h -- C ABI of libtrianglemix_hip.so (MI355X / gfx950).
 *
 * The library replaces the TriangleMix hot path of the reference fork:
 *   - the Python O(N^2) mask loops   (vllm/attention/trianglemix_attention.py:53-261)
 *   - the eager masked attention     (vllm/attention/ascend_trianglemix.py:52-108)
 * and is what a TORCH_LIBRARY binding in the style of the reference's
 * csrc/rocm/torch_bindings.cpp:14-57 / csrc/rocm/ops.h:17-26 forwards to
 * (tensors become pointers + strides, the current HIP stream is passed in).
 *
 * Conventions
 *   - Plain C: pointers, sizes, enums as int.  No torch types.
 *   - Every pointer named *_dev / in a struct marked (device) is device memory
 *     of the current HIP device; *_host pointers are host memory.
 *   - Every function returns TMX_OK (0) or a TMX_ERR_* code; tmx_last_error()
 *     gives the message of the calling thread's last failure.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *     Calls enqueue work and return; nothing synchronises with the host and
 *     nothing is allocated on the device.
 *   - Positions: request b owns query rows cu_seqlens_q[b] .. cu_seqlens_q[b+1]-1;
 *     they are the LAST q_len positions of a sequence of seq_lens[b] tokens, so
 *     query row r sits at position i = seq_lens[b] - q_len + r (the context of
 *     seq_lens[b] - q_len earlier tokens lives only in the K/V storage).
 *   - Mask predicate (reference trianglemix_attention.py:82-89,126-129,192-196):
 *       allowed(i, j) = j <= i && (j <= si || i - j <= sl || N - i < last)
 *     with N = seq_lens[b]; TMX_MODE_CAUSAL is j <= i (trianglemix_attention.py:218-219),
 *     TMX_MODE_NONE attends to every key (ascend_trianglemix.py:96 with no mask).
 */
#ifndef TRIANGLEMIX_HIP_H_
#define TRIANGLEMIX_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMX_ABI_VERSION 6
/* int32 entries of tmx_index_t.scratch for a batch of `batch` requests and a capacity of R q-blocks */
#define TMX_INDEX_SCRATCH_INTS(batch, R) ((batch) + 1 + 2 * ((R) + 1) + 8 * ((R) / 1024 + 1))

enum {
  TMX_OK = 0,
  TMX_ERR_INVALID_ARG = 1,  /* bad pointer / size / enum value */
  TMX_ERR_UNSUPPORTED = 2,  /* valid request the kernels do not cover (head size, dtype ...) */
  TMX_ERR_HIP = 3,          /* a HIP runtime call failed */
  TMX_ERR_CAPACITY = 4      /* caller-provided buffer capacities too small */
};

/* element types */
enum { TMX_F32 = 0, TMX_BF16 = 1, TMX_F16 = 2, TMX_FP8_E4M3 = 3 };

/* attention / index modes */
enum { TMX_MODE_NONE = 0, TMX_MODE_CAUSAL = 1, TMX_MODE_TRIANGLE = 2 };
/* Query rows from which a request of a causal launch goes to the dense causal kernel (shorter ones, decode and
 * speculative decode, stay on the work list). */
#define TMX_DENSE_MIN_ROWS 17

/* dense mask kinds for tmx_dense_mask */
enum {
  TMX_MASK_STREAMING = 0, /* trianglemix_attention.py:53-91   */
  TMX_MASK_LAST = 1,      /* trianglemix_attention.py:93-131  */
  TMX_MASK_MIDDLE = 2,    /* trianglemix_attention.py:133-171 */
  TMX_MASK_TRIANGLE = 3,  /* trianglemix_attention.py:173-198 */
  TMX_MASK_DENSE = 4      /* trianglemix_attention.py:200-219 */
};

int tmx_abi_version(void);
const char* tmx_last_error(void);

/* Number of query rows per q-block the attention kernel uses for a given
 * grouped-head ratio (num_heads / num_kv_heads), and its key-tile width.
 * The index passed to tmx_attention must be built with exactly these. */
int tmx_attention_block_m(int num_heads, int num_kv_heads);
int tmx_attention_block_n(void);

/* ------------------------------------------------------------------------
 * Dense [n, n] mask materialisation -- API parity with
 * TriangleMixMaskGenerator._create_*_mask and .get_attention_mask
 * (trianglemix_attention.py:53-261).  Writes n rows of n elements of `dtype`
 * (TMX_F32 / TMX_BF16 / TMX_F16) at out_dev with a row stride in elements.
 * additive == 0: 1 where attended, 0 elsewhere;
 * additive != 0: 0 where attended, -inf elsewhere (:254-258).
 * ---------------------------------------------------------------------- */
int tmx_dense_mask(void* out_dev, int64_t row_stride, int n, int si, int sl, int last,
                   int kind, int dtype, int additive, void* stream);

/* AscendNPUTriangleMixOptimizer.optimize_mask_for_npu (ascend_trianglemix.py:131-141):
 * in place, every -inf of `count` contiguous elements becomes `value`. */
int tmx_replace_neg_inf(void* mask_dev, int64_t count, int dtype, float value, void* stream);

/* ------------------------------------------------------------------------
 * Packed block-sparse index.
 *
 * The batch is cut into q-blocks of block_m rows, request by request (a
 * q-block never spans two requests).  For q-block r the index lists, in
 * ascending order, every key tile t (keys [t*block_n, (t+1)*block_n)) that
 * holds at least one attended (row, key) pair.  CSR layout, int32, shared by
 * all heads:
 *     row_ptr[R + 1], col_idx[nnz]
 * plus per q-block metadata and a work list for the attention kernel:
 *     qb_meta[R][2]    = (request b, first row offset inside the request)
 *     part_ptr[R + 1]  = first partial slot of each q-block (see below)
 *     items[I][8]      = (q-block, first entry in col_idx, tile count, partial slot,
 *                         request b, first row in the packed q tensor, position of that row, seq_len)
 *     split_qb[S]      = the q-blocks whose tile list was split, ascending
 * A q-block with more than chunk_tiles tiles is split into pieces of at most
 * chunk_tiles tiles; each piece is one item with its own partial slot, merged
 * by the combine pass of tmx_attention.  Unsplit q-blocks have slot -1.  The
 * item list starts with the pieces of all split q-blocks (the long jobs:
 * q-blocks DESCENDING, pieces of one q-block ascending) and goes on with the
 * unsplit q-blocks, DESCENDING: the dense trailing rows of every request start
 * first.
 *
 * Chunked prefill: the reference builds its mask for the whole sequence at
 * once (N = seq_len, trianglemix_attention.py:126-129, 330-332).  When a
 * prompt arrives in several steps, full_lens[b] (optional) is the length the
 * sequence will have once its prompt is complete; N = max(seq_len[b],
 * full_lens[b]) then decides the dense trailing rows (N - i < last) and the
 * length gate, so that every step attends exactly as the one-step prefill
 * of the whole prompt does.  Without full_lens, N = seq_len[b].
 * ---------------------------------------------------------------------- */
typedef struct tmx_index_sizes_t {
  int64_t num_qblocks; /* R */
  int64_t nnz_tiles;   /* entries of col_idx */
  int64_t num_items;   /* I */
  int64_t num_partials;/* partial slots = part_ptr[R] */
  int64_t num_split_qblocks; /* S */
  int64_t max_tiles_per_qblock;
} tmx_index_sizes_t;

/* Host-side sizing from the host mirrors of the batch (vLLM keeps them:
 * CommonAttentionMetadata.query_start_loc_cpu / seq_lens_cpu,
 * vllm/v1/attention/backends/utils.py:53-60).  No device work. */
int tmx_index_sizes(const int32_t* cu_seqlens_q_host, const int32_t* seq_lens_host,
                    const int32_t* full_lens_host /* [batch] or NULL */, int batch,
                    int si, int sl, int last, int block_m, int block_n, int mode, int tri_min_len,
                    int chunk_tiles, tmx_index_sizes_t* out);

/* Scheduling advice (host only; no counterpart in the reference, whose masks are dense): the
 * chunk_tiles to build the index with, for a batch and a head grouping.  units_per_item =
 * num_kv_heads * (group size / heads per workgroup) = num_kv_heads for group sizes that are
 * multiples of 4; num_cus <= 0 queries the current device.  Measured on MI355X: a prefill with sparse
 * and dense rows runs fastest when the dense rows make one piece per CU (8192 tokens: 16 tiles,
 * 32768: 64, 65536: 128 for Qwen3-8B's heads; decode rows in the same step ask for pieces of up to 32
 * tiles); batches of dense rows only get one piece per workgroup slot of the chip (two per CU), within
 * 8..512 tiles for decode steps and 64..256 for causal prompts. */
int tmx_suggest_chunk_tiles(const int32_t* cu_seqlens_q_host, const int32_t* seq_lens_host,
                            const int32_t* full_lens_host /* or NULL */, int batch, int si, int sl, int last,
                            int block_m, int block_n, int mode, int tri_min_len, int units_per_item, int num_cus,
                            int* out);

/* The same for an index that leaves the long requests of a causal launch to the dense causal kernel
 * (tmx_index_t.dense_rows_from): requests with dense_rows_from query rows or more count no q-blocks, tiles,
 * items or partial slots.  dense_rows_from == 0: tmx_index_sizes. */
int tmx_index_sizes_without_dense(const int32_t* cu_seqlens_q_host, const int32_t* seq_lens_host,
                                  const int32_t* full_lens_host /* or NULL */, int batch, int si, int sl, int last,
                                  int block_m, int block_n, int mode, int tri_min_len, int chunk_tiles,
                                  int dense_rows_from, tmx_index_sizes_t* out);

typedef struct tmx_index_t {
  /* all (device), int32, sized from tmx_index_sizes_t */
  int32_t* row_ptr;   /* [R + 1] */
  int32_t* col_idx;   /* [nnz] */
  int32_t* qb_meta;   /* [R][2] */
  int32_t* part_ptr;  /* [R + 1] */
  int32_t* items;     /* [I][8], 16-byte aligned */
  int32_t* split_qb;  /* [S] (may be NULL when S == 0) */
  int32_t* totals;    /* [8]: R, nnz, I, partials, S, max tiles per q-block, 0, 0 -- as the kernel counted them */
  int32_t* scratch;   /* [batch + 1 + 2 * (R + 1) + 8 * (R / 1024 + 1)] work area of the scan
                         (TMX_INDEX_SCRATCH_INTS) */
  /* capacities the buffers above were allocated with */
  int64_t cap_qblocks, cap_nnz, cap_items, cap_split;
  int32_t block_m, block_n, mode, chunk_tiles;
  int32_t si, sl, last, batch;
  /* TMX_MODE_TRIANGLE only: requests with seq_len <= tri_min_len fall back to
   * TMX_MODE_CAUSAL (the reference helper's length gate,
   * trianglemix_attention.py:330-332, applied per request); 0 = no gate. */
  int32_t tri_min_len;
  /* Optional page resolution (paged K/V, block_n == 32): when tile_pages and
   * pages_block_table are set before the build, entry e of col_idx also gets
   * tile_pages[e] = (physical page of the tile's keys 0..15, of its keys 16..31)
   * looked up in the request's block-table row, so that the attention kernel
   * reads no block table.  Valid for the block table / page size it was built
   * from; tmx_attention falls back to its own lookups for any other. */
  int32_t pages_page_size;
  int32_t* tile_pages;                /* (device) [nnz][2] or NULL */
  const int32_t* pages_block_table;   /* (device) [batch, pages_block_table_stride] or NULL */
  int64_t pages_block_table_stride;
  /* Optional (see "Chunked prefill" above): set before the build, read again by tmx_attention. */
  const int32_t* full_lens;           /* (device) [batch] or NULL */
  const int32_t* full_lens_host;      /* host mirror, for the capacity check of the build; or NULL */
  /* TMX_MODE_CAUSAL only, 0 or TMX_DENSE_MIN_ROWS: requests with at least this many query rows get NO q-blocks (no
   * tiles, work items or partial slots; sizes from tmx_index_sizes_without_dense).  tmx_attention computes those
   * requests with its dense causal kernel, which reads no index -- a 32k-token prompt of a layer outside the
   * triangle set otherwise lists 525k tiles and asks for 4.6 GB of partial results that nobody reads (128k tokens:
   * 8.4M tiles, 71 GB).  An attention call that cannot run the dense kernel (bias, head size other than 128, query
   * heads per K/V head not a multiple of 4, more than 512 requests) refuses such an index. */
  int32_t dense_rows_from;
  int32_t reserved;
} tmx_index_t;

/* Builds the index on the device from the device copies of the batch
 * description.  idx->block_m .. idx->batch and the capacities must be filled
 * in by the caller; buffers must be at least as large as tmx_index_sizes says
 * (checked against the capacities with the host-computed sizes when the host
 * mirrors are given; pass NULL mirrors to skip that check). */
int tmx_build_block_index(tmx_index_t* idx, const int32_t* cu_seqlens_q_dev, const int32_t* seq_lens_dev,
                          const int32_t* cu_seqlens_q_host, const int32_t* seq_lens_host, void* stream);

/* ------------------------------------------------------------------------
 * Masked attention (prefill and decode), grouped heads, ragged batch.
 * Replaces AscendTriangleMixAttention.forward (ascend_trianglemix.py:52-108)
 * and is the kernel behind AttentionImpl.forward
 * (vllm/attention/backends/abstract.py:158-171; layout conventions of
 * vllm/v1/attention/backends/rocm_attn.py:275-426).
 * ---------------------------------------------------------------------- */
typedef struct tmx_attention_args_t {
  uint32_t struct_size; /* sizeof(tmx_attention_args_t), ABI check */
  int32_t io_dtype;     /* q / k / v element type: TMX_BF16 or TMX_F16 */
  int32_t out_dtype;    /* TMX_BF16 / TMX_F16 (== io_dtype) or TMX_F32 */
  int32_t mode;         /* TMX_MODE_* ; must equal index->mode */
  int32_t si, sl, last; /* must equal the index parameters */
  float scale;          /* softmax scale, > 0 (the row maximum is taken before scaling) */

  int32_t batch, num_heads, num_kv_heads, head_dim;
  int32_t total_q;      /* rows of q / out */

  void* out;            /* (device) [total_q, num_heads, head_dim] */
  int64_t out_stride_t, out_stride_h;
  const void* q;        /* (device) [total_q, num_heads, head_dim] */
  int64_t q_stride_t, q_stride_h; /* strides in elements; head_dim is contiguous */

  /* Keys / values.  block_table == NULL: contiguous rows, request b's key j is
   * row cu_seqlens_k[b] + j of k / v ([rows, num_kv_heads, head_dim]).
   * block_table != NULL: paged cache [num_blocks, page_size, num_kv_heads, head_dim],
   * request b's key j is in page block_table[b * block_table_stride + j / page_size],
   * slot j % page_size; strides kv_stride_page / kv_stride_t / kv_stride_h. */
  const void* k;
  const void* v;
  int64_t kv_stride_page, kv_stride_t, kv_stride_h;
  const int32_t* block_table; /* (device) or NULL */
  int64_t block_table_stride;
  int32_t page_size;
  const int32_t* cu_seqlens_k; /* (device) [batch + 1], contiguous mode only */
  /* K/V element type: 0 (or io_dtype) = the 16-bit type of q; TMX_FP8_E4M3 = an fp8 (e4m3fn) paged cache as
   * tmx_reshape_and_cache_fp8 writes it (kv_cache_dtype "fp8"): strides then count bytes, and the kernel
   * reads value = fp8 x k_scale / v_scale, rounded once to the 16-bit type, inside its tile loop (the
   * k_descale / v_descale of vllm/v1/attention/backends/rocm_attn.py:421-422).  Paged layout, head_dim 128,
   * no bias. */
  int32_t kv_dtype;
  float k_scale, v_scale;
  /* Optional second source of keys / values (paged 16-bit cache only): the step's own key / value rows, the
   * tensors tmx_reshape_and_cache takes ([total_q, num_kv_heads, head_dim], row t belongs to query row t; strides
   * in elements).  Request b's keys from position seq_lens[b] - q_len on are then read from these rows and only
   * the earlier ones from the cache, so the call does not depend on the cache write of the same step: the
   * reference writes first and reads the cache back (vllm/v1/attention/backends/rocm_attn.py:345-354, 398-426);
   * here the write may run beside the attention on another stream.  Same values either way. */
  const void* k_new;
  const void* v_new;
  int64_t new_stride_t, new_stride_h;
  /* Optional, with k_new / v_new: the cache write of the step, fused into this call.  Row t of k_new / v_new is
   * also stored to flat slot cache_slots[t] of the paged cache k / v (slot = page * page_size + offset, negative:
   * skipped): the contract of tmx_reshape_and_cache, which then need not be called.  The cache must be writable;
   * the stores are complete when the call's work on the stream is. */
  const int64_t* cache_slots; /* (device) [total_q] or NULL */

  const int32_t* cu_seqlens_q; /* (device) [batch + 1] */
  const int32_t* seq_lens;     /* (device) [batch] */

  /* Optional additive bias added to the scaled scores before the softmax (the attn_mask argument of
   * ascend_trianglemix.py:57,84,96-97, anything that broadcasts to [batch, heads, N, N]): the element for
   * request b, head h, position i, key j is bias[b * bias_stride_b + h * bias_stride_h + i * bias_stride + j];
   * a mask shared by requests and / or heads has zero strides there. */
  const void* bias;
  int32_t bias_dtype; /* TMX_F32 / TMX_BF16 / TMX_F16 */
  int64_t bias_stride; /* row stride in elements */
  int32_t bias_rows, bias_cols;
  int64_t bias_stride_b, bias_stride_h;

  const tmx_index_t* index;
  int64_t num_items, num_partials, num_split_qblocks; /* from tmx_index_sizes */
  int64_t nnz_tiles; /* from tmx_index_sizes (entries of col_idx); 0 = unknown, the capacity is used. Only
                        the static schedule's weights read it: results never depend on it. */

  /* Work area for split q-blocks (device, float32):
   * partial_o [num_partials][num_heads][block_m][head_dim], partial_ml [num_partials][num_heads][block_m][2].
   * May be NULL when num_partials == 0. */
  float* partial_o;
  float* partial_ml;

  /* Measurement aids (0 in normal use): TMX_ATTN_SKIP_COMBINE leaves split
   * q-blocks as partials, TMX_ATTN_ONLY_COMBINE runs the merge pass alone, so
   * that each kernel can be timed on its own with stream events. */
  /* TMX_ATTN_SHORT_QUERIES is a hint, results do not depend on it: no request of the step has more
   * than 16 query rows (decode, speculative decode); the kernel then lets the waves without rows
   * fetch the K/V tiles. */
  uint32_t flags;
  uint32_t reserved;

  /* Optional staging area of the dense causal kernel (device; NULL: not used), kv_stage_rows rows:
   * [2][kv_stage_rows][num_kv_heads][128] 16-bit values.  A causal launch (TMX_MODE_CAUSAL, head size 128) first
   * gathers the key and value rows of its long requests into it -- out of the pages, the step's own rows or an fp8
   * cache, converted once -- and its tiles then reach LDS by LDS-DMA: same results, about 7 % less time at 32k
   * tokens.  kv_stage_rows must be at least tmx_dense_stage_rows() of the batch (a smaller area is never written
   * past its end: the requests' rows are then left unwritten).  Contents are scratch: nothing is kept between
   * calls, and calls that share an area must not overlap. */
  void* kv_stage;
  int64_t kv_stage_rows;
} tmx_attention_args_t;

/* Rows of the staging area (tmx_attention_args_t.kv_stage) for a batch, from the host copies of cu_seqlens_q /
 * seq_lens: the sequence lengths of the requests the dense causal kernel takes (17 query rows or more), each
 * rounded up to a multiple of 64 -- or 0 when the gather would not pay (it copies every key row once; the kernel
 * reads every key row once per 64 query rows): fewer than min_rows_per_request query rows per such request on
 * average (negative: the default, 4096; 0: always).  No counterpart in the reference, whose dense layers run
 * torch's eager product (vllm/attention/ascend_trianglemix.py:88-106). */
int64_t tmx_dense_stage_rows(const int32_t* cu_seqlens_q_host, const int32_t* seq_lens_host, int batch,
                             int min_rows_per_request);

#define TMX_ATTN_SKIP_COMBINE 1u
#define TMX_ATTN_ONLY_COMBINE 2u
#define TMX_ATTN_SHORT_QUERIES 4u
/* Measurement aid: every q-block goes through the work list (the band workgroups that otherwise take the
 * sparse q-blocks in canonical position are not launched).  Results do not depend on it. */
#define TMX_ATTN_NO_BAND 8u
/* Measurement aid: a causal launch keeps every request on the work list (the dense causal kernel that
 * otherwise computes the requests with 17 query rows or more is not launched).  Same results within the
 * tolerance of the attention (the two kernels sum in different orders). */
#define TMX_ATTN_NO_DENSE 16u
/* Measurement aid: one-row requests stay on the work list (the decode kernel that otherwise computes the
 * one-row requests whose row is dense is not launched).  Same results within the tolerance of the attention. */
#define TMX_ATTN_NO_DECODE 32u
/* Hint from a caller that knows its batch (vLLM's max_query_len == 1): EVERY request of the step has exactly
 * one query row.  The work-list kernel is then not launched at all when the decode kernel covers the step
 * (head size 128, no bias, a causal or unmasked launch or a pattern with last >= 2 and no full_lens).  A step
 * that sets it with a longer request in the batch leaves that request's rows unwritten. */
#define TMX_ATTN_DECODE_ONLY 64u
/* Measurement aid: the decode kernel forms its weighted sum on the vector units instead of the matrix cores
 * (its first implementation; same results within the tolerance of the attention). */
#define TMX_ATTN_DECODE_VALU 128u
/* The index has NOT been built yet: this call builds it (args->index describes the buffers exactly as for
 * tmx_build_block_index, from args->cu_seqlens_q / seq_lens) and attends with it.  Replaces the pair
 * "mask generator, then attention" of the reference (vllm/attention/trianglemix_attention.py:310-340 ahead of
 * vllm/attention/ascend_trianglemix.py:88-106) by ONE call: where the launch has workgroups that do not read the
 * index (the band workgroups of a long triangle prompt), no other kernel of the call needs it first, the batch has
 * at most 16 requests and the index fits one workgroup's build (up to 1024 q-blocks, head size 128, 16-bit K/V),
 * workgroup 0 of the attention launch builds it while those workgroups already stream, and the workgroups that
 * walk the work list start when it is complete; every other step gets the build as a launch of its own ahead of
 * the attention.  (Measured on MI355X the in-launch form saves the build launch and gives the time back at the
 * end of the attention launch: no faster at 32k tokens, slower at 8k; callers after speed build first.)
 * Same index and same results as the two calls; the counts in args (num_items,
 * num_partials, num_split_qblocks, nnz_tiles) come from tmx_index_sizes as always.  Afterwards the index is
 * built: later calls of the step (the other layers) pass it without the flag. */
#define TMX_ATTN_BUILD_INDEX 256u

int tmx_attention(const tmx_attention_args_t* args, void* stream);

/* ------------------------------------------------------------------------
 * tmx_attention for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments
 * and results as tmx_attention, except that args->partial_o / partial_ml are ignored: the work area of
 * the split q-blocks is a fixed array of TMX_STATIC_WORKSPACE_BYTES inside the library's device image
 * (one copy per GPU, held until the process ends).  Steps whose partials need more
 * (num_partials * num_heads * block_m * (head_dim + 2) * 4 bytes, the first part rounded up to 256)
 * fail with TMX_ERR_CAPACITY and launch nothing.  The call allocates nothing and frees nothing (see
 * tmx_static_workspace_warmup for its first call on a GPU).
 * All calls on one GPU share the array: two calls that overlap in time on one GPU (different streams)
 * race on it; calls on one stream are ordered and safe.  Nothing in it needs clearing between calls
 * (every partial row is written by its piece before the merge pass reads it).
 * ---------------------------------------------------------------------- */
#define TMX_STATIC_WORKSPACE_BYTES (32u << 20)
size_t tmx_static_workspace_bytes(void);
int tmx_attention_static_ws(const tmx_attention_args_t* args, void* stream);
/* The first tmx_attention_static_ws call on a GPU asks the runtime for the device address of the array
 * (hipGetSymbolAddress, which may load the library's code object on that GPU: not something to do while a
 * stream is being captured).  Callers that capture call this once per GPU beforehand, with that GPU current;
 * afterwards tmx_attention_static_ws only launches kernels.  Idempotent, no counterpart in the reference. */
int tmx_static_workspace_warmup(void);

/* ------------------------------------------------------------------------
 * A step of one-row requests (decode) with a launch whose shape does not depend on the requests' lengths:
 * what a caller needs to capture the step in a HIP graph and replay it with other lengths and block tables,
 * as the reference's ROCm backend allows (cudagraph_support = AttentionCGSupport.ALWAYS,
 * vllm/v1/attention/backends/rocm_attn.py:63; its decode call: rocm_attn.py:398-426).  No index, no host
 * copies of the batch: request b owns query row b, which sits at position seq_lens[b] - 1 and attends to
 * every key of its sequence -- what the reference's pattern gives for every decode row (the last row always
 * lies among the trailing dense rows, trianglemix_attention.py:126-129, provided last >= 2) and what a layer
 * outside the triangle set computes.  Paged K/V, head_dim 128, query-head groups that are multiples of 4;
 * 16-bit or fp8 (e4m3fn) caches; k_new / v_new / cache_slots as in tmx_attention_args_t (row b = request b's
 * newest key / value, of the q dtype; the call then also stores it to its slot).  With an fp8 cache the stored
 * bytes are fp8(x / scale), exactly what tmx_reshape_and_cache_fp8 stores, and the row's own key / value is
 * attended to as the cache holds it -- the call equals that write followed by the attention over the cache
 * (the reference's pair, rocm_attn.py:345-358), bit for bit, in one launch.
 *
 * The launch depends on the batch size alone.  Its pieces are laid out on the device, by every workgroup for
 * itself, from seq_lens: with T the 32-key tiles of all requests, pieces are P = max(8, ceil(T * units / slots))
 * tiles long (slots = two workgroups per CU; units = K/V head chunks per piece), so that a step fills the chip
 * whatever its lengths are; the grid provides for slots / units + batch pieces, which no step of that batch size
 * exceeds, and workgroups beyond the step's pieces return at once.  seq_lens[b] <= 0 marks a padding row: nothing
 * is read and out row b is left alone.  Requests that take more than one piece go through partial results in the
 * caller's work area and a merge launch: partial_o [rows][head_dim] floats, partial_ml [rows][2] floats, rows =
 * tmx_decode_workspace_rows(batch, ...) (num_cus <= 0 asks the current device; -1 on a HIP error).  A captured
 * launch replays with the pointers it was captured with: keep the work area.  Nothing is allocated, nothing
 * synchronises with the host.
 * ---------------------------------------------------------------------- */
typedef struct tmx_decode_args_t {
  uint32_t struct_size; /* sizeof(tmx_decode_args_t), ABI check */
  int32_t io_dtype;     /* q element type: TMX_BF16 or TMX_F16 */
  int32_t out_dtype;    /* == io_dtype, or TMX_F32 */
  float scale;
  int32_t batch, num_heads, num_kv_heads, head_dim;
  void* out;            /* (device) [batch, num_heads, head_dim] */
  int64_t out_stride_t, out_stride_h;
  const void* q;        /* (device) [batch, num_heads, head_dim] */
  int64_t q_stride_t, q_stride_h;
  const void* k;        /* paged cache [num_blocks, page_size, num_kv_heads, head_dim] */
  const void* v;
  int64_t kv_stride_page, kv_stride_t, kv_stride_h; /* elements (fp8: bytes) */
  const int32_t* block_table; /* (device) [batch, block_table_stride] */
  int64_t block_table_stride;
  int32_t page_size;
  int32_t kv_dtype;     /* 0 / io_dtype, or TMX_FP8_E4M3 */
  float k_scale, v_scale;
  const void* k_new;    /* optional: [batch, num_kv_heads, head_dim] of the q dtype, row b = request b's newest key */
  const void* v_new;
  int64_t new_stride_t, new_stride_h;
  const int64_t* cache_slots; /* optional, with k_new / v_new: (device) [batch], negative = skipped */
  const int32_t* seq_lens;    /* (device) [batch] */
  float* partial_o;     /* work area (see above) */
  float* partial_ml;
  int64_t workspace_rows; /* rows the work area holds */
  uint32_t flags, reserved; /* 0 */
} tmx_decode_args_t;

int64_t tmx_decode_workspace_rows(int batch, int num_heads, int num_kv_heads, int kv_dtype, int num_cus);
int tmx_decode_attention(const tmx_decode_args_t* args, void* stream);

/* ------------------------------------------------------------------------
 * The eager attention of the reference in fp32 (ascend_trianglemix.py:88-106 run with fp32 tensors, the
 * configuration the reference itself runs on the CPU): out = softmax(q k^T * scale + mask) v, everything in fp32.
 * q / k / v / out: contiguous [batch, n, num_heads, head_dim] float32 (no grouped heads in this class); the
 * pattern is given by mode / si / sl / last as for tmx_attention (TMX_MODE_NONE: no mask at all, what the
 * reference computes up to 2048 tokens); bias (optional): added to the scaled scores, element for (b, h, i, j)
 * at bias[b * bias_stride_b + h * bias_stride_h + i * bias_stride_i + j] (zero strides broadcast).  A fully
 * masked row gives NaN, as the reference's softmax does.  A compatibility path on the vector units: sized for
 * the reference's test shapes, not a fast path (16-bit inputs take tmx_attention).
 * ---------------------------------------------------------------------- */
int tmx_eager_attention_f32(const float* q, const float* k, const float* v, float* out, int32_t batch, int32_t n,
                            int32_t num_heads, int32_t head_dim, float scale, int32_t mode, int32_t si, int32_t sl,
                            int32_t last, const void* bias, int32_t bias_dtype, int64_t bias_stride_b,
                            int64_t bias_stride_h, int64_t bias_stride_i, void* stream);

/* ------------------------------------------------------------------------
 * K/V cache write: copies key/value rows of this step into the paged cache,
 * row t to flat slot slot_mapping[t] (slot = page * page_size + offset;
 * negative slots are skipped).  Same contract as the reference's
 * reshape_and_cache_flash op (csrc/cache_kernels.cu:282-355, called at
 * vllm/v1/attention/backends/rocm_attn.py:345-354), 16-bit cache only.
 * ---------------------------------------------------------------------- */
int tmx_reshape_and_cache(const void* key, const void* value, int64_t kv_in_stride_t, int64_t kv_in_stride_h,
                          void* key_cache, void* value_cache, int64_t cache_stride_page, int64_t cache_stride_t,
                          int64_t cache_stride_h, const int64_t* slot_mapping_dev, int32_t num_tokens,
                          int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------
 * fp8 (e4m3fn, the format vllm/platforms/rocm.py:402-406 selects on gfx950)
 * K/V cache, kv_cache_dtype "fp8".
 *
 * tmx_reshape_and_cache_fp8: the cache write of reshape_and_cache_flash with an
 * fp8 cache (csrc/cache_kernels.cu:282-355 + the scaled conversion of
 * csrc/quantization/fp8/amd/quant_utils.cuh): byte = fp8(x / scale), nearest
 * even, finite values saturate at +-448.  Cache strides in elements (bytes).
 *
 * tmx_gather_dequant_kv_fp8: the keys / values a step attends to, converted
 * back (fp8 * scale, one rounding to the 16-bit type) into contiguous rows
 * k_out / v_out [cu_seqlens_k[batch], num_kv_heads, head_dim]; request b's key
 * j is row cu_seqlens_k[b] + j.  tmx_attention then runs in its contiguous
 * mode on them: the descale step of the reference's attention kernel
 * (k_descale / v_descale, vllm/v1/attention/backends/rocm_attn.py:421-422),
 * done ahead of the kernel instead of inside it.
 * ---------------------------------------------------------------------- */
int tmx_reshape_and_cache_fp8(const void* key, const void* value, int64_t kv_in_stride_t, int64_t kv_in_stride_h,
                              void* key_cache, void* value_cache, int64_t cache_stride_page, int64_t cache_stride_t,
                              int64_t cache_stride_h, const int64_t* slot_mapping_dev, int32_t num_tokens,
                              int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t in_dtype,
                              float k_scale, float v_scale, void* stream);

int tmx_gather_dequant_kv_fp8(const void* key_cache, const void* value_cache, int64_t cache_stride_page,
                              int64_t cache_stride_t, int64_t cache_stride_h, const int32_t* block_table_dev,
                              int64_t block_table_stride, int32_t page_size, const int32_t* seq_lens_dev,
                              const int32_t* cu_seqlens_k_dev, int32_t batch, int32_t max_seq_len,
                              int32_t num_kv_heads, int32_t head_dim, void* k_out, void* v_out, int32_t out_dtype,
                              float k_scale, float v_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TRIANGLEMIX_HIP_H_ */
