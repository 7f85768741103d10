// Dense causal attention kernel (tmx_dense.hip): parameters and launcher, shared with the dispatch in
// tmx_attention.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmx {

// Requests with at least this many query rows take the dense kernel; shorter ones (decode, speculative
// decode) stay on the work-list kernel, whose 16-row waves waste nothing on them.
constexpr int kDenseMinRows = 17;  // (TMX_DENSE_MIN_ROWS of the public header: checked where both are seen)
// The kernel keeps one block count per request in LDS.
constexpr int kDenseMaxBatch = 512;

struct DenseParams {
  const void* q;
  void* out;
  const void* k;
  const void* v;
  const int32_t* block_table;
  const int32_t* cu_k;
  const int32_t* cu_q;
  const int32_t* seq_lens;
  const void* k_new;  // the step's own K / V rows, or NULL (see tmx_attention_args_t)
  const void* v_new;
  int new_stride_t, new_stride_h;
  const long long* cache_slots;  // fused cache write (see fused_cache_write in tmx_attention.hip), or NULL
  int q_stride_t, q_stride_h, out_stride_t, out_stride_h;  // elements
  int kv_stride_page, kv_stride_t, kv_stride_h, block_table_stride;
  int page_size, page_shift;
  int H, Hkv, G;   // G = H / Hkv, a multiple of 4
  int Y;           // K/V head chunks: Hkv * G / 4
  int batch, total_q;
  int out_dtype;
  float scale_log2;
  int kv_fp8;               // K/V are an fp8 (e4m3fn) paged cache: strides in bytes, value = fp8 x scale
  float k_scale, v_scale;
  // Staging area, or NULL: [2][stage_rows][Hkv][128] 16-bit values.  A pass ahead of the kernel gathers the K rows
  // (first half) and the V rows (second half, as fp16) of the requests the kernel takes into it, request after
  // request, each padded to a multiple of 64 rows; the kernel's tiles then come into LDS by LDS-DMA.
  void* stage;
  long long stage_rows;
};

int launch_dense_causal(const DenseParams& p, bool bf16, bool paged, hipStream_t s);

// Rows of the staging area for a batch: the sequence lengths of the requests with at least kDenseMinRows query
// rows, each rounded up to a multiple of 64 -- or 0 when staging does not pay: it copies every key row once, the
// kernel reads every key row once per 64 query rows, so it pays from a few thousand query rows per request on
// (`min_rows_per_request` on average; negative: the default).
constexpr int kDenseStageMinRows = 4096;
long long dense_stage_rows(const int32_t* cu_q_host, const int32_t* seq_lens_host, int batch, int min_rows_per_request);

}  // namespace tmx
