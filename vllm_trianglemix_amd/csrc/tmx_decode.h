// Decode-row kernel (tmx_decode.hip): parameters, launcher and the rule that says which work items it
// computes, shared with the dispatch and the work-list kernel in tmx_attention.hip (which passes over them).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tmx_common.h"

namespace tmx {

// A request with one query row whose row attends to every key up to its own position: any row of a causal or
// unmasked launch, and in the triangle pattern the rows among the trailing `last` ones (N - i < last) or of a
// sequence at or below the length gate.
TMX_HD bool decode_row(int mode, int last, int tri_min_len, int q_len, int pos, int n_full) {
  if (q_len != 1) return false;
  return effective_mode(mode, n_full, tri_min_len) != TMX_MODE_TRIANGLE || (n_full - pos) < last;
}

// With other requests in the batch the decode kernel's grid still covers every work item (workgroups of
// other items return at once); beyond this many workgroups the launch is not worth it and decode rows stay
// on the work-list kernel.
constexpr long long kDecodeMaxMixedGrid = 8192;

struct DecodeParams {
  const void* q;
  void* out;
  const void* k;
  const void* v;
  const int32_t* block_table;
  const int32_t* cu_k;
  const int32_t* cu_q;
  const int32_t* full_lens;  // [batch] length of each sequence once its prompt is complete, or NULL
  const int32_t* items;      // the index's work items, 8 ints each
  const int32_t* col_idx;
  float* partial_o;
  float* partial_ml;
  const void* k_new;  // the step's own K / V rows, or NULL (see tmx_attention_args_t)
  const void* v_new;
  const long long* cache_slots;  // fused cache write of the step's rows, or NULL
  int new_stride_t, new_stride_h;
  int total_q;
  int q_stride_t, q_stride_h, out_stride_t, out_stride_h;  // elements
  int kv_stride_page, kv_stride_t, kv_stride_h, block_table_stride;
  int page_size, page_shift;  // a power of two, at least 16
  int H, Hkv, G;              // G = H / Hkv, a multiple of 4
  int valu_sum;               // weighted sum on the vector units instead of the matrix cores (TMX_ATTN_DECODE_VALU)
  int hw;                     // query heads per workgroup: 8 when G is a multiple of 8, else 4
  int Y;                      // K/V head chunks: Hkv * G / hw
  int block_m;                // rows per q-block of the index (layout of the partial slots)
  int num_items;
  int mode, last, tri_min_len;
  int out_dtype;
  float scale_log2;
  int kv_fp8;  // K/V are an fp8 (e4m3fn) paged cache: strides in bytes, value = fp8 x scale
  float k_scale, v_scale;
  // Fixed-grid form (tmx_decode_attention): no index.  The step's pieces are laid out on the device, by every
  // workgroup for itself, from seq_lens alone: T = all requests' 32-key tiles, piece length P = max(8, ceil(T *
  // plan_units / plan_slots)) tiles (one piece per workgroup slot of the chip), request b has ceil(tiles_b / P)
  // pieces, and the pieces of all requests are numbered in request order.  Workgroup (piece w, K/V head chunk)
  // of the grid's pieces_cap x Y finds the request that owns piece w; a request with more than one piece leaves
  // partial results in slot w (one row each: block_m = 1) for the merge launch.  Since sum_b ceil(tiles_b / P) <=
  // T / P + batch <= plan_slots / plan_units + batch, pieces_cap = that bound covers every step of the batch size.
  const int32_t* seq_lens;
  int pieces_cap, plan_units, plan_slots;
};

int launch_decode(const DecodeParams& p, bool bf16, bool paged, hipStream_t s);
// the fixed-grid form: batch x pieces_cap x Y workgroups, then the merge launch when pieces_cap > 1
int launch_decode_fixed(const DecodeParams& p, bool bf16, int batch, hipStream_t s);

}  // namespace tmx
