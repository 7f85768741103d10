// Error reporting, version and host-side index sizing of libtrianglemix_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include <climits>

#include "tmx_common.h"

namespace tmx {
static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
}  // namespace tmx

extern "C" {

int tmx_abi_version(void) { return TMX_ABI_VERSION; }

const char* tmx_last_error(void) { return tmx::g_error; }

int tmx_attention_block_m(int num_heads, int num_kv_heads) {
  if (num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads != 0) return -1;
  return tmx::block_m_for_group(num_heads / num_kv_heads);
}

int tmx_attention_block_n(void) { return tmx::kBlockN; }

int tmx_index_sizes(const int32_t* cu_q, const int32_t* seq_lens, const int32_t* full_lens, int batch, int si,
                    int sl, int last, int block_m, int block_n, int mode, int tri_min_len, int chunk_tiles,
                    tmx_index_sizes_t* out) {
  return tmx_index_sizes_without_dense(cu_q, seq_lens, full_lens, batch, si, sl, last, block_m, block_n, mode, tri_min_len,
                                       chunk_tiles, 0, out);
}

int tmx_index_sizes_without_dense(const int32_t* cu_q, const int32_t* seq_lens, const int32_t* full_lens, int batch, int si,
                                  int sl, int last, int block_m, int block_n, int mode, int tri_min_len, int chunk_tiles,
                                  int dense_rows_from, tmx_index_sizes_t* out) {
  TMX_CHECK_ARG(out != nullptr, "tmx_index_sizes: out is NULL");
  TMX_CHECK_ARG(dense_rows_from >= 0, "tmx_index_sizes: dense_rows_from is negative");
  TMX_CHECK_ARG(batch >= 0 && (batch == 0 || (cu_q && seq_lens)), "tmx_index_sizes: NULL batch arrays");
  TMX_CHECK_ARG(block_m > 0 && block_n > 0 && chunk_tiles > 0, "tmx_index_sizes: block sizes must be positive");
  TMX_CHECK_ARG(mode >= TMX_MODE_NONE && mode <= TMX_MODE_TRIANGLE, "tmx_index_sizes: bad mode %d", mode);
  int64_t R = 0, nnz = 0, items = 0, parts = 0, split = 0, mx = 0;
  for (int b = 0; b < batch; ++b) {
    const int ql = cu_q[b + 1] - cu_q[b];
    const int n = seq_lens[b];
    TMX_CHECK_ARG(ql >= 0 && n >= ql, "tmx_index_sizes: request %d has q_len %d, seq_len %d", b, ql, n);
    if (dense_rows_from > 0 && ql >= dense_rows_from) continue;  // left to the dense causal kernel: no q-blocks
    const int ctx = n - ql;
    const int nf = full_lens ? tmx::imax(n, full_lens[b]) : n;
    const int m = tmx::effective_mode(mode, nf, tri_min_len);
    const int nq = tmx::ceil_div(ql, block_m);
    for (int k = 0; k < nq;) {
      const int r0 = k * block_m, r1 = tmx::imin(r0 + block_m, ql);
      const int cnt = tmx::tile_ranges(m, ctx + r0, ctx + r1 - 1, n, nf, si, sl, last, block_n).count();
      const int64_t reps = tmx::same_count_run(m, k, ctx, ql, nf, si, sl, last, block_m, block_n);
      const int pieces = tmx::imax(1, tmx::ceil_div(cnt, chunk_tiles));
      R += reps;
      nnz += reps * cnt;
      items += reps * pieces;
      if (pieces > 1) {
        parts += reps * pieces;
        split += reps;
      }
      if (cnt > mx) mx = cnt;
      k += (int)reps;
    }
  }
  TMX_CHECK_ARG(nnz < INT_MAX && items < INT_MAX && R < INT_MAX, "tmx_index_sizes: index exceeds int32 range");
  out->num_qblocks = R;
  out->nnz_tiles = nnz;
  out->num_items = items;
  out->num_partials = parts;
  out->num_split_qblocks = split;
  out->max_tiles_per_qblock = mx;
  return TMX_OK;
}

int tmx_suggest_chunk_tiles(const int32_t* cu_q, const int32_t* seq_lens, const int32_t* full_lens, int batch, int si,
                            int sl, int last, int block_m, int block_n, int mode, int tri_min_len,
                            int units_per_item, int num_cus, int* out) {
  TMX_CHECK_ARG(out != nullptr, "tmx_suggest_chunk_tiles: out is NULL");
  TMX_CHECK_ARG(batch >= 0 && (batch == 0 || (cu_q && seq_lens)), "tmx_suggest_chunk_tiles: NULL batch arrays");
  TMX_CHECK_ARG(block_m > 0 && block_n > 0 && units_per_item > 0, "tmx_suggest_chunk_tiles: sizes must be positive");
  if (num_cus <= 0) {
    int dev = 0;
    TMX_CHECK_HIP(hipGetDevice(&dev));
    TMX_CHECK_HIP(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  // dense rows (the trailing rows of the pattern, every row of a causal request) against sparse rows; the dense
  // rows of requests with up to 16 query rows (decode) apart: their pieces stream K/V for next to no arithmetic
  long long dense_long = 0, dense_short = 0, sparse_blocks = 0;
  int sparse_max = 0;
  bool decode_only = true;
  for (int b = 0; b < batch; ++b) {
    const int ql = cu_q[b + 1] - cu_q[b];
    const int n = seq_lens[b];
    TMX_CHECK_ARG(ql >= 0 && n >= ql, "tmx_suggest_chunk_tiles: request %d has q_len %d, seq_len %d", b, ql, n);
    if (ql > 1) decode_only = false;
    const int ctx = n - ql;
    const int nf = full_lens ? tmx::imax(n, full_lens[b]) : n;
    const int m = tmx::effective_mode(mode, nf, tri_min_len);
    const int nq = tmx::ceil_div(ql, block_m);
    for (int k = 0; k < nq;) {
      const int r0 = k * block_m;
      const int i_max = ctx + tmx::imin(r0 + block_m, ql) - 1;
      const int cnt = tmx::tile_ranges(m, ctx + r0, i_max, n, nf, si, sl, last, block_n).count();
      const int reps = tmx::same_count_run(m, k, ctx, ql, nf, si, sl, last, block_m, block_n);  // > 1: sparse rows
      if (m != TMX_MODE_TRIANGLE || (nf - i_max) < last) {
        (ql > 16 ? dense_long : dense_short) += cnt;
      } else {
        sparse_blocks += reps;
        sparse_max = tmx::imax(sparse_max, cnt);
      }
      k += reps;
    }
  }
  const long long dense_tiles = dense_long + dense_short;
  const long long slots = 2ll * num_cus;  // two workgroups are resident per CU
  int chunk;
  if (dense_tiles == 0) {
    chunk = tmx::imax(64, sparse_max);  // nothing to split
  } else if (sparse_blocks == 0) {
    // only dense rows: one piece per workgroup slot of the chip when that is possible.
    // Decode steps (measured, 256 CUs, 32 / 8 heads): 1 x 8192 keys 25 us with 8-tile pieces against 106 us with
    // 128; 4 x 16384: 56 us (32 tiles) against 109 us; 64 x 8192: 389 us (256 tiles) against 402 us.  Prompts
    // attended causally: pieces no shorter than 64 tiles (their partial results cost more than the balance gains:
    // 8192 tokens unsplit 0.90 ms against 1.10 ms in 64-tile pieces) and no longer than 256 (32768 tokens: 17.1 ms
    // against 18.5 ms with 512).
    const long long c = (dense_tiles * units_per_item + slots - 1) / slots;
    if (decode_only) chunk = (int)(c < 8 ? 8 : (c > 512 ? 512 : c));
    else chunk = (int)(c < 64 ? 64 : (c > 256 ? 256 : c));
  } else {
    // sparse and dense rows: one piece of the prompts' dense rows per CU, so that every CU runs one arithmetic-bound
    // piece beside one workgroup of memory-bound sparse rows.  Decode rows in the same step ask for little: pieces
    // of up to 32 tiles, one per workgroup slot (8192-token prompt + 32 decode rows over 4096 keys: 180 us with
    // 32 tiles, 220 us with 72, 384 us with 256).
    long long c = (dense_long * units_per_item + num_cus - 1) / num_cus;
    long long cd = (dense_short * units_per_item + slots - 1) / slots;
    if (cd > 32) cd = 32;
    if (cd > c) c = cd;
    chunk = (int)(c < 8 ? 8 : (c > 512 ? 512 : c));
    chunk = tmx::imax(chunk, sparse_max);  // sparse rows stay whole
  }
  *out = chunk;
  return TMX_OK;
}

}  // extern "C"
