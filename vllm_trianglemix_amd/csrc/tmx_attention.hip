// tmx_attention and tmx_attention_static_ws: argument checks, the choice of kernels for a step, and their launches.
//
// A step is served by up to four kernels, all on the caller's stream:
//   dense_causal_kernel  (tmx_dense.hip)        whole-prompt causal layers: every row attends to all earlier keys
//   decode_kernel        (tmx_decode.hip)       requests with one query row (and the dense rows of short ones)
//   attn_kernel          (tmx_attn_kernel.hip)  everything the block index lists: work-list + band workgroups
//   combine_kernel       (tmx_attn_combine.hip) merges the pieces of split q-blocks
// Replaces the dense eager path of the reference (vllm/attention/ascend_trianglemix.py:88-106) and the kernel
// behind AttentionImpl.forward (vllm/v1/attention/backends/rocm_attn.py:275-426).
#include "tmx_attn.h"

namespace tmx {

// attention kernel, then the merge pass
static int launch(const AttnParams& p, bool bf16, int head_dim, bool paged, bool has_bias, bool fp8, int64_t num_split, uint32_t flags, hipStream_t s) {
  if (p.units_total > 0 && !(flags & TMX_ATTN_ONLY_COMBINE)) {
    const int rc = launch_attention_kernel(p, bf16, head_dim, paged, has_bias, fp8, s);
    if (rc != TMX_OK) return rc;
  }
  if (num_split > 0 && !(flags & TMX_ATTN_SKIP_COMBINE)) return launch_combine_kernel(p, bf16, head_dim, num_split, s);
  return TMX_OK;
}

// The schedule's exchange rate: how many light (unsplit) items keep a workgroup busy as long as one piece
// of a split q-block does.  A piece is chunk_tiles arithmetic-bound tile steps; a light item costs its tile
// steps plus a fixed part (query rows in, output rows out, set-up) worth about six steps when it runs beside
// a piece (measured on MI355X: 64-tile pieces against 3-tile items balance at 6..8 items per piece).
// TMX_SCHED_W overrides it in diagnostic builds (tuning runs).
static int sched_light_per_piece(const tmx_attention_args_t* a, const tmx_index_t* ix) {
#ifdef TMX_STAMPS
  static const char* env = getenv("TMX_SCHED_W");
  if (env != nullptr) return atoi(env);
#endif
  const long long pieces = a->num_partials, light = a->num_items - a->num_partials;
  if (pieces <= 0 || light <= 0) return 0;
  // listed tiles: the count the caller passes (tmx_index_sizes), else the buffer's capacity as an upper bound
  const long long nnz = a->nnz_tiles > 0 ? a->nnz_tiles : ix->cap_nnz;
  const long long light_tiles = nnz - pieces * ix->chunk_tiles;  // lower bound: last pieces are shorter
  const double avg = light_tiles > light ? (double)light_tiles / (double)light : 1.0;
  const double w = (double)ix->chunk_tiles / (6.0 + avg);
  return w > 64.0 ? 64 : (int)(w + 0.5);
}

static bool aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; }

}  // namespace tmx

extern "C" int tmx_attention(const tmx_attention_args_t* a, void* stream) {
  using namespace tmx;
  TMX_CHECK_ARG(a != nullptr, "tmx_attention: args is NULL");
  TMX_CHECK_ARG(a->struct_size == sizeof(tmx_attention_args_t), "tmx_attention: struct_size %u != %zu (ABI mismatch)",
                a->struct_size, sizeof(tmx_attention_args_t));
  TMX_CHECK_SUPPORTED(a->io_dtype == TMX_BF16 || a->io_dtype == TMX_F16,
                      "tmx_attention: q/k/v dtype %d not supported (bf16 and fp16 are)", a->io_dtype);
  TMX_CHECK_SUPPORTED(a->out_dtype == a->io_dtype || a->out_dtype == TMX_F32,
                      "tmx_attention: out dtype %d must equal the input dtype or be fp32", a->out_dtype);
  TMX_CHECK_SUPPORTED(a->head_dim == 64 || a->head_dim == 128, "tmx_attention: head_dim %d not supported (64, 128)",
                      a->head_dim);
  TMX_CHECK_ARG(a->num_heads > 0 && a->num_kv_heads > 0 && a->num_heads % a->num_kv_heads == 0,
                "tmx_attention: num_heads %d must be a positive multiple of num_kv_heads %d", a->num_heads,
                a->num_kv_heads);
  TMX_CHECK_ARG(a->mode >= TMX_MODE_NONE && a->mode <= TMX_MODE_TRIANGLE, "tmx_attention: bad mode %d", a->mode);
  TMX_CHECK_ARG(a->scale > 0.f && a->scale < 3.0e38f, "tmx_attention: scale %g must be positive and finite",
                (double)a->scale);
  TMX_CHECK_ARG(a->batch >= 0 && a->total_q >= 0 && a->num_items >= 0 && a->num_partials >= 0 &&
                    a->num_split_qblocks >= 0,
                "tmx_attention: negative size");
  const tmx_index_t* ix = a->index;
  // (an index without the dense causal kernel's requests may be empty: that kernel still has work)
  if (a->total_q == 0 || (a->num_items == 0 && !(ix != nullptr && ix->dense_rows_from != 0))) return TMX_OK;
  TMX_CHECK_ARG(ix != nullptr, "tmx_attention: index is NULL");
  const int G = a->num_heads / a->num_kv_heads;
  TMX_CHECK_ARG(ix->block_n == kBlockN && ix->block_m == block_m_for_group(G),
                "tmx_attention: index built with block_m %d, block_n %d; this head grouping needs %d, %d",
                ix->block_m, ix->block_n, block_m_for_group(G), kBlockN);
  TMX_CHECK_ARG(ix->mode == a->mode && ix->si == a->si && ix->sl == a->sl && ix->last == a->last &&
                    ix->batch == a->batch,
                "tmx_attention: mask parameters differ from the ones the index was built with");
  TMX_CHECK_SUPPORTED(a->num_items < (1ll << 30), "tmx_attention: more than 2^30 work items");
  TMX_CHECK_ARG(a->num_items <= ix->cap_items && a->num_split_qblocks <= ix->cap_split,
                "tmx_attention: num_items / num_split_qblocks exceed the index capacities");
  // (every partial slot belongs to a piece, and every piece is an item)
  TMX_CHECK_ARG(a->num_partials <= a->num_items, "tmx_attention: num_partials %lld exceeds num_items %lld",
                (long long)a->num_partials, (long long)a->num_items);
  TMX_CHECK_ARG(a->q && a->out && a->k && a->v && a->cu_seqlens_q && a->seq_lens, "tmx_attention: NULL tensor");
  TMX_CHECK_ARG(ix->items && ix->qb_meta && ix->col_idx && ix->part_ptr, "tmx_attention: NULL index buffer");
  TMX_CHECK_ARG(a->num_partials == 0 || (a->partial_o && a->partial_ml && a->num_split_qblocks > 0 && ix->split_qb),
                "tmx_attention: split q-blocks need partial_o / partial_ml / split_qb");
  TMX_CHECK_ARG(aligned16(a->q) && aligned16(a->k) && aligned16(a->v) && aligned16(a->out) &&
                    aligned16(a->partial_o),
                "tmx_attention: q / k / v / out / partial_o must be 16-byte aligned");
  TMX_CHECK_ARG(a->q_stride_t % 8 == 0 && a->q_stride_h % 8 == 0 && a->out_stride_t % 8 == 0 &&
                    a->out_stride_h % 8 == 0 && a->kv_stride_t % 8 == 0 && a->kv_stride_h % 8 == 0 &&
                    a->kv_stride_page % 8 == 0,
                "tmx_attention: strides must be multiples of 8 elements");
  const bool paged = a->block_table != nullptr;
  if (paged)
    TMX_CHECK_SUPPORTED(a->page_size >= 16 && (a->page_size & (a->page_size - 1)) == 0,
                        "tmx_attention: page_size %d not supported (powers of two >= 16 are)", a->page_size);
  else TMX_CHECK_ARG(a->cu_seqlens_k != nullptr, "tmx_attention: contiguous K/V needs cu_seqlens_k");
  const bool has_bias = a->bias != nullptr;
  const bool fp8 = a->kv_dtype == TMX_FP8_E4M3;
  TMX_CHECK_SUPPORTED(a->kv_dtype == 0 || a->kv_dtype == a->io_dtype || fp8,
                      "tmx_attention: kv_dtype %d not supported (0 / the q dtype, or TMX_FP8_E4M3)", a->kv_dtype);
  if (fp8) {
    TMX_CHECK_SUPPORTED(paged && !has_bias && a->head_dim == 128,
                        "tmx_attention: an fp8 K/V cache needs the paged layout, head_dim 128 and no bias "
                        "(other cases: tmx_gather_dequant_kv_fp8 ahead of the call)");
    TMX_CHECK_ARG(a->k_scale > 0.f && a->v_scale > 0.f && a->k_scale < 3.0e38f && a->v_scale < 3.0e38f,
                  "tmx_attention: k_scale / v_scale must be positive and finite");
    TMX_CHECK_ARG(a->kv_stride_t % 16 == 0 && a->kv_stride_h % 16 == 0 && a->kv_stride_page % 16 == 0,
                  "tmx_attention: fp8 cache strides must be multiples of 16 bytes");
  }
  const bool two_src = a->k_new != nullptr || a->v_new != nullptr;
  if (two_src) {
    TMX_CHECK_ARG(a->k_new != nullptr && a->v_new != nullptr, "tmx_attention: k_new and v_new go together");
    TMX_CHECK_SUPPORTED(paged && !fp8, "tmx_attention: k_new / v_new need a 16-bit paged cache (an fp8 cache holds other "
                                       "values than the step's rows)");
    TMX_CHECK_ARG(aligned16(a->k_new) && aligned16(a->v_new) && a->new_stride_t % 8 == 0 && a->new_stride_h % 8 == 0 &&
                      a->new_stride_t >= 0 && a->new_stride_h >= 0 && a->new_stride_t < (1ll << 24) &&
                      a->new_stride_h <= 0x7FFFFFFF,
                  "tmx_attention: k_new / v_new must be 16-byte aligned with strides that are multiples of 8 elements");
  }
  if (a->cache_slots != nullptr) {
    TMX_CHECK_ARG(two_src, "tmx_attention: cache_slots needs k_new / v_new (the rows to store)");
    TMX_CHECK_SUPPORTED((long long)a->total_q * a->num_kv_heads * (a->head_dim / 8) < (1ll << 31),
                        "tmx_attention: too many rows for the fused cache write");
  }
  if (has_bias)
    TMX_CHECK_ARG((a->bias_dtype == TMX_F32 || a->bias_dtype == TMX_BF16 || a->bias_dtype == TMX_F16) &&
                      a->bias_rows > 0 && a->bias_cols > 0 && a->bias_stride >= a->bias_cols && a->bias_stride_b >= 0 &&
                      a->bias_stride_h >= 0,
                  "tmx_attention: bad bias description");

  AttnParams p;
  p.q = a->q;
  p.out = a->out;
  p.k = a->k;
  p.v = a->v;
  p.bias = a->bias;
  p.block_table = a->block_table;
  p.cu_k = a->cu_seqlens_k;
  p.cu_q = a->cu_seqlens_q;
  p.seq_lens = a->seq_lens;
  p.col_idx = ix->col_idx;
  p.qb_meta = ix->qb_meta;
  p.items = ix->items;
  p.part_ptr = ix->part_ptr;
  p.split_qb = ix->split_qb;
  p.partial_o = a->partial_o;
  p.partial_ml = a->partial_ml;
  p.stamps = nullptr;
#ifdef TMX_STAMPS
  // diagnostic build: the stamps go to the second half of a (max, sum) buffer of at least 2^21 floats, which only
  // tools/stamps.py provides -- it says so with TMX_STAMPS_BUFFER=1; any other caller's buffer is left alone
  if (const char* env = getenv("TMX_STAMPS_BUFFER"))
    if (env[0] == '1' && a->partial_ml != nullptr) p.stamps = reinterpret_cast<unsigned long long*>(a->partial_ml + (1 << 19));
#endif
  const int64_t lim = 0x7FFFFFFF;
  TMX_CHECK_SUPPORTED(a->q_stride_t <= lim && a->q_stride_h <= lim && a->out_stride_t <= lim &&
                          a->out_stride_h <= lim && a->kv_stride_page <= lim && a->kv_stride_t <= lim &&
                          a->kv_stride_h <= lim && a->block_table_stride <= lim && a->q_stride_t >= 0 &&
                          a->q_stride_h >= 0 && a->out_stride_t >= 0 && a->out_stride_h >= 0 &&
                          a->kv_stride_page >= 0 && a->kv_stride_t >= 0 && a->kv_stride_h >= 0 &&
                          a->block_table_stride >= 0,
                      "tmx_attention: strides must be non-negative and below 2^31 elements");
  // per-lane byte offsets inside a tile (32 rows) / a wave's query block (16 rows) are 32-bit
  TMX_CHECK_SUPPORTED(a->kv_stride_t < (1ll << 24) && a->q_stride_t < (1ll << 24),
                      "tmx_attention: row strides of q and k/v must be below 2^24 elements");
  p.q_stride_t = (int)a->q_stride_t;
  p.q_stride_h = (int)a->q_stride_h;
  p.out_stride_t = (int)a->out_stride_t;
  p.out_stride_h = (int)a->out_stride_h;
  p.kv_stride_page = (int)a->kv_stride_page;
  p.kv_stride_t = (int)a->kv_stride_t;
  p.kv_stride_h = (int)a->kv_stride_h;
  p.block_table_stride = (int)a->block_table_stride;
  p.tile_pages = nullptr;
  p.full_lens = ix->full_lens;
  p.k_new = a->k_new;
  p.v_new = a->v_new;
  p.new_stride_t = (int)a->new_stride_t;
  p.new_stride_h = (int)a->new_stride_h;
  p.cache_slots = reinterpret_cast<const long long*>(a->cache_slots);
  p.total_q = a->total_q;
  if (paged && ix->tile_pages != nullptr && ix->pages_block_table == a->block_table &&
      ix->pages_page_size == a->page_size && ix->pages_block_table_stride == a->block_table_stride)
    p.tile_pages = ix->tile_pages;
  p.bias_stride = a->bias_stride;
  p.bias_stride_b = a->bias_stride_b;
  p.bias_stride_h = a->bias_stride_h;
  p.page_size = paged ? a->page_size : 1;
  p.page_shift = paged ? __builtin_ctz(a->page_size) : 0;
  p.H = a->num_heads;
  p.Hkv = a->num_kv_heads;
  p.G = G;
  p.Gw = heads_per_wg(G);
  p.gw_shift = __builtin_ctz(p.Gw);
  p.block_m = ix->block_m;
  p.mode = a->mode;
  p.si = a->si;
  p.sl = a->sl;
  p.last = a->last;
  p.tri_min_len = ix->tri_min_len;
  p.bias_rows = a->bias_rows;
  p.bias_cols = a->bias_cols;
  p.bias_dtype = a->bias_dtype;
  p.out_dtype = a->out_dtype;
  p.scale_log2 = a->scale * kLog2e;
  p.k_scale = fp8 ? a->k_scale : 1.f;
  p.v_scale = fp8 ? a->v_scale : 1.f;

  p.Y = a->num_kv_heads * (G / p.Gw);
  p.num_items = (int)a->num_items;
  p.issue_hi = (a->flags & TMX_ATTN_SHORT_QUERIES) ? 1 : 0;
  p.num_heavy = (int)(a->num_partials < a->num_items ? a->num_partials : a->num_items);
  p.sched_w = sched_light_per_piece(a, ix);
  p.units_total = (long long)a->num_items * p.Y;
  // The band path (sparse q-blocks in canonical position on workgroups of their own) covers the default
  // pattern's shape: a window of 17..32 keys below the row, at most 16 sink columns, four query heads per
  // workgroup, 32-row q-blocks that the index does not split.
  p.num_qblocks = (int)(a->num_items - a->num_partials + a->num_split_qblocks);
  p.band_on = (a->mode == TMX_MODE_TRIANGLE && !has_bias && p.Gw == 4 && p.block_m == 32 && a->sl >= 17 && a->sl <= 32 &&
               a->si >= -1 && a->si <= 15 && a->last >= 0 && ix->chunk_tiles >= 3 && p.num_qblocks >= 8 &&
               !(a->flags & (TMX_ATTN_SHORT_QUERIES | TMX_ATTN_NO_BAND)))
                  ? 1
                  : 0;
  if (p.band_on) p.sched_w = 0;
  // Dense causal kernel: long requests of a causal launch (the layers outside the triangle set) go to the
  // kernel built for the matrix cores; this launch keeps the requests with few query rows.
  p.dense_on = (a->mode == TMX_MODE_CAUSAL && !has_bias && G % 4 == 0 && a->head_dim == 128 && a->batch <= kDenseMaxBatch &&
                a->total_q >= kDenseMinRows && !(a->flags & (TMX_ATTN_SHORT_QUERIES | TMX_ATTN_NO_DENSE | TMX_ATTN_ONLY_COMBINE)))
                   ? 1
                   : 0;
  static_assert(kDenseMinRows == TMX_DENSE_MIN_ROWS, "the public constant is the kernel's");
  // an index without the long requests of a causal launch: the dense kernel has to be there for them
  if (ix->dense_rows_from != 0) {
    TMX_CHECK_ARG(ix->dense_rows_from == kDenseMinRows && a->mode == TMX_MODE_CAUSAL,
                  "tmx_attention: index->dense_rows_from is 0 or %d, the latter with TMX_MODE_CAUSAL only", kDenseMinRows);
    // (TMX_ATTN_SHORT_QUERIES: the caller says that no request has more than 16 rows -- the index left none out)
    TMX_CHECK_SUPPORTED(p.dense_on || a->total_q < kDenseMinRows || (a->flags & TMX_ATTN_SHORT_QUERIES),
                        "tmx_attention: the index leaves requests of %d query rows or more to the dense causal kernel, which "
                        "this call cannot run (bias, head size other than 128, query heads per K/V head not a multiple of 4, "
                        "more than %d requests, or a flag that keeps it off)", kDenseMinRows, kDenseMaxBatch);
  }
  p.band_rl = 0;
  p.band_tail0 = 0;
  p.band_rls = 0;
  p.batch = a->batch;
  p.chunk_tiles = ix->chunk_tiles;
  p.walk_shift = 0;
  p.n_generic = 0;
  p.exp_mode = 0;
  p.xcd_generic = p.xcd_band = 0;
  hipStream_t s = (hipStream_t)stream;
  // query heads per decode workgroup (all of them share the K/V loads), its units per item, and whether the step
  // has a decode launch (below)
  const int hw_d = G % 8 == 0 ? 8 : 4;
  const int Yd = a->num_kv_heads * (G / hw_d > 0 ? G / hw_d : 1);
  const bool pure = a->total_q == a->batch;
  const bool decode_will_run = !has_bias && G % 4 == 0 && a->head_dim == 128 && (!paged || a->page_size >= 16) &&
                               (a->mode != TMX_MODE_TRIANGLE || a->last >= 2) &&
                               !(a->flags & (TMX_ATTN_NO_DECODE | TMX_ATTN_ONLY_COMBINE)) &&
                               (pure || (a->batch >= 2 && (long long)a->num_items * Yd <= kDecodeMaxMixedGrid));
  // The index is still to be built (TMX_ATTN_BUILD_INDEX).  Inside the attention launch when that launch has band
  // workgroups to run meanwhile, nothing ahead of it reads the index (no dense causal or decode launch in this
  // call) and one workgroup can build it; else as a launch of its own, first.
  p.ix_build = 0;
  p.ix_flag = nullptr;
  p.ix_epoch = 0;
  p.ix = IndexParams{};
  if (a->flags & TMX_ATTN_BUILD_INDEX) {
    // (batches of a few requests: the band workgroups, which cannot read the index yet, walk the batch description
    // to place a q-block)
    const bool in_launch = p.band_on && !p.dense_on && !decode_will_run && ix->cap_qblocks <= 1024 && p.units_total > 0 &&
                           a->batch <= 16 &&
                           attention_kernel_builds_index(a->head_dim, has_bias, fp8) &&
                           !(a->flags & (TMX_ATTN_ONLY_COMBINE | TMX_ATTN_SHORT_QUERIES));
    if (in_launch) {
      const int rc = index_params_from(ix, a->cu_seqlens_q, a->seq_lens, &p.ix);
      if (rc != TMX_OK) return rc;
      // the epoch goes to an 8-byte slot of the build's work area that the single-workgroup build leaves alone (the
      // per-block totals of the blocked form: at least 8 ints); epochs never repeat within a process and carry a
      // tag in their upper half, so that no earlier content of the buffer reads as "built"
      static std::atomic<unsigned long long> epoch_counter{0};
      p.ix_flag = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(p.ix.blk_tot) + 7) & ~(uintptr_t)7);
      p.ix_epoch = (0x544D5849ull << 32) | (epoch_counter.fetch_add(1, std::memory_order_relaxed) + 1);
      p.ix_build = 1;
    } else {
      const int rc = tmx_build_block_index(const_cast<tmx_index_t*>(ix), a->cu_seqlens_q, a->seq_lens, nullptr, nullptr, stream);
      if (rc != TMX_OK) return rc;
    }
  }
  if (p.dense_on) {
    DenseParams d;
    d.q = a->q;
    d.out = a->out;
    d.k = a->k;
    d.v = a->v;
    d.block_table = a->block_table;
    d.cu_k = a->cu_seqlens_k;
    d.cu_q = a->cu_seqlens_q;
    d.seq_lens = a->seq_lens;
    d.k_new = a->k_new;
    d.v_new = a->v_new;
    d.new_stride_t = p.new_stride_t;
    d.new_stride_h = p.new_stride_h;
    d.cache_slots = p.cache_slots;  // the dense kernel, launched first, stores the step's rows
    p.cache_slots = nullptr;
    d.q_stride_t = p.q_stride_t;
    d.q_stride_h = p.q_stride_h;
    d.out_stride_t = p.out_stride_t;
    d.out_stride_h = p.out_stride_h;
    d.kv_stride_page = p.kv_stride_page;
    d.kv_stride_t = p.kv_stride_t;
    d.kv_stride_h = p.kv_stride_h;
    d.block_table_stride = p.block_table_stride;
    d.page_size = p.page_size;
    d.page_shift = p.page_shift;
    d.H = p.H;
    d.Hkv = p.Hkv;
    d.G = G;
    d.Y = a->num_kv_heads * (G / 4);
    d.batch = a->batch;
    d.total_q = a->total_q;
    d.out_dtype = a->out_dtype;
    d.scale_log2 = p.scale_log2;
    d.kv_fp8 = fp8 ? 1 : 0;
    d.k_scale = p.k_scale;
    d.v_scale = p.v_scale;
    TMX_CHECK_ARG(a->kv_stage == nullptr || (aligned16(a->kv_stage) && a->kv_stage_rows >= 0),
                  "tmx_attention: kv_stage must be 16-byte aligned and kv_stage_rows non-negative");
    d.stage = a->kv_stage;
    d.stage_rows = a->kv_stage != nullptr ? a->kv_stage_rows : 0;
    const int rc = launch_dense_causal(d, a->io_dtype == TMX_BF16, paged, s);
    if (rc != TMX_OK) return rc;
  }
  if (a->num_items == 0) return TMX_OK;  // (an index that left every request to the dense kernel)
  // Decode kernel: one-row requests whose row is dense (every decode step of the default pattern, every row of
  // a causal layer) stream their K/V straight into registers there.  A step of one-row requests only is all
  // the kernel's; in a mixed step its grid also covers the other requests' items (workgroups that return at
  // once), which pays only while that grid is small.
  p.decode_on = 0;
  uint32_t flags = a->flags;
  {
    if (decode_will_run) p.decode_on = 1;
    if (p.decode_on) {
      DecodeParams d;
      d.q = a->q;
      d.out = a->out;
      d.k = a->k;
      d.v = a->v;
      d.block_table = a->block_table;
      d.cu_k = a->cu_seqlens_k;
      d.cu_q = a->cu_seqlens_q;
      d.full_lens = p.full_lens;
      d.items = p.items;
      d.col_idx = p.col_idx;
      d.partial_o = p.partial_o;
      d.partial_ml = p.partial_ml;
      d.k_new = a->k_new;
      d.v_new = a->v_new;
      d.cache_slots = p.cache_slots;  // launched first, the decode kernel stores the step's rows
      p.cache_slots = nullptr;
      d.new_stride_t = p.new_stride_t;
      d.new_stride_h = p.new_stride_h;
      d.total_q = a->total_q;
      d.q_stride_t = p.q_stride_t;
      d.q_stride_h = p.q_stride_h;
      d.out_stride_t = p.out_stride_t;
      d.out_stride_h = p.out_stride_h;
      d.kv_stride_page = p.kv_stride_page;
      d.kv_stride_t = p.kv_stride_t;
      d.kv_stride_h = p.kv_stride_h;
      d.block_table_stride = p.block_table_stride;
      d.page_size = p.page_size;
      d.page_shift = p.page_shift;
      d.H = p.H;
      d.Hkv = p.Hkv;
      d.G = G;
      d.hw = hw_d;
      d.valu_sum = (a->flags & TMX_ATTN_DECODE_VALU) ? 1 : 0;
      d.Y = Yd;
      d.block_m = p.block_m;
      d.num_items = p.num_items;
      d.mode = p.mode;
      d.last = p.last;
      d.tri_min_len = p.tri_min_len;
      d.out_dtype = a->out_dtype;
      d.scale_log2 = p.scale_log2;
      d.kv_fp8 = fp8 ? 1 : 0;
      d.k_scale = p.k_scale;
      d.v_scale = p.v_scale;
      const int rc = launch_decode(d, a->io_dtype == TMX_BF16, paged, s);
      if (rc != TMX_OK) return rc;
      // every request has one row and every such row is dense: nothing is left for the work-list kernel
      const bool all_dense_rows = a->mode != TMX_MODE_TRIANGLE || (a->last >= 2 && p.full_lens == nullptr);
      if ((a->flags & TMX_ATTN_DECODE_ONLY) && pure && all_dense_rows) flags |= TMX_ATTN_ONLY_COMBINE;
    }
  }
  return launch(p, a->io_dtype == TMX_BF16, a->head_dim, paged, has_bias, fp8, a->num_split_qblocks, flags, s);
}

// ---------------------------------------------------------------------------------------------------
// The same call for callers that may not allocate: the work area of the split q-blocks lives in the
// library's own device memory.  Every device the library runs on holds one copy of the array (at its own
// address) until the process ends.
namespace tmx {

constexpr size_t kStaticWsBytes = (size_t)TMX_STATIC_WORKSPACE_BYTES;
static __device__ __align__(256) unsigned char static_ws[kStaticWsBytes];

// Device address of this device's copy of static_ws.  The array's name is only a handle on the host:
// the address comes from hipGetSymbolAddress, asked once per device with that device current.
static int static_ws_base(unsigned char** out) {
  static std::atomic<unsigned char*> base[kMaxDevices];
  int dev = 0;
  TMX_CHECK_HIP(hipGetDevice(&dev));
  TMX_CHECK_SUPPORTED(dev >= 0 && dev < kMaxDevices, "tmx_attention_static_ws: device ordinal %d out of range", dev);
  unsigned char* ptr = base[dev].load(std::memory_order_acquire);
  if (ptr == nullptr) {
    TMX_CHECK_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(static_ws)));
    TMX_CHECK_ARG(ptr != nullptr, "tmx_attention_static_ws: no device address for the static work area");
    base[dev].store(ptr, std::memory_order_release);  // every thread gets the same answer for a device
  }
  *out = ptr;
  return TMX_OK;
}

}  // namespace tmx

extern "C" size_t tmx_static_workspace_bytes(void) { return tmx::kStaticWsBytes; }

extern "C" int tmx_static_workspace_warmup(void) {
  unsigned char* base = nullptr;
  return tmx::static_ws_base(&base);
}

extern "C" int tmx_attention_static_ws(const tmx_attention_args_t* a, void* stream) {
  using namespace tmx;
  TMX_CHECK_ARG(a != nullptr, "tmx_attention_static_ws: args is NULL");
  TMX_CHECK_ARG(a->struct_size == sizeof(tmx_attention_args_t),
                "tmx_attention_static_ws: struct_size %u != %zu (ABI mismatch)", a->struct_size,
                sizeof(tmx_attention_args_t));
  tmx_attention_args_t b = *a;
  b.partial_o = nullptr;
  b.partial_ml = nullptr;
  if (a->num_partials > 0) {
    TMX_CHECK_ARG(a->index != nullptr && a->num_heads > 0 && a->head_dim > 0 && a->index->block_m > 0,
                  "tmx_attention_static_ws: bad sizes");
    // bounds ahead of the products below, so that none of them can wrap (a slot is a piece, a piece an item)
    TMX_CHECK_ARG(a->num_partials <= a->num_items && a->num_items < (1ll << 30) && a->num_heads < (1 << 16) &&
                      a->head_dim <= 1024 && a->index->block_m <= 1024,
                  "tmx_attention_static_ws: num_partials %lld / num_heads %d / head_dim %d out of range",
                  (long long)a->num_partials, a->num_heads, a->head_dim);
    // partial_o [P][H][block_m][D] floats, then partial_ml [P][H][block_m][2] floats on the next 256-byte boundary
    const unsigned long long rows = (unsigned long long)a->num_partials * (unsigned long long)a->num_heads *
                                    (unsigned long long)a->index->block_m;
    const unsigned long long o_bytes = (rows * (unsigned long long)a->head_dim * 4ull + 255ull) & ~255ull;
    const unsigned long long need = o_bytes + rows * 8ull;
    if (need > kStaticWsBytes) {
      set_error("tmx_attention_static_ws: the split q-blocks of this step need %llu bytes of work area, the library "
                "holds %zu (use tmx_attention with a workspace of your own, or longer pieces)", need, kStaticWsBytes);
      return TMX_ERR_CAPACITY;
    }
    unsigned char* base = nullptr;
    const int rc = static_ws_base(&base);
    if (rc != TMX_OK) return rc;
    b.partial_o = reinterpret_cast<float*>(base);
    b.partial_ml = reinterpret_cast<float*>(base + o_bytes);
  }
  return tmx_attention(&b, stream);
}