// TORCH_LIBRARY binding of libtrianglemix_hip.so, in the style the reference registers its own
// ROCm ops (csrc/rocm/torch_bindings.cpp:14-57, csrc/rocm/ops.h:17-26, csrc/core/registration.h:13-27):
// schema strings with `Tensor!` for mutated outputs, implementations on the CUDA dispatch key (HIP
// tensors dispatch there), outputs preallocated by the Python caller as vllm/_custom_ops.py:153-196
// does.  Tensors are unpacked to the pointers, strides and sizes of the C ABI; the work is enqueued
// on the current stream of the tensors' device; nothing synchronises with the host.
//
// Built as _trianglemix_C.so (plain C++, no device code) and loaded with
// torch.ops.load_library(); the ops then live under torch.ops._trianglemix_C.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <optional>
#include <vector>

#include "trianglemix_hip.h"

namespace {

using at::Tensor;

int dtype_code(const Tensor& t) {
  switch (t.scalar_type()) {
    case at::kFloat: return TMX_F32;
    case at::kBFloat16: return TMX_BF16;
    case at::kHalf: return TMX_F16;
    default: TORCH_CHECK(false, "trianglemix: unsupported dtype ", t.scalar_type());
  }
}

void check(int code) { TORCH_CHECK(code == TMX_OK, "trianglemix: ", tmx_last_error()); }

// PyTorch-ROCm presents HIP devices under the "cuda" device type; these are its guard and stream types for them.
using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;
void* stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

int32_t* i32(const Tensor& t) {
  TORCH_CHECK(t.scalar_type() == at::kInt && t.is_contiguous(), "trianglemix: index tensors must be contiguous int32");
  return t.data_ptr<int32_t>();
}

int32_t* i32_opt(const std::optional<Tensor>& t) { return t.has_value() ? i32(*t) : nullptr; }

// The eight index buffers travel as one list: row_ptr, col_idx, qb_meta, part_ptr, items, split_qb, totals, scratch.
void fill_index(tmx_index_t& ix, const std::vector<Tensor>& b, const std::optional<Tensor>& tile_pages,
                const std::optional<Tensor>& pages_block_table, int64_t page_size, int64_t si, int64_t sl,
                int64_t last, int64_t block_m, int64_t block_n, int64_t mode, int64_t tri_min_len,
                int64_t chunk_tiles, int64_t batch, const std::optional<Tensor>& full_lens = std::nullopt,
                const std::optional<Tensor>& full_lens_cpu = std::nullopt, int64_t dense_rows_from = 0) {
  TORCH_CHECK(b.size() == 8, "trianglemix: the index is a list of 8 tensors");
  ix = tmx_index_t{};
  ix.row_ptr = i32(b[0]);
  ix.col_idx = i32(b[1]);
  ix.qb_meta = i32(b[2]);
  ix.part_ptr = i32(b[3]);
  ix.items = i32(b[4]);
  ix.split_qb = i32(b[5]);
  ix.totals = i32(b[6]);
  ix.scratch = i32(b[7]);
  ix.cap_qblocks = b[0].numel() - 1;
  TORCH_CHECK(b[7].numel() >= TMX_INDEX_SCRATCH_INTS(batch, ix.cap_qblocks),
              "trianglemix: the index's scratch tensor has ", b[7].numel(), " entries, ",
              TMX_INDEX_SCRATCH_INTS(batch, ix.cap_qblocks), " are needed (batch + 1 + 2 (R + 1) + 8 (R / 1024 + 1))");
  ix.cap_nnz = b[1].numel();
  ix.cap_items = b[4].numel() / 8;
  ix.cap_split = b[5].numel();
  ix.block_m = (int32_t)block_m;
  ix.block_n = (int32_t)block_n;
  ix.mode = (int32_t)mode;
  ix.chunk_tiles = (int32_t)chunk_tiles;
  ix.si = (int32_t)si;
  ix.sl = (int32_t)sl;
  ix.last = (int32_t)last;
  ix.batch = (int32_t)batch;
  ix.tri_min_len = (int32_t)tri_min_len;
  if (tile_pages.has_value() && pages_block_table.has_value()) {
    ix.tile_pages = i32(*tile_pages);
    TORCH_CHECK(pages_block_table->scalar_type() == at::kInt && pages_block_table->stride(1) == 1,
                "trianglemix: block_table must be int32 with contiguous rows");
    ix.pages_block_table = pages_block_table->data_ptr<int32_t>();
    ix.pages_block_table_stride = pages_block_table->stride(0);
    ix.pages_page_size = (int32_t)page_size;
  }
  ix.full_lens = i32_opt(full_lens);
  ix.full_lens_host = i32_opt(full_lens_cpu);
  ix.dense_rows_from = (int32_t)dense_rows_from;
}

// Exact buffer sizes from the host copies of the batch: [R, nnz, items, partials, split q-blocks, max tiles].
std::vector<int64_t> index_sizes(const Tensor& cu_seqlens_q_cpu, const Tensor& seq_lens_cpu, int64_t si, int64_t sl,
                                 int64_t last, int64_t block_m, int64_t block_n, int64_t mode, int64_t tri_min_len,
                                 int64_t chunk_tiles, const std::optional<Tensor>& full_lens_cpu, int64_t dense_rows_from) {
  TORCH_CHECK(cu_seqlens_q_cpu.is_cpu() && seq_lens_cpu.is_cpu() && (!full_lens_cpu.has_value() || full_lens_cpu->is_cpu()),
              "trianglemix: host tensors expected");
  tmx_index_sizes_t sz;
  check(tmx_index_sizes_without_dense(i32(cu_seqlens_q_cpu), i32(seq_lens_cpu), i32_opt(full_lens_cpu), (int)seq_lens_cpu.numel(),
                                      (int)si, (int)sl, (int)last, (int)block_m, (int)block_n, (int)mode, (int)tri_min_len,
                                      (int)chunk_tiles, (int)dense_rows_from, &sz));
  return {sz.num_qblocks, sz.nnz_tiles, sz.num_items, sz.num_partials, sz.num_split_qblocks, sz.max_tiles_per_qblock};
}

// Piece length advice for a batch and a head grouping (host only).
int64_t suggest_chunk_tiles(const Tensor& cu_seqlens_q_cpu, const Tensor& seq_lens_cpu, int64_t si, int64_t sl,
                            int64_t last, int64_t block_m, int64_t block_n, int64_t mode, int64_t tri_min_len,
                            int64_t units_per_item, int64_t num_cus, const std::optional<Tensor>& full_lens_cpu) {
  TORCH_CHECK(cu_seqlens_q_cpu.is_cpu() && seq_lens_cpu.is_cpu(), "trianglemix: host tensors expected");
  int out = 0;
  check(tmx_suggest_chunk_tiles(i32(cu_seqlens_q_cpu), i32(seq_lens_cpu), i32_opt(full_lens_cpu),
                                (int)seq_lens_cpu.numel(), (int)si, (int)sl, (int)last, (int)block_m, (int)block_n,
                                (int)mode, (int)tri_min_len, (int)units_per_item, (int)num_cus, &out));
  return out;
}

void build_block_index(std::vector<Tensor> index, const std::optional<Tensor>& tile_pages, const Tensor& cu_seqlens_q,
                       const Tensor& seq_lens, const Tensor& cu_seqlens_q_cpu, const Tensor& seq_lens_cpu,
                       const std::optional<Tensor>& block_table, int64_t page_size, int64_t si, int64_t sl,
                       int64_t last, int64_t block_m, int64_t block_n, int64_t mode, int64_t tri_min_len,
                       int64_t chunk_tiles, const std::optional<Tensor>& full_lens,
                       const std::optional<Tensor>& full_lens_cpu, int64_t dense_rows_from) {
  const DeviceGuard guard(cu_seqlens_q.device());
  tmx_index_t ix;
  fill_index(ix, index, tile_pages, block_table, page_size, si, sl, last, block_m, block_n, mode, tri_min_len,
             chunk_tiles, seq_lens.numel(), full_lens, full_lens_cpu, dense_rows_from);
  check(tmx_build_block_index(&ix, i32(cu_seqlens_q), i32(seq_lens), i32(cu_seqlens_q_cpu), i32(seq_lens_cpu),
                              stream_of(cu_seqlens_q)));
}

// Every tensor that reaches the library as a device pointer must live on the GPU of `ref`.
void on_gpu(const Tensor& t, const Tensor& ref, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.device() == ref.device(), "trianglemix: ", name, " must be on the HIP device of the query");
}
void on_gpu(const std::optional<Tensor>& t, const Tensor& ref, const char* name) {
  if (t.has_value()) on_gpu(*t, ref, name);
}

void attention(Tensor& out, const Tensor& query, const Tensor& key_cache, const Tensor& value_cache,
               const std::optional<Tensor>& block_table, const Tensor& cu_seqlens_q, const Tensor& seq_lens,
               const std::optional<Tensor>& cu_seqlens_k, std::vector<Tensor> index,
               const std::optional<Tensor>& tile_pages, const std::optional<Tensor>& partial_o,
               const std::optional<Tensor>& partial_ml, const std::optional<Tensor>& bias, double scale, int64_t si,
               int64_t sl, int64_t last, int64_t mode, int64_t tri_min_len, int64_t chunk_tiles, int64_t num_items,
               int64_t num_partials, int64_t num_split_qblocks, const std::optional<Tensor>& full_lens,
               int64_t flags, int64_t nnz_tiles, const std::optional<Tensor>& k_new, const std::optional<Tensor>& v_new,
               const std::optional<Tensor>& cache_slots, double k_scale, double v_scale, int64_t index_block_m,
               int64_t index_block_n, const std::optional<Tensor>& kv_stage, int64_t index_dense_rows_from) {
  TORCH_CHECK(query.is_cuda(), "trianglemix: query must be on a HIP device (there is no CPU path)");
  const DeviceGuard guard(query.device());
  TORCH_CHECK(query.dim() == 3 && out.dim() == 3 && query.sizes() == out.sizes(), "trianglemix: query / out must both be [T, H, D]");
  const bool paged = block_table.has_value();
  TORCH_CHECK(key_cache.dim() == (paged ? 4 : 3), "trianglemix: K/V must be [blocks, page, Hkv, D] (paged) or [rows, Hkv, D]");
  on_gpu(out, query, "out");
  on_gpu(key_cache, query, "key_cache");
  on_gpu(value_cache, query, "value_cache");
  on_gpu(cu_seqlens_q, query, "cu_seqlens_q");
  on_gpu(seq_lens, query, "seq_lens");
  on_gpu(block_table, query, "block_table");
  on_gpu(cu_seqlens_k, query, "cu_seqlens_k");
  on_gpu(tile_pages, query, "tile_pages");
  on_gpu(partial_o, query, "partial_o");
  on_gpu(partial_ml, query, "partial_ml");
  on_gpu(bias, query, "bias");
  on_gpu(full_lens, query, "full_lens");
  on_gpu(k_new, query, "k_new");
  on_gpu(v_new, query, "v_new");
  on_gpu(cache_slots, query, "cache_slots");
  for (const Tensor& t : index) on_gpu(t, query, "index");
  const bool fp8 = key_cache.element_size() == 1;  // uint8 / float8_e4m3fn: an fp8 cache read in place
  TORCH_CHECK(fp8 || (key_cache.scalar_type() == query.scalar_type() && value_cache.scalar_type() == query.scalar_type()),
              "trianglemix: q, k, v must share a dtype of bfloat16 or float16 (or K/V be an fp8 cache)");
  TORCH_CHECK(!fp8 || (paged && !bias.has_value() && query.size(2) == 128 && value_cache.element_size() == 1),
              "trianglemix: an fp8 K/V cache is read in place only in the paged layout, with head size 128 and no bias");
  TORCH_CHECK(key_cache.sizes() == value_cache.sizes() && key_cache.strides() == value_cache.strides(),
              "trianglemix: k and v must have the same shape and strides");
  TORCH_CHECK(query.stride(2) == 1 && out.stride(2) == 1 && key_cache.stride(-1) == 1,
              "trianglemix: the head dimension must be contiguous");
  TORCH_CHECK(key_cache.size(-1) == query.size(2), "trianglemix: k/v head size differs from q");
  tmx_index_t ix;
  const int64_t hkv = key_cache.size(paged ? 2 : 1);
  // the geometry the index was built with (0: the kernel's own; the library rejects any other)
  const int64_t block_m = index_block_m > 0 ? index_block_m : tmx_attention_block_m((int)query.size(1), (int)hkv);
  fill_index(ix, index, tile_pages, block_table, paged ? key_cache.size(1) : 0, si, sl, last, block_m,
             index_block_n > 0 ? index_block_n : tmx_attention_block_n(), mode, tri_min_len, chunk_tiles, seq_lens.numel(),
             full_lens, std::nullopt, index_dense_rows_from);
  tmx_attention_args_t a{};
  a.struct_size = sizeof(a);
  a.io_dtype = dtype_code(query);
  a.out_dtype = dtype_code(out);
  a.mode = (int32_t)mode;
  a.si = (int32_t)si;
  a.sl = (int32_t)sl;
  a.last = (int32_t)last;
  a.scale = (float)scale;
  a.batch = (int32_t)seq_lens.numel();
  a.num_heads = (int32_t)query.size(1);
  a.num_kv_heads = (int32_t)hkv;
  a.head_dim = (int32_t)query.size(2);
  a.total_q = (int32_t)query.size(0);
  a.out = out.data_ptr();
  a.out_stride_t = out.stride(0);
  a.out_stride_h = out.stride(1);
  a.q = query.data_ptr();
  a.q_stride_t = query.stride(0);
  a.q_stride_h = query.stride(1);
  a.k = key_cache.data_ptr();
  a.v = value_cache.data_ptr();
  if (paged) {
    a.kv_stride_page = key_cache.stride(0);
    a.kv_stride_t = key_cache.stride(1);
    a.kv_stride_h = key_cache.stride(2);
    TORCH_CHECK(block_table->scalar_type() == at::kInt && block_table->dim() == 2 && block_table->stride(1) == 1 &&
                    block_table->size(0) >= seq_lens.numel(),
                "trianglemix: block_table must be int32 [B, max_blocks] with contiguous rows that cover the batch");
    a.block_table = block_table->data_ptr<int32_t>();
    a.block_table_stride = block_table->stride(0);
    a.page_size = (int32_t)key_cache.size(1);
  } else {
    TORCH_CHECK(cu_seqlens_k.has_value(), "trianglemix: contiguous K/V needs cu_seqlens_k");
    a.kv_stride_t = key_cache.stride(0);
    a.kv_stride_h = key_cache.stride(1);
    a.cu_seqlens_k = i32(*cu_seqlens_k);
  }
  if (fp8) {
    a.kv_dtype = TMX_FP8_E4M3;
    a.k_scale = (float)k_scale;
    a.v_scale = (float)v_scale;
  } else {
    a.k_scale = a.v_scale = 1.0f;
  }
  TORCH_CHECK(k_new.has_value() == v_new.has_value(), "trianglemix: k_new and v_new go together");
  if (k_new.has_value()) {
    TORCH_CHECK(paged && !fp8, "trianglemix: k_new / v_new need a 16-bit paged cache");
    TORCH_CHECK(k_new->dim() == 3 && k_new->sizes() == v_new->sizes() && k_new->strides() == v_new->strides() &&
                    k_new->stride(2) == 1 && k_new->scalar_type() == query.scalar_type() &&
                    v_new->scalar_type() == query.scalar_type() && k_new->size(0) >= query.size(0) &&
                    k_new->size(1) == hkv && k_new->size(2) == query.size(2),
                "trianglemix: k_new / v_new must be [T, Hkv, D] rows of the q dtype with equal strides");
    a.k_new = k_new->data_ptr();
    a.v_new = v_new->data_ptr();
    a.new_stride_t = k_new->stride(0);
    a.new_stride_h = k_new->stride(1);
  }
  if (cache_slots.has_value()) {
    TORCH_CHECK(k_new.has_value() && cache_slots->scalar_type() == at::kLong && cache_slots->is_contiguous() &&
                    cache_slots->numel() >= query.size(0),
                "trianglemix: cache_slots must be contiguous int64 [T] and comes with k_new / v_new");
    a.cache_slots = cache_slots->data_ptr<int64_t>();
  }
  a.cu_seqlens_q = i32(cu_seqlens_q);
  a.seq_lens = i32(seq_lens);
  if (bias.has_value()) {
    a.bias = bias->data_ptr();
    a.bias_dtype = dtype_code(*bias);
    TORCH_CHECK(bias->dim() == 2 || bias->dim() == 4, "trianglemix: bias must be [rows, cols] or [1|B, 1|H, rows, cols]");
    const int64_t d0 = bias->dim() - 2;
    TORCH_CHECK(bias->stride(d0 + 1) == 1, "trianglemix: bias rows must be contiguous");
    a.bias_stride = bias->stride(d0);
    a.bias_rows = (int32_t)bias->size(d0);
    a.bias_cols = (int32_t)bias->size(d0 + 1);
    if (bias->dim() == 4) {
      TORCH_CHECK((bias->size(0) == 1 || bias->size(0) == seq_lens.numel()) && (bias->size(1) == 1 || bias->size(1) == query.size(1)),
                  "trianglemix: a 4-D bias is [1|batch, 1|heads, rows, cols]");
      a.bias_stride_b = bias->size(0) > 1 ? bias->stride(0) : 0;
      a.bias_stride_h = bias->size(1) > 1 ? bias->stride(1) : 0;
    }
  }
  a.index = &ix;
  a.num_items = num_items;
  a.num_partials = num_partials;
  a.num_split_qblocks = num_split_qblocks;
  a.nnz_tiles = nnz_tiles;  // 0: unknown (the library then falls back to the capacity of col_idx)
  if (num_partials > 0) {
    const int64_t need = num_partials * query.size(1) * block_m;
    TORCH_CHECK(partial_o.has_value() && partial_ml.has_value() && partial_o->scalar_type() == at::kFloat &&
                    partial_ml->scalar_type() == at::kFloat && partial_o->is_contiguous() && partial_ml->is_contiguous() &&
                    partial_o->numel() >= need * query.size(2) && partial_ml->numel() >= need * 2,
                "trianglemix: attention workspace too small for this index");
  }
  a.partial_o = partial_o.has_value() ? partial_o->data_ptr<float>() : nullptr;
  a.partial_ml = partial_ml.has_value() ? partial_ml->data_ptr<float>() : nullptr;
  a.flags = (uint32_t)flags;
  if (kv_stage.has_value()) {
    // staging area of the dense causal kernel: [2][rows][Hkv][128] 16-bit values
    on_gpu(*kv_stage, query, "kv_stage");
    TORCH_CHECK(kv_stage->dim() == 4 && kv_stage->size(0) == 2 && kv_stage->size(2) == hkv && kv_stage->size(3) == 128 &&
                    kv_stage->element_size() == 2 && kv_stage->is_contiguous(),
                "trianglemix: kv_stage must be a contiguous 16-bit [2, rows, Hkv, 128] tensor");
    a.kv_stage = kv_stage->data_ptr();
    a.kv_stage_rows = kv_stage->size(1);
  }
  check(tmx_attention(&a, stream_of(query)));
}

// A step of one-row requests with a launch that depends on the batch size alone (tmx_decode_attention): what the
// backend calls for uniform single-token decode steps, and what replays under a graph.
void decode_attention(Tensor& out, const Tensor& query, const Tensor& key_cache, const Tensor& value_cache,
                      const Tensor& block_table, const Tensor& seq_lens, Tensor& partial_o, Tensor& partial_ml,
                      double scale, const std::optional<Tensor>& k_new, const std::optional<Tensor>& v_new,
                      const std::optional<Tensor>& cache_slots, double k_scale, double v_scale) {
  TORCH_CHECK(query.is_cuda(), "trianglemix: query must be on a HIP device (there is no CPU path)");
  const DeviceGuard guard(query.device());
  TORCH_CHECK(query.dim() == 3 && out.dim() == 3 && query.sizes() == out.sizes() && query.stride(2) == 1 && out.stride(2) == 1,
              "trianglemix: query / out must both be [B, H, D] with a contiguous head dimension");
  TORCH_CHECK(key_cache.dim() == 4 && key_cache.sizes() == value_cache.sizes() && key_cache.strides() == value_cache.strides() &&
                  key_cache.stride(3) == 1 && key_cache.size(3) == query.size(2),
              "trianglemix: paged K/V: caches [num_blocks, page_size, Hkv, D] with equal strides");
  on_gpu(out, query, "out");
  on_gpu(key_cache, query, "key_cache");
  on_gpu(value_cache, query, "value_cache");
  on_gpu(block_table, query, "block_table");
  on_gpu(seq_lens, query, "seq_lens");
  on_gpu(partial_o, query, "partial_o");
  on_gpu(partial_ml, query, "partial_ml");
  on_gpu(k_new, query, "k_new");
  on_gpu(v_new, query, "v_new");
  on_gpu(cache_slots, query, "cache_slots");
  const int64_t B = query.size(0);
  TORCH_CHECK(seq_lens.scalar_type() == at::kInt && seq_lens.is_contiguous() && seq_lens.numel() >= B,
              "trianglemix: seq_lens must be contiguous int32 [B]");
  TORCH_CHECK(block_table.scalar_type() == at::kInt && block_table.dim() == 2 && block_table.stride(1) == 1 &&
                  block_table.size(0) >= B,
              "trianglemix: block_table must be int32 [B, max_blocks] with contiguous rows that cover the batch");
  const bool fp8 = key_cache.element_size() == 1;
  TORCH_CHECK(fp8 ? value_cache.element_size() == 1
                  : (key_cache.scalar_type() == query.scalar_type() && value_cache.scalar_type() == query.scalar_type()),
              "trianglemix: q, k, v must share a dtype of bfloat16 or float16 (or K/V be an fp8 cache)");
  TORCH_CHECK(partial_o.scalar_type() == at::kFloat && partial_ml.scalar_type() == at::kFloat && partial_o.is_contiguous() &&
                  partial_ml.is_contiguous(),
              "trianglemix: the decode work area is two contiguous float32 tensors");
  tmx_decode_args_t a{};
  a.struct_size = sizeof(a);
  a.io_dtype = dtype_code(query);
  a.out_dtype = dtype_code(out);
  a.scale = (float)scale;
  a.batch = (int32_t)B;
  a.num_heads = (int32_t)query.size(1);
  a.num_kv_heads = (int32_t)key_cache.size(2);
  a.head_dim = (int32_t)query.size(2);
  a.out = out.data_ptr();
  a.out_stride_t = out.stride(0);
  a.out_stride_h = out.stride(1);
  a.q = query.data_ptr();
  a.q_stride_t = query.stride(0);
  a.q_stride_h = query.stride(1);
  a.k = key_cache.data_ptr();
  a.v = value_cache.data_ptr();
  a.kv_stride_page = key_cache.stride(0);
  a.kv_stride_t = key_cache.stride(1);
  a.kv_stride_h = key_cache.stride(2);
  a.block_table = block_table.data_ptr<int32_t>();
  a.block_table_stride = block_table.stride(0);
  a.page_size = (int32_t)key_cache.size(1);
  a.kv_dtype = fp8 ? TMX_FP8_E4M3 : 0;
  a.k_scale = fp8 ? (float)k_scale : 1.0f;
  a.v_scale = fp8 ? (float)v_scale : 1.0f;
  TORCH_CHECK(k_new.has_value() == v_new.has_value(), "trianglemix: k_new and v_new go together");
  if (k_new.has_value()) {
    TORCH_CHECK(k_new->dim() == 3 && k_new->sizes() == v_new->sizes() && k_new->strides() == v_new->strides() &&
                    k_new->stride(2) == 1 && k_new->scalar_type() == query.scalar_type() &&
                    v_new->scalar_type() == query.scalar_type() && k_new->size(0) >= B &&
                    k_new->size(1) == key_cache.size(2) && k_new->size(2) == query.size(2),
                "trianglemix: k_new / v_new must be [B, Hkv, D] rows of the q dtype with equal strides");
    a.k_new = k_new->data_ptr();
    a.v_new = v_new->data_ptr();
    a.new_stride_t = k_new->stride(0);
    a.new_stride_h = k_new->stride(1);
  }
  if (cache_slots.has_value()) {
    TORCH_CHECK(k_new.has_value() && cache_slots->scalar_type() == at::kLong && cache_slots->is_contiguous() &&
                    cache_slots->numel() >= B,
                "trianglemix: cache_slots must be contiguous int64 [B] and comes with k_new / v_new");
    a.cache_slots = cache_slots->data_ptr<int64_t>();
  }
  a.seq_lens = seq_lens.data_ptr<int32_t>();
  a.partial_o = partial_o.data_ptr<float>();
  a.partial_ml = partial_ml.data_ptr<float>();
  a.workspace_rows = std::min<int64_t>(partial_o.numel() / query.size(2), partial_ml.numel() / 2);
  check(tmx_decode_attention(&a, stream_of(query)));
}

void reshape_and_cache(const Tensor& key, const Tensor& value, Tensor& key_cache, Tensor& value_cache,
                       const Tensor& slot_mapping) {
  const DeviceGuard guard(key.device());
  TORCH_CHECK(slot_mapping.scalar_type() == at::kLong && slot_mapping.is_contiguous(),
              "trianglemix: slot_mapping must be contiguous int64");
  const int64_t n = std::min<int64_t>(slot_mapping.numel(), key.size(0));
  check(tmx_reshape_and_cache(key.data_ptr(), value.data_ptr(), key.stride(0), key.stride(1), key_cache.data_ptr(),
                              value_cache.data_ptr(), key_cache.stride(0), key_cache.stride(1), key_cache.stride(2),
                              slot_mapping.data_ptr<int64_t>(), (int32_t)n, (int32_t)key.size(1), (int32_t)key.size(2),
                              (int32_t)key_cache.size(1), dtype_code(key), stream_of(key)));
}

// fp8 (e4m3fn) caches arrive as uint8 or float8_e4m3fn tensors; strides of a one-byte type are bytes.
void* fp8_ptr(const Tensor& t) {
  TORCH_CHECK(t.element_size() == 1, "trianglemix: an fp8 cache is a one-byte tensor (uint8 / float8_e4m3fn)");
  return t.data_ptr();
}

void reshape_and_cache_fp8(const Tensor& key, const Tensor& value, Tensor& key_cache, Tensor& value_cache,
                           const Tensor& slot_mapping, double k_scale, double v_scale) {
  const DeviceGuard guard(key.device());
  TORCH_CHECK(slot_mapping.scalar_type() == at::kLong && slot_mapping.is_contiguous(),
              "trianglemix: slot_mapping must be contiguous int64");
  const int64_t n = std::min<int64_t>(slot_mapping.numel(), key.size(0));
  check(tmx_reshape_and_cache_fp8(key.data_ptr(), value.data_ptr(), key.stride(0), key.stride(1), fp8_ptr(key_cache),
                                  fp8_ptr(value_cache), key_cache.stride(0), key_cache.stride(1), key_cache.stride(2),
                                  slot_mapping.data_ptr<int64_t>(), (int32_t)n, (int32_t)key.size(1),
                                  (int32_t)key.size(2), (int32_t)key_cache.size(1), dtype_code(key), (float)k_scale,
                                  (float)v_scale, stream_of(key)));
}

void gather_dequant_kv_fp8(Tensor& k_rows, Tensor& v_rows, const Tensor& key_cache, const Tensor& value_cache,
                           const Tensor& block_table, const Tensor& seq_lens, const Tensor& cu_seqlens_k,
                           int64_t max_seq_len, double k_scale, double v_scale) {
  const DeviceGuard guard(key_cache.device());
  TORCH_CHECK(k_rows.is_contiguous() && v_rows.is_contiguous() && k_rows.dim() == 3,
              "trianglemix: k_rows / v_rows must be contiguous [rows, Hkv, D]");
  check(tmx_gather_dequant_kv_fp8(fp8_ptr(key_cache), fp8_ptr(value_cache), key_cache.stride(0), key_cache.stride(1),
                                  key_cache.stride(2), i32(block_table), block_table.stride(0),
                                  (int32_t)key_cache.size(1), i32(seq_lens), i32(cu_seqlens_k),
                                  (int32_t)seq_lens.numel(), (int32_t)max_seq_len, (int32_t)key_cache.size(2),
                                  (int32_t)key_cache.size(3), k_rows.data_ptr(), v_rows.data_ptr(), dtype_code(k_rows),
                                  (float)k_scale, (float)v_scale, stream_of(key_cache)));
}

}  // namespace

TORCH_LIBRARY(_trianglemix_C, m) {
  m.def("trianglemix_index_sizes(Tensor cu_seqlens_q_cpu, Tensor seq_lens_cpu, int si, int sl, int last, int block_m, "
        "int block_n, int mode, int tri_min_len, int chunk_tiles, Tensor? full_lens_cpu=None, int dense_rows_from=0) -> int[]");
  m.impl("trianglemix_index_sizes", c10::DispatchKey::CPU, &index_sizes);
  m.def("trianglemix_suggest_chunk_tiles(Tensor cu_seqlens_q_cpu, Tensor seq_lens_cpu, int si, int sl, int last, "
        "int block_m, int block_n, int mode, int tri_min_len, int units_per_item, int num_cus, "
        "Tensor? full_lens_cpu=None) -> int");
  m.impl("trianglemix_suggest_chunk_tiles", c10::DispatchKey::CPU, &suggest_chunk_tiles);
  m.def("trianglemix_build_block_index(Tensor(a!)[] index, Tensor? tile_pages, Tensor cu_seqlens_q, Tensor seq_lens, "
        "Tensor cu_seqlens_q_cpu, Tensor seq_lens_cpu, Tensor? block_table, int page_size, int si, int sl, int last, "
        "int block_m, int block_n, int mode, int tri_min_len, int chunk_tiles, Tensor? full_lens=None, "
        "Tensor? full_lens_cpu=None, int dense_rows_from=0) -> ()");
  m.impl("trianglemix_build_block_index", c10::DispatchKey::CUDA, &build_block_index);
  m.def("trianglemix_attention(Tensor! out, Tensor query, Tensor key_cache, Tensor value_cache, Tensor? block_table, "
        "Tensor cu_seqlens_q, Tensor seq_lens, Tensor? cu_seqlens_k, Tensor[] index, Tensor? tile_pages, "
        "Tensor? partial_o, Tensor? partial_ml, Tensor? bias, float scale, int si, int sl, int last, int mode, "
        "int tri_min_len, int chunk_tiles, int num_items, int num_partials, int num_split_qblocks, "
        "Tensor? full_lens=None, int flags=0, int nnz_tiles=0, Tensor? k_new=None, Tensor? v_new=None, "
        "Tensor? cache_slots=None, float k_scale=1.0, float v_scale=1.0, int index_block_m=0, int index_block_n=0, "
        "Tensor? kv_stage=None, int index_dense_rows_from=0) -> ()");
  m.impl("trianglemix_attention", c10::DispatchKey::CUDA, &attention);
  m.def("trianglemix_decode_attention(Tensor! out, Tensor query, Tensor key_cache, Tensor value_cache, Tensor block_table, "
        "Tensor seq_lens, Tensor! partial_o, Tensor! partial_ml, float scale, Tensor? k_new=None, Tensor? v_new=None, "
        "Tensor? cache_slots=None, float k_scale=1.0, float v_scale=1.0) -> ()");
  m.impl("trianglemix_decode_attention", c10::DispatchKey::CUDA, &decode_attention);
  m.def("trianglemix_reshape_and_cache(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, "
        "Tensor slot_mapping) -> ()");
  m.impl("trianglemix_reshape_and_cache", c10::DispatchKey::CUDA, &reshape_and_cache);
  m.def("trianglemix_reshape_and_cache_fp8(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, "
        "Tensor slot_mapping, float k_scale, float v_scale) -> ()");
  m.impl("trianglemix_reshape_and_cache_fp8", c10::DispatchKey::CUDA, &reshape_and_cache_fp8);
  m.def("trianglemix_gather_dequant_kv_fp8(Tensor! k_rows, Tensor! v_rows, Tensor key_cache, Tensor value_cache, "
        "Tensor block_table, Tensor seq_lens, Tensor cu_seqlens_k, int max_seq_len, float k_scale, float v_scale) -> ()");
  m.impl("trianglemix_gather_dequant_kv_fp8", c10::DispatchKey::CUDA, &gather_dequant_kv_fp8);
}

// The extension is also a Python module, so that a plain `import` of the shared object runs the
// TORCH_LIBRARY block above: the reference's REGISTER_EXTENSION (csrc/core/registration.h:13-27), which is
// how vllm/platforms/rocm.py:32-40 loads vllm._C / vllm._rocm_C.
#include <Python.h>

extern "C" __attribute__((visibility("default"))) PyObject* PyInit__trianglemix_C() {
  static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_trianglemix_C", nullptr, 0, nullptr};
  return PyModule_Create(&module);
}
