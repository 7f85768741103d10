// K/V cache write: this step's key/value rows into the paged cache by slot.
//
// Same contract as the reference's reshape_and_cache_flash op
// (csrc/cache_kernels.cu:282-355, launch at :665-666; called from
// vllm/v1/attention/backends/rocm_attn.py:345-354) for a 16-bit cache
// laid out [num_blocks, page_size, num_kv_heads, head_dim]: token t goes to
// page slot_mapping[t] / page_size, offset slot_mapping[t] % page_size;
// negative slots (padding) are skipped.  One workgroup per token, 16-byte
// loads and stores.
#include "tmx_common.h"

namespace tmx {

__global__ __launch_bounds__(256) void reshape_and_cache_kernel(
    const uint16_t* __restrict__ key, const uint16_t* __restrict__ value, int64_t in_stride_t, int64_t in_stride_h,
    uint16_t* __restrict__ key_cache, uint16_t* __restrict__ value_cache, int64_t stride_page, int64_t stride_t,
    int64_t stride_h, const int64_t* __restrict__ slot_mapping, int num_kv_heads, int head_dim, int page_size) {
  const int t = blockIdx.x;
  const int64_t slot = slot_mapping[t];
  if (slot < 0) return;
  const int64_t page = slot / page_size, in_pg = slot - page * page_size;
  const int cpr = head_dim / 8;  // 16-byte chunks per head row
  for (int c = threadIdx.x; c < num_kv_heads * cpr; c += blockDim.x) {
    const int h = c / cpr, cc = c - h * cpr;
    const int64_t src = (int64_t)t * in_stride_t + (int64_t)h * in_stride_h + cc * 8;
    const int64_t dst = page * stride_page + in_pg * stride_t + (int64_t)h * stride_h + cc * 8;
    *reinterpret_cast<uint4*>(key_cache + dst) = *reinterpret_cast<const uint4*>(key + src);
    *reinterpret_cast<uint4*>(value_cache + dst) = *reinterpret_cast<const uint4*>(value + src);
  }
}

}  // namespace tmx

extern "C" int tmx_reshape_and_cache(const void* key, const void* value, int64_t in_stride_t, int64_t in_stride_h,
                                     void* key_cache, void* value_cache, int64_t stride_page, int64_t stride_t,
                                     int64_t stride_h, const int64_t* slot_mapping, int32_t num_tokens,
                                     int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t dtype,
                                     void* stream) {
  TMX_CHECK_ARG(num_tokens >= 0 && num_kv_heads > 0 && head_dim > 0 && page_size > 0,
                "tmx_reshape_and_cache: bad sizes");
  TMX_CHECK_SUPPORTED(dtype == TMX_BF16 || dtype == TMX_F16, "tmx_reshape_and_cache: 16-bit caches only (dtype %d)",
                      dtype);
  TMX_CHECK_SUPPORTED(head_dim % 8 == 0, "tmx_reshape_and_cache: head_dim %d must be a multiple of 8", head_dim);
  if (num_tokens == 0) return TMX_OK;
  TMX_CHECK_ARG(key && value && key_cache && value_cache && slot_mapping, "tmx_reshape_and_cache: NULL pointer");
  TMX_CHECK_ARG(in_stride_t % 8 == 0 && in_stride_h % 8 == 0 && stride_page % 8 == 0 && stride_t % 8 == 0 &&
                    stride_h % 8 == 0,
                "tmx_reshape_and_cache: strides must be multiples of 8 elements");
  TMX_CHECK_ARG(((uintptr_t)key % 16 == 0) && ((uintptr_t)value % 16 == 0) && ((uintptr_t)key_cache % 16 == 0) &&
                    ((uintptr_t)value_cache % 16 == 0),
                "tmx_reshape_and_cache: pointers must be 16-byte aligned");
  const int work = num_kv_heads * (head_dim / 8);
  const int threads = work >= 256 ? 256 : (work > 64 ? 128 : 64);
  tmx::reshape_and_cache_kernel<<<num_tokens, threads, 0, (hipStream_t)stream>>>(
      (const uint16_t*)key, (const uint16_t*)value, in_stride_t, in_stride_h, (uint16_t*)key_cache,
      (uint16_t*)value_cache, stride_page, stride_t, stride_h, slot_mapping, num_kv_heads, head_dim, page_size);
  TMX_CHECK_HIP(hipGetLastError());
  return TMX_OK;
}

// ---------------------------------------------------------------------------
// fp8 (e4m3fn) K/V cache.
//
// Write: the reference's reshape_and_cache_flash with kv_cache_dtype "fp8"
// (csrc/cache_kernels.cu:282-355 with the scaled conversion of
// csrc/quantization/fp8/amd/quant_utils.cuh): cache byte = fp8(x / scale),
// round to nearest even, finite values saturating at +-448.
// Read: the attention kernels take 16-bit K/V, so the keys and values a step
// attends to are gathered from the pages and converted back (fp8 -> float is
// exact, times the scale, rounded once to the 16-bit type) into contiguous rows
// [sum of seq_lens, num_kv_heads, head_dim] that tmx_attention reads in its
// contiguous mode with cu_seqlens_k (the descale the reference passes to its
// attention kernel as k_descale / v_descale, rocm_attn.py:421-422).
// ---------------------------------------------------------------------------
namespace tmx {

typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float load16(const uint16_t* p, int dtype) {
  return dtype == TMX_BF16 ? __uint_as_float((uint32_t)*p << 16) : (float)*reinterpret_cast<const _Float16*>(p);
}
__device__ __forceinline__ float sat448(float x) { return fminf(fmaxf(x, -448.f), 448.f); }  // NaN stays NaN

__global__ __launch_bounds__(256) void reshape_and_cache_fp8_kernel(
    const uint16_t* __restrict__ key, const uint16_t* __restrict__ value, int64_t in_stride_t, int64_t in_stride_h,
    uint8_t* __restrict__ key_cache, uint8_t* __restrict__ value_cache, int64_t stride_page, int64_t stride_t,
    int64_t stride_h, const int64_t* __restrict__ slot_mapping, int num_kv_heads, int head_dim, int page_size,
    int in_dtype, float k_scale, float v_scale) {
  const int t = blockIdx.x;
  const int64_t slot = slot_mapping[t];
  if (slot < 0) return;
  const int64_t page = slot / page_size, in_pg = slot - page * page_size;
  const int cpr = head_dim / 8;  // 8 elements per thread: 16 bytes in, 8 bytes out
  for (int c = threadIdx.x; c < num_kv_heads * cpr; c += blockDim.x) {
    const int h = c / cpr, cc = c - h * cpr;
    const int64_t src = (int64_t)t * in_stride_t + (int64_t)h * in_stride_h + cc * 8;
    const int64_t dst = page * stride_page + in_pg * stride_t + (int64_t)h * stride_h + cc * 8;
    const uint4 kin = *reinterpret_cast<const uint4*>(key + src);
    const uint4 vin = *reinterpret_cast<const uint4*>(value + src);
    const uint2 ko = quant8_fp8(kin, in_dtype == TMX_BF16, k_scale), vo = quant8_fp8(vin, in_dtype == TMX_BF16, v_scale);
    *reinterpret_cast<uint2*>(key_cache + dst) = ko;
    *reinterpret_cast<uint2*>(value_cache + dst) = vo;
  }
}

__device__ __forceinline__ uint32_t pack16(float a, float b, int dtype) {
  if (dtype == TMX_BF16) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf2));
  }
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, h2));
}

// grid (token blocks of the longest request, batch); a thread converts 16 cache bytes of K and of V
__global__ __launch_bounds__(256) void gather_dequant_kernel(
    const uint8_t* __restrict__ key_cache, const uint8_t* __restrict__ value_cache, int64_t stride_page,
    int64_t stride_t, int64_t stride_h, const int32_t* __restrict__ block_table, int64_t block_table_stride,
    int page_shift, const int32_t* __restrict__ seq_lens, const int32_t* __restrict__ cu_k,
    uint16_t* __restrict__ k_out, uint16_t* __restrict__ v_out, int num_kv_heads, int head_dim, int out_dtype,
    float k_scale, float v_scale, int tokens_per_block) {
  const int b = blockIdx.y;
  const int n = seq_lens[b];
  const int tok0 = blockIdx.x * tokens_per_block;
  if (tok0 >= n) return;
  const int cpr = head_dim / 16;
  const int per_tok = num_kv_heads * cpr;
  const int64_t row0 = cu_k[b];
  const int32_t* bt = block_table + (int64_t)b * block_table_stride;
  const int page_mask = (1 << page_shift) - 1;
  for (int w = threadIdx.x; w < tokens_per_block * per_tok; w += blockDim.x) {
    const int tl = w / per_tok, c = w - tl * per_tok;
    const int j = tok0 + tl;
    if (j >= n) break;
    const int h = c / cpr, cc = c - h * cpr;
    const int64_t src = (int64_t)bt[j >> page_shift] * stride_page + (int64_t)(j & page_mask) * stride_t +
                        (int64_t)h * stride_h + cc * 16;
    const int64_t dst = ((row0 + j) * num_kv_heads + h) * head_dim + cc * 16;
    const uint4 kin = *reinterpret_cast<const uint4*>(key_cache + src);
    const uint4 vin = *reinterpret_cast<const uint4*>(value_cache + src);
    const uint32_t* kw = reinterpret_cast<const uint32_t*>(&kin);
    const uint32_t* vw = reinterpret_cast<const uint32_t*>(&vin);
    uint32_t ko[8], vo[8];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const f32x2_t k0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)kw[x], false), k1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)kw[x], true);
      const f32x2_t v0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)vw[x], false), v1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)vw[x], true);
      ko[2 * x] = pack16(k0[0] * k_scale, k0[1] * k_scale, out_dtype);
      ko[2 * x + 1] = pack16(k1[0] * k_scale, k1[1] * k_scale, out_dtype);
      vo[2 * x] = pack16(v0[0] * v_scale, v0[1] * v_scale, out_dtype);
      vo[2 * x + 1] = pack16(v1[0] * v_scale, v1[1] * v_scale, out_dtype);
    }
    uint4* kd = reinterpret_cast<uint4*>(k_out + dst);
    uint4* vd = reinterpret_cast<uint4*>(v_out + dst);
    kd[0] = make_uint4(ko[0], ko[1], ko[2], ko[3]);
    kd[1] = make_uint4(ko[4], ko[5], ko[6], ko[7]);
    vd[0] = make_uint4(vo[0], vo[1], vo[2], vo[3]);
    vd[1] = make_uint4(vo[4], vo[5], vo[6], vo[7]);
  }
}

}  // namespace tmx

extern "C" int tmx_reshape_and_cache_fp8(const void* key, const void* value, int64_t in_stride_t, int64_t in_stride_h,
                                         void* key_cache, void* value_cache, int64_t stride_page, int64_t stride_t,
                                         int64_t stride_h, const int64_t* slot_mapping, int32_t num_tokens,
                                         int32_t num_kv_heads, int32_t head_dim, int32_t page_size, int32_t in_dtype,
                                         float k_scale, float v_scale, void* stream) {
  TMX_CHECK_ARG(num_tokens >= 0 && num_kv_heads > 0 && head_dim > 0 && page_size > 0,
                "tmx_reshape_and_cache_fp8: bad sizes");
  TMX_CHECK_SUPPORTED(in_dtype == TMX_BF16 || in_dtype == TMX_F16,
                      "tmx_reshape_and_cache_fp8: key / value must be bf16 or fp16 (dtype %d)", in_dtype);
  TMX_CHECK_SUPPORTED(head_dim % 8 == 0, "tmx_reshape_and_cache_fp8: head_dim %d must be a multiple of 8", head_dim);
  TMX_CHECK_ARG(k_scale > 0.f && v_scale > 0.f && k_scale < 3.0e38f && v_scale < 3.0e38f,
                "tmx_reshape_and_cache_fp8: scales must be positive and finite");
  if (num_tokens == 0) return TMX_OK;
  TMX_CHECK_ARG(key && value && key_cache && value_cache && slot_mapping, "tmx_reshape_and_cache_fp8: NULL pointer");
  TMX_CHECK_ARG(in_stride_t % 8 == 0 && in_stride_h % 8 == 0 && stride_page % 8 == 0 && stride_t % 8 == 0 &&
                    stride_h % 8 == 0,
                "tmx_reshape_and_cache_fp8: strides must be multiples of 8 elements");
  TMX_CHECK_ARG(((uintptr_t)key % 16 == 0) && ((uintptr_t)value % 16 == 0) && ((uintptr_t)key_cache % 8 == 0) &&
                    ((uintptr_t)value_cache % 8 == 0),
                "tmx_reshape_and_cache_fp8: key / value must be 16-byte, the caches 8-byte aligned");
  const int work = num_kv_heads * (head_dim / 8);
  const int threads = work >= 256 ? 256 : (work > 64 ? 128 : 64);
  tmx::reshape_and_cache_fp8_kernel<<<num_tokens, threads, 0, (hipStream_t)stream>>>(
      (const uint16_t*)key, (const uint16_t*)value, in_stride_t, in_stride_h, (uint8_t*)key_cache,
      (uint8_t*)value_cache, stride_page, stride_t, stride_h, slot_mapping, num_kv_heads, head_dim, page_size,
      in_dtype, k_scale, v_scale);
  TMX_CHECK_HIP(hipGetLastError());
  return TMX_OK;
}

extern "C" int tmx_gather_dequant_kv_fp8(const void* key_cache, const void* value_cache, int64_t stride_page,
                                         int64_t stride_t, int64_t stride_h, const int32_t* block_table,
                                         int64_t block_table_stride, int32_t page_size, const int32_t* seq_lens,
                                         const int32_t* cu_seqlens_k, int32_t batch, int32_t max_seq_len,
                                         int32_t num_kv_heads, int32_t head_dim, void* k_out, void* v_out,
                                         int32_t out_dtype, float k_scale, float v_scale, void* stream) {
  TMX_CHECK_ARG(batch >= 0 && max_seq_len >= 0 && num_kv_heads > 0 && head_dim > 0,
                "tmx_gather_dequant_kv_fp8: bad sizes");
  TMX_CHECK_SUPPORTED(out_dtype == TMX_BF16 || out_dtype == TMX_F16,
                      "tmx_gather_dequant_kv_fp8: output must be bf16 or fp16 (dtype %d)", out_dtype);
  TMX_CHECK_SUPPORTED(head_dim % 16 == 0, "tmx_gather_dequant_kv_fp8: head_dim %d must be a multiple of 16", head_dim);
  TMX_CHECK_SUPPORTED(page_size > 0 && (page_size & (page_size - 1)) == 0,
                      "tmx_gather_dequant_kv_fp8: page_size %d must be a power of two", page_size);
  TMX_CHECK_ARG(k_scale > 0.f && v_scale > 0.f && k_scale < 3.0e38f && v_scale < 3.0e38f,
                "tmx_gather_dequant_kv_fp8: scales must be positive and finite");
  if (batch == 0 || max_seq_len == 0) return TMX_OK;
  TMX_CHECK_ARG(key_cache && value_cache && block_table && seq_lens && cu_seqlens_k && k_out && v_out,
                "tmx_gather_dequant_kv_fp8: NULL pointer");
  TMX_CHECK_ARG(stride_page % 16 == 0 && stride_t % 16 == 0 && stride_h % 16 == 0,
                "tmx_gather_dequant_kv_fp8: cache strides must be multiples of 16 bytes");
  TMX_CHECK_ARG(((uintptr_t)key_cache % 16 == 0) && ((uintptr_t)value_cache % 16 == 0) && ((uintptr_t)k_out % 16 == 0) &&
                    ((uintptr_t)v_out % 16 == 0),
                "tmx_gather_dequant_kv_fp8: pointers must be 16-byte aligned");
  const int per_tok = num_kv_heads * (head_dim / 16);
  int tokens_per_block = 256 / per_tok;
  if (tokens_per_block < 1) tokens_per_block = 1;
  tokens_per_block *= 4;  // four rounds per thread
  const unsigned gx = (unsigned)((max_seq_len + tokens_per_block - 1) / tokens_per_block);
  tmx::gather_dequant_kernel<<<dim3(gx, (unsigned)batch), 256, 0, (hipStream_t)stream>>>(
      (const uint8_t*)key_cache, (const uint8_t*)value_cache, stride_page, stride_t, stride_h, block_table,
      block_table_stride, __builtin_ctz(page_size), seq_lens, cu_seqlens_k, (uint16_t*)k_out, (uint16_t*)v_out,
      num_kv_heads, head_dim, out_dtype, k_scale, v_scale, tokens_per_block);
  TMX_CHECK_HIP(hipGetLastError());
  return TMX_OK;
}
