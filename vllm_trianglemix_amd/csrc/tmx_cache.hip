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
