// Dense [n, n] mask materialisation and the -inf replacement pass.
//
// API-parity kernels for TriangleMixMaskGenerator._create_*_mask /
// get_attention_mask (vllm/attention/trianglemix_attention.py:53-261) and
// AscendNPUTriangleMixOptimizer.optimize_mask_for_npu
// (vllm/attention/ascend_trianglemix.py:131-141).  Pure HBM-write work:
// every lane evaluates the predicate for 8 consecutive columns and stores them
// with one (16-bit types) or two (fp32) 16-byte stores.
#include <cstring>

#include "tmx_common.h"

namespace tmx {

__device__ __forceinline__ bool kind_allowed(int kind, int i, int j, int n, int si, int sl, int last) {
  if (j > i) return false;
  const bool stream = (j <= si) || (i - j) <= sl;
  switch (kind) {
    case TMX_MASK_STREAMING: return stream;
    case TMX_MASK_LAST: return !stream && (n - i) < last;
    case TMX_MASK_MIDDLE: return !stream && (n - i) >= last;
    case TMX_MASK_TRIANGLE: return stream || (n - i) < last;
    default: return true;  // TMX_MASK_DENSE
  }
}

template <typename T>
struct Vals;
template <>
struct Vals<float> {
  static __device__ __forceinline__ float make(bool on, bool additive) {
    return additive ? (on ? 0.0f : -__builtin_inff()) : (on ? 1.0f : 0.0f);
  }
};
template <>
struct Vals<uint16_t> {  // bf16 bit patterns
  static __device__ __forceinline__ uint16_t make(bool on, bool additive) {
    return additive ? (on ? 0x0000 : 0xFF80) : (on ? 0x3F80 : 0x0000);
  }
};
struct HalfBits {};
template <>
struct Vals<HalfBits> {  // fp16 bit patterns
  static __device__ __forceinline__ uint16_t make(bool on, bool additive) {
    return additive ? (on ? 0x0000 : 0xFC00) : (on ? 0x3C00 : 0x0000);
  }
};

template <typename Tag, typename Store>
__global__ __launch_bounds__(256) void dense_mask_kernel(Store* __restrict__ out, int64_t row_stride, int n, int si,
                                                         int sl, int last, int kind, int additive) {
  const int chunks = ceil_div(n, 256 * 8);
  const int i = blockIdx.x / chunks;
  const int j0 = ((blockIdx.x % chunks) * 256 + threadIdx.x) * 8;
  if (j0 >= n) return;
  Store* row = out + (int64_t)i * row_stride;
  Store v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = Vals<Tag>::make(kind_allowed(kind, i, j0 + e, n, si, sl, last), additive != 0);
  const bool aligned = ((reinterpret_cast<uintptr_t>(row + j0) & 15) == 0) && (j0 + 8 <= n);
  if (aligned) {
    if constexpr (sizeof(Store) == 4) {
      float4* p = reinterpret_cast<float4*>(row + j0);
      p[0] = make_float4(v[0], v[1], v[2], v[3]);
      p[1] = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      uint4 w;
      w.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
      w.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
      w.z = (uint32_t)v[4] | ((uint32_t)v[5] << 16);
      w.w = (uint32_t)v[6] | ((uint32_t)v[7] << 16);
      *reinterpret_cast<uint4*>(row + j0) = w;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (j0 + e < n) row[j0 + e] = v[e];
  }
}

template <typename Store, uint32_t NEG_INF_BITS>
__global__ __launch_bounds__(256) void replace_neg_inf_kernel(Store* __restrict__ m, int64_t count, Store value) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride) {
    Store x = m[e];
    uint32_t bits;
    if constexpr (sizeof(Store) == 4) bits = __float_as_uint(x);
    else bits = x;
    if (bits == NEG_INF_BITS) m[e] = value;
  }
}

static uint16_t f32_to_bf16_bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}

static uint16_t f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}

}  // namespace tmx

extern "C" {

int tmx_dense_mask(void* out, int64_t row_stride, int n, int si, int sl, int last, int kind, int dtype,
                   int additive, void* stream) {
  TMX_CHECK_ARG(n >= 0, "tmx_dense_mask: n = %d", n);
  TMX_CHECK_ARG(kind >= TMX_MASK_STREAMING && kind <= TMX_MASK_DENSE, "tmx_dense_mask: bad kind %d", kind);
  TMX_CHECK_ARG(dtype == TMX_F32 || dtype == TMX_BF16 || dtype == TMX_F16, "tmx_dense_mask: bad dtype %d", dtype);
  if (n == 0) return TMX_OK;
  TMX_CHECK_ARG(out != nullptr && row_stride >= n, "tmx_dense_mask: NULL output or row stride < n");
  TMX_CHECK_ARG((int64_t)n * tmx::ceil_div(n, 256 * 8) < 0x7FFFFFFFLL, "tmx_dense_mask: n = %d too large", n);
  dim3 grid((unsigned)(n * tmx::ceil_div(n, 256 * 8))), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TMX_F32)
    tmx::dense_mask_kernel<float, float><<<grid, block, 0, s>>>((float*)out, row_stride, n, si, sl, last, kind, additive);
  else if (dtype == TMX_BF16)
    tmx::dense_mask_kernel<uint16_t, uint16_t><<<grid, block, 0, s>>>((uint16_t*)out, row_stride, n, si, sl, last,
                                                                     kind, additive);
  else
    tmx::dense_mask_kernel<tmx::HalfBits, uint16_t><<<grid, block, 0, s>>>((uint16_t*)out, row_stride, n, si, sl,
                                                                          last, kind, additive);
  TMX_CHECK_HIP(hipGetLastError());
  return TMX_OK;
}

int tmx_replace_neg_inf(void* mask, int64_t count, int dtype, float value, void* stream) {
  TMX_CHECK_ARG(count >= 0, "tmx_replace_neg_inf: count < 0");
  TMX_CHECK_ARG(dtype == TMX_F32 || dtype == TMX_BF16 || dtype == TMX_F16, "tmx_replace_neg_inf: bad dtype %d", dtype);
  if (count == 0) return TMX_OK;
  TMX_CHECK_ARG(mask != nullptr, "tmx_replace_neg_inf: NULL mask");
  const int blocks = (int)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TMX_F32)
    tmx::replace_neg_inf_kernel<float, 0xFF800000u><<<blocks, 256, 0, s>>>((float*)mask, count, value);
  else if (dtype == TMX_BF16)
    tmx::replace_neg_inf_kernel<uint16_t, 0xFF80u><<<blocks, 256, 0, s>>>((uint16_t*)mask, count,
                                                                          tmx::f32_to_bf16_bits(value));
  else
    tmx::replace_neg_inf_kernel<uint16_t, 0xFC00u><<<blocks, 256, 0, s>>>((uint16_t*)mask, count,
                                                                          tmx::f32_to_f16_bits(value));
  TMX_CHECK_HIP(hipGetLastError());
  return TMX_OK;
}

}  // extern "C"
