// fp32 form of the reference's eager attention (vllm/attention/ascend_trianglemix.py:88-106: matmul, + mask,
// softmax, matmul, all in the dtype of the inputs), for fp32 inputs of the standalone module -- the
// configuration the reference itself runs on the CPU.  Everything is computed in fp32 on the vector units (the
// 16-bit matrix-core kernels would round the inputs), one wave per query row: lanes are keys for the scores
// and head-dim columns for the weighted sum.  A compatibility path, sized for the reference's own test shapes.
#include "tmx_common.h"

namespace tmx {

struct EagerF32Params {
  const float* q;
  const float* k;
  const float* v;
  float* out;
  const void* bias;
  long long bias_stride_b, bias_stride_h, bias_stride_i;
  int bias_dtype;
  int batch, n, heads, d;
  int mode, si, sl, last;
  float scale;
};

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

constexpr int kEagerMaxD = 256;

__global__ __launch_bounds__(256) void eager_f32_kernel(const EagerF32Params p) {
  __shared__ float qs[4][kEagerMaxD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long row_id = (long long)blockIdx.x * 4 + wave;
  const long long total = (long long)p.batch * p.heads * p.n;
  if (row_id >= total) return;  // (no workgroup-wide barrier below: every wave works on its own)
  const int i = (int)(row_id % p.n);
  const int h = (int)((row_id / p.n) % p.heads);
  const int b = (int)(row_id / ((long long)p.n * p.heads));
  const long long tok_stride = (long long)p.heads * p.d;
  const float* qrow = p.q + ((long long)b * p.n + i) * tok_stride + (long long)h * p.d;
  for (int d = lane; d < p.d; d += 64) qs[wave][d] = qrow[d];
  __builtin_amdgcn_wave_barrier();
  const float* kb = p.k + (long long)b * p.n * tok_stride + (long long)h * p.d;
  const float* vb = p.v + (long long)b * p.n * tok_stride + (long long)h * p.d;
  const int j_end = p.mode == TMX_MODE_NONE ? p.n : i + 1;
  float m = -__builtin_inff(), l = 0.f;
  float o[kEagerMaxD / 64] = {0.f, 0.f, 0.f, 0.f};
  for (int j0 = 0; j0 < j_end; j0 += 64) {
    const int j = j0 + lane;
    const bool valid = j < j_end && allowed(p.mode, i, j, p.n, p.si, p.sl, p.last);
    if (__builtin_amdgcn_ballot_w64(valid) == 0) continue;  // e.g. the excluded middle section of a sparse row
    float s = -__builtin_inff();
    if (valid) {
      const float4* kr = reinterpret_cast<const float4*>(kb + (long long)j * tok_stride);
      float acc = 0.f;
      for (int d4 = 0; d4 < p.d / 4; ++d4) {
        const float4 kk = kr[d4];
        const float4 qq = *reinterpret_cast<const float4*>(&qs[wave][4 * d4]);
        acc = __builtin_fmaf(qq.x, kk.x, acc);
        acc = __builtin_fmaf(qq.y, kk.y, acc);
        acc = __builtin_fmaf(qq.z, kk.z, acc);
        acc = __builtin_fmaf(qq.w, kk.w, acc);
      }
      s = acc * p.scale;
      if (p.bias != nullptr) {
        const long long off = (long long)b * p.bias_stride_b + (long long)h * p.bias_stride_h + (long long)i * p.bias_stride_i + j;
        float bv;
        if (p.bias_dtype == TMX_F32) bv = reinterpret_cast<const float*>(p.bias)[off];
        else {
          const uint16_t raw = reinterpret_cast<const uint16_t*>(p.bias)[off];
          bv = p.bias_dtype == TMX_BF16 ? __uint_as_float((uint32_t)raw << 16) : (float)__builtin_bit_cast(_Float16, raw);
        }
        s += bv;
      }
    }
    const float m_new = fmaxf(m, wave_max(s));
    if (m_new == -__builtin_inff()) continue;  // everything so far masked by the bias
    const float pj = valid ? expf(s - m_new) : 0.f;  // exp(-inf - finite) = 0 for entries the bias masks
    const float alpha = expf(m - m_new);              // first chunk: exp(-inf) = 0
    l = l * alpha + wave_sum(pj);
#pragma unroll
    for (int t = 0; t < kEagerMaxD / 64; ++t) o[t] *= alpha;
    m = m_new;
    const int cnt = imin(64, j_end - j0);
    for (int jj = 0; jj < cnt; ++jj) {
      const float w = __shfl(pj, jj, 64);
      if (w == 0.f) continue;  // wave-uniform
      const float* vr = vb + (long long)(j0 + jj) * tok_stride;
#pragma unroll
      for (int t = 0; t < kEagerMaxD / 64; ++t)
        if (lane + 64 * t < p.d) o[t] = __builtin_fmaf(w, vr[lane + 64 * t], o[t]);
    }
  }
  float* orow = p.out + ((long long)b * p.n + i) * tok_stride + (long long)h * p.d;
  const float inv = 1.0f / l;  // a fully masked row: 0 * inf = NaN, as softmax over all -inf gives
#pragma unroll
  for (int t = 0; t < kEagerMaxD / 64; ++t)
    if (lane + 64 * t < p.d) orow[lane + 64 * t] = o[t] * inv;
}

}  // namespace tmx

extern "C" int tmx_eager_attention_f32(const float* q, const float* k, const float* v, float* out, int32_t batch, int32_t n,
                                       int32_t num_heads, int32_t head_dim, float scale, int32_t mode, int32_t si, int32_t sl,
                                       int32_t last, const void* bias, int32_t bias_dtype, int64_t bias_stride_b,
                                       int64_t bias_stride_h, int64_t bias_stride_i, void* stream) {
  using namespace tmx;
  TMX_CHECK_ARG(batch >= 0 && n >= 0 && num_heads > 0 && head_dim > 0, "tmx_eager_attention_f32: bad sizes");
  TMX_CHECK_SUPPORTED(head_dim % 4 == 0 && head_dim <= kEagerMaxD, "tmx_eager_attention_f32: head_dim %d not supported "
                      "(multiples of 4 up to %d)", head_dim, kEagerMaxD);
  TMX_CHECK_ARG(mode >= TMX_MODE_NONE && mode <= TMX_MODE_TRIANGLE, "tmx_eager_attention_f32: bad mode %d", mode);
  TMX_CHECK_ARG(scale > 0.f && scale < 3.0e38f, "tmx_eager_attention_f32: scale must be positive and finite");
  if (batch == 0 || n == 0) return TMX_OK;
  TMX_CHECK_ARG(q && k && v && out, "tmx_eager_attention_f32: NULL tensor");
  TMX_CHECK_ARG(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15) == 0,
                "tmx_eager_attention_f32: q / k / v must be 16-byte aligned");
  if (bias != nullptr)
    TMX_CHECK_ARG((bias_dtype == TMX_F32 || bias_dtype == TMX_BF16 || bias_dtype == TMX_F16) && bias_stride_b >= 0 &&
                      bias_stride_h >= 0 && bias_stride_i >= 0,
                  "tmx_eager_attention_f32: bad bias description");
  EagerF32Params p;
  p.q = q;
  p.k = k;
  p.v = v;
  p.out = out;
  p.bias = bias;
  p.bias_stride_b = bias_stride_b;
  p.bias_stride_h = bias_stride_h;
  p.bias_stride_i = bias_stride_i;
  p.bias_dtype = bias_dtype;
  p.batch = batch;
  p.n = n;
  p.heads = num_heads;
  p.d = head_dim;
  p.mode = mode;
  p.si = si;
  p.sl = sl;
  p.last = last;
  p.scale = scale;
  const long long rows = (long long)batch * num_heads * n;
  TMX_CHECK_SUPPORTED((rows + 3) / 4 < (1ll << 31), "tmx_eager_attention_f32: too many rows");
  eager_f32_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(p);
  TMX_CHECK_HIP(hipGetLastError());
  return TMX_OK;
}
