// Merge pass of the attention: the pieces of every split q-block (partial rows with their running maximum and sum)
// become the output rows.  Launched after the attention kernel on the same stream.
#include "tmx_attn.h"

namespace tmx {

// Merge the pieces of every split q-block: out = sum_p O_p 2^(m_p - M) / sum_p l_p 2^(m_p - M).
// One workgroup per (split q-block, head, 8 rows): a thread owns 4 output columns of one row.  The pass is a chain
// of dependent trips to memory, so it is kept short: the (max, sum) pairs AND the partial rows of up to 16 pieces
// are asked for in one go, together with the request's bounds (the partial buffers hold block_m rows per slot, so
// the loads need no bounds; only the store does); more than 16 pieces fold in by rounds with a running maximum.
template <typename TR, int D>
__global__ __launch_bounds__(8 * (D / 4)) void combine_kernel(const AttnParams p) {
  constexpr int C4 = D / 4;
  const int groups = (p.block_m + 7) / 8;
  const int qb = p.split_qb[blockIdx.x / groups];
  const int row = (blockIdx.x % groups) * 8 + threadIdx.x / C4;
  const int d4 = threadIdx.x % C4;
  const int head = blockIdx.y;
  const int slot0 = p.part_ptr[qb], np = p.part_ptr[qb + 1] - slot0;
  const int b = p.qb_meta[2 * qb], row0 = p.qb_meta[2 * qb + 1];
  if (row >= p.block_m) return;
  // the dense causal kernel wrote this request's rows: its pieces were never computed (before any of them is read:
  // a long causal prompt has thousands of pieces)
  if (p.dense_on && p.cu_q[b + 1] - p.cu_q[b] >= kDenseMinRows) return;
  const long long stride = (long long)p.H * p.block_m;  // partial rows between consecutive pieces
  const long long prow0 = ((long long)slot0 * p.H + head) * p.block_m + row;
  const float2* mlp = reinterpret_cast<const float2*>(p.partial_ml) + prow0;
  const float4* op = reinterpret_cast<const float4*>(p.partial_o + prow0 * D) + d4;
  const long long ostride = stride * C4;  // float4 elements between consecutive pieces
  float M = kMinusBig, l = 0.f;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int UN = 16;
  for (int pc = 0; pc < np; pc += UN) {
    float2 ml[UN];
    float4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int q = imin(pc + u, np - 1);
      ml[u] = mlp[(long long)q * stride];
      v[u] = __builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(op + (long long)q * ostride)));  // read once
    }
    float Mr = M;
#pragma unroll
    for (int u = 0; u < UN; ++u) Mr = fmaxf(Mr, ml[u].x);  // (a repeated last piece changes nothing)
    if (pc > 0) {  // later rounds: bring what is summed so far to the new maximum
      const float alpha = __builtin_amdgcn_exp2f(M - Mr);
      l *= alpha;
      o.x *= alpha;
      o.y *= alpha;
      o.z *= alpha;
      o.w *= alpha;
    }
    M = Mr;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const float w = (pc + u < np) ? __builtin_amdgcn_exp2f(ml[u].x - M) : 0.f;
      l += w * ml[u].y;
      o.x += w * v[u].x;
      o.y += w * v[u].y;
      o.z += w * v[u].z;
      o.w += w * v[u].w;
    }
  }
  const int q_start = p.cu_q[b];
  const int q_len = p.cu_q[b + 1] - q_start;
  if (row >= q_len - row0) return;
  const float inv = 1.0f / l;  // every piece empty for this row: 0 * inf = NaN, as for an unsplit row
  o.x *= inv;
  o.y *= inv;
  o.z *= inv;
  o.w *= inv;
  const long long off = (long long)(q_start + row0 + row) * p.out_stride_t + (long long)head * p.out_stride_h + 4 * d4;
  if (p.out_dtype == TMX_F32) {
    store_stream16(reinterpret_cast<float*>(p.out) + off, o);
  } else {
    s16x4 h;
    h[0] = TR::cvt(o.x);
    h[1] = TR::cvt(o.y);
    h[2] = TR::cvt(o.z);
    h[3] = TR::cvt(o.w);
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + off) = __builtin_bit_cast(uint2, h);
  }
}

template <typename TR, int D>
static int launch_combine_for(const AttnParams& p, int64_t num_split, hipStream_t s) {
  const unsigned groups = (unsigned)((p.block_m + 7) / 8);
  combine_kernel<TR, D><<<dim3((unsigned)num_split * groups, (unsigned)p.H), 8 * (D / 4), 0, s>>>(p);
  TMX_CHECK_HIP(hipGetLastError());
  return TMX_OK;
}

int launch_combine_kernel(const AttnParams& p, bool bf16, int head_dim, int64_t num_split, hipStream_t s) {
  if (bf16) return head_dim == 64 ? launch_combine_for<Bf16Traits, 64>(p, num_split, s) : launch_combine_for<Bf16Traits, 128>(p, num_split, s);
  return head_dim == 64 ? launch_combine_for<F16Traits, 64>(p, num_split, s) : launch_combine_for<F16Traits, 128>(p, num_split, s);
}

}  // namespace tmx
