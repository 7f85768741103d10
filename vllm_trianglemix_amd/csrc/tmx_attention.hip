// Masked attention for gfx950 (MI355X): prefill and decode, grouped heads,
// ragged batch, contiguous or paged K/V, driven by the packed block index.
//
// Replaces the dense eager path of the reference
// (vllm/attention/ascend_trianglemix.py:88-106: matmul -> +mask -> softmax ->
// matmul over all N^2 pairs) with a flash-style kernel that only visits the
// key tiles the index lists and re-evaluates the mask predicate per element
// inside them.
//
// Work decomposition
//   grid.x = work item (q-block piece from the index), grid.y = K/V head x
//   head chunk.  A workgroup is 4 waves; wave w owns 16 query rows of one
//   query head: head = group base + (w % heads_per_wg), row block = w / heads_per_wg,
//   so all waves of a workgroup share the same K/V head and the K/V tiles are
//   staged once in LDS for all of them.
//
// Per 32-key tile and wave (all MFMA are 16x16x32, fp32 accumulate):
//   S^T[key][row]  = K_tile . Q^T      A = K rows from LDS (ds_read_b128, XOR-swizzled
//                                      16-byte chunks), B = Q fragments held in registers
//   online softmax  per lane: the lane's column of S^T is ONE query row
//                   (row = lane & 15), its 8 values are 8 keys; the row maximum
//                   needs two cross-lane steps (lanes l, l^16, l^32, l^48)
//   O^T[d][row]    += V_tile^T . P^T   A = V columns by ds_read_b64_tr_b16 (hardware
//                                      transpose), B = P packed to 16 bit in registers
// Both products sum over an index whose order is free (head dim for S, keys for
// O), so the fragments are laid out for the loads: Q/K fragments are contiguous
// 16-byte pieces of a row, P needs no cross-lane movement between the two
// products, and every lane ends with 8 consecutive output columns per store.
#include "tmx_common.h"

namespace tmx {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

struct Bf16Traits {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ short cvt(float x) {
    __bf16 h = (__bf16)x;
    return __builtin_bit_cast(short, h);
  }
  static __device__ __forceinline__ float to_float(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
};

struct F16Traits {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ short cvt(float x) {
    _Float16 h = (_Float16)x;
    return __builtin_bit_cast(short, h);
  }
  static __device__ __forceinline__ float to_float(uint16_t b) {
    return (float)__builtin_bit_cast(_Float16, b);
  }
};

struct AttnParams {
  const void* q;
  void* out;
  const void* k;
  const void* v;
  const void* bias;
  const int32_t* block_table;
  const int32_t* cu_k;
  const int32_t* cu_q;
  const int32_t* seq_lens;
  const int32_t* col_idx;
  const int32_t* qb_meta;
  const int32_t* items;
  const int32_t* part_ptr;
  const int32_t* split_qb;
  float* partial_o;
  float* partial_ml;
  int64_t q_stride_t, q_stride_h, out_stride_t, out_stride_h;
  int64_t kv_stride_page, kv_stride_t, kv_stride_h, block_table_stride, bias_stride;
  int page_size, page_shift;  // page_shift >= 0 when page_size is a power of two
  int H, Hkv, G, Gw, block_m;
  int mode, si, sl, last, tri_min_len;
  int bias_rows, bias_cols, bias_dtype, out_dtype;
  float scale_log2;
};

// 16-byte chunk swizzle of the K image: rows of D 16-bit elements, read by
// ds_read_b128 with 16 different rows per lane group.
template <int D>
__device__ __forceinline__ int k_chunk_swizzle(int chunk, int row) {
  if constexpr (D == 128) return chunk ^ (row & 15);
  else return chunk ^ ((row >> 1) & 7);  // D == 64: two rows share a 256-byte bank row
}

// 8-byte slot swizzle of the V image, read by ds_read_b64_tr_b16: the 32 lanes
// of a half-wave read 8 key rows x 4 slots; the XOR spreads the 8 rows over the
// slot bits the read pattern leaves free.
template <int D>
__device__ __forceinline__ int v_slot_xor(int row) {
  if constexpr (D == 128) return (row & 1) | (((row >> 1) & 3) << 3);
  else return ((row >> 1) & 1) | (((row >> 2) & 1) << 3);
}

// slot (8-byte unit) of output block db, chunk p of the transposed read
template <int D>
__device__ __forceinline__ constexpr int v_db_bits(int db) {
  return 8 * (db >> 1) + (db & 1);
}

__device__ __forceinline__ float load_bias(const void* bias, int dtype, int64_t off) {
  if (dtype == TMX_F32) return reinterpret_cast<const float*>(bias)[off];
  const uint16_t b = reinterpret_cast<const uint16_t*>(bias)[off];
  return dtype == TMX_BF16 ? Bf16Traits::to_float(b) : F16Traits::to_float(b);
}

template <typename TR, int D, bool PAGED, bool HAS_BIAS>
__global__ __launch_bounds__(256) void attn_kernel(AttnParams p) {
  constexpr int KS = D / 32;          // k-steps of S^T = K Q^T
  constexpr int DB = D / 16;          // 16-column output blocks
  constexpr int ROWB = D * 2;         // bytes per K/V row in LDS
  constexpr int CPR = D / 8;          // 16-byte chunks per row
  constexpr int TILE = 32 * ROWB;     // bytes of one K (or V) tile
  constexpr int LPT = 32 * CPR / 256; // 16-byte chunks each thread stages per tile and tensor
  using frag = typename TR::frag;

  __shared__ __attribute__((aligned(16))) char lds[4 * TILE];  // [buffer][K | V]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int qi = lane & 15, g = lane >> 4;

  const int4 item = reinterpret_cast<const int4*>(p.items)[blockIdx.x];
  const int qb = item.x, col_start = item.y, ntiles = item.z, slot = item.w;
  const int b = p.qb_meta[2 * qb], row0 = p.qb_meta[2 * qb + 1];
  const int q_start = p.cu_q[b];
  const int q_len = p.cu_q[b + 1] - q_start;
  const int n = p.seq_lens[b];
  const int ctx = n - q_len;
  const int mode = effective_mode(p.mode, n, p.tri_min_len);

  const int chunks = p.G / p.Gw;
  const int kvh = blockIdx.y / chunks;
  const int hc = blockIdx.y - kvh * chunks;
  const int hg = wave % p.Gw, rb = wave / p.Gw;
  const int head = kvh * p.G + hc * p.Gw + hg;

  const int wrow0 = row0 + 16 * rb;                   // first row of this wave (inside the request)
  const int wrow1 = imin(wrow0 + 16, q_len);          // one past its last valid row
  const bool wave_has_rows = wrow0 < q_len;
  const int my_row = wrow0 + qi;
  const bool row_valid = my_row < q_len;
  const int pos_i = ctx + my_row;

  // ---- Q fragments (B operand of S^T): 16 contiguous bytes per k-step ----
  frag qf[KS];
  {
    const uint16_t* qp = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)(q_start + my_row) * p.q_stride_t +
                         (int64_t)head * p.q_stride_h + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 w = make_uint4(0, 0, 0, 0);
      if (row_valid) w = *reinterpret_cast<const uint4*>(qp + 32 * ks);
      qf[ks] = __builtin_bit_cast(frag, w);
    }
  }

  // ---- K/V staging: global -> registers -> LDS ----
  const int64_t k_row0 = PAGED ? 0 : (int64_t)p.cu_k[b];
  const int32_t* bt = PAGED ? p.block_table + (int64_t)b * p.block_table_stride : nullptr;
  const uint16_t* kbase = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)kvh * p.kv_stride_h;
  const uint16_t* vbase = reinterpret_cast<const uint16_t*>(p.v) + (int64_t)kvh * p.kv_stride_h;
  uint4 kreg[LPT], vreg[LPT];

  auto issue_loads = [&](int tile) {
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const int c = tid + 256 * u;
      const int row = c / CPR, cc = c % CPR;
      const int j = tile * kBlockN + row;
      kreg[u] = make_uint4(0, 0, 0, 0);
      vreg[u] = make_uint4(0, 0, 0, 0);
      if (j < n) {
        int64_t off;
        if constexpr (PAGED) {
          const int pg = p.page_shift >= 0 ? (j >> p.page_shift) : (j / p.page_size);
          const int in_pg = j - pg * p.page_size;
          off = (int64_t)bt[pg] * p.kv_stride_page + (int64_t)in_pg * p.kv_stride_t;
        } else {
          off = (k_row0 + j) * p.kv_stride_t;
        }
        off += cc * 8;
        kreg[u] = *reinterpret_cast<const uint4*>(kbase + off);
        vreg[u] = *reinterpret_cast<const uint4*>(vbase + off);
      }
    }
  };
  auto write_lds = [&](int buf) {
    char* kb_ = lds + buf * 2 * TILE;
    char* vb_ = kb_ + TILE;
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const int c = tid + 256 * u;
      const int row = c / CPR, cc = c % CPR;
      *reinterpret_cast<uint4*>(kb_ + row * ROWB + 16 * k_chunk_swizzle<D>(cc, row)) = kreg[u];
      const int x = v_slot_xor<D>(row);
      uint4 w = vreg[u];
      if (x & 1) w = make_uint4(w.z, w.w, w.x, w.y);
      *reinterpret_cast<uint4*>(vb_ + row * ROWB + 8 * (((2 * cc) ^ x) & ~1)) = w;
    }
  };

  // ---- per-lane LDS read addresses ----
  // K: row 16*kb + qi, chunk 4*ks + g (swizzled)
  int k_off[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_off[ks] = qi * ROWB + 16 * k_chunk_swizzle<D>(4 * ks + g, qi);
  // V transposed read: lane 4*q' + pp of each 16-lane group addresses key row 4*g + q' (+16 for the
  // second half of the tile), slot 2*pp + (bits of the output block)
  const int vq = (lane & 15) >> 2, vp = lane & 3;
  const int v_row = 4 * g + vq;
  const int v_x = v_slot_xor<D>(v_row);  // identical for row and row + 16
  const int v_base = v_row * ROWB + 8 * (2 * vp);

  f32x4 acc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) acc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -__builtin_inff();
  float l_run = 0.f;

  const int i_min = ctx + wrow0, i_max = ctx + wrow1 - 1;  // positions of this wave's rows

  if (ntiles > 0) {
    issue_loads(p.col_idx[col_start]);
    write_lds(0);
  }
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int tile = p.col_idx[col_start + t];
    const bool more = t + 1 < ntiles;
    if (more) issue_loads(p.col_idx[col_start + t + 1]);

    // does any (row, key) pair of this wave fall into the tile?
    bool needed = wave_has_rows;
    if (needed && mode != TMX_MODE_NONE) {
      const int j0 = tile * kBlockN, j1 = j0 + kBlockN - 1;
      needed = j0 <= i_max;
      if (needed && mode == TMX_MODE_TRIANGLE && !((n - i_max) < p.last))
        needed = (j0 <= imin(i_max, p.si)) || (p.sl >= 0 && j1 >= i_min - p.sl);
    }

    if (needed) {
      const char* kt = lds + (t & 1) * 2 * TILE;
      const char* vt = kt + TILE;
      f32x4 s[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        s[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const uint4 w = *reinterpret_cast<const uint4*>(kt + kb * 16 * ROWB + k_off[ks]);
          s[kb] = TR::mfma(__builtin_bit_cast(frag, w), qf[ks], s[kb]);
        }
      }
      // scale, bias, mask
      float x[8];
      float tmax = -__builtin_inff();
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = tile * kBlockN + 16 * kb + 4 * g + r;
          float val = s[kb][r] * p.scale_log2;
          bool ok = row_valid && j < n && allowed(mode, pos_i, j, n, p.si, p.sl, p.last);
          if constexpr (HAS_BIAS) {
            if (ok && pos_i < p.bias_rows && j < p.bias_cols)
              val += kLog2e * load_bias(p.bias, p.bias_dtype, (int64_t)pos_i * p.bias_stride + j);
          }
          val = ok ? val : -__builtin_inff();
          x[kb * 4 + r] = val;
          tmax = fmaxf(tmax, val);
        }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m_run, tmax);
      float alpha = 1.f, m_safe = 0.f;
      if (m_new != -__builtin_inff()) {
        alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_safe = m_new;
      }
      m_run = m_new;
      float psum = 0.f;
      s16x8 pk;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float pe = __builtin_amdgcn_exp2f(x[e] - m_safe);
        psum += pe;
        pk[e] = TR::cvt(pe);
      }
      l_run = l_run * alpha + psum;
      const frag pf = __builtin_bit_cast(frag, pk);
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        acc[db] *= alpha;
        const int so = 8 * (v_db_bits<D>(db) ^ v_x);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(vt + v_base + so));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(vt + 16 * ROWB + v_base + so));
        const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        acc[db] = TR::mfma(__builtin_bit_cast(frag, vv), pf, acc[db]);
      }
    }

    if (more) write_lds((t + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue ----
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  if (!row_valid) return;

  if (slot >= 0) {
    // piece of a split q-block: unnormalised fp32 accumulator + (max, sum) for the combine pass
    const int64_t prow = ((int64_t)slot * p.H + head) * p.block_m + (16 * rb + qi);
    float* po = p.partial_o + prow * D;
#pragma unroll
    for (int c = 0; c < DB / 2; ++c) {
      float4* dst = reinterpret_cast<float4*>(po + 32 * c + 8 * g);
      dst[0] = make_float4(acc[2 * c][0], acc[2 * c][1], acc[2 * c][2], acc[2 * c][3]);
      dst[1] = make_float4(acc[2 * c + 1][0], acc[2 * c + 1][1], acc[2 * c + 1][2], acc[2 * c + 1][3]);
    }
    if (g == 0) {
      p.partial_ml[prow * 2] = m_run;
      p.partial_ml[prow * 2 + 1] = l_tot;
    }
    return;
  }

  const float inv = 1.0f / l_tot;  // a fully masked row gives 0 * inf = NaN, as softmax over all -inf does
  const int64_t orow = (int64_t)(q_start + my_row) * p.out_stride_t + (int64_t)head * p.out_stride_h;
  if (p.out_dtype == TMX_F32) {
    float* po = reinterpret_cast<float*>(p.out) + orow;
#pragma unroll
    for (int c = 0; c < DB / 2; ++c) {
      float4* dst = reinterpret_cast<float4*>(po + 32 * c + 8 * g);
      dst[0] = make_float4(acc[2 * c][0] * inv, acc[2 * c][1] * inv, acc[2 * c][2] * inv, acc[2 * c][3] * inv);
      dst[1] = make_float4(acc[2 * c + 1][0] * inv, acc[2 * c + 1][1] * inv, acc[2 * c + 1][2] * inv,
                           acc[2 * c + 1][3] * inv);
    }
  } else {
    uint16_t* po = reinterpret_cast<uint16_t*>(p.out) + orow;
#pragma unroll
    for (int c = 0; c < DB / 2; ++c) {
      s16x8 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = TR::cvt(acc[2 * c][r] * inv);
        o[4 + r] = TR::cvt(acc[2 * c + 1][r] * inv);
      }
      *reinterpret_cast<uint4*>(po + 32 * c + 8 * g) = __builtin_bit_cast(uint4, o);
    }
  }
}

// Merge the pieces of every split q-block: out = sum_p O_p 2^(m_p - M) / sum_p l_p 2^(m_p - M).
template <typename TR, int D>
__global__ __launch_bounds__(256) void combine_kernel(AttnParams p) {
  const int qb = p.split_qb[blockIdx.x];
  const int head = blockIdx.y;
  const int slot0 = p.part_ptr[qb], np = p.part_ptr[qb + 1] - slot0;
  const int b = p.qb_meta[2 * qb], row0 = p.qb_meta[2 * qb + 1];
  const int q_start = p.cu_q[b];
  const int q_len = p.cu_q[b + 1] - q_start;
  const int rows = imin(p.block_m, q_len - row0);
  constexpr int C4 = D / 4;
  for (int idx = threadIdx.x; idx < rows * C4; idx += 256) {
    const int row = idx / C4, d4 = idx - row * C4;
    float M = -__builtin_inff();
    for (int pc = 0; pc < np; ++pc) {
      const int64_t prow = ((int64_t)(slot0 + pc) * p.H + head) * p.block_m + row;
      M = fmaxf(M, p.partial_ml[prow * 2]);
    }
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    float l = 0.f;
    for (int pc = 0; pc < np; ++pc) {
      const int64_t prow = ((int64_t)(slot0 + pc) * p.H + head) * p.block_m + row;
      const float mp = p.partial_ml[prow * 2];
      if (mp == -__builtin_inff()) continue;  // this piece saw no attended key for the row
      const float w = __builtin_amdgcn_exp2f(mp - M);
      l += w * p.partial_ml[prow * 2 + 1];
      const float4 v = reinterpret_cast<const float4*>(p.partial_o + prow * D)[d4];
      o.x += w * v.x;
      o.y += w * v.y;
      o.z += w * v.z;
      o.w += w * v.w;
    }
    const float inv = 1.0f / l;
    o.x *= inv;
    o.y *= inv;
    o.z *= inv;
    o.w *= inv;
    const int64_t off = (int64_t)(q_start + row0 + row) * p.out_stride_t + (int64_t)head * p.out_stride_h + 4 * d4;
    if (p.out_dtype == TMX_F32) {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + off) = o;
    } else {
      s16x4 h;
      h[0] = TR::cvt(o.x);
      h[1] = TR::cvt(o.y);
      h[2] = TR::cvt(o.z);
      h[3] = TR::cvt(o.w);
      *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + off) = __builtin_bit_cast(uint2, h);
    }
  }
}

template <typename TR, int D>
static int launch(const AttnParams& p, bool paged, bool has_bias, dim3 grid, int64_t num_split, uint32_t flags,
                  hipStream_t s) {
  if (grid.x > 0 && !(flags & TMX_ATTN_ONLY_COMBINE)) {
    if (paged) {
      if (has_bias) attn_kernel<TR, D, true, true><<<grid, 256, 0, s>>>(p);
      else attn_kernel<TR, D, true, false><<<grid, 256, 0, s>>>(p);
    } else {
      if (has_bias) attn_kernel<TR, D, false, true><<<grid, 256, 0, s>>>(p);
      else attn_kernel<TR, D, false, false><<<grid, 256, 0, s>>>(p);
    }
    TMX_CHECK_HIP(hipGetLastError());
  }
  if (num_split > 0 && !(flags & TMX_ATTN_SKIP_COMBINE)) {
    combine_kernel<TR, D><<<dim3((unsigned)num_split, (unsigned)p.H), 256, 0, s>>>(p);
    TMX_CHECK_HIP(hipGetLastError());
  }
  return TMX_OK;
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
  TMX_CHECK_ARG(a->batch >= 0 && a->total_q >= 0 && a->num_items >= 0 && a->num_partials >= 0 &&
                    a->num_split_qblocks >= 0,
                "tmx_attention: negative size");
  if (a->total_q == 0 || a->num_items == 0) return TMX_OK;
  const tmx_index_t* ix = a->index;
  TMX_CHECK_ARG(ix != nullptr, "tmx_attention: index is NULL");
  const int G = a->num_heads / a->num_kv_heads;
  TMX_CHECK_ARG(ix->block_n == kBlockN && ix->block_m == block_m_for_group(G),
                "tmx_attention: index built with block_m %d, block_n %d; this head grouping needs %d, %d",
                ix->block_m, ix->block_n, block_m_for_group(G), kBlockN);
  TMX_CHECK_ARG(ix->mode == a->mode && ix->si == a->si && ix->sl == a->sl && ix->last == a->last &&
                    ix->batch == a->batch,
                "tmx_attention: mask parameters differ from the ones the index was built with");
  TMX_CHECK_ARG(a->num_items <= ix->cap_items && a->num_split_qblocks <= ix->cap_split,
                "tmx_attention: num_items / num_split_qblocks exceed the index capacities");
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
  if (paged) TMX_CHECK_ARG(a->page_size > 0, "tmx_attention: page_size %d", a->page_size);
  else TMX_CHECK_ARG(a->cu_seqlens_k != nullptr, "tmx_attention: contiguous K/V needs cu_seqlens_k");
  const bool has_bias = a->bias != nullptr;
  if (has_bias)
    TMX_CHECK_ARG((a->bias_dtype == TMX_F32 || a->bias_dtype == TMX_BF16 || a->bias_dtype == TMX_F16) &&
                      a->bias_rows > 0 && a->bias_cols > 0 && a->bias_stride >= a->bias_cols,
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
  p.q_stride_t = a->q_stride_t;
  p.q_stride_h = a->q_stride_h;
  p.out_stride_t = a->out_stride_t;
  p.out_stride_h = a->out_stride_h;
  p.kv_stride_page = a->kv_stride_page;
  p.kv_stride_t = a->kv_stride_t;
  p.kv_stride_h = a->kv_stride_h;
  p.block_table_stride = a->block_table_stride;
  p.bias_stride = a->bias_stride;
  p.page_size = paged ? a->page_size : 1;
  p.page_shift = -1;
  if (paged && (a->page_size & (a->page_size - 1)) == 0) p.page_shift = __builtin_ctz(a->page_size);
  p.H = a->num_heads;
  p.Hkv = a->num_kv_heads;
  p.G = G;
  p.Gw = heads_per_wg(G);
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

  const dim3 grid((unsigned)a->num_items, (unsigned)(a->num_kv_heads * (G / p.Gw)));
  TMX_CHECK_ARG(grid.y <= 65535, "tmx_attention: too many K/V head chunks (%u)", grid.y);
  hipStream_t s = (hipStream_t)stream;
  if (a->io_dtype == TMX_BF16) {
    if (a->head_dim == 128) return launch<Bf16Traits, 128>(p, paged, has_bias, grid, a->num_split_qblocks, a->flags, s);
    return launch<Bf16Traits, 64>(p, paged, has_bias, grid, a->num_split_qblocks, a->flags, s);
  }
  if (a->head_dim == 128) return launch<F16Traits, 128>(p, paged, has_bias, grid, a->num_split_qblocks, a->flags, s);
  return launch<F16Traits, 64>(p, paged, has_bias, grid, a->num_split_qblocks, a->flags, s);
}
