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
typedef __attribute__((ext_vector_type(4))) int i32x4;
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

// Read through the scalar (constant) path: the index, the block table and the
// batch description are wave-uniform and never written while the kernel runs.
template <typename T>
__device__ __forceinline__ T sload(const T* ptr) {
  return *reinterpret_cast<const __attribute__((address_space(4))) T*>(
      (const __attribute__((address_space(4))) void*)ptr);
}

// 16-byte chunk swizzle of the K image: rows of D 16-bit elements, read by
// ds_read_b128 with 16 different rows per lane group.
template <int D>
__device__ __forceinline__ int k_chunk_swizzle(int row) {
  if constexpr (D == 128) return row & 15;
  else return (row >> 1) & 7;  // D == 64: two rows share a 256-byte bank row
}

// 16-byte chunk swizzle of the V image, read by ds_read_b64_tr_b16: the 32
// lanes of a half-wave read 8 key rows x 32 contiguous bytes; the XOR moves
// the 8 rows to 8 different 32-byte groups of the 256-byte bank row.
template <int D>
__device__ __forceinline__ int v_chunk_swizzle(int row) {
  if constexpr (D == 128) return (row & 7) << 1;
  else return ((row >> 1) & 3) << 1;
}

__device__ __forceinline__ float load_bias(const void* bias, int dtype, int64_t off) {
  if (dtype == TMX_F32) return reinterpret_cast<const float*>(bias)[off];
  const uint16_t b = reinterpret_cast<const uint16_t*>(bias)[off];
  return dtype == TMX_BF16 ? Bf16Traits::to_float(b) : F16Traits::to_float(b);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// Wait until at most `tiles` of this wave's most recent tile transfers are
// still in flight (DPT LDS-DMA instructions per tile and wave).
template <int DPT, int MAXT>
__device__ __forceinline__ void wait_tiles_in_flight(int tiles) {
  if constexpr (MAXT >= 3) {
    if (tiles >= 3) { wait_vmcnt<3 * DPT>(); return; }
  }
  if constexpr (MAXT >= 2) {
    if (tiles == 2) { wait_vmcnt<2 * DPT>(); return; }
  }
  if constexpr (MAXT >= 1) {
    if (tiles == 1) { wait_vmcnt<DPT>(); return; }
  }
  wait_vmcnt<0>();
}

// One LDS-DMA instruction: every lane fetches 16 bytes from its own global
// address; the wave's 1 KiB lands at LDS byte address lds_addr + 16 * lane.
// Issued from inline asm on purpose: the compiler would otherwise wait for ALL
// transfers in flight (vmcnt(0)) before every LDS read; here the waits are
// counted by hand (wait_tiles_in_flight) and followed by a workgroup barrier.
__device__ __forceinline__ void glds16(const void* src, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_addr)
      : "memory");
}

template <typename TR, int D, int NS, bool PAGED, bool HAS_BIAS>
__global__ __launch_bounds__(256) void attn_kernel(AttnParams p) {
  constexpr int KS = D / 32;            // k-steps of S^T = K Q^T
  constexpr int DB = D / 16;            // 16-column output blocks
  constexpr int ROWB = D * 2;           // bytes per K/V row in LDS
  constexpr int CPR = D / 8;            // 16-byte chunks per row
  constexpr int TILE = 32 * ROWB;       // bytes of one K (or V) tile
  constexpr int RPP = 1024 / ROWB;      // rows per 1-KiB LDS-DMA piece (one wave instruction)
  constexpr int U = TILE / 1024 / 4;    // pieces per wave, tile and tensor
  constexpr int DPT = 2 * U;            // LDS-DMA instructions per wave and tile (K and V)
  using frag = typename TR::frag;

  __shared__ __attribute__((aligned(16))) char lds[NS * 2 * TILE];  // ring of [K | V] tile slots

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 15, g = lane >> 4;

  // ---- work item: 8 ints, read through the scalar path ----
  const i32x4 it0 = sload(reinterpret_cast<const i32x4*>(p.items) + 2 * blockIdx.x);
  const i32x4 it1 = sload(reinterpret_cast<const i32x4*>(p.items) + 2 * blockIdx.x + 1);
  const int col_start = it0.y, ntiles = it0.z, slot = it0.w;
  const int b = it1.x, q_tok0 = it1.y, pos0 = it1.z, n = it1.w;
  const int rows_valid = imin(p.block_m, n - pos0);
  const int mode = effective_mode(p.mode, n, p.tri_min_len);

  const int chunks = p.G / p.Gw;
  const int kvh = blockIdx.y / chunks;
  const int hc = blockIdx.y - kvh * chunks;
  const int hg = wave % p.Gw, rb = wave / p.Gw;
  const int head = kvh * p.G + hc * p.Gw + hg;

  const int wrow0 = 16 * rb;                           // first row of this wave inside the q-block
  const int wrow1 = imin(wrow0 + 16, rows_valid);      // one past its last valid row
  const bool wave_has_rows = wrow0 < rows_valid;
  const int my_row = wrow0 + qi;
  const bool row_valid = my_row < rows_valid;
  const int pos_i = pos0 + my_row;

  // ---- Q fragments (B operand of S^T): 16 contiguous bytes per k-step ----
  frag qf[KS];
  {
    const uint16_t* qp = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)(q_tok0 + my_row) * p.q_stride_t +
                         (int64_t)head * p.q_stride_h + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 w = make_uint4(0, 0, 0, 0);
      if (row_valid) w = *reinterpret_cast<const uint4*>(qp + 32 * ks);
      qf[ks] = __builtin_bit_cast(frag, w);
    }
  }

  // ---- K/V tiles: global -> LDS ring by LDS-DMA, NS - 1 tiles ahead ----
  const int64_t k_row0 = PAGED ? 0 : (int64_t)sload(p.cu_k + b);
  const int32_t* bt = PAGED ? p.block_table + (int64_t)b * p.block_table_stride : nullptr;
  const uint16_t* kbase = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)kvh * p.kv_stride_h;
  const uint16_t* vbase = reinterpret_cast<const uint16_t*>(p.v) + (int64_t)kvh * p.kv_stride_h;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)lds));
  const int last_key = n - 1;
  const int last_page = PAGED ? (last_key >> p.page_shift) : 0;

  auto issue_tile = [&](int tile, int ring_slot) {
    const unsigned kdst = lds_base + ring_slot * 2 * TILE;
    const unsigned vdst = kdst + TILE;
    int64_t pg_off0 = 0, pg_off1 = 0;
    int pg_first = 0;
    if constexpr (PAGED) {
      pg_first = (tile * kBlockN) >> p.page_shift;
      pg_off0 = (int64_t)sload(bt + pg_first) * p.kv_stride_page;
      pg_off1 = (int64_t)sload(bt + imin(pg_first + 1, last_page)) * p.kv_stride_page;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int piece = wave * U + u;
      const int row = piece * RPP + lane / CPR;
      const int cc = lane % CPR;
      // keys past the end of the sequence are masked later; clamp their source to a valid row
      const int j = imin(tile * kBlockN + row, last_key);
      int64_t off;
      if constexpr (PAGED) {
        const int pg = j >> p.page_shift;
        off = (pg == pg_first ? pg_off0 : pg_off1) + (int64_t)(j & (p.page_size - 1)) * p.kv_stride_t;
      } else {
        off = (k_row0 + j) * p.kv_stride_t;
      }
      glds16(kbase + off + 8 * (cc ^ k_chunk_swizzle<D>(row)), kdst + piece * 1024);
      glds16(vbase + off + 8 * (cc ^ v_chunk_swizzle<D>(row)), vdst + piece * 1024);
    }
  };

  // ---- per-lane LDS read addresses ----
  // K: row 16*kb + qi, chunk 4*ks + g (swizzled)
  int k_off[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_off[ks] = qi * ROWB + 16 * ((4 * ks + g) ^ k_chunk_swizzle<D>(qi));
  // V transposed read: lane 4*q' + pp of each 16-lane group addresses key row 4*g + q' (+16 for the
  // second half of the tile), columns 16*db + 4*pp .. +3
  const int vq = (lane & 15) >> 2, vp = lane & 3;
  const int v_row = 4 * g + vq;
  const int v_x = v_chunk_swizzle<D>(v_row) >> 1;  // XOR on the 32-byte group index == the output block db
  const int v_base = v_row * ROWB + 8 * vp;

  f32x4 acc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) acc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -__builtin_inff();
  float l_run = 0.f;

  const int i_min = pos0 + wrow0, i_max = pos0 + wrow1 - 1;  // positions of this wave's rows
  const bool wave_any_last = (n - i_max) < p.last;           // some row is in the dense trailing region
  const bool wave_all_last = (n - i_min) < p.last;           // every row is
  // per-lane form of the predicate: attended iff j <= key_hi and not (hole_lo < j < hole_lo + 1 + hole_width)
  int key_hi = -1;
  int hole_lo = 0;
  unsigned hole_width = 0;
  if (row_valid) {
    key_hi = (mode == TMX_MODE_NONE) ? n - 1 : imin(pos_i, n - 1);
    if (mode == TMX_MODE_TRIANGLE && !((n - pos_i) < p.last)) {
      // keys si < j < pos_i - sl are the excluded middle section
      hole_lo = p.si;
      const long long w = (long long)pos_i - p.sl - (long long)p.si - 1;
      hole_width = w > 0 ? (unsigned)w : 0u;
      if (p.si < -1) {  // a negative sink size excludes nothing below key 0; start the hole at -1
        hole_lo = -1;
        const long long w2 = (long long)pos_i - p.sl;
        hole_width = w2 > 0 ? (unsigned)w2 : 0u;
      }
    }
  }

  int issued = 0;
  for (; issued < ntiles && issued < NS - 1; ++issued) issue_tile(sload(p.col_idx + col_start + issued), issued);
  // Settle the Q registers here, once: a first use inside the loop would make the compiler wait for
  // every transfer in flight on each iteration.
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));

  for (int t = 0; t < ntiles; ++t) {
    const int tile = sload(p.col_idx + col_start + t);
    wait_tiles_in_flight<DPT, NS - 2>(issued - 1 - t);
    __builtin_amdgcn_s_barrier();
    if (issued < ntiles) {
      issue_tile(sload(p.col_idx + col_start + issued), issued % NS);
      ++issued;
    }

    // Classify the tile against this wave's rows (all wave-uniform): skip it when no (row, key) pair
    // is attended, and skip the per-element predicate when every pair is.
    const int j0 = tile * kBlockN, j1 = j0 + kBlockN - 1;
    bool needed = wave_has_rows, full = false;
    if (mode == TMX_MODE_NONE) {
      full = j1 < n;
    } else if (needed) {
      needed = j0 <= i_max;
      full = j1 <= i_min;
      if (mode == TMX_MODE_TRIANGLE && !wave_all_last) {
        if (!wave_any_last) needed = needed && ((j0 <= imin(i_max, p.si)) || (p.sl >= 0 && j1 >= i_min - p.sl));
        full = full && (j1 <= p.si || j0 >= i_max - p.sl);
      }
    }

    if (needed) {
      const char* kt = lds + (t % NS) * 2 * TILE;
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
      // scale, bias, mask (branch free: key j is attended iff j <= key_hi and not hole_lo < j < hole_hi)
      float x[8];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[kb * 4 + r] = s[kb][r] * p.scale_log2;
      if constexpr (HAS_BIAS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int j = j0 + 16 * (e >> 2) + 4 * g + (e & 3);
          if (row_valid && j < p.bias_cols && pos_i < p.bias_rows && j <= key_hi)
            x[e] += kLog2e * load_bias(p.bias, p.bias_dtype, (int64_t)pos_i * p.bias_stride + j);
        }
      }
      if (!full) {
        const int jl = j0 + 4 * g;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int j = jl + 16 * (e >> 2) + (e & 3);
          const bool in_hole = (unsigned)(j - hole_lo - 1) < hole_width;
          x[e] = (j <= key_hi && !in_hole) ? x[e] : -__builtin_inff();
        }
      }
      float tmax = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), fmaxf(fmaxf(x[4], x[5]), fmaxf(x[6], x[7])));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m_run, tmax);
      // alpha rescales what was accumulated under the old maximum; nothing was when m_run is -inf
      float alpha = 1.f, m_safe = 0.f;
      if (m_new != -__builtin_inff()) {
        m_safe = m_new;
        if (m_run != -__builtin_inff()) alpha = __builtin_amdgcn_exp2f(m_run - m_new);
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
      if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {  // wave-uniform: some row's maximum moved
#pragma unroll
        for (int db = 0; db < DB; ++db) acc[db] *= alpha;
      }
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const int so = 32 * (db ^ v_x);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(vt + v_base + so));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(vt + 16 * ROWB + v_base + so));
        const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        acc[db] = TR::mfma(__builtin_bit_cast(frag, vv), pf, acc[db]);
      }
    }
  }

  // ---- epilogue: lane (qi, g) holds columns 16*db + 4*g .. +3 of row qi for every db ----
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);

  if (slot >= 0) {
    // piece of a split q-block: unnormalised fp32 accumulator + (max, sum) for the combine pass
    if (!row_valid) return;
    const int64_t prow = ((int64_t)slot * p.H + head) * p.block_m + my_row;
    float* po = p.partial_o + prow * D + 4 * g;
#pragma unroll
    for (int db = 0; db < DB; ++db)
      *reinterpret_cast<float4*>(po + 16 * db) = make_float4(acc[db][0], acc[db][1], acc[db][2], acc[db][3]);
    if (g == 0) {
      p.partial_ml[prow * 2] = m_run;
      p.partial_ml[prow * 2 + 1] = l_tot;
    }
    return;
  }

  const float inv = 1.0f / l_tot;  // a fully masked row gives 0 * inf = NaN, as softmax over all -inf does
  const int64_t orow = (int64_t)(q_tok0 + my_row) * p.out_stride_t + (int64_t)head * p.out_stride_h;
  if (p.out_dtype == TMX_F32) {
    if (!row_valid) return;
    float* po = reinterpret_cast<float*>(p.out) + orow + 4 * g;
#pragma unroll
    for (int db = 0; db < DB; ++db)
      *reinterpret_cast<float4*>(po + 16 * db) =
          make_float4(acc[db][0] * inv, acc[db][1] * inv, acc[db][2] * inv, acc[db][3] * inv);
  } else {
    // 16-bit output: lanes l and l ^ 16 exchange one 8-byte piece per pair of output blocks so that
    // every lane stores 16 contiguous bytes (8 columns)
    uint16_t* po = reinterpret_cast<uint16_t*>(p.out) + orow + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g);
#pragma unroll
    for (int c = 0; c < DB / 2; ++c) {
      s16x4 a, bq;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[r] = TR::cvt(acc[2 * c][r] * inv);
        bq[r] = TR::cvt(acc[2 * c + 1][r] * inv);
      }
      uint2 au = __builtin_bit_cast(uint2, a), bu = __builtin_bit_cast(uint2, bq);
      auto r0 = __builtin_amdgcn_permlane16_swap(au.x, bu.x, false, false);
      auto r1 = __builtin_amdgcn_permlane16_swap(au.y, bu.y, false, false);
      // even 16-lane rows: [own block 2c | partner's block 2c]; odd rows: [partner's block 2c+1 | own]
      if (row_valid) *reinterpret_cast<uint4*>(po + 32 * c) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
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

constexpr int kRingSlots = 4;  // K/V tile slots in LDS; NS - 1 tiles are in flight

template <typename TR, int D>
static int launch(const AttnParams& p, bool paged, bool has_bias, dim3 grid, int64_t num_split, uint32_t flags,
                  hipStream_t s) {
  if (grid.x > 0 && !(flags & TMX_ATTN_ONLY_COMBINE)) {
    if (paged) {
      if (has_bias) attn_kernel<TR, D, kRingSlots, true, true><<<grid, 256, 0, s>>>(p);
      else attn_kernel<TR, D, kRingSlots, true, false><<<grid, 256, 0, s>>>(p);
    } else {
      if (has_bias) attn_kernel<TR, D, kRingSlots, false, true><<<grid, 256, 0, s>>>(p);
      else attn_kernel<TR, D, kRingSlots, false, false><<<grid, 256, 0, s>>>(p);
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
  if (paged)
    TMX_CHECK_SUPPORTED(a->page_size >= 16 && (a->page_size & (a->page_size - 1)) == 0,
                        "tmx_attention: page_size %d not supported (powers of two >= 16 are)", a->page_size);
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
  p.page_shift = paged ? __builtin_ctz(a->page_size) : 0;
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
