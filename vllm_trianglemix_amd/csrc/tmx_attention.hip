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
  unsigned long long* stamps;  // diagnostic builds only
  const int32_t* tile_pages;   // [nnz][2] physical pages of each listed tile's halves, or NULL
  int q_stride_t, q_stride_h, out_stride_t, out_stride_h;  // elements; checked to fit 31 bits
  int kv_stride_page, kv_stride_t, kv_stride_h, block_table_stride;
  int64_t bias_stride;
  int page_size, page_shift;  // page_size is a power of two
  int H, Hkv, G, Gw, block_m;
  int Y;                      // units per work item: K/V heads x head chunks
  int num_items;
  long long units_total;      // work items x Y
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

// s_waitcnt takes an immediate: wait until at most n (rounded down to even, capped at 2 * HI) of this
// wave's vector-memory operations are outstanding.  Waiting for fewer than asked is always safe.
template <int LO, int HI>
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  if constexpr (LO == HI) {
    wait_vmcnt<2 * LO>();
  } else {
    constexpr int MID = (LO + HI + 1) / 2;
    if (n >= 2 * MID) wait_vmcnt_dyn<MID, HI>(n);
    else wait_vmcnt_dyn<LO, MID - 1>(n);
  }
}

// One LDS-DMA instruction: every lane fetches 16 bytes from its own global
// address; the wave's 1 KiB lands at LDS byte address lds_addr + 16 * lane.
// Issued from inline asm on purpose: the compiler would otherwise wait for ALL
// transfers in flight (vmcnt(0)) before every LDS read; here the waits are
// counted by hand and followed by a workgroup barrier.
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

// Diagnostic builds only (-DTMX_STAMPS): per-wave cycle sums of the loop's segments, written to a
// buffer of their own.  No stamp executes in the shipped library.
#ifdef TMX_STAMPS
#define TMX_STAMP(var)                                                                   \
  do {                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);                                                   \
  } while (0)
#define TMX_SEG(idx, t0, t1) seg[idx] += (t1) - (t0)
#else
#define TMX_STAMP(var) do { } while (0)
#define TMX_SEG(idx, t0, t1) do { } while (0)
#endif

// What one workgroup needs to know about a unit of work (all wave-uniform):
// a q-block piece from the index; the K/V head is fixed per workgroup.
struct Unit {
  int valid;
  int col_start, ntiles, slot, b, q_tok0, pos0, n;
};

// Persistent, software-pipelined kernel.  Every workgroup walks units
// blockIdx.x, blockIdx.x + gridDim.x, ... and streams their inputs through ONE
// ring of NS LDS entries of equal size: per unit first RB "Q entries" (the
// 16 * RB rows of every wave), then one entry per key tile ([K | V], 32 keys).
// Entries are fetched by LDS-DMA NS - 1 entries ahead of their use, across
// unit boundaries, so the loads of the next unit are in flight while the
// current one is computed and stored.  A unit is (work item, K/V head chunk);
// the grid is a multiple of the chunks per item, so a workgroup keeps its chunk.
template <typename TR, int D, int NS, int RB, bool PAGED, bool HAS_BIAS>
__global__ __launch_bounds__(256, 2) void attn_kernel(AttnParams p) {
  constexpr int KS = D / 32;            // k-steps of S^T = K Q^T
  constexpr int DB = D / 16;            // 16-column output blocks
  constexpr int ROWB = D * 2;           // bytes per Q/K/V row in LDS
  constexpr int CPR = D / 8;            // 16-byte chunks per row
  constexpr int ENTRY = 64 * ROWB;      // ring entry: K tile + V tile, or 4 waves x 16 Q rows
  constexpr int HALF = ENTRY / 2;
  constexpr int RPP = 1024 / ROWB;      // rows per 1-KiB LDS-DMA piece (one wave instruction)
  constexpr int DPT = ENTRY / 1024 / 4; // LDS-DMA instructions per wave and entry
  constexpr int U = DPT / 2;            // ... per tensor of a K/V entry
  using frag = typename TR::frag;

  __shared__ __attribute__((aligned(16))) char lds[NS * ENTRY];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 15, g = lane >> 4;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)lds));

  const int hg = wave % p.Gw;           // this wave's head inside the workgroup's head group
  const int rbw = (wave / p.Gw) * RB;   // its first 16-row block inside the q-block
  const int W = gridDim.x;
  // fixed for the workgroup: K/V head and first query head of its chunk
  const int kvc = blockIdx.x % p.Y;
  const int chunks = p.G / p.Gw;
  const int kvh = kvc / chunks;
  const int head = kvh * p.G + (kvc - kvh * chunks) * p.Gw + hg;
  const int item0 = blockIdx.x / p.Y, item_step = W / p.Y;  // W is a multiple of Y, or every workgroup has one unit

  auto load_unit = [&](int k) -> Unit {
    Unit u;
    const long long item = item0 + (long long)k * item_step;
    u.valid = (item < p.num_items) && (k == 0 || item_step > 0);
    u.col_start = u.ntiles = u.slot = u.b = u.q_tok0 = u.pos0 = 0;
    u.n = 1;
    if (u.valid) {
      const i32x4 it0 = sload(reinterpret_cast<const i32x4*>(p.items) + 2 * item);
      const i32x4 it1 = sload(reinterpret_cast<const i32x4*>(p.items) + 2 * item + 1);
      u.col_start = it0.y;
      u.ntiles = it0.z;
      u.slot = it0.w;
      u.b = it1.x;
      u.q_tok0 = it1.y;
      u.pos0 = it1.z;
      u.n = it1.w;
    }
    return u;
  };

  // ------------------------------------------------------------------ issue side
  Unit iu = load_unit(0);
  Unit iu_next = load_unit(1);
  int ik = 0;   // ordinal of iu among this workgroup's units
  int ie = 0;   // entries of iu already issued
  int gi = 0;   // entries issued so far (ring position)
  // descriptor of the next K/V entry to issue, fetched one step ahead: key tile and the page offsets
  // (elements) of its two 16-key halves
  int d_tile = 0;
  long long d_off0 = 0, d_off1 = 0;
  long long i_krow0 = 0;  // contiguous K/V: first row of iu's request

  const int piece_row = lane / CPR;  // row of this lane inside a 1-KiB piece
  const int piece_cc = lane % CPR;   // its 16-byte chunk
  const uint16_t* qbase = reinterpret_cast<const uint16_t*>(p.q) + (long long)head * p.q_stride_h;
  const uint16_t* kbase = reinterpret_cast<const uint16_t*>(p.k) + (long long)kvh * p.kv_stride_h;
  const uint16_t* vbase = reinterpret_cast<const uint16_t*>(p.v) + (long long)kvh * p.kv_stride_h;

  // start fetching what the NEXT issue of a K/V entry needs (no wait here)
  auto prefetch_desc = [&]() {
    if (!iu.valid || ie < RB || ie >= RB + iu.ntiles) return;
    const int pos = iu.col_start + (ie - RB);
    d_tile = sload(p.col_idx + pos);
    if constexpr (PAGED) {
      if (p.tile_pages != nullptr) {
        d_off0 = (long long)sload(p.tile_pages + 2 * pos) * p.kv_stride_page;
        d_off1 = (long long)sload(p.tile_pages + 2 * pos + 1) * p.kv_stride_page;
      } else {
        const int32_t* bt = p.block_table + (long long)iu.b * p.block_table_stride;
        const int pg_first = (d_tile * kBlockN) >> p.page_shift;
        d_off0 = (long long)sload(bt + pg_first) * p.kv_stride_page;
        d_off1 = (long long)sload(bt + imin(pg_first + 1, (iu.n - 1) >> p.page_shift)) * p.kv_stride_page;
      }
    }
  };

  auto issue_one = [&]() {
    const unsigned dst = lds_base + ((unsigned)gi % NS) * ENTRY;
    if (ie < RB) {
      // Q entry: this wave's 16 rows of row block rbw + ie, head `head`
      const int rows_valid = imin(p.block_m, iu.n - iu.pos0);
#pragma unroll
      for (int x = 0; x < DPT; ++x) {
        const int r16 = x * RPP + piece_row;  // row inside the 16-row block
        // rows past the end of the request are never stored; read a valid row instead
        const int r = imin(16 * (rbw + ie) + r16, rows_valid - 1);
        glds16(qbase + (long long)(iu.q_tok0 + r) * p.q_stride_t + 8 * (piece_cc ^ k_chunk_swizzle<D>(r16)),
               dst + wave * (ENTRY / 4) + x * 1024);
      }
    } else {
      const int last_key = iu.n - 1;
      const int pg_first = PAGED ? ((d_tile * kBlockN) >> p.page_shift) : 0;
#pragma unroll
      for (int x = 0; x < U; ++x) {
        const int piece = wave * U + x;
        const int row = piece * RPP + piece_row;
        // keys past the end of the sequence are masked later; clamp their source to a valid row
        const int j = imin(d_tile * kBlockN + row, last_key);
        long long off;
        if constexpr (PAGED) {
          off = ((j >> p.page_shift) == pg_first ? d_off0 : d_off1) + (j & (p.page_size - 1)) * p.kv_stride_t;
        } else {
          off = (i_krow0 + j) * p.kv_stride_t;
        }
        glds16(kbase + off + 8 * (piece_cc ^ k_chunk_swizzle<D>(row)), dst + piece * 1024);
        glds16(vbase + off + 8 * (piece_cc ^ v_chunk_swizzle<D>(row)), dst + HALF + piece * 1024);
      }
    }
    ++ie;
    ++gi;
    if (ie == RB + iu.ntiles) {
      iu = iu_next;
      ++ik;
      iu_next = load_unit(ik + 1);
      ie = 0;
      if constexpr (!PAGED) {
        if (iu.valid) i_krow0 = sload(p.cu_k + iu.b);
      }
    }
    prefetch_desc();
  };

  if constexpr (!PAGED) {
    if (iu.valid) i_krow0 = sload(p.cu_k + iu.b);
  }
#pragma unroll 1
  for (int x = 0; x < NS - 1; ++x)
    if (iu.valid) issue_one();

  // ------------------------------------------------------------------ compute side
  // per-lane LDS read offsets inside an entry
  int k_off[KS];  // K / Q row qi, chunk 4 * ks + g (swizzled)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_off[ks] = qi * ROWB + 16 * ((4 * ks + g) ^ k_chunk_swizzle<D>(qi));
  // V transposed read: lane 4 * q' + pp of each 16-lane group addresses key row 4 * g + q' (+16 for the
  // second half of the tile), columns 16 * db + 4 * pp .. +3
  const int v_row = 4 * g + ((lane & 15) >> 2);
  const int v_x = v_chunk_swizzle<D>(v_row) >> 1;  // XOR on the 32-byte group index == the output block db
  const int v_base = HALF + v_row * ROWB + 8 * (lane & 3);
  const int koff_lane = 4 * g;  // this lane's first key inside a 16-key half tile

  Unit cu = load_unit(0);
  Unit cu_next = load_unit(1);
  int ck = 0;
  int gs = 0;          // entries consumed so far
  int store_mark = 0;  // entries with ring position < store_mark were issued before the last stores
  int store_ops = 0;   // vector-memory instructions of those stores (this wave)

#ifdef TMX_STAMPS
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long ts0, ts1, ts2, ts3, ts4;
  TMX_STAMP(ts4);
  const unsigned long long t_begin = ts4;
#endif

  while (cu.valid) {
    const int rows_valid = imin(p.block_m, cu.n - cu.pos0);
    const int mode = effective_mode(p.mode, cu.n, p.tri_min_len);
    const int n = cu.n;

    // ---- per row block: wave-uniform row ranges and the per-lane form of the predicate ----
    // key j is attended by this lane's row iff j <= key_hi and not (hole_lo < j < hole_lo + 1 + hole_width)
    int i_min[RB], i_max[RB], key_hi[RB];
    unsigned hole_width[RB];
    bool has_rows[RB], any_last[RB], all_last[RB], row_valid[RB];
    const int hole_lo = p.si < -1 ? -1 : p.si;
#pragma unroll
    for (int e = 0; e < RB; ++e) {
      const int wrow0 = 16 * (rbw + e);
      const int wrow1 = imin(wrow0 + 16, rows_valid);
      has_rows[e] = wrow0 < rows_valid;
      i_min[e] = cu.pos0 + wrow0;
      i_max[e] = cu.pos0 + wrow1 - 1;
      any_last[e] = (n - i_max[e]) < p.last;
      all_last[e] = (n - i_min[e]) < p.last;
      row_valid[e] = wrow0 + qi < rows_valid;
      const int pos_i = cu.pos0 + wrow0 + qi;
      key_hi[e] = -1;
      hole_width[e] = 0;
      if (row_valid[e]) {
        key_hi[e] = (mode == TMX_MODE_NONE) ? n - 1 : imin(pos_i, n - 1);
        if (mode == TMX_MODE_TRIANGLE && !((n - pos_i) < p.last)) {
          // keys si < j < pos_i - sl are the excluded middle section
          const long long w = (long long)pos_i - p.sl - (long long)hole_lo - 1;
          hole_width[e] = w > 0 ? (unsigned)w : 0u;
        }
      }
    }

    frag qf[RB][KS];
    f32x4 acc[RB][DB];
    float m_run[RB], l_run[RB];
#pragma unroll
    for (int e = 0; e < RB; ++e) {
      m_run[e] = -__builtin_inff();
      l_run[e] = 0.f;
#pragma unroll
      for (int db = 0; db < DB; ++db) acc[e][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int n_entries = RB + cu.ntiles;
    int tile = cu.ntiles > 0 ? sload(p.col_idx + cu.col_start) : 0;
#pragma unroll 1
    for (int ent = 0; ent < n_entries; ++ent) {
      // ---- wait for ring entry gs, make it visible to the workgroup, refill the slot freed last step
      TMX_STAMP(ts0);
      wait_vmcnt_dyn<0, 15>(DPT * (gi - gs - 1) + (gs < store_mark ? store_ops : 0));
      TMX_STAMP(ts1);
      __builtin_amdgcn_s_barrier();
      TMX_STAMP(ts2);
      if (iu.valid) issue_one();
      TMX_STAMP(ts3);
      TMX_SEG(0, ts0, ts1);
      TMX_SEG(1, ts1, ts2);
      TMX_SEG(2, ts2, ts3);
      const char* es = lds + ((unsigned)gs % NS) * ENTRY;
      ++gs;

      if (ent < RB) {
        // Q fragments of row block `ent` (B operand of S^T)
        const char* qs = es + wave * (ENTRY / 4);
#pragma unroll
        for (int e = 0; e < RB; ++e)
          if (e == ent) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
              qf[e][ks] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(qs + k_off[ks]));
          }
        continue;
      }
      const int t = ent - RB;
      const int j0 = tile * kBlockN;
      if (t + 1 < cu.ntiles) tile = sload(p.col_idx + cu.col_start + t + 1);  // next iteration's tile

      // Classify each 16-key half of the tile against each row block (wave-uniform): `need`: some
      // (row, key) pair is attended; `full`: every pair is, so the per-element predicate is skipped.
      bool need[RB][2], full[RB], kb_used[2] = {false, false};
#pragma unroll
      for (int e = 0; e < RB; ++e) {
        full[e] = false;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const int h0 = j0 + 16 * kb, h1 = h0 + 15;
          bool nd = has_rows[e] && h0 < n;
          if (mode != TMX_MODE_NONE && nd) {
            nd = h0 <= i_max[e];
            if (mode == TMX_MODE_TRIANGLE && !any_last[e])
              nd = nd && ((h0 <= imin(i_max[e], p.si)) || (p.sl >= 0 && h1 >= i_min[e] - p.sl));
          }
          need[e][kb] = nd;
          kb_used[kb] = kb_used[kb] || nd;
        }
        const int j1 = j0 + kBlockN - 1;
        if (mode == TMX_MODE_NONE) {
          full[e] = j1 < n;
        } else {
          full[e] = j1 <= i_min[e];
          if (mode == TMX_MODE_TRIANGLE && !all_last[e]) full[e] = full[e] && (j1 <= p.si || j0 >= i_max[e] - p.sl);
        }
      }
      if (!kb_used[0] && !kb_used[1]) continue;
      TMX_STAMP(ts0);

      // S^T = K Q^T: K fragments are read once for all row blocks; unused halves are skipped
      f32x4 s[RB][2];
#pragma unroll
      for (int e = 0; e < RB; ++e) s[e][0] = s[e][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        if (!kb_used[kb]) continue;
        frag kf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          kf[ks] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(es + kb * 16 * ROWB + k_off[ks]));
#pragma unroll
        for (int e = 0; e < RB; ++e) {
          if (!need[e][kb]) continue;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) s[e][kb] = TR::mfma(kf[ks], qf[e][ks], s[e][kb]);
        }
      }

      // online softmax per row block; P stays in registers as the B operand of the second product
      frag pf[RB];
#pragma unroll
      for (int e = 0; e < RB; ++e) {
        s16x8 pk = {0, 0, 0, 0, 0, 0, 0, 0};
        if (need[e][0] || need[e][1]) {
          float x[8];
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            if (need[e][kb]) {
#pragma unroll
              for (int r = 0; r < 4; ++r) x[kb * 4 + r] = s[e][kb][r] * p.scale_log2;
              if constexpr (HAS_BIAS) {
                const int pos_i = i_min[e] + qi;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int j = j0 + 16 * kb + koff_lane + r;
                  if (row_valid[e] && j < p.bias_cols && pos_i < p.bias_rows && j <= key_hi[e])
                    x[kb * 4 + r] += kLog2e * load_bias(p.bias, p.bias_dtype, (long long)pos_i * p.bias_stride + j);
                }
              }
              if (!full[e]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int j = j0 + 16 * kb + koff_lane + r;
                  const bool in_hole = (unsigned)(j - hole_lo - 1) < hole_width[e];
                  x[kb * 4 + r] = (j <= key_hi[e] && !in_hole) ? x[kb * 4 + r] : -__builtin_inff();
                }
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) x[kb * 4 + r] = -__builtin_inff();
            }
          }
          float tmax = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), fmaxf(fmaxf(x[4], x[5]), fmaxf(x[6], x[7])));
          tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
          tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
          const float m_new = fmaxf(m_run[e], tmax);
          // alpha rescales what was accumulated under the old maximum; nothing was while m_run is -inf
          float alpha = 1.f, m_safe = 0.f;
          if (m_new != -__builtin_inff()) {
            m_safe = m_new;
            if (m_run[e] != -__builtin_inff()) alpha = __builtin_amdgcn_exp2f(m_run[e] - m_new);
          }
          m_run[e] = m_new;
          float psum = 0.f;
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            if (!need[e][kb]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pe = __builtin_amdgcn_exp2f(x[kb * 4 + r] - m_safe);
              psum += pe;
              pk[kb * 4 + r] = TR::cvt(pe);
            }
          }
          l_run[e] = l_run[e] * alpha + psum;
          if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {  // wave-uniform: some row's maximum moved
#pragma unroll
            for (int db = 0; db < DB; ++db) acc[e][db] *= alpha;
          }
        }
        pf[e] = __builtin_bit_cast(frag, pk);
      }

      // O^T += V^T P^T (V fragments are read once, transposed by the LDS read)
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const int so = v_base + 32 * (db ^ v_x);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(es + so));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(es + 16 * ROWB + so));
        const frag vf = __builtin_bit_cast(frag, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int e = 0; e < RB; ++e) acc[e][db] = TR::mfma(vf, pf[e], acc[e][db]);
      }
      TMX_STAMP(ts1);
      TMX_SEG(3, ts0, ts1);
    }
    TMX_STAMP(ts0);

    // ---- epilogue: lane (qi, g) holds columns 16 * db + 4 * g .. +3 of its row for every db ----
    int stores = 0;
#pragma unroll
    for (int e = 0; e < RB; ++e) {
      float l_tot = l_run[e] + __shfl_xor(l_run[e], 16, 64);
      l_tot += __shfl_xor(l_tot, 32, 64);
      const int my_row = 16 * (rbw + e) + qi;
      if (!has_rows[e]) continue;  // wave-uniform
      if (cu.slot >= 0) {
        // piece of a split q-block: unnormalised fp32 accumulator + (max, sum) for the combine pass
        const long long prow = ((long long)cu.slot * p.H + head) * p.block_m + my_row;
        float* po = p.partial_o + prow * D + 4 * g;
        if (row_valid[e]) {
#pragma unroll
          for (int db = 0; db < DB; ++db)
            *reinterpret_cast<float4*>(po + 16 * db) =
                make_float4(acc[e][db][0], acc[e][db][1], acc[e][db][2], acc[e][db][3]);
          if (g == 0) *reinterpret_cast<float2*>(p.partial_ml + prow * 2) = make_float2(m_run[e], l_tot);
        }
        stores += DB;  // the (max, sum) store may be skipped by lanes; counting fewer only waits longer
        continue;
      }
      const float inv = 1.0f / l_tot;  // a fully masked row gives 0 * inf = NaN, as softmax over all -inf does
      const long long orow = (long long)(cu.q_tok0 + my_row) * p.out_stride_t + (long long)head * p.out_stride_h;
      if (p.out_dtype == TMX_F32) {
        float* po = reinterpret_cast<float*>(p.out) + orow + 4 * g;
        if (row_valid[e]) {
#pragma unroll
          for (int db = 0; db < DB; ++db)
            *reinterpret_cast<float4*>(po + 16 * db) = make_float4(acc[e][db][0] * inv, acc[e][db][1] * inv,
                                                                  acc[e][db][2] * inv, acc[e][db][3] * inv);
        }
        stores += DB;
      } else {
        // 16-bit output: lanes l and l ^ 16 exchange one 8-byte piece per pair of output blocks so
        // that every lane stores 16 contiguous bytes (8 columns)
        uint16_t* po = reinterpret_cast<uint16_t*>(p.out) + orow + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g);
#pragma unroll
        for (int c = 0; c < DB / 2; ++c) {
          s16x4 a, bq;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            a[r] = TR::cvt(acc[e][2 * c][r] * inv);
            bq[r] = TR::cvt(acc[e][2 * c + 1][r] * inv);
          }
          const uint2 au = __builtin_bit_cast(uint2, a), bu = __builtin_bit_cast(uint2, bq);
          const auto r0 = __builtin_amdgcn_permlane16_swap(au.x, bu.x, false, false);
          const auto r1 = __builtin_amdgcn_permlane16_swap(au.y, bu.y, false, false);
          // even 16-lane rows: [own block 2c | partner's block 2c]; odd rows: [partner's block 2c+1 | own]
          if (row_valid[e]) *reinterpret_cast<uint4*>(po + 32 * c) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
        stores += DB / 2;
      }
    }
    store_mark = gi;
    store_ops = stores;
    TMX_STAMP(ts1);
    TMX_SEG(4, ts0, ts1);

    cu = cu_next;
    ++ck;
    cu_next = load_unit(ck + 1);
    TMX_STAMP(ts2);
    TMX_SEG(5, ts1, ts2);
  }
#ifdef TMX_STAMPS
  TMX_STAMP(ts4);
  seg[6] = ts4 - t_begin;
  seg[7] = (unsigned long long)gs;
  if (p.stamps != nullptr && lane == 0) {
    unsigned long long* dst = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int x = 0; x < 8; ++x) dst[x] = seg[x];
  }
#endif
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
    const int64_t off = (long long)(q_start + row0 + row) * p.out_stride_t + (long long)head * p.out_stride_h + 4 * d4;
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

constexpr int kRingSlots = 4;  // LDS ring entries; NS - 1 are in flight

// Persistent grid: as many workgroups as stay resident (LDS and registers allow a fixed number per
// CU), rounded down to a multiple of the units per item so that a workgroup keeps one K/V head.
template <typename K>
static int persistent_grid(K kernel, const AttnParams& p, unsigned* grid_out) {
  int dev = 0, cus = 0, per_cu = 0;
  TMX_CHECK_HIP(hipGetDevice(&dev));
  TMX_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  TMX_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0));
  if (per_cu < 1) per_cu = 1;
  long long w = (long long)cus * per_cu;
  if (w > p.units_total) w = p.units_total;
  if (w > p.Y) w -= w % p.Y;
  *grid_out = (unsigned)w;
  return TMX_OK;
}

template <typename TR, int D, bool PAGED, bool HAS_BIAS>
static int launch_attn(const AttnParams& p, hipStream_t s) {
  auto kernel = attn_kernel<TR, D, kRingSlots, kRowBlocksPerWave, PAGED, HAS_BIAS>;
  static unsigned cached_grid_cap[16] = {0};  // per device: resident workgroups of this instantiation
  int dev = 0;
  TMX_CHECK_HIP(hipGetDevice(&dev));
  unsigned grid = 0;
  int rc = persistent_grid(kernel, p, &grid);
  if (rc != TMX_OK) return rc;
  (void)cached_grid_cap;
  kernel<<<grid, 256, 0, s>>>(p);
  TMX_CHECK_HIP(hipGetLastError());
  return TMX_OK;
}

template <typename TR, int D>
static int launch(const AttnParams& p, bool paged, bool has_bias, int64_t num_split, uint32_t flags, hipStream_t s) {
  if (p.units_total > 0 && !(flags & TMX_ATTN_ONLY_COMBINE)) {
    int rc;
    if (paged) rc = has_bias ? launch_attn<TR, D, true, true>(p, s) : launch_attn<TR, D, true, false>(p, s);
    else rc = has_bias ? launch_attn<TR, D, false, true>(p, s) : launch_attn<TR, D, false, false>(p, s);
    if (rc != TMX_OK) return rc;
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
  p.stamps = nullptr;
#ifdef TMX_STAMPS
  p.stamps = reinterpret_cast<unsigned long long*>(a->partial_ml);  // diagnostic build: stamps replace the partial sums
#endif
  const int64_t lim = 0x7FFFFFFF;
  TMX_CHECK_SUPPORTED(a->q_stride_t <= lim && a->q_stride_h <= lim && a->out_stride_t <= lim &&
                          a->out_stride_h <= lim && a->kv_stride_page <= lim && a->kv_stride_t <= lim &&
                          a->kv_stride_h <= lim && a->block_table_stride <= lim && a->q_stride_t >= 0 &&
                          a->q_stride_h >= 0 && a->out_stride_t >= 0 && a->out_stride_h >= 0 &&
                          a->kv_stride_page >= 0 && a->kv_stride_t >= 0 && a->kv_stride_h >= 0 &&
                          a->block_table_stride >= 0,
                      "tmx_attention: strides must be non-negative and below 2^31 elements");
  p.q_stride_t = (int)a->q_stride_t;
  p.q_stride_h = (int)a->q_stride_h;
  p.out_stride_t = (int)a->out_stride_t;
  p.out_stride_h = (int)a->out_stride_h;
  p.kv_stride_page = (int)a->kv_stride_page;
  p.kv_stride_t = (int)a->kv_stride_t;
  p.kv_stride_h = (int)a->kv_stride_h;
  p.block_table_stride = (int)a->block_table_stride;
  p.tile_pages = nullptr;
  if (paged && ix->tile_pages != nullptr && ix->pages_block_table == a->block_table &&
      ix->pages_page_size == a->page_size && ix->pages_block_table_stride == a->block_table_stride)
    p.tile_pages = ix->tile_pages;
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

  p.Y = a->num_kv_heads * (G / p.Gw);
  p.num_items = (int)a->num_items;
  p.units_total = (long long)a->num_items * p.Y;
  hipStream_t s = (hipStream_t)stream;
  if (a->io_dtype == TMX_BF16) {
    if (a->head_dim == 128) return launch<Bf16Traits, 128>(p, paged, has_bias, a->num_split_qblocks, a->flags, s);
    return launch<Bf16Traits, 64>(p, paged, has_bias, a->num_split_qblocks, a->flags, s);
  }
  if (a->head_dim == 128) return launch<F16Traits, 128>(p, paged, has_bias, a->num_split_qblocks, a->flags, s);
  return launch<F16Traits, 64>(p, paged, has_bias, a->num_split_qblocks, a->flags, s);
}
