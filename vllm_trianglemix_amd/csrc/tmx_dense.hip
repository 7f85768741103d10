// Dense causal attention for gfx950 (MI355X): the layers outside the triangle set attend causally
// (reference: the dense mask of vllm/attention/trianglemix_attention.py:200-219 behind the eager product of
// vllm/attention/ascend_trianglemix.py:88-106), and a long prompt's causal attention is matrix-core bound:
// N^2/2 pairs per head against N rows of K/V.  This kernel is shaped for the matrix cores, where the work-list
// kernel (tmx_attention.hip) is shaped for sparse rows:
//
//   workgroup = 8 waves = 4 query heads of one K/V head x 64 query rows (two 32-row blocks); wave w owns
//   32 rows of one head.  Key tiles of 64 keys, K and V staged once per workgroup in LDS (double buffered,
//   fetched one tile ahead into registers, written after the tile's arithmetic).
//   S^T[key][row] = K . Q^T with v_mfma_f32_32x32x16 (A = K rows from LDS by ds_read_b128 through an XOR
//   swizzle, B = the wave's query rows kept in registers): a lane ends with 32 scores of ONE query row, so
//   the softmax needs one cross-lane step (lane l and l ^ 32 share a row).
//   O^T[d][row] += V^T . P^T, again 32x32x16: the score accumulators, converted to 16 bit in place, ARE the
//   B operand (no LDS round trip); V^T comes out of LDS transposed (ds_read_b64_tr_b16).
//   Per 64-key tile and wave: 32 matrix instructions of 32 cycles on 64 KB of operands read from LDS, against
//   24 of 16 cycles on 16 KB in the work-list kernel's 16-row x 32-key step.
//
// Probabilities enter the second product as ONE 16-bit term.  bf16 has 8 significant bits, too few for the
// 1e-3 bar (a peaked row's error is 2^-9 |v|), so for bf16 inputs V is converted to fp16 on its way into LDS
// (exact for |v| <= 65504 down to 2^-14; smaller magnitudes keep an absolute error below 2^-25) and the
// probabilities are fp16 (11 bits, rounded to nearest).  fp16 inputs need no conversion.
//
// Requests with fewer than kDenseMinRows query rows (decode, speculative decode) stay on the work-list kernel.
//
// Two forms of the tile fetch.  Register form: every thread loads 32 bytes of K and of V a tile ahead, converts
// (V to fp16, an fp8 cache to 16 bit) and stores them to LDS after the tile's arithmetic.  Staged form (STAGED,
// when the caller lends a staging area): dense_stage_kernel first gathers the K/V rows of the kernel's requests
// -- from pages, the step's own tensors or an fp8 cache -- into contiguous 16-bit rows, V already as fp16, and the
// tiles then come into LDS by LDS-DMA: no registers, no conversion and no address arithmetic per tile (measured
// 7 % faster at 32k tokens; the gather costs under 1 % there).  Same values in LDS, same results.
#include "tmx_common.h"

#include <atomic>
#include "tmx_dense.h"

#include <type_traits>

namespace tmx {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

constexpr int kDM = 64;        // query rows per workgroup
constexpr int kDN = 64;        // keys per tile
constexpr int kDRowB = 256;    // bytes per K / V row in LDS (head size 128, 16-bit)
constexpr int kDTile = kDN * kDRowB;  // one tensor of one tile: 16 KB
constexpr float kDMinusBig = -1.0e30f;
#ifndef TMX_DENSE_QK_AHEAD
#define TMX_DENSE_QK_AHEAD 4
#endif
constexpr int kQkAhead = TMX_DENSE_QK_AHEAD;  // K fragments in flight in the first product (1..8: the wait counts are written out)
#ifndef TMX_DENSE_PV_AHEAD
#define TMX_DENSE_PV_AHEAD 3
#endif
constexpr int kPvAhead = TMX_DENSE_PV_AHEAD;  // V fragments (two transposed reads each) in flight ahead of the second product's instruction (1..7)
constexpr float kDSlack = 8.0f;  // the running maximum moves only when a tile exceeds it by more than this (log2 units)

template <typename T>
__device__ __forceinline__ T dsload(const T* ptr) {
  return *reinterpret_cast<const __attribute__((address_space(4))) T*>((const __attribute__((address_space(4))) void*)ptr);
}

template <bool BF16>
__device__ __forceinline__ f32x16 mfma_qk(uint4 a, uint4 b, f32x16 c) {
  if constexpr (BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_pv(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// two floats -> one dword of two fp16, round to nearest even
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, f16x2_t));
}
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
// one dword of two bf16 -> one dword of two fp16 (exact inside fp16's range)
__device__ __forceinline__ uint32_t bf16x2_to_f16x2(uint32_t w) {
  return pack_f16(__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u));
}
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ uint4 lds_b128(unsigned addr) {
  __builtin_assume((addr & 15u) == 0);  // (lets a register + immediate address keep the 16-byte form)
  return __builtin_bit_cast(uint4, *(const __attribute__((address_space(3))) u32x4_t*)(uintptr_t)addr);
}
// 16 bytes from LDS at a per-lane address plus a compile-time offset, by an instruction the compiler neither reorders
// nor waits for: the caller counts it in lgkmcnt itself.  (Written out because the compiler turns a 16-byte read at
// register + immediate into two 8-byte ones at half the rate.)
template <unsigned OFF>
__device__ __forceinline__ void lds_b128_at(int addr, u32x4_t& dst) {
  static_assert(OFF < 65536, "the offset field has 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_store_b128(unsigned addr, uint4 v) {
  *(__attribute__((address_space(3))) u32x4_t*)(uintptr_t)addr = __builtin_bit_cast(u32x4_t, v);
}
__device__ __forceinline__ s16x4_t lds_tr16(unsigned addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)addr);
}
// The transposed 8-byte read at a per-lane address plus a compile-time offset, like lds_b128_at: neither reordered nor
// waited for by the compiler.
template <unsigned OFF>
__device__ __forceinline__ void lds_tr16_at(int addr, s16x4_t& dst) {
  static_assert(OFF < 65536, "the offset field has 16 bits");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// One LDS-DMA instruction: lane i fetches 16 bytes at (wave-uniform base + its own 32-bit byte offset) into LDS byte
// lds_addr + 16 i.  Issued from inline asm: the caller waits for vmcnt itself, ahead of the tile's barrier.
__device__ __forceinline__ void lds_dma16(const void* sbase, int voff, unsigned lds_addr) {
  const unsigned long long b = (unsigned long long)(uintptr_t)sbase;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(sb), "s"(__builtin_amdgcn_readfirstlane(lds_addr))
      : "memory", "m0");
}

// The requests a dense launch takes and where they start, in LDS: blk_ptr[b] = first 64-row block of request b,
// key_ptr[b] = its first row in the staging area (sequence lengths rounded up to 64 rows).  Requests with fewer than
// kDenseMinRows query rows have neither.  All threads of the workgroup (NT of them) call it; batch <= kDenseMaxBatch.
template <int NT>
__device__ __forceinline__ void dense_request_scan(const DenseParams& p, int* blk_ptr, int* key_ptr) {
  const int tid = threadIdx.x;
  for (int b = tid; b < p.batch; b += NT) {
    const int q_len = p.cu_q[b + 1] - p.cu_q[b];
    const bool takes = q_len >= kDenseMinRows;
    blk_ptr[b + 1] = takes ? (q_len + kDM - 1) / kDM : 0;
    key_ptr[b + 1] = takes ? (p.seq_lens[b] + kDN - 1) / kDN * kDN : 0;
  }
  if (tid == 0) blk_ptr[0] = key_ptr[0] = 0;
  __syncthreads();
  // inclusive scans of [1 .. batch]
  for (int step = 1; step < p.batch; step <<= 1) {
    int add_b[(kDenseMaxBatch + NT - 1) / NT], add_k[(kDenseMaxBatch + NT - 1) / NT];
#pragma unroll
    for (int e = 0; e < (kDenseMaxBatch + NT - 1) / NT; ++e) {
      const int i = tid + e * NT;
      add_b[e] = add_k[e] = 0;
      if (i < p.batch && i >= step) {
        add_b[e] = blk_ptr[i + 1 - step];
        add_k[e] = key_ptr[i + 1 - step];
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < (kDenseMaxBatch + NT - 1) / NT; ++e) {
      const int i = tid + e * NT;
      if (i < p.batch) {
        blk_ptr[i + 1] += add_b[e];
        key_ptr[i + 1] += add_k[e];
      }
    }
    __syncthreads();
  }
}

// KV8: K and V are an fp8 (e4m3fn) paged cache; a thread then carries 16 bytes (16 values) of a row per tensor
// and the staging step descales them: value = fp8 x scale, rounded once to the 16-bit type of q (what the
// reference's kernels do with k_descale / v_descale, vllm/v1/attention/backends/rocm_attn.py:421-422).
// STAGED: K/V are read from the staging area (PAGED, KV8 false: dense_stage_kernel dealt with the source).
template <bool BF16, bool PAGED, bool KV8, bool STAGED = false>
__global__ __launch_bounds__(512, 2) void dense_causal_kernel(const DenseParams p) {
  static_assert(!KV8 || PAGED, "an fp8 cache is paged");
  static_assert(!STAGED || (!PAGED && !KV8), "the staged form has one source");
  constexpr int KVB = KV8 ? 1 : 2;  // bytes per K/V element in memory
  __shared__ __attribute__((aligned(1024))) char lds[4 * kDTile];  // [buffer][K | V]
  __shared__ int blk_ptr[kDenseMaxBatch + 1];                       // first 64-row block of every request
  __shared__ int key_ptr[kDenseMaxBatch + 1];                       // first staging row of every request

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int hg = wave & 3, rb = wave >> 2;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)lds));
  __builtin_assume((lds_base & 1023u) == 0);  // (the array's alignment, lost through the scalar copy)

  // ---- the step's cache write, fused (16-bit cache; the launch reads the step's keys from k_new / v_new) ----
  if constexpr (PAGED && !KV8) {
    if (p.cache_slots != nullptr) {
      const int units = p.total_q * p.Hkv * 16;  // (row, head, 16-byte chunk)
      const uint16_t* kn = reinterpret_cast<const uint16_t*>(p.k_new);
      const uint16_t* vn = reinterpret_cast<const uint16_t*>(p.v_new);
      uint16_t* kc = reinterpret_cast<uint16_t*>(const_cast<void*>(p.k));
      uint16_t* vc = reinterpret_cast<uint16_t*>(const_cast<void*>(p.v));
      for (int u = (int)blockIdx.x * 512 + tid; u < units; u += (int)gridDim.x * 512) {
        const int c = u & 15, th = u >> 4;
        const int hh = th % p.Hkv, t = th / p.Hkv;
        const long long slot = p.cache_slots[t];
        if (slot < 0) continue;
        const long long src = (long long)t * p.new_stride_t + (long long)hh * p.new_stride_h + 8 * c;
        const long long dst = (slot >> p.page_shift) * p.kv_stride_page + (slot & (p.page_size - 1)) * (long long)p.kv_stride_t +
                              (long long)hh * p.kv_stride_h + 8 * c;
        const uint4 kx = *reinterpret_cast<const uint4*>(kn + src);
        const uint4 vx = *reinterpret_cast<const uint4*>(vn + src);
        *reinterpret_cast<uint4*>(kc + dst) = kx;
        *reinterpret_cast<uint4*>(vc + dst) = vx;
      }
    }
  }

  // ---- the launch's work: 64-row blocks of the requests this kernel takes, x K/V head chunks ----
  dense_request_scan<512>(p, blk_ptr, key_ptr);
  if constexpr (STAGED) {
    if (key_ptr[p.batch] > p.stage_rows) return;  // (a staging area smaller than the batch needs: nothing was staged)
  }
  const int num_blocks = blk_ptr[p.batch];
  const int lanes_per_unit = (int)gridDim.x / p.Y;  // workgroups that share a K/V head chunk (grid is a multiple of Y)
  const int kvc = (int)blockIdx.x % p.Y;
  const int wg_lane = (int)blockIdx.x / p.Y;
  const int chunks = p.G / 4;
  const int kvh = kvc / chunks;
  const int head = kvh * p.G + (kvc - kvh * chunks) * 4 + hg;

  // ---- per-lane constants ----
  // (All LDS addresses below are a per-lane register that already holds the LDS base, plus a compile-time offset: the
  // tile loop is unrolled over its two buffers so that no read or write needs an address computation.)
  // K fragment (A of S^T = K Q^T): key row 32 kb + r, 16-byte chunk 2 ks + h, stored at chunk ^ (row & 15)
  int koff[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) koff[ks] = (int)lds_base + r * kDRowB + 16 * ((2 * ks + h) ^ (r & 15));
  // V^T fragment (A of O^T = V^T P^T): two transposed 4-key x 16-column blocks; lane 4 q + pp of a 16-lane group
  // addresses key row q of the block, columns 4 pp .. 4 pp + 3 of the group's 16; chunk index ^ ((key & 3) << 2)
  const int tq = (lane & 15) >> 2, tpp = lane & 3, dhalf = (lane >> 4) & 1;
  int voff[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
    voff[db] = (int)lds_base + kDTile + (4 * h + tq) * kDRowB + 16 * ((((db ^ tq) << 2)) | (2 * dhalf + (tpp >> 1))) + 8 * (tpp & 1);
  // staging: thread t carries 32 bytes (chunks 2 c2, 2 c2 + 1) of key row t / 8 of K and of V
  const int srow = tid >> 3, sc2 = tid & 7;
  const unsigned st_k0 = lds_base + srow * kDRowB + 16 * ((2 * sc2) ^ (srow & 15));
  const unsigned st_k1 = lds_base + srow * kDRowB + 16 * ((2 * sc2 + 1) ^ (srow & 15));
  const unsigned st_v0 = lds_base + kDTile + srow * kDRowB + 16 * ((2 * sc2) ^ ((srow & 3) << 2));
  const unsigned st_v1 = lds_base + kDTile + srow * kDRowB + 16 * ((2 * sc2 + 1) ^ ((srow & 3) << 2));
  const int kv_row_bytes = KVB * p.kv_stride_t;
  const char* const k_head = reinterpret_cast<const char*>(p.k) + (long long)KVB * kvh * p.kv_stride_h + 16 * KVB * sc2;
  const char* const v_head = reinterpret_cast<const char*>(p.v) + (long long)KVB * kvh * p.kv_stride_h + 16 * KVB * sc2;
  // the same without the lane's column: wave-uniform bases of the cache-only fetch (scalar base + 32-bit lane offset)
  const char* const k_head_s = reinterpret_cast<const char*>(p.k) + (long long)KVB * kvh * p.kv_stride_h;
  const char* const v_head_s = reinterpret_cast<const char*>(p.v) + (long long)KVB * kvh * p.kv_stride_h;
  bool two_sources = false;  // some keys of the launch are rows of the step's own tensors
  if constexpr (PAGED && !KV8) two_sources = p.k_new != nullptr;
  // staged form: wave w brings key rows 8 w .. 8 w + 7 of a tile, two LDS-DMA instructions of 4 rows per tensor;
  // lane i lands at chunk position i & 15 of row i >> 4 and therefore fetches the chunk the swizzle keeps there
  const int stage_row_bytes = p.Hkv * kDRowB;
  const char* const stage_k = reinterpret_cast<const char*>(p.stage) + (long long)kvh * kDRowB;
  const char* const stage_v = stage_k + p.stage_rows * (long long)stage_row_bytes;
  int dma_k[2], dma_v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int R = 8 * wave + 4 * j + (lane >> 4);
    dma_k[j] = R * stage_row_bytes + 16 * ((lane & 15) ^ (R & 15));
    dma_v[j] = R * stage_row_bytes + 16 * ((lane & 15) ^ ((R & 3) << 2));
  }

  const int rounds = (num_blocks + lanes_per_unit - 1) / lanes_per_unit;
#pragma unroll 1
  for (int round = 0; round < rounds; ++round) {
    // blocks in descending length within a request; odd rounds run the workgroups backwards, so that every
    // workgroup's sum of block lengths comes out about equal
    const int ord = round * lanes_per_unit + ((round & 1) ? lanes_per_unit - 1 - wg_lane : wg_lane);
    if (ord >= num_blocks) continue;
    // request of block `ord`: the last b with blk_ptr[b] <= ord
    int lo = 0, hi = p.batch;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (blk_ptr[mid] <= ord) lo = mid;
      else hi = mid;
    }
    const int b = __builtin_amdgcn_readfirstlane(lo);
    const int nb = blk_ptr[b + 1] - blk_ptr[b];
    const int blk = __builtin_amdgcn_readfirstlane(nb - 1 - (ord - blk_ptr[b]));
    const int q_start = dsload(p.cu_q + b);
    const int q_len = dsload(p.cu_q + b + 1) - q_start;
    const int n = dsload(p.seq_lens + b);
    const int ctx = n - q_len;
    const int r0 = kDM * blk;
    const int rows_valid = imin(kDM, q_len - r0);
    const int w_r0 = r0 + 32 * rb;                       // first row of this wave
    const bool w_rows = 32 * rb < rows_valid;             // the wave has rows at all
    const int my_row = imin(w_r0 + r, q_len - 1);         // (rows past the end compute on a valid row, unstored)
    const bool row_ok = w_r0 + r < q_len;
    const int pos = ctx + my_row;                          // this lane's query position
    const int w_pos0 = ctx + w_r0;                         // smallest position of the wave
    const int num_tiles = (ctx + r0 + rows_valid - 1) / kDN + 1;
    const int w_tiles = w_rows ? (ctx + imin(w_r0 + 31, q_len - 1)) / kDN + 1 : 0;  // tiles this wave needs

    // ---- query rows of the wave: B operand fragments, chunk 2 ks + h of row r ----
    uint4 qf[8];
    {
      const char* qrow = reinterpret_cast<const char*>(p.q) + 2ll * ((long long)(q_start + my_row) * p.q_stride_t + (long long)head * p.q_stride_h) + 16 * h;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(qrow + 32 * ks);
    }

    const long long krow0 = (PAGED || STAGED) ? 0 : (long long)dsload(p.cu_k + b);  // (staged: cu_k may be NULL)
    const int32_t* const bt = PAGED ? p.block_table + (long long)b * p.block_table_stride : nullptr;
    // physical page of the wave's keys in tile t (its 8 keys share a page); looked up a tile before it is used, so
    // that no fetch waits for the lookup (any t: the key is clamped to the sequence)
    auto page_of = [&](int t) -> int {
      if constexpr (PAGED) return dsload(bt + (imin(kDN * t + 8 * wave, n - 1) >> p.page_shift));
      return 0;
    };
    // this thread's 32 bytes of K and of V of tile t, `page` = page_of(t) (keys past the end read the last key:
    // masked later)
    auto fetch = [&](int t, int page, uint4 (&kreg)[2], uint4 (&vreg)[2]) {
      const int key = imin(kDN * t + srow, n - 1);
      if constexpr (PAGED) {
        if (!two_sources) {
          const long long base = (long long)KVB * page * p.kv_stride_page;  // wave-uniform
          const unsigned lane_off = (unsigned)((key & (p.page_size - 1)) * kv_row_bytes + 16 * KVB * sc2);
          const char* ksrc = k_head_s + base;
          const char* vsrc = v_head_s + base;
          kreg[0] = *reinterpret_cast<const uint4*>(ksrc + lane_off);
          vreg[0] = *reinterpret_cast<const uint4*>(vsrc + lane_off);
          if constexpr (!KV8) {
            kreg[1] = *reinterpret_cast<const uint4*>(ksrc + lane_off + 16);
            vreg[1] = *reinterpret_cast<const uint4*>(vsrc + lane_off + 16);
          }
          return;
        }
      }
      long long off;
      if constexpr (PAGED) {
        const long long pg = page;
        off = (long long)KVB * pg * p.kv_stride_page + (long long)(key & (p.page_size - 1)) * kv_row_bytes;
      } else {
        off = (krow0 + key) * kv_row_bytes;
      }
      const char* ksrc = k_head + off;
      const char* vsrc = v_head + off;
      if constexpr (PAGED && !KV8) {
        // two sources: keys from position ctx on are rows of the step's own tensors
        if (p.k_new != nullptr && key >= ctx) {
          const long long noff = (long long)(q_start - ctx + key) * (2 * p.new_stride_t) + 2ll * kvh * p.new_stride_h + 32 * sc2;
          ksrc = reinterpret_cast<const char*>(p.k_new) + noff;
          vsrc = reinterpret_cast<const char*>(p.v_new) + noff;
        }
      }
      kreg[0] = *reinterpret_cast<const uint4*>(ksrc);
      vreg[0] = *reinterpret_cast<const uint4*>(vsrc);
      if constexpr (!KV8) {
        kreg[1] = *reinterpret_cast<const uint4*>(ksrc + 16);
        vreg[1] = *reinterpret_cast<const uint4*>(vsrc + 16);
      }
    };
    auto stage = [&](unsigned buf, uint4 (&kreg)[2], uint4 (&vreg)[2]) {
      if constexpr (KV8) {
        // 16 fp8 values per tensor -> two 16-byte chunks of 16-bit values
        const uint32_t kw[4] = {kreg[0].x, kreg[0].y, kreg[0].z, kreg[0].w};
        const uint32_t vw[4] = {vreg[0].x, vreg[0].y, vreg[0].z, vreg[0].w};
        uint32_t ko[8], vo[8];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const auto klo = __builtin_amdgcn_cvt_pk_f32_fp8((int)kw[x], false), khi = __builtin_amdgcn_cvt_pk_f32_fp8((int)kw[x], true);
          const auto vlo = __builtin_amdgcn_cvt_pk_f32_fp8((int)vw[x], false), vhi = __builtin_amdgcn_cvt_pk_f32_fp8((int)vw[x], true);
          if constexpr (BF16) {
            ko[2 * x] = pack_bf16(klo[0] * p.k_scale, klo[1] * p.k_scale);
            ko[2 * x + 1] = pack_bf16(khi[0] * p.k_scale, khi[1] * p.k_scale);
            vo[2 * x] = pack_bf16(vlo[0] * p.v_scale, vlo[1] * p.v_scale);   // (converted on to fp16 below)
            vo[2 * x + 1] = pack_bf16(vhi[0] * p.v_scale, vhi[1] * p.v_scale);
          } else {
            ko[2 * x] = pack_f16(klo[0] * p.k_scale, klo[1] * p.k_scale);
            ko[2 * x + 1] = pack_f16(khi[0] * p.k_scale, khi[1] * p.k_scale);
            vo[2 * x] = pack_f16(vlo[0] * p.v_scale, vlo[1] * p.v_scale);
            vo[2 * x + 1] = pack_f16(vhi[0] * p.v_scale, vhi[1] * p.v_scale);
          }
        }
        kreg[0] = make_uint4(ko[0], ko[1], ko[2], ko[3]);
        kreg[1] = make_uint4(ko[4], ko[5], ko[6], ko[7]);
        vreg[0] = make_uint4(vo[0], vo[1], vo[2], vo[3]);
        vreg[1] = make_uint4(vo[4], vo[5], vo[6], vo[7]);
      }
      if constexpr (BF16) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          vreg[x].x = bf16x2_to_f16x2(vreg[x].x);
          vreg[x].y = bf16x2_to_f16x2(vreg[x].y);
          vreg[x].z = bf16x2_to_f16x2(vreg[x].z);
          vreg[x].w = bf16x2_to_f16x2(vreg[x].w);
        }
      }
      lds_store_b128(buf + st_k0, kreg[0]);
      lds_store_b128(buf + st_k1, kreg[1]);
      lds_store_b128(buf + st_v0, vreg[0]);
      lds_store_b128(buf + st_v1, vreg[1]);
    };

    // staged form: tile t of the request is rows key_ptr[b] + 64 t .. + 63 of the staging area (whole tiles: the
    // request's rows are padded to a multiple of 64)
    const long long stage_req = STAGED ? (long long)key_ptr[b] * stage_row_bytes : 0;
    auto fetch_dma = [&](int t, unsigned buf) {
      const long long adv = stage_req + (long long)t * (kDN * stage_row_bytes);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        lds_dma16(stage_k + adv, dma_k[j], lds_base + buf + (8 * wave + 4 * j) * kDRowB);
        lds_dma16(stage_v + adv, dma_v[j], lds_base + buf + kDTile + (8 * wave + 4 * j) * kDRowB);
      }
    };

    f32x16 acc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[db][i] = 0.f;
    float m_run = kDMinusBig, l_run = 0.f;

    // every wave is done with the previous block's tiles
    __syncthreads();
    int page_next = 0;
    if constexpr (STAGED) {
      fetch_dma(0, 0u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      uint4 kreg[2], vreg[2];
      fetch(0, page_of(0), kreg, vreg);
      stage(0u, kreg, vreg);
      page_next = page_of(1);
    }
    __syncthreads();

    auto tile = [&](int t, auto buffer) {
      constexpr unsigned cur = decltype(buffer)::value * 2 * kDTile;
      constexpr unsigned nxt = (1 - decltype(buffer)::value) * 2 * kDTile;
      uint4 kreg[2], vreg[2];
      const bool more = t + 1 < num_tiles;
      if constexpr (STAGED) {
        if (more) fetch_dma(t + 1, nxt);
      } else {
        if (more) fetch(t + 1, page_next, kreg, vreg);
        page_next = page_of(t + 2);
      }

      if (t < w_tiles) {
        // ---- S^T = K Q^T: two 32-key blocks ----
        f32x16 s[2];
        // The K fragments of kQkAhead steps are in flight at a time, read by instructions the compiler cannot reorder
        // (register + immediate offset: no address arithmetic) and waited for by count: left to itself it reads one
        // fragment, waits for it and multiplies, sixteen times over.
        {
          f32x16 zero;  // (a constant operand of each block's first instruction: no registers are cleared)
#pragma unroll
          for (int i = 0; i < 16; ++i) zero[i] = 0.f;
          u32x4_t kf[kQkAhead];
          // step i: key block i & 1, k-step i >> 1 -- consecutive matrix instructions accumulate into different
          // blocks (a chain of eight on one block, then eight on the other: 1.4 % slower)
          auto rd = [&](int i, u32x4_t& dst) {
            if ((i & 1) == 0) lds_b128_at<cur>(koff[i >> 1], dst);
            else lds_b128_at<cur + 32 * kDRowB>(koff[i >> 1], dst);
          };
#pragma unroll
          for (int i = 0; i < kQkAhead; ++i) rd(i, kf[i]);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            u32x4_t& f = kf[i % kQkAhead];
            const int left = 15 - i < kQkAhead - 1 ? 15 - i : kQkAhead - 1;  // reads that may stay in flight
            static_assert(kQkAhead >= 1 && kQkAhead <= 8, "wait counts are written out up to 7");
            if (left == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f)::"memory");
            else if (left == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f)::"memory");
            else if (left == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f)::"memory");
            else if (left == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f)::"memory");
            else if (left == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f)::"memory");
            else if (left == 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(f)::"memory");
            else if (left == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f)::"memory");
            else asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(f)::"memory");
            s[i & 1] = mfma_qk<BF16>(__builtin_bit_cast(uint4, f), qf[i >> 1], (i >> 1) == 0 ? zero : s[i & 1]);
            if (i + kQkAhead < 16) rd(i + kQkAhead, f);
          }
        }
        // The V^T fragments of the second product, read the same way: fragment i (k-step i >> 2 of the tile's four,
        // column block i & 3) is two transposed 8-byte reads; kPvAhead fragments are in flight ahead of the
        // instruction that takes one, the first ones from here on, behind the maximum and the exponentials.  (Left
        // to itself the compiler issues a fragment's reads right in front of its instruction and waits: every one
        // of the sixteen then exposes the LDS latency.)
        s16x4_t vlo[kPvAhead + 1], vhi[kPvAhead + 1];
        auto rdv = [&](int i, s16x4_t& lo, s16x4_t& hi) {
          const int a = voff[i & 3];
          if ((i >> 2) == 0) {
            lds_tr16_at<cur>(a, lo);
            lds_tr16_at<cur + 8 * kDRowB>(a, hi);
          } else if ((i >> 2) == 1) {
            lds_tr16_at<cur + 16 * kDRowB>(a, lo);
            lds_tr16_at<cur + 24 * kDRowB>(a, hi);
          } else if ((i >> 2) == 2) {
            lds_tr16_at<cur + 32 * kDRowB>(a, lo);
            lds_tr16_at<cur + 40 * kDRowB>(a, hi);
          } else {
            lds_tr16_at<cur + 48 * kDRowB>(a, lo);
            lds_tr16_at<cur + 56 * kDRowB>(a, hi);
          }
        };
#pragma unroll
        for (int i = 0; i < kPvAhead; ++i) rdv(i, vlo[i], vhi[i]);
        // register i of block kb is key 64 t + 32 kb + (i & 3) + 8 (i >> 2) + 4 h
        const int key0 = kDN * t + 4 * h;
        if (kDN * t + kDN - 1 > w_pos0) {  // the tile reaches past some row's diagonal
          const int d = pos - key0;        // attended iff key offset <= d
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i)
              s[kb][i] = (32 * kb + (i & 3) + 8 * (i >> 2) <= d) ? s[kb][i] : -__builtin_inff();
        }
        float tmax = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, s[kb][i]);
        // the running maximum moves only when some row's tile maximum exceeds it by more than the slack:
        // the usual tile has no cross-lane step and no pass over the accumulators
        if (__builtin_amdgcn_ballot_w64(tmax * p.scale_log2 > m_run + kDSlack) != 0) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
          tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
          const float m_tile = tmax * p.scale_log2;
          const float m_new = (m_tile > m_run + kDSlack) ? m_tile : m_run;
          const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
          m_run = m_new;
          l_run *= alpha;
          if (t > 0) {
#pragma unroll
            for (int db = 0; db < 4; ++db) acc[db] *= alpha;
          }
        }
        // probabilities, 16 bit in place: registers 8 u .. 8 u + 7 of block kb are the B fragment of k-step u
        uint4 pf[2][2];
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][8 * u + j], p.scale_log2, -m_run));
              const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][8 * u + j + 1], p.scale_log2, -m_run));
              psum += p0 + p1;
              w[j >> 1] = pack_f16(p0, p1);
            }
            pf[kb][u] = make_uint4(w[0], w[1], w[2], w[3]);
          }
        l_run += psum;
        // ---- O^T += V^T P^T: element j of lane half h in k-step u of block kb is key 32 kb + 16 u + 8 (j >> 2) + 4 h + (j & 3) ----
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          constexpr int R = kPvAhead + 1;
          if (i + kPvAhead < 16) rdv(i + kPvAhead, vlo[(i + kPvAhead) % R], vhi[(i + kPvAhead) % R]);
          s16x4_t& lo4 = vlo[i % R];
          s16x4_t& hi4 = vhi[i % R];
          const int left = 15 - i < kPvAhead ? 15 - i : kPvAhead;  // fragments that may stay in flight (two reads each)
          static_assert(kPvAhead >= 1 && kPvAhead <= 7, "the counter has four bits");
          if (left == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo4), "+v"(hi4)::"memory");
          else if (left == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(lo4), "+v"(hi4)::"memory");
          else if (left == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(lo4), "+v"(hi4)::"memory");
          else if (left == 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(lo4), "+v"(hi4)::"memory");
          else if (left == 4) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(lo4), "+v"(hi4)::"memory");
          else if (left == 5) asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(lo4), "+v"(hi4)::"memory");
          else if (left == 6) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(lo4), "+v"(hi4)::"memory");
          else asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(lo4), "+v"(hi4)::"memory");
          const uint4 vf = __builtin_bit_cast(uint4, (s16x8_t)__builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
          acc[i & 3] = mfma_pv(vf, pf[i >> 3][(i >> 2) & 1], acc[i & 3]);
        }
      }

      if constexpr (STAGED) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile's rows are in LDS
      } else {
        if (more) stage(nxt, kreg, vreg);
      }
      __syncthreads();
    };
#pragma unroll 1
    for (int t = 0; t < num_tiles; t += 2) {
      tile(t, std::integral_constant<int, 0>{});
      if (t + 1 < num_tiles) tile(t + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue: lane (row r, half h) holds columns 32 db + 8 e + 4 h .. + 3 in registers 4 e .. 4 e + 3 of acc[db] ----
    if (w_rows) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
      const float l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      const float inv = 1.0f / l_tot;
      const long long orow = (long long)(q_start + my_row) * p.out_stride_t + (long long)head * p.out_stride_h;
      if (p.out_dtype == TMX_F32) {
        float* po = reinterpret_cast<float*>(p.out) + orow + 4 * h;
        if (row_ok) {
#pragma unroll
          for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              *reinterpret_cast<float4*>(po + 32 * db + 8 * e) =
                  make_float4(acc[db][4 * e] * inv, acc[db][4 * e + 1] * inv, acc[db][4 * e + 2] * inv, acc[db][4 * e + 3] * inv);
        }
      } else {
        // 16-bit output: lanes l and l ^ 32 exchange 8-byte pieces so that every lane stores 16 contiguous bytes:
        // of a pair of column groups (e, e + 1) the lower half-wave ends with all 8 columns of group e, the upper
        // half-wave with those of group e + 1
        uint16_t* po = reinterpret_cast<uint16_t*>(p.out) + orow;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            uint32_t a0, a1, b0, b1;
            if (p.out_dtype == TMX_BF16) {
              a0 = pack_bf16(acc[db][4 * e] * inv, acc[db][4 * e + 1] * inv);
              a1 = pack_bf16(acc[db][4 * e + 2] * inv, acc[db][4 * e + 3] * inv);
              b0 = pack_bf16(acc[db][4 * e + 4] * inv, acc[db][4 * e + 5] * inv);
              b1 = pack_bf16(acc[db][4 * e + 6] * inv, acc[db][4 * e + 7] * inv);
            } else {
              a0 = pack_f16(acc[db][4 * e] * inv, acc[db][4 * e + 1] * inv);
              a1 = pack_f16(acc[db][4 * e + 2] * inv, acc[db][4 * e + 3] * inv);
              b0 = pack_f16(acc[db][4 * e + 4] * inv, acc[db][4 * e + 5] * inv);
              b1 = pack_f16(acc[db][4 * e + 6] * inv, acc[db][4 * e + 7] * inv);
            }
            // swap: the upper half-wave's a with the lower half-wave's b
            const auto x0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            const auto x1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            // lower half: (own a, partner's a) = columns 0..3 | 4..7 of group e; upper half: (partner's b, own b) of group e + 1
            if (row_ok)
              *reinterpret_cast<uint4*>(po + 32 * db + 8 * (e + h)) = make_uint4(x0[0], x1[0], x0[1], x1[1]);
          }
      }
    }
  }
}


// The pass ahead of the staged form: K and V rows of the requests the dense kernel takes, gathered into the staging
// area -- row key_ptr[b] + j = key j of request b, from the cache page, the contiguous rows, the step's own tensors
// (keys from position seq_len - q_len on, when the call brings them) or an fp8 cache (descaled, rounded once to the
// type of q: what the register form's staging step does) -- K in the type of q, V as fp16 (the conversion the
// register form makes on the way into LDS).  Rows between a request's last key and the next multiple of 64 repeat
// the last key: the kernel masks them by position.  A thread moves 16 bytes of a K row and of a V row.  The step's
// fused cache write, which the register form does at its start, is done here.
template <bool BF16, bool PAGED, bool KV8>
__global__ __launch_bounds__(256) void dense_stage_kernel(const DenseParams p) {
  static_assert(!KV8 || PAGED, "an fp8 cache is paged");
  constexpr int KVB = KV8 ? 1 : 2;
  __shared__ int blk_ptr[kDenseMaxBatch + 1];
  __shared__ int key_ptr[kDenseMaxBatch + 1];
  const int tid = threadIdx.x;

  if constexpr (PAGED && !KV8) {
    if (p.cache_slots != nullptr) {
      const int units = p.total_q * p.Hkv * 16;  // (row, head, 16-byte chunk)
      const uint16_t* kn = reinterpret_cast<const uint16_t*>(p.k_new);
      const uint16_t* vn = reinterpret_cast<const uint16_t*>(p.v_new);
      uint16_t* kc = reinterpret_cast<uint16_t*>(const_cast<void*>(p.k));
      uint16_t* vc = reinterpret_cast<uint16_t*>(const_cast<void*>(p.v));
      for (int u = (int)blockIdx.x * 256 + tid; u < units; u += (int)gridDim.x * 256) {
        const int c = u & 15, th = u >> 4;
        const int hh = th % p.Hkv, t = th / p.Hkv;
        const long long slot = p.cache_slots[t];
        if (slot < 0) continue;
        const long long src = (long long)t * p.new_stride_t + (long long)hh * p.new_stride_h + 8 * c;
        const long long dst = (slot >> p.page_shift) * p.kv_stride_page + (slot & (p.page_size - 1)) * (long long)p.kv_stride_t +
                              (long long)hh * p.kv_stride_h + 8 * c;
        const uint4 kx = *reinterpret_cast<const uint4*>(kn + src);
        const uint4 vx = *reinterpret_cast<const uint4*>(vn + src);
        *reinterpret_cast<uint4*>(kc + dst) = kx;
        *reinterpret_cast<uint4*>(vc + dst) = vx;
      }
    }
  }

  dense_request_scan<256>(p, blk_ptr, key_ptr);
  const int total = key_ptr[p.batch];
  if (total > p.stage_rows) return;  // (the kernel behind it makes the same test)
  const int row_bytes = p.Hkv * kDRowB;
  char* const out_k = reinterpret_cast<char*>(p.stage);
  char* const out_v = out_k + p.stage_rows * (long long)row_bytes;
  // 16 threads per (staging row, K/V head): one 16-byte chunk of the 16-bit row each
  const unsigned pairs = (unsigned)total * (unsigned)p.Hkv;  // (the launcher keeps stage_rows x Hkv below 2^31)
  const int c = tid & 15;
  for (unsigned th = blockIdx.x * 16u + (unsigned)(tid >> 4); th < pairs; th += gridDim.x * 16u) {
    const int hh = (int)(th % (unsigned)p.Hkv);
    const int sr = (int)(th / (unsigned)p.Hkv);
    // request of staging row sr: the last b with key_ptr[b] <= sr (requests without rows have empty ranges)
    int lo = 0, hi = p.batch;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (key_ptr[mid] <= sr) lo = mid;
      else hi = mid;
    }
    const int b = lo;
    const int n = p.seq_lens[b];
    const int q_start = p.cu_q[b];
    const int ctx = n - (p.cu_q[b + 1] - q_start);
    const int key = imin(sr - key_ptr[b], n - 1);
    long long off;  // elements (bytes of an fp8 cache) from the start of k / v
    if constexpr (PAGED) {
      const long long page = p.block_table[(long long)b * p.block_table_stride + (key >> p.page_shift)];
      off = page * p.kv_stride_page + (long long)(key & (p.page_size - 1)) * p.kv_stride_t;
    } else {
      off = ((long long)p.cu_k[b] + key) * p.kv_stride_t;
    }
    off += (long long)hh * p.kv_stride_h + 8 * c;
    const char* ksrc = reinterpret_cast<const char*>(p.k) + KVB * off;
    const char* vsrc = reinterpret_cast<const char*>(p.v) + KVB * off;
    if constexpr (PAGED && !KV8) {
      if (p.k_new != nullptr && key >= ctx) {
        const long long noff = 2 * ((long long)(q_start - ctx + key) * p.new_stride_t + (long long)hh * p.new_stride_h + 8 * c);
        ksrc = reinterpret_cast<const char*>(p.k_new) + noff;
        vsrc = reinterpret_cast<const char*>(p.v_new) + noff;
      }
    }
    uint4 kx, vx;
    if constexpr (KV8) {
      const uint2 k8 = *reinterpret_cast<const uint2*>(ksrc);
      const uint2 v8 = *reinterpret_cast<const uint2*>(vsrc);
      const uint32_t kw[2] = {k8.x, k8.y}, vw[2] = {v8.x, v8.y};
      uint32_t ko[4], vo[4];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const auto klo = __builtin_amdgcn_cvt_pk_f32_fp8((int)kw[x], false), khi = __builtin_amdgcn_cvt_pk_f32_fp8((int)kw[x], true);
        const auto vlo = __builtin_amdgcn_cvt_pk_f32_fp8((int)vw[x], false), vhi = __builtin_amdgcn_cvt_pk_f32_fp8((int)vw[x], true);
        if constexpr (BF16) {
          ko[2 * x] = pack_bf16(klo[0] * p.k_scale, klo[1] * p.k_scale);
          ko[2 * x + 1] = pack_bf16(khi[0] * p.k_scale, khi[1] * p.k_scale);
          vo[2 * x] = pack_bf16(vlo[0] * p.v_scale, vlo[1] * p.v_scale);  // (converted on to fp16 below)
          vo[2 * x + 1] = pack_bf16(vhi[0] * p.v_scale, vhi[1] * p.v_scale);
        } else {
          ko[2 * x] = pack_f16(klo[0] * p.k_scale, klo[1] * p.k_scale);
          ko[2 * x + 1] = pack_f16(khi[0] * p.k_scale, khi[1] * p.k_scale);
          vo[2 * x] = pack_f16(vlo[0] * p.v_scale, vlo[1] * p.v_scale);
          vo[2 * x + 1] = pack_f16(vhi[0] * p.v_scale, vhi[1] * p.v_scale);
        }
      }
      kx = make_uint4(ko[0], ko[1], ko[2], ko[3]);
      vx = make_uint4(vo[0], vo[1], vo[2], vo[3]);
    } else {
      kx = *reinterpret_cast<const uint4*>(ksrc);
      vx = *reinterpret_cast<const uint4*>(vsrc);
    }
    if constexpr (BF16) {
      vx.x = bf16x2_to_f16x2(vx.x);
      vx.y = bf16x2_to_f16x2(vx.y);
      vx.z = bf16x2_to_f16x2(vx.z);
      vx.w = bf16x2_to_f16x2(vx.w);
    }
    const long long dst = (long long)sr * row_bytes + hh * kDRowB + 16 * c;
    *reinterpret_cast<uint4*>(out_k + dst) = kx;
    *reinterpret_cast<uint4*>(out_v + dst) = vx;
  }
}

}  // namespace tmx

namespace tmx {

int launch_dense_causal(const DenseParams& p, bool bf16, bool paged, hipStream_t s) {
  static std::atomic<unsigned> cus_cache[64];  // CUs per device, asked once (host threads of different GPUs may meet here)
  int dev = 0;
  TMX_CHECK_HIP(hipGetDevice(&dev));
  int cus = (dev >= 0 && dev < 64) ? (int)cus_cache[dev].load(std::memory_order_relaxed) : 0;
  if (cus == 0) {
    TMX_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (dev >= 0 && dev < 64) cus_cache[dev].store((unsigned)cus, std::memory_order_relaxed);
  }
  if (paged) {
    // (the cache-only fetch adds a 32-bit lane offset, row inside the page x row pitch, to a wave-uniform base)
    const long long span = (long long)(p.page_size - 1) * (p.kv_fp8 ? 1 : 2) * p.kv_stride_t + 256;
    if (span >= (1ll << 31)) {
      set_error("dense causal kernel: a page of the K/V cache spans %lld bytes (2 GiB or more)", span);
      return TMX_ERR_UNSUPPORTED;
    }
  }
  // one workgroup per CU (the registers of 8 waves x 256 fill a CU), a multiple of the K/V head chunks, and no
  // more workgroups than there are (64-row block, chunk) units at most
  long long grid = cus - cus % p.Y;
  if (grid < p.Y) grid = p.Y;
  const long long units = ((long long)p.total_q / kDM + p.batch) * p.Y;  // upper bound of the blocks x chunks
  if (grid > units) grid = units - units % p.Y;
  if (grid < p.Y) grid = p.Y;
  if (p.kv_fp8 && !paged) {
    set_error("dense causal kernel: an fp8 K/V cache is paged");
    return TMX_ERR_INVALID_ARG;
  }
  if (p.stage != nullptr && p.stage_rows > 0 && p.stage_rows * (long long)p.Hkv < (1ll << 31)) {
    // staged form: gather + convert first (a copy: a few workgroups per CU saturate the memory system), then the
    // kernel with one source
    const unsigned sgrid = (unsigned)(cus * 8);
    if (p.kv_fp8) {
      if (bf16) dense_stage_kernel<true, true, true><<<sgrid, 256, 0, s>>>(p);
      else dense_stage_kernel<false, true, true><<<sgrid, 256, 0, s>>>(p);
    } else if (bf16) {
      if (paged) dense_stage_kernel<true, true, false><<<sgrid, 256, 0, s>>>(p);
      else dense_stage_kernel<true, false, false><<<sgrid, 256, 0, s>>>(p);
    } else {
      if (paged) dense_stage_kernel<false, true, false><<<sgrid, 256, 0, s>>>(p);
      else dense_stage_kernel<false, false, false><<<sgrid, 256, 0, s>>>(p);
    }
    TMX_CHECK_HIP(hipGetLastError());
    if (bf16) dense_causal_kernel<true, false, false, true><<<(unsigned)grid, 512, 0, s>>>(p);
    else dense_causal_kernel<false, false, false, true><<<(unsigned)grid, 512, 0, s>>>(p);
  } else if (p.kv_fp8) {
    if (bf16) dense_causal_kernel<true, true, true><<<(unsigned)grid, 512, 0, s>>>(p);
    else dense_causal_kernel<false, true, true><<<(unsigned)grid, 512, 0, s>>>(p);
  } else if (bf16) {
    if (paged) dense_causal_kernel<true, true, false><<<(unsigned)grid, 512, 0, s>>>(p);
    else dense_causal_kernel<true, false, false><<<(unsigned)grid, 512, 0, s>>>(p);
  } else {
    if (paged) dense_causal_kernel<false, true, false><<<(unsigned)grid, 512, 0, s>>>(p);
    else dense_causal_kernel<false, false, false><<<(unsigned)grid, 512, 0, s>>>(p);
  }
  TMX_CHECK_HIP(hipGetLastError());
  return TMX_OK;
}

long long dense_stage_rows(const int32_t* cu_q, const int32_t* seq_lens, int batch, int min_rows_per_request) {
  if (min_rows_per_request < 0) min_rows_per_request = kDenseStageMinRows;
  long long rows = 0, q_rows = 0, requests = 0;
  for (int b = 0; b < batch; ++b) {
    const int q_len = cu_q[b + 1] - cu_q[b];
    if (q_len < kDenseMinRows) continue;
    rows += ((long long)seq_lens[b] + kDN - 1) / kDN * kDN;
    q_rows += q_len;
    ++requests;
  }
  if (requests == 0 || batch > kDenseMaxBatch || q_rows < requests * (long long)min_rows_per_request) return 0;
  return rows;
}

}  // namespace tmx

extern "C" int64_t tmx_dense_stage_rows(const int32_t* cu_seqlens_q_host, const int32_t* seq_lens_host, int batch,
                                        int min_rows_per_request) {
  if (batch <= 0 || cu_seqlens_q_host == nullptr || seq_lens_host == nullptr) return 0;
  return tmx::dense_stage_rows(cu_seqlens_q_host, seq_lens_host, batch, min_rows_per_request);
}
