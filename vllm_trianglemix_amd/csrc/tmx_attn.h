// Shared by the attention kernels (work-list / band workgroups, merge pass) and the host code that launches them:
// the MFMA traits of the two 16-bit types, the kernel parameter block, LDS / LDS-DMA helpers, the mapping of
// workgroups to K/V head chunks, and the rule that decides which q-blocks the band workgroups take.
#pragma once
#include "tmx_common.h"
#include "tmx_attn_switches.h"
#include "tmx_index_scan.h"  // the index build as the first thing workgroup 0 does (TMX_ATTN_BUILD_INDEX)
#include "tmx_dense.h"   // kDenseMinRows: which requests the dense causal kernel takes
#include "tmx_decode.h"  // decode_row: which rows the decode kernel takes

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace tmx {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned nt_u32x4;

// Output rows are written once per launch and read by nobody in it: stored with the non-temporal policy they do
// not displace the K/V lines, which are read twice (band runs and the pieces of the trailing rows), from the L2
// and the Infinity Cache (32k step, one box: 180.0 -> 176.6 us).  Query rows keep the default policy: a load
// instruction of a wave covers half a cache line per row and the next k-step's load the other half (loaded
// non-temporally the step went back to 178.5 us).
__device__ __forceinline__ void store_stream16(void* ptr, uint4 v) {
  __builtin_nontemporal_store(__builtin_bit_cast(nt_u32x4, v), reinterpret_cast<nt_u32x4*>(ptr));
}
__device__ __forceinline__ void store_stream16(void* ptr, float4 v) {
  __builtin_nontemporal_store(__builtin_bit_cast(nt_u32x4, v), reinterpret_cast<nt_u32x4*>(ptr));
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

struct Bf16Traits {
  using frag = bf16x8;
  // 16-key product: A / B hold 4 values per lane (keys 4 * (lane >> 4) .. + 3)
  static __device__ __forceinline__ f32x4 mfma16(s16x4 a, uint2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
  }
  // two floats -> one dword of two bf16 (round to nearest even): one v_cvt_pk_bf16_f32
  static __device__ __forceinline__ uint32_t cvt2(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
  }
  static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ short cvt(float x) {
    __bf16 h = (__bf16)x;
    return __builtin_bit_cast(short, h);
  }
  static __device__ __forceinline__ float to_float(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
  // 8 mantissa bits are too few for the probabilities: P = hi + lo, hi = p truncated to bf16 (exact
  // remainder), lo = the remainder rounded to bf16; the second product doubles as error compensation.
  static constexpr bool kSplitP = true;
  static __device__ __forceinline__ void pack2(float p0, float p1, uint32_t& hi, uint32_t& lo) {
    const uint32_t u0 = __float_as_uint(p0), u1 = __float_as_uint(p1);
    hi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);  // (u1 & 0xffff0000) | (u0 >> 16)
    const float r0 = p0 - __uint_as_float(u0 & 0xffff0000u), r1 = p1 - __uint_as_float(u1 & 0xffff0000u);
    lo = cvt2(r0, r1);
  }
  static __device__ __forceinline__ frag ones() {
    return __builtin_bit_cast(frag, make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u));
  }
};

struct F16Traits {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 mfma16(s16x4 a, uint2 b, f32x4 c) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t cvt2(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, f16x2));
  }
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
  static constexpr bool kSplitP = false;  // 11 mantissa bits: one term
  static __device__ __forceinline__ void pack2(float p0, float p1, uint32_t& hi, uint32_t& lo) {
    hi = cvt2(p0, p1);
    lo = 0;
  }
  static __device__ __forceinline__ frag ones() {
    return __builtin_bit_cast(frag, make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u));
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
  const int32_t* full_lens;    // [batch] length of each sequence once its prompt is complete, or NULL
  const void* k_new;           // the step's own K / V rows ([total_q, Hkv, D], row t belongs to query row t), or NULL:
  const void* v_new;           // keys from position seq_len - q_len on are read there instead of the paged cache
  int new_stride_t, new_stride_h;
  const long long* cache_slots;  // [total_q] flat cache slot of every row of k_new / v_new (negative: none), or NULL
  int total_q;
  int q_stride_t, q_stride_h, out_stride_t, out_stride_h;  // elements; checked to fit 31 bits
  int kv_stride_page, kv_stride_t, kv_stride_h, block_table_stride;
  int64_t bias_stride;
  int64_t bias_stride_b, bias_stride_h;  // per request / per head offsets of the bias (0: shared)
  int page_size, page_shift;  // page_size is a power of two
  int H, Hkv, G, Gw, gw_shift, block_m;  // Gw = 1 << gw_shift query heads per workgroup
  int Y;                      // units per work item: K/V heads x head chunks
  int num_items;
  int issue_hi;               // waves 4..7 fetch the K/V entries (TMX_ATTN_SHORT_QUERIES)
  int num_heavy, sched_w;     // the first num_heavy items are pieces of split q-blocks; light items per piece (schedule)
  int band_on;                // sparse q-blocks in canonical position run on the band path (their own workgroups)
  int band_rl;                // q-blocks per band workgroup
  int band_tail0, band_rls;   // q-blocks from band_tail0 on are left to the work-list workgroups, band_rls each,
                              // for when their list is done (band_rls = 0: none, band_tail0 = num_qblocks)
  int n_generic;              // workgroups 0 .. n_generic - 1 walk the work list, the others are band workgroups
  int num_qblocks;            // q-blocks of the batch
  int batch, chunk_tiles;     // requests of the step; tiles per piece of a split q-block
  int walk_shift;             // pieces of split q-blocks walk their tile list in 1 << walk_shift interleaved parts
  int dense_on;               // requests of kDenseMinRows query rows or more are computed by the dense causal kernel
  int decode_on;              // one-row requests whose row is dense are computed by the decode kernel
  int xcd_generic, xcd_band;  // workgroup -> K/V head chunk mapping that puts the chunks of a K/V head on one XCD
  int exp_mode;               // tuning builds, bits: 1 = work-list workgroups return at once, 2 = band workgroups do,
                              // 4 / 8 / 16 = band workgroups skip their output stores / query loads / key fetches, 32 = default-policy
                              // output stores, 64 / 128 / 256 = whole-line / whole-row / quad-contiguous output stores (wrong placement: timing only)
  long long units_total;      // work items x Y
  int mode, si, sl, last, tri_min_len;
  int bias_rows, bias_cols, bias_dtype, out_dtype;
  float scale_log2;
  float k_scale, v_scale;     // fp8 K/V cache: value = fp8 x scale
  // In-launch index build (ix_build): workgroup 0 builds the index from `ix`, then stores ix_epoch to *ix_flag
  // (device scope, release); the other work-list workgroups wait for that value before they read anything of the
  // index.  The band workgroups of that form do not read the index (they place a q-block from cu_q instead of qb_meta).
  int ix_build;
  unsigned long long* ix_flag;
  unsigned long long ix_epoch;
  IndexParams ix;
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

// The value, hidden from the optimiser: what is computed from it is computed where it is written.
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
// This lane's number inside its wave, computed on the spot (two v_mbcnt): rare paths inside the main loops
// use it instead of a value derived from threadIdx at kernel entry, which would stay live (or be spilled and
// reloaded, with a wait for every transfer in flight) across the whole loop.
__device__ __forceinline__ int lane_now() {
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

template <typename F>
__device__ __forceinline__ F lds_read_b128(unsigned addr) {
  return __builtin_bit_cast(F, *(const __attribute__((address_space(3))) uint4*)(uintptr_t)addr);
}
__device__ __forceinline__ s16x4 lds_read_tr16_b64(unsigned addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)addr);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// Running maximum before any key was seen: finite, so that exp2(m_old - m_new) and exp2(-inf - m)
// need no special cases (scores are finite; a fully masked row still ends with sum 0 -> NaN).
constexpr float kMinusBig = -1.0e30f;

// Workgroup shape of the attention kernel: 4 (head, 32-row) slots x 2 sixteen-row blocks = 8 waves
// of 16 query rows each; all of them share the K/V tiles of one K/V head.
constexpr int kWaves = 2 * kWavesPerWG;
constexpr int kRingSlots = 3;  // K/V ring entries of the attention kernel (plus a private Q region per wave)
// Key tiles per softmax step.
constexpr int kGroupMax = 1;
// The running row maximum is kept until a tile's maximum exceeds it by more than this (log2 units).
constexpr float kMaxSlack = 8.0f;

// One LDS-DMA instruction: every lane fetches 16 bytes from (wave-uniform 64-bit base + its own
// 32-bit byte offset); the wave's 1 KiB lands at LDS byte address lds_addr + 16 * lane.
// Issued from inline asm on purpose: the compiler would otherwise wait for ALL transfers in flight
// (vmcnt(0)) before every LDS read; here the waits are counted by hand and followed by a barrier.
__device__ __forceinline__ void glds16_s(const void* sbase, int voff, unsigned lds_addr) {
  // wave-uniform operands are pinned to scalar registers (a uniform value the compiler happened
  // to compute on the vector side would not assemble)
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
// Per-lane 64-bit address form (ragged edges only).
__device__ __forceinline__ void glds16(const void* src, unsigned lds_addr) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off"
      :
      : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_addr))
      : "memory", "m0");
}

// s_waitcnt takes an immediate: wait until at most `n` of this wave's vector-memory operations are
// outstanding, rounded DOWN to a multiple of STEP and capped at 4 * STEP (waiting for fewer than
// asked is always safe).  The common steady-state value is tested first.
template <int STEP>
__device__ __forceinline__ void wait_vmcnt_levels(int n) {
  if (n >= 2 * STEP && n < 3 * STEP) wait_vmcnt<2 * STEP>();
  else if (n >= 4 * STEP) wait_vmcnt<4 * STEP>();
  else if (n >= 3 * STEP) wait_vmcnt<3 * STEP>();
  else if (n >= STEP) wait_vmcnt<STEP>();
  else wait_vmcnt<0>();
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

// The cache write of the step, fused into the attention launch (the reference's reshape_and_cache_flash,
// csrc/cache_kernels.cu:282-355, called ahead of its attention kernel at
// vllm/v1/attention/backends/rocm_attn.py:345-354): before its own work every workgroup copies its share of the
// step's key / value rows to their cache slots.  Nothing in this launch reads those slots -- the keys of the
// step come from k_new / v_new -- so the copy needs no ordering against the attention; the next launch on the
// stream sees the finished cache.
template <int D>
__device__ __forceinline__ void fused_cache_write(const AttnParams& p) {
  constexpr int CPR = D / 8;  // 16-byte chunks per row of one head
  const int units = p.total_q * p.Hkv * CPR;  // (row, head, chunk), below 2^31 (checked on the host)
  const int stride = (int)gridDim.x * (int)blockDim.x;
  const uint16_t* kn = reinterpret_cast<const uint16_t*>(p.k_new);
  const uint16_t* vn = reinterpret_cast<const uint16_t*>(p.v_new);
  uint16_t* kc = reinterpret_cast<uint16_t*>(const_cast<void*>(p.k));
  uint16_t* vc = reinterpret_cast<uint16_t*>(const_cast<void*>(p.v));
  for (int u = (int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x; u < units; u += stride) {
    const int c = u % CPR, th = u / CPR;
    const int h = th % p.Hkv, t = th / p.Hkv;
    const long long slot = p.cache_slots[t];
    if (slot < 0) continue;
    const long long src = (long long)t * p.new_stride_t + (long long)h * p.new_stride_h + 8 * c;
    const long long dst = (slot >> p.page_shift) * p.kv_stride_page + (slot & (p.page_size - 1)) * (long long)p.kv_stride_t +
                          (long long)h * p.kv_stride_h + 8 * c;
    const uint4 kx = *reinterpret_cast<const uint4*>(kn + src);
    const uint4 vx = *reinterpret_cast<const uint4*>(vn + src);
    *reinterpret_cast<uint4*>(kc + dst) = kx;
    *reinterpret_cast<uint4*>(vc + dst) = vx;
  }
}

// K/V head chunk and lane of workgroup `id` of a role.  Plain mapping: chunk = id % Y.  A K/V head with more than
// four query heads is served by several chunks, which read the same keys: when the K/V heads divide the 8 XCDs
// (consecutive workgroup ids go round the XCDs) and the role's workgroup count is a multiple of 8 x chunks, the
// chunks of a head are placed 8 ids apart -- on the same XCD, started together, walking the same keys -- so that
// the second one finds them in that XCD's L2 instead of fetching them again.
struct UnitLane {
  int kvc, lane;
};
__device__ __forceinline__ UnitLane unit_of(int id, int Y, int Hkv, int chunks, int xcd_map) {
  if (!xcd_map) return UnitLane{id % Y, id / Y};
  // 2: the Y units of a lane on ONE XCD (ids 8 apart), lanes round the XCDs: the workgroups of an XCD read whole
  // rows of the query / output / cache tensors between them
  if (xcd_map == 2) return UnitLane{(id >> 3) % Y, (id & 7) + 8 * (id / (8 * Y))};
  const int x = id & 7, c = (id >> 3) % chunks, t = (id >> 3) / chunks;
  return UnitLane{(x % Hkv) * chunks + c, t * (8 / Hkv) + x / Hkv};
}

// ---------------------------------------------------------------------------------------------------
// Band path.  In a long triangle request almost every q-block is in "canonical position": 32 full rows
// starting at a multiple of 32, none of them among the trailing dense rows, its window clear of the sink
// columns' tile.  Such a q-block attends to the 16-key group of the sink columns and to the 64 keys
// [pos0 - 32, pos0 + 32); each of its two 16-row halves to the sink group and 48 of those keys.  That is a
// fixed shape: ONE softmax step over four 16-key groups per wave, masks that depend on the lane only, and
// consecutive q-blocks share three quarters of their keys.  A band workgroup takes a run of consecutive
// q-blocks of one K/V head chunk and keeps a rolling window of 16-key halves ([K | V], LDS-DMA) in LDS:
// per q-block it fetches 32 new keys, the query rows go straight to registers (fetched one q-block ahead,
// once the score products no longer need the previous ones), one barrier per q-block.
//
// Whether a q-block takes this path is decided by the same function in the band workgroups (which walk
// q-blocks) and in the workgroups that walk the work list (which skip such items).
struct BandRule {
  int sl, last, mode, tri_min_len;
};

// q-block at position pos0 of a request with n keys so far, n_full once its prompt is complete
__device__ __forceinline__ bool band_qblock(const BandRule& r, int pos0, int n, int n_full) {
  return (pos0 & 31) == 0 && pos0 >= 64 && n - pos0 >= 32 && (n_full - (pos0 + 31)) >= r.last &&
         effective_mode(r.mode, n_full, r.tri_min_len) == TMX_MODE_TRIANGLE;
}

constexpr int kMaxDevices = 64;  // per-device host caches (resident workgroups, the static work area's address)

// ---- host side: one launcher per kernel file ----
// Work-list + band workgroups (tmx_attn_kernel.hip).  bf16: the 16-bit type of q / k / v (else fp16).
int launch_attention_kernel(const AttnParams& p, bool bf16, int head_dim, bool paged, bool has_bias, bool fp8, hipStream_t s);
// Whether launch_attention_kernel has the form that builds the index in the launch for this shape (p.ix_build).
bool attention_kernel_builds_index(int head_dim, bool has_bias, bool fp8);
// Merge pass over the pieces of the split q-blocks (tmx_attn_combine.hip).
int launch_combine_kernel(const AttnParams& p, bool bf16, int head_dim, int64_t num_split, hipStream_t s);

}  // namespace tmx
