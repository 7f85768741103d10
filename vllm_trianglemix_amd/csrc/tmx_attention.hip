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
//   A persistent grid of (CUs x resident workgroups per CU) workgroups; workgroup b serves the
//   K/V head chunk b % Y for its whole life and walks a static share of the work list (q-block
//   pieces from the index; see the schedule inside the kernel).  A workgroup is 8 waves; wave w
//   owns 16 query rows of one query head: slot = w & 3 picks (head of the K/V group, 32-row
//   block), w >> 2 the 16-row half, so all waves share the same K/V head and every K/V tile is
//   staged once in LDS (a ring of 3 tiles filled by LDS-DMA) for all of them.
//
// Per 32-key tile and wave (all MFMA are 16x16x32, fp32 accumulate):
//   S^T[key][row]  = K_tile . Q^T      A = K rows from LDS (ds_read_b128, XOR-swizzled
//                                      16-byte chunks), B = Q fragments held in registers
//   online softmax  per lane: the lane's column of S^T is ONE query row
//                   (row = lane & 15), its 8 values are 8 keys; a new row maximum
//                   (rare: the running one is kept within a slack) needs two
//                   cross-lane steps (lanes l, l^16, l^32, l^48)
//   O^T[d][row]    += V_tile^T . P^T   A = V columns by ds_read_b64_tr_b16 (hardware
//                                      transpose), B = P packed to 16 bit in registers
// Both products sum over an index whose order is free (head dim for S, keys for
// O), so the fragments are laid out for the loads: Q/K fragments are contiguous
// 16-byte pieces of a row, P needs no cross-lane movement between the two
// products, and every lane ends with 8 consecutive output columns per store.
#include "tmx_common.h"
#include "tmx_dense.h"
#include "tmx_decode.h"

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
// 0: the compiler keeps the band workgroups' sixteen mask comparisons as lane masks in scalar registers (and brings
// them back through v_readlane, two per selection).  1: compared and selected in the loop -- 44 fewer spilled scalar
// registers, but the three thresholds then live in vector registers the loop does not have (4 spilled): measured
// 163 us against 160 for the launch, so off.
#ifndef TMX_LIST_PRIO
#define TMX_LIST_PRIO 0  // wave priority of the work-list workgroups in a launch that has band workgroups (1, 3: no effect measured)
#endif
#ifndef TMX_BAND_PRIO
#define TMX_BAND_PRIO 0  // wave priority of the band workgroups (0: as the work-list workgroups; 1, 3: the work list starves, 184 us against 161)
#endif
#ifndef TMX_BAND_MASKS_IN_LOOP
#define TMX_BAND_MASKS_IN_LOOP 0
#endif
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

// bf16 launches: the V rows are turned into fp16 inside LDS by the waves that fetched them (every value of the
// bf16 range that fp16 holds converts exactly; larger ones saturate at +-65504, smaller ones than 2^-24 vanish)
// and multiplied by ONE fp16 term of the probabilities (11 bits) instead of two bf16 terms (band workgroups: 34
// matrix instructions per q-block and wave instead of 50; work list: 24 per tile instead of 40), and the row
// sums come from the matrix cores as well (a ones operand in V's place: they add up the same rounded
// probabilities as the weighted sum, so that rounding a probability only shifts weight between keys).
// 0: V stays bf16, two-term probabilities.
#ifndef TMX_V16
#define TMX_V16 1
#endif
// The same for the workgroups that walk the work list: off.  There the conversion falls to the four waves that
// fetch the tiles and run the scalar walk, which every tile step already waits for: measured at 32k tokens,
// Qwen3-8B heads, same box, the kernel went from 157.5 to 165.8 us with it (parity tests green either way).
#ifndef TMX_V16_LIST
#define TMX_V16_LIST 0
#endif
// Partial results of split q-blocks stored with the non-temporal policy (A/B switch).
#ifndef TMX_PARTIAL_NT
#define TMX_PARTIAL_NT 0
#endif

// kvc: the workgroup's K/V head chunk; [qb, qb_end): its run of q-blocks.
// FP8: K and V are an fp8 (e4m3fn) paged cache.  The wave that fetches eight rows of a half (1 KiB of fp8 bytes)
// lands them in the first half of those rows' own place in the window and, one q-block ahead of their use like the
// V conversion of bf16 launches, expands them there into the 16-bit image the arithmetic reads: value x scale
// rounded once to the 16-bit type of q (the descale of the reference's attention kernels,
// vllm/v1/attention/backends/rocm_attn.py:421-422); V of a bf16 launch goes on from that bf16 value to fp16 (exact).
template <typename TR, int D, bool PAGED, bool FP8 = false>
__device__ __forceinline__ void band_body(const AttnParams& p, const unsigned lds_base, const int kvc, int qb, const int qb_end,
                                          const bool own_stamps) {
  constexpr int KS = D / 32, DB = D / 16, ROWB = D * 2, CPR = D / 8;
  constexpr int RPP = 1024 / ROWB;   // rows per 1-KiB LDS-DMA piece
  constexpr int HALFB = 32 * ROWB;   // a 16-key half: [K 16 rows | V 16 rows]
  constexpr int NSLOT = 8;           // rolling window: 4 halves in use, 4 arriving (two q-blocks ahead)
  constexpr int KVB = FP8 ? 1 : 2;   // bytes per K/V element in memory
  constexpr int PWB = FP8 ? 1 : 8 / RPP;  // pieces per wave and pair of halves
  static_assert(!FP8 || (D == 128 && PAGED), "fp8 K/V: paged cache, head size 128");
  constexpr bool V16 = TR::kSplitP && TMX_V16 != 0;  // V converted to fp16 in LDS, one-term fp16 probabilities
  constexpr bool PRECISE = TR::kSplitP && !V16;
  using frag = typename TR::frag;
  static_assert((NSLOT + 1) * HALFB <= kRingSlots * 64 * ROWB + kWaves * 16 * ROWB, "band window does not fit the kernel's LDS");
  // 16-bit output rows leave through a 1-KiB stage per wave behind the window (head size 128: the window takes 72
  // of the kernel's 80 KiB): rounds of eight half rows, written by the lanes that hold them, read back one 16-byte
  // piece per lane in row order, so that a store instruction covers 128 contiguous bytes of each of eight rows
  // (adjacent lanes, adjacent bytes) instead of a 64-byte piece of each of 16 rows, four lanes 16 apart each.  Same bytes, other shape: measured with the
  // placement alone changed, band workgroups only 137.7 -> 123.1 us, the whole 32k launch 163.8 -> 153.5 us.
  constexpr bool ROWST = (NSLOT + 1) * HALFB + kWaves * 1024 <= kRingSlots * 64 * ROWB + kWaves * 16 * ROWB;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 15, g = lane >> 4;
  const int hg = wave & 3, hf = wave >> 2;  // head of the group, 16-row half of the q-block
  const int chunks = p.G / p.Gw;
  const int kvh = kvc / chunks;
  const int head = kvh * p.G + (kvc - kvh * chunks) * p.Gw + hg;
  const int q_row_bytes = 2 * p.q_stride_t, kv_row_bytes = KVB * p.kv_stride_t;

  // ---- LDS-DMA share of this wave: 8 rows of one half of each pair, as 1-KiB pieces ----
  // Two pieces per wave (head size 128): four rows of K and the same four rows of V, so that every wave has the
  // same share of the V conversion.  One piece (head size 64): eight rows of K (waves 0, 1 of a group of four)
  // or of V (waves 2, 3).
  // fp8: one piece too, eight rows of 128 bytes.
  const int pc_t[2] = {PWB == 2 ? 0 : ((wave & 3) >> 1), 1};                    // piece x: 0 = K rows, 1 = V rows
  const int pc_r = PWB == 2 ? RPP * (wave & 3) : 8 * (wave & 1);                // its first row inside the half
  const bool wave_has_v = PWB == 2 || pc_t[0] == 1;
  const bool wave_converts = FP8 || (V16 && wave_has_v);
  const char* kv_src[PWB];
  const char* new_src[PWB];
  int kv_voff[PWB];
  unsigned dma_off[PWB];
  const int new_row_bytes = 2 * p.new_stride_t;
#pragma unroll
  for (int x = 0; x < PWB; ++x) {
    if constexpr (FP8) {
      // (the fp8 rows land as they are: lane L's 16 bytes are bytes 16 (L % 8) .. of row pc_r + L / 8)
      kv_voff[x] = (pc_r + lane / 8) * kv_row_bytes + 16 * (lane % 8);
    } else {
      const int row = pc_r + lane / CPR;
      kv_voff[x] = row * kv_row_bytes + 16 * ((lane % CPR) ^ (pc_t[x] ? v_chunk_swizzle<D>(row) : k_chunk_swizzle<D>(row)));
    }
    kv_src[x] = reinterpret_cast<const char*>(pc_t[x] ? p.v : p.k) + (long long)KVB * kvh * p.kv_stride_h;
    new_src[x] = reinterpret_cast<const char*>(pc_t[x] ? p.v_new : p.k_new) + 2ll * kvh * p.new_stride_h;
    dma_off[x] = (unsigned)(pc_t[x] * 16 * ROWB + pc_r * ROWB);
  }
  const unsigned conv_off = dma_off[PWB - 1];  // this wave's V piece inside a half
  const unsigned sink_base = lds_base + NSLOT * HALFB;
  auto slot_base = [&](int half_no) { return lds_base + ((unsigned)half_no % NSLOT) * HALFB; };  // (half numbers are >= 0)

  // ---- compute side: per-lane read offsets inside a half, lane-constant masks ----
  const int k_base = qi * ROWB + 16 * (g ^ k_chunk_swizzle<D>(qi));
  const int v_row = 4 * g + ((lane & 15) >> 2);
  const int v_base = 16 * ROWB + v_row * ROWB + 8 * (lane & 3) + 32 * (v_chunk_swizzle<D>(v_row) >> 1);
  // this lane's row is row qi of the wave's 16, its 4 keys of a 16-key group are 4 g .. 4 g + 3.  Group A
  // starts 32 keys below the wave's first row, B 16 below, C at it: distance row - key = qi + {32, 16, 0} - key.
  // sink group: key <= si; A: row - key = qi + 32 - key <= sl; B: qi + 16 - key <= sl; C: key <= qi.  In terms
  // of the lane's element c (key = 4 g + c): c <= c_sink, c >= c_win (A), c >= c_win - 16 (B), c <= c_diag.
  int c_sink = p.si - 4 * g, c_win = qi + 32 - p.sl - 4 * g, c_diag = qi - 4 * g;  // (never changed: see the masks)
  const float mult = p.scale_log2;
  // query rows and output rows: wave-uniform base of the wave's 16 rows + this lane's 32-bit offset
  const char* const q_head = reinterpret_cast<const char*>(p.q) + 2ll * head * p.q_stride_h;
  auto load_q = [&](int q_tok0, frag (&q)[KS]) {
#ifdef TMX_TUNING
    if (p.exp_mode & 8) q_tok0 = q_tok0 & 31;  // every q-block reads the same rows (cached)
#endif
    const char* rows = q_head + (long long)(q_tok0 + 16 * hf) * q_row_bytes;
    // (the lane's 32-bit offset is computed here, from the lane number: a 64-bit per-lane pointer kept across
    // the loop was the first thing the register allocator spilled)
    const int l = opaque(lane);
    const unsigned qo = (unsigned)((l & 15) * q_row_bytes + 16 * (l >> 4));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      q[ks] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(rows + (qo + 64u * ks)));
  };

  bool fresh = true;  // nothing of this workgroup is in flight or being read
#ifdef TMX_STAMPS
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long ts0, ts1, ts2, ts3, ts4, ts5, ts6;
  TMX_STAMP(ts0);
  const unsigned long long t_begin = ts0;
#endif
#pragma unroll 1
  while (qb < qb_end) {
    // ---- next segment: consecutive band q-blocks of one request inside [qb, qb_end) ----
    const int b = sload(p.qb_meta + 2 * qb), row0 = sload(p.qb_meta + 2 * qb + 1);
    const int q_start = sload(p.cu_q + b);
    const int q_len = sload(p.cu_q + b + 1) - q_start;
    const int n = sload(p.seq_lens + b);
    const int nf = p.full_lens != nullptr ? imax(n, sload(p.full_lens + b)) : n;
    const int ctx = n - q_len;
    const int blocks_left = (q_len - row0 + 31) / 32;  // q-blocks of this request from qb on
    // band q-blocks of the request are k_lo .. k_hi - 1 (k = row0 / 32)
    int k_lo = 0, k_hi = 0;
    if ((ctx & 31) == 0 && effective_mode(p.mode, nf, p.tri_min_len) == TMX_MODE_TRIANGLE) {
      k_lo = ctx >= 64 ? 0 : (64 - ctx) / 32;
      const int lim = nf - p.last - 31 - ctx;
      k_hi = lim >= 0 ? imin(q_len / 32, lim / 32 + 1) : 0;
    }
    const int k0 = row0 / 32;
    const int k_first = imax(k0, k_lo);
    const int k_stop = imin(k_hi, k0 + imin(blocks_left, qb_end - qb));
    const int qb_next = qb + imin(blocks_left, qb_end - qb);  // where the walk goes on after this request
    if (k_first >= k_stop) {
      qb = qb_next;
      continue;
    }
    const long long krow0 = PAGED ? 0 : (long long)sload(p.cu_k + b);
    const int32_t* const bt = PAGED ? p.block_table + (long long)b * p.block_table_stride : nullptr;
    // Byte offset of the 16-key half that starts at key0 (one page, or part of one, of a paged cache).  Two
    // sources: halves from position ctx on are rows of the step's own tensors, flagged by bit 0 of the offset
    // (offsets are multiples of 16; a band request's context is a multiple of 32, so no half holds both kinds).
    auto half_offset = [&](int key0) -> long long {
      if constexpr (PAGED) {
        if (!FP8 && p.k_new != nullptr && key0 >= ctx) return ((long long)(q_start - ctx + key0) * new_row_bytes) | 1;
        return (long long)KVB * sload(bt + (key0 >> p.page_shift)) * p.kv_stride_page + (long long)(key0 & (p.page_size - 1)) * kv_row_bytes;
      } else {
        return (krow0 + key0) * kv_row_bytes;
      }
    };
    auto issue_at = [&](long long off, unsigned half_base) {
      if (PAGED && (off & 1)) {
        const int lrow = opaque(lane) / CPR;
#pragma unroll
        for (int x = 0; x < PWB; ++x)
          glds16_s(new_src[x] + (off - 1), kv_voff[x] + (pc_r + lrow) * (new_row_bytes - kv_row_bytes), half_base + dma_off[x]);
        return;
      }
#pragma unroll
      for (int x = 0; x < PWB; ++x) glds16_s(kv_src[x] + off, kv_voff[x], half_base + dma_off[x]);
    };
    // This wave's V piece of the half at half_base, bf16 -> fp16 in place (the piece has landed: the caller
    // waited for it; the next barrier shows the result to the other waves).  A lane converts the 16 bytes it
    // fetched.  Round-toward-zero conversion: in-range values are exact either way, larger ones saturate.
    auto convert_v = [&](unsigned half_base) {
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      const int l = opaque(lane);
      const unsigned a = half_base + conv_off + 16 * l;
      const uint4 raw = lds_read_b128<uint4>(a);
      const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
      if constexpr (FP8) {
        // 16 fp8 values of row crow -> two 16-byte chunks of its 16-bit image, swizzled as the arithmetic reads it.
        // Every lane of the wave has its raw bytes before any lane writes (the piece and its image are this wave's).
        const bool cv = pc_t[0] != 0;
        const int crow = pc_r + (l >> 3), cc = l & 7;
        const float sc = cv ? p.v_scale : p.k_scale;
        uint32_t o[8];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)rw[x], false);
          const auto hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)rw[x], true);
          o[2 * x] = TR::cvt2(lo[0] * sc, lo[1] * sc);
          o[2 * x + 1] = TR::cvt2(hi[0] * sc, hi[1] * sc);
        }
        if constexpr (V16) {
          if (cv) {
            // the bf16 values (what the descale produces), carried on to fp16
#pragma unroll
            for (int x = 0; x < 8; ++x)
              o[x] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(o[x] << 16), __uint_as_float(o[x] & 0xffff0000u)));
          }
        }
        const int swz = cv ? v_chunk_swizzle<D>(crow) : k_chunk_swizzle<D>(crow);
        const unsigned img = half_base + (cv ? 16 * ROWB : 0) + crow * ROWB;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(img + 16 * ((2 * cc) ^ swz)) = u32x4{o[0], o[1], o[2], o[3]};
        *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(img + 16 * ((2 * cc + 1) ^ swz)) = u32x4{o[4], o[5], o[6], o[7]};
      } else {
        uint32_t o[4];
#pragma unroll
        for (int x = 0; x < 4; ++x)
          o[x] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(rw[x] << 16), __uint_as_float(rw[x] & 0xffff0000u)));
        *(__attribute__((address_space(3))) u32x4*)(uintptr_t)a = u32x4{o[0], o[1], o[2], o[3]};
      }
    };
    // pair of halves: waves 0..3 fetch the first, waves 4..7 the second (a key below 0: nothing)
    auto issue_pair = [&](int key_a, unsigned base_a, int key_b, unsigned base_b) {
      int key = hf ? key_b : key_a;
#ifdef TMX_TUNING
      if ((p.exp_mode & 16) && key >= 0) key &= 127;  // every q-block fetches the same keys (cached)
#endif
      if (key >= 0) issue_at(half_offset(key), hf ? base_b : base_a);
    };

    if (!fresh) {
      // every wave is done with the halves of the previous segment before they are overwritten
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    fresh = false;
    int k = k_first;
    int pos0 = ctx + 32 * k;
    int h0 = pos0 / 16 - 2;  // the q-block's halves are h0 .. h0 + 3 (16-key groups of the request)
    // Run-ahead: while q-block k is computed, the keys of k + 1 and k + 2 and the query rows of k + 1 are on
    // their way (one band workgroup beside an arithmetic-bound piece on the same CU has to keep the CU's
    // share of the memory system busy by itself).  qa holds the query rows of k, k + 2, ..., qb_ those of
    // k + 1, k + 3, ...: the rows of the next q-block are asked for at the top of a q-block, into the set
    // that q-block does not use.
    frag qa[KS], qb_[KS];
    issue_pair(0, sink_base, 16 * h0, slot_base(h0));
    issue_pair(16 * (h0 + 1), slot_base(h0 + 1), 16 * (h0 + 2), slot_base(h0 + 2));
    issue_pair(16 * (h0 + 3), slot_base(h0 + 3), -1, 0);
    load_q(q_start + 32 * k, qa);
    if (k + 1 < k_stop) issue_pair(16 * (h0 + 4), slot_base(h0 + 4), 16 * (h0 + 5), slot_base(h0 + 5));
    bool first = true;
    // V16: the halves this wave fetched above (their V piece is converted ahead of the first barrier), and
    // the half whose V piece it converts during the q-block before the one that reads it
    const unsigned cv0 = hf ? slot_base(h0) : sink_base, cv1 = hf ? slot_base(h0 + 2) : slot_base(h0 + 1);
    const unsigned cv2 = slot_base(h0 + 3);
    unsigned cv_pending = slot_base(h0 + 4 + hf);

    // one q-block: its query rows are in `cur`, the next q-block's go to `nxt`
    auto one_qblock = [&](const frag (&cur)[KS], frag (&nxt)[KS]) {
      const bool has1 = k + 1 < k_stop, has2 = k + 2 < k_stop;
      TMX_STAMP(ts0);
      if (has1) load_q(q_start + 32 * (k + 1), nxt);
      // where this wave's half of the q-block after the next one lives (a block-table lookup: asked for
      // here, used after the score products)
#ifdef TMX_TUNING
      const long long off2 = has2 ? half_offset((p.exp_mode & 16) ? (16 * (h0 + 6 + hf)) & 127 : 16 * (h0 + 6 + hf)) : 0;
#else
      const long long off2 = has2 ? half_offset(16 * (h0 + 6 + hf)) : 0;
#endif
      // Wait for this q-block's halves and query rows: what the wave issued later may stay in flight -- the
      // next q-block's pieces, the output rows stored last, the next q-block's query rows.
      {
        const int st = first ? 0 : (p.out_dtype == TMX_F32 ? DB : DB / 2);
        const int n = (has1 ? PWB + KS : 0) + st;
        if (n == PWB + KS + DB / 2) wait_vmcnt<PWB + KS + DB / 2>();
        else if (n == PWB + KS + DB) wait_vmcnt<PWB + KS + DB>();
        else if (n == PWB + KS) wait_vmcnt<PWB + KS>();
        else if (n == DB / 2) wait_vmcnt<DB / 2>();
        else if (n == DB) wait_vmcnt<DB>();
        else wait_vmcnt<0>();
      }
      const bool was_first = first;
      if constexpr (V16 || FP8) {
        if (first && wave_converts) {
          convert_v(cv0);
          convert_v(cv1);
          if (!hf) convert_v(cv2);
        }
      }
      first = false;
      TMX_STAMP(ts1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      TMX_STAMP(ts2);

      // ---- scores of the four groups: sink, A, B, C ----
      const unsigned gb[4] = {sink_base, slot_base(h0 + hf), slot_base(h0 + hf + 1), slot_base(h0 + hf + 2)};
      f32x4 sc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) sc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      // The K fragments of two k-steps are in flight at a time (eight LDS reads, 32 registers), read by
      // instructions the compiler cannot reorder and waited for by count: left to itself the register-minimising
      // scheduler reads one fragment, waits for it, multiplies, sixteen times over.
      auto scores = [&](const frag (&qf)[KS]) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        u32x4 kf[2][4];
        auto fetch = [&](int ks, u32x4 (&dst)[4]) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            asm volatile("ds_read_b128 %0, %1" : "=v"(dst[u]) : "v"((gb[u] + k_base) ^ (64 * ks)) : "memory");
        };
        fetch(0, kf[0]);
        if constexpr (KS > 1) fetch(1, kf[1]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          u32x4(&cur)[4] = kf[ks & 1];
          if (ks + 1 < KS) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])::"memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])::"memory");
#pragma unroll
          for (int u = 0; u < 4; ++u) sc[u] = TR::mfma(__builtin_bit_cast(frag, cur[u]), qf[ks], sc[u]);
          if (ks + 2 < KS) fetch(ks + 2, cur);
        }
      };
      scores(cur);
      // the halves of the q-block before this one are free: fetch the keys of the q-block after the next one
      asm volatile("" : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(sc[3])::"memory");
      TMX_STAMP(ts3);
      if constexpr (V16 || FP8) {
        if (has1) {
          // The V piece this wave fetched during the previous q-block (or in the set-up) belongs to the next
          // q-block: it has had a whole q-block to land.  Issued after it: the previous q-block's output rows
          // and the next q-block's query rows.
          const int st = was_first ? 0 : (p.out_dtype == TMX_F32 ? DB : DB / 2);
          if (st == DB / 2) wait_vmcnt<KS + DB / 2>();
          else if (st == DB) wait_vmcnt<KS + DB>();
          else wait_vmcnt<KS>();
          if (wave_converts) convert_v(cv_pending);
        }
      }
      if (has2) {
        issue_at(off2, slot_base(h0 + 6 + hf));
        cv_pending = slot_base(h0 + 6 + hf);
      }
      TMX_STAMP(ts4);

      // ---- one softmax step over the 16 scores of the lane's row ----
      const float ninf = -__builtin_inff();
#if TMX_BAND_MASKS_IN_LOOP
      // (the thresholds are hidden from the compiler here, so that it cannot move the comparisons out of the loop)
      asm volatile("" : "+v"(c_sink), "+v"(c_win), "+v"(c_diag));
#endif
      const int cs = c_sink, cw = c_win, cd = c_diag;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sc[0][c] = c <= cs ? sc[0][c] : ninf;
        sc[1][c] = c >= cw ? sc[1][c] : ninf;
        sc[2][c] = c >= cw - 16 ? sc[2][c] : ninf;
        sc[3][c] = c <= cd ? sc[3][c] : ninf;
      }
      float tmax = fmaxf(fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3])),
                         fmaxf(fmaxf(sc[1][0], sc[1][1]), fmaxf(sc[1][2], sc[1][3])));
      tmax = fmaxf(tmax, fmaxf(fmaxf(fmaxf(sc[2][0], sc[2][1]), fmaxf(sc[2][2], sc[2][3])),
                               fmaxf(fmaxf(sc[3][0], sc[3][1]), fmaxf(sc[3][2], sc[3][3]))));
      {
        const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
        const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
      }
      const float mneg = -tmax * mult;  // the row's own key is always attended: finite
      uint32_t ph[8], pl[8];
      float psum = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
          const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[u][c], mult, mneg));
          const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[u][c + 1], mult, mneg));
          if constexpr (V16) {
            ph[2 * u + (c >> 1)] = F16Traits::cvt2(p0, p1);
          } else {
            psum += p0 + p1;
            TR::pack2(p0, p1, ph[2 * u + (c >> 1)], pl[2 * u + (c >> 1)]);
          }
        }

      TMX_STAMP(ts5);
      // ---- O^T = V^T P^T over two 32-key steps: (sink, A) and (B, C) ----
      f32x4 acc[DB];
#pragma unroll
      for (int db = 0; db < DB; ++db) acc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
      // V16: the row sums come from the matrix cores too (a ones operand in V's place): they add up the SAME
      // rounded probabilities as the weighted sum -- a constant V comes back unchanged, the rounding of a
      // probability only shifts weight between keys -- and every lane ends with its row's total.
      f32x4 lsum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const frag pf = __builtin_bit_cast(frag, make_uint4(ph[4 * st], ph[4 * st + 1], ph[4 * st + 2], ph[4 * st + 3]));
        const frag pq = __builtin_bit_cast(frag, make_uint4(pl[4 * st], pl[4 * st + 1], pl[4 * st + 2], pl[4 * st + 3]));
        if constexpr (V16) lsum = F16Traits::mfma(F16Traits::ones(), __builtin_bit_cast(f16x8, pf), lsum);
        const unsigned va = gb[2 * st] + v_base, vb = gb[2 * st + 1] + v_base;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const s16x4 lo4 = lds_read_tr16_b64(va ^ (32 * db));
          const s16x4 hi4 = lds_read_tr16_b64(vb ^ (32 * db));
          const frag vf = __builtin_bit_cast(frag, (s16x8)__builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
          if constexpr (V16) acc[db] = F16Traits::mfma(__builtin_bit_cast(f16x8, vf), __builtin_bit_cast(f16x8, pf), acc[db]);
          else acc[db] = TR::mfma(vf, pf, acc[db]);
          if constexpr (PRECISE) acc[db] = TR::mfma(vf, pq, acc[db]);
          if ((db & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most four V fragments fetched ahead
        }
      }

      // ---- epilogue: lane (qi, g) holds columns 16 * db + 4 * g .. + 3 of its row ----
      float l_tot;
      if constexpr (V16) {
        l_tot = lsum[0];
      } else {
        const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(psum), __float_as_uint(psum), false, false);
        l_tot = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
        const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
        l_tot = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
      }
      const float inv = 1.0f / l_tot;
      // (element offsets of the wave's 16 rows stay below 2^31: the strides are checked on the host)
      const long long orows = (long long)(q_start + 32 * k + 16 * hf) * p.out_stride_t + (long long)head * p.out_stride_h;
#ifdef TMX_TUNING
      if ((p.exp_mode & 4) && inv != 12345.f) {
        // no stores: the wait counts below still hold (they only wait longer)
      } else
#endif
      if (p.out_dtype == TMX_F32) {
        // (wave-uniform base + a 32-bit lane offset computed here: nothing of it stays live across the loop)
        const int l = opaque(lane);
        char* const ob = reinterpret_cast<char*>(p.out) + 4 * orows;
        const unsigned lo = 4u * (unsigned)((l & 15) * p.out_stride_t + 4 * (l >> 4));
#pragma unroll
        for (int db = 0; db < DB; ++db)
          store_stream16(ob + lo + 64 * db, make_float4(acc[db][0] * inv, acc[db][1] * inv, acc[db][2] * inv, acc[db][3] * inv));
      } else if (ROWST
#ifdef TMX_TUNING
                 && !(p.exp_mode & (32 | 64 | 128 | 256 | 512))
#endif
      ) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        // two passes, one per half of the row (ROWB / 2 bytes = DB / 4 pieces per lane), each in two rounds of
        // 2 RPP rows: the stage holds 2 RPP half rows; a store instruction covers 2 RPP rows x one half row
#ifndef TMX_ROW_PASSES
#define TMX_ROW_PASSES 2
#endif
        constexpr int NPASS = TMX_ROW_PASSES;     // passes over the row: 2 = half rows, 1 = whole rows
        constexpr int HP = DB / 2 / NPASS;        // pieces per lane and pass
        constexpr int HROWB = ROWB / NPASS;       // bytes of a row part
        constexpr int RR = NPASS * RPP;           // rows per round (the stage holds 1 KiB)
        const int l = opaque(lane);
        const unsigned stg = lds_base + (NSLOT + 1) * HALFB + wave * 1024;  // this wave's stage (nobody else's)
        // (rows of the stage are HROWB bytes apart: rows two apart would meet in the same LDS banks, so the 16-byte
        // chunks of a row are XOR-swizzled by the row, on the way in and on the way out)
        const int wsw = NPASS == 2 ? ((qi % RR) >> 1) & 3 : 0;
        const unsigned waddr = stg + (qi % RR) * HROWB + 2 * ((g & 1) ? 16 + 4 * (g - 1) : 4 * g);
        // wave-uniform base of the wave's 16 output rows + this lane's 32-bit byte offset
        char* const ob = reinterpret_cast<char*>(p.out) + 2 * orows;
        const unsigned lo = 2u * (unsigned)((l / (CPR / NPASS)) * p.out_stride_t + 8 * (l % (CPR / NPASS)));
#pragma unroll
        for (int hh = 0; hh < NPASS; ++hh) {
          u32x4 rc[HP];  // piece c of this lane's row: 8 columns at 32 c + {0, 16, 8, 24}[g]
#pragma unroll
          for (int x = 0; x < HP; ++x) {
            const int c = hh * HP + x;
            const f32x4 oa = acc[2 * c] * inv, ob = acc[2 * c + 1] * inv;
            const uint2 au = make_uint2(TR::cvt2(oa[0], oa[1]), TR::cvt2(oa[2], oa[3]));
            const uint2 bu = make_uint2(TR::cvt2(ob[0], ob[1]), TR::cvt2(ob[2], ob[3]));
            const auto r0 = __builtin_amdgcn_permlane16_swap(au.x, bu.x, false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(au.y, bu.y, false, false);
            rc[x] = u32x4{r0[0], r1[0], r0[1], r1[1]};
          }
          // All rounds of the pass are issued back to back: LDS operations of a wave execute in order, so a
          // round's read sees that round's writes and the next round's writes come after it; only the stores
          // wait, each for its own read.  The reads are pinned by inline asm: lanes exchange data through the
          // stage, which the compiler, reasoning about one thread, does not see -- it moved a plain read into
          // the branch of the writing lanes, "unchanged for everyone else".
          constexpr int NR = 16 / RR;
          u32x4 v[NR];
#pragma unroll
          for (int j = 0; j < NR; ++j) {
            if (qi / RR == j) {
#pragma unroll
              for (int x = 0; x < HP; ++x)
                *(__attribute__((address_space(3))) u32x4*)(uintptr_t)((waddr + 64 * x) ^ (16 * wsw)) = rc[x];
            }
            const int rsw = NPASS == 2 ? ((l / (CPR / NPASS)) >> 1) & 3 : 0;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"((stg + 16 * l) ^ (16 * rsw)) : "memory");
          }
#pragma unroll
          for (int j = 0; j < NR; ++j) {
            // (the count is of this wave's LDS operations after read j: the later rounds' reads; the masked
            // writes between them only make the wait longer for the lanes that issued them)
            if (NR - 1 - j >= 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(v[j])::"memory");
            else if (NR - 1 - j == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(v[j])::"memory");
            else if (NR - 1 - j == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(v[j])::"memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[j])::"memory");
            store_stream16(ob + (2ll * j * RR * p.out_stride_t + hh * HROWB) + lo, __builtin_bit_cast(uint4, v[j]));
          }
        }
      } else {
        uint16_t* po = reinterpret_cast<uint16_t*>(p.out) + orows + (qi * p.out_stride_t + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g));
#pragma unroll
        for (int c = 0; c < DB / 2; ++c) {
          const f32x4 oa = acc[2 * c] * inv, ob = acc[2 * c + 1] * inv;
          const uint2 au = make_uint2(TR::cvt2(oa[0], oa[1]), TR::cvt2(oa[2], oa[3]));
          const uint2 bu = make_uint2(TR::cvt2(ob[0], ob[1]), TR::cvt2(ob[2], ob[3]));
          const auto r0 = __builtin_amdgcn_permlane16_swap(au.x, bu.x, false, false);
          const auto r1 = __builtin_amdgcn_permlane16_swap(au.y, bu.y, false, false);
#ifdef TMX_TUNING
          if (p.exp_mode & 32) {
            *reinterpret_cast<uint4*>(po + 32 * c) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            continue;
          }
          if (p.exp_mode & 64) {
            // timing experiment (wrong placement, same bytes): every store instruction covers 8 rows x 128 bytes
            // (whole cache lines) instead of 16 rows x 64 bytes
            uint16_t* pe = reinterpret_cast<uint16_t*>(p.out) + orows + (((qi & 7) + 8 * (c & 1)) * p.out_stride_t + 64 * (c >> 1) + 32 * (qi >> 3) + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g));
            store_stream16(pe, make_uint4(r0[0], r1[0], r0[1], r1[1]));
            continue;
          }
          if (p.exp_mode & 256) {
            // timing experiment (wrong placement, same bytes): four adjacent lanes store 64 contiguous bytes, a
            // 16-lane group covers 4 rows x 64 bytes, the instruction the same 16 rows x 64 bytes as the shipped one
            const int x = qi & 3;
            uint16_t* pe = reinterpret_cast<uint16_t*>(p.out) + orows + (((qi & 12) | g) * p.out_stride_t + 32 * c + ((x & 1) ? 16 + 4 * (x - 1) : 4 * x));
            store_stream16(pe, make_uint4(r0[0], r1[0], r0[1], r1[1]));
            continue;
          }
          if (p.exp_mode & 128) {
            // timing experiment (wrong placement, same bytes): adjacent lanes store adjacent 16-byte pieces, an
            // instruction covers 4 whole rows of 256 bytes
            const int l = opaque(lane);
            uint16_t* pe = reinterpret_cast<uint16_t*>(p.out) + orows + (((l >> 4) + 4 * c) * p.out_stride_t + 8 * (l & 15));
            store_stream16(pe, make_uint4(r0[0], r1[0], r0[1], r1[1]));
            continue;
          }
#endif
          store_stream16(po + 32 * c, make_uint4(r0[0], r1[0], r0[1], r1[1]));
        }
      }
      TMX_STAMP(ts6);
      TMX_SEG(0, ts0, ts1);
      TMX_SEG(1, ts1, ts2);
      TMX_SEG(2, ts2, ts3);
      TMX_SEG(3, ts3, ts4);
      TMX_SEG(4, ts4, ts5);
      TMX_SEG(5, ts5, ts6);
#ifdef TMX_STAMPS
      seg[7] += 1;
#endif
      ++k;
      pos0 += 32;
      h0 += 2;
    };
    // the two register sets swap roles from one q-block to the next
#pragma unroll 1
    for (;;) {
      one_qblock(qa, qb_);
      if (k >= k_stop) break;
      one_qblock(qb_, qa);
      if (k >= k_stop) break;
    }
    qb = qb_next;
  }
#ifdef TMX_STAMPS
  TMX_STAMP(ts0);
  seg[6] = ts0 - t_begin;
  if (p.stamps != nullptr && lane == 0 && own_stamps) {
    unsigned long long* dst = p.stamps + ((size_t)blockIdx.x * kWaves + wave) * 8;
    for (int x = 0; x < 8; ++x) dst[x] = seg[x];
  }
#endif
}

// Persistent, software-pipelined kernel.  Every workgroup walks work items
// item0, item0 + step, ... for ONE K/V head chunk.  Their key tiles ([K | V],
// 32 keys, one LDS entry each) stream through a ring of NS entries shared by
// the workgroup, fetched by LDS-DMA (every wave a fixed share of each entry) up
// to NS entries ahead of their use and across item boundaries.  The 16 query
// rows of a wave are read by that wave alone, so they go through a private LDS
// region per wave with no workgroup synchronisation: the rows of item k + 1 are
// fetched as soon as the fragments of item k sit in registers.
// FP8: K and V are an fp8 (e4m3fn) paged cache.  The ring then holds the tiles as they are in memory (half the
// bytes, NS = 4 entries), and every tile step opens with a conversion stage: the eight waves turn the entry of
// the step, one 1-KiB piece each, into the 16-bit image the arithmetic reads (value x scale, rounded once to
// the 16-bit type: the descale of the reference's attention kernels, vllm/v1/attention/backends/rocm_attn.py:421-422),
// with a second barrier between the conversion and the arithmetic.
template <typename TR, int D, int NS, bool PAGED, bool HAS_BIAS, bool FP8>
__global__ __launch_bounds__(64 * kWaves, 4) void attn_kernel(const AttnParams p) {
  constexpr int KS = D / 32;            // k-steps of S^T = K Q^T
  constexpr int DB = D / 16;            // 16-column output blocks
  constexpr int ROWB = D * 2;           // bytes per Q row, and per K/V row of the 16-bit image the arithmetic reads
  constexpr int CPR = D / 8;            // 16-byte chunks per such row
  constexpr int KVB = FP8 ? 1 : 2;      // bytes per K/V element in memory and in the ring
  constexpr int KROWB = D * KVB;        // bytes per K/V row in the ring
  constexpr int KCPR = KROWB / 16;
  constexpr int ENTRY = 64 * KROWB;     // ring entry: K tile + V tile
  constexpr int HALF = 32 * ROWB;       // V part of the 16-bit image of a tile
  constexpr int IMG = FP8 ? 64 * ROWB : 0;  // fp8: the image of the entry being computed, behind the ring
  constexpr int QREG = 16 * ROWB;       // a wave's private region: its 16 Q rows
  constexpr int QP = QREG / 1024;       // LDS-DMA pieces of that region
  constexpr int RPP = 1024 / ROWB;      // Q rows per 1-KiB LDS-DMA piece (one wave instruction)
  constexpr int KRPP = 1024 / KROWB;    // K/V rows per piece
  constexpr int PE = ENTRY / 1024;      // pieces per entry
  constexpr int PT = PE / 2;            // ... per tensor of a K/V entry
  static_assert(!FP8 || (D == 128 && PAGED && !HAS_BIAS), "fp8 K/V: paged cache, head size 128");
  // kIssuers waves fetch the K/V entries (and run the scalar walk that finds them); the others only meet them
  // at the barrier.  Waves 0..3 by default (measured 4-5 % faster on prefill than 4..7); waves 4..7 when the
  // caller says that no request has more than 16 query rows (decode): they own rows 16..31 of a q-block and
  // have nothing else to do there, so the walk runs beside the arithmetic of waves 0..3 (decode -11 %).  The walk is the same in every wave, so fewer walkers leave more issue slots to
  // the arithmetic (measured: 4 walkers against 8: dense-causal -10 %, sparse rows -5 %; 2 against 4: no change).
  constexpr int kIssuers = 4;
  constexpr int PW = PE / kIssuers;     // ... fetched by each issuing wave
  constexpr bool V16L = TR::kSplitP && !FP8 && TMX_V16_LIST != 0;  // V tiles turned into fp16 inside the ring, one-term P
  constexpr bool PRECISE = TR::kSplitP && !V16L;               // P as a sum of two 16-bit terms
  constexpr bool SPLIT = V16L;          // an issuing wave fetches K rows and V rows (else: rows of one tensor)
  static_assert(PW >= 1 && NS >= 3 && (!SPLIT || PW % 2 == 0), "ring too small");
  using frag = typename TR::frag;

  __shared__ __attribute__((aligned(1024))) char lds[NS * ENTRY + IMG + kWaves * QREG];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 15, g = lane >> 4;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)lds));
  // Two roles in one launch: workgroups 0 .. n_generic - 1 walk the work list (below), the others take runs
  // of sparse q-blocks in canonical position (band_body).  One launch, so that the arithmetic-bound pieces of
  // the dense rows and the memory-bound sparse q-blocks share the CUs.
  const int bid = (int)blockIdx.x, ngrid = p.n_generic;
  if constexpr (PAGED && !FP8) {
    if (p.cache_slots != nullptr) fused_cache_write<D>(p);
  }
  if constexpr (!HAS_BIAS) {
    if (bid >= ngrid) {
#ifdef TMX_TUNING
      if (p.exp_mode & 2) return;
#endif
      const UnitLane bl = unit_of(bid - ngrid, p.Y, p.Hkv, p.G / p.Gw, p.xcd_band);
      const int qb0 = bl.lane * p.band_rl;
#if TMX_BAND_PRIO
      // the band workgroups are the last to finish: their waves go first where a SIMD has to choose
      __builtin_amdgcn_s_setprio(TMX_BAND_PRIO);
#endif
      band_body<TR, D, PAGED, FP8>(p, lds_base, bl.kvc, qb0, imin(p.band_tail0, qb0 + p.band_rl), true);
      return;
    }
  }
#if TMX_LIST_PRIO
  // a work-list tile step is a chain (scalar walk, fetch, barrier, matrix instructions) on which eight waves wait;
  // the band waves beside it mostly wait for memory
  if (p.band_on) __builtin_amdgcn_s_setprio(TMX_LIST_PRIO);
#endif
#ifdef TMX_TUNING
  if (p.exp_mode & 1) return;
#endif

  const int sub = wave & 3, half = wave >> 2;
  const int hg = sub & (p.Gw - 1);                  // this wave's head inside the workgroup's head group
  const int rb16 = (sub >> p.gw_shift) * 2 + half;   // its 16-row block inside the q-block
  // fixed for the workgroup: K/V head and query heads (the grid is a multiple of Y, or has <= Y workgroups)
  const int chunks = p.G / p.Gw;
  const UnitLane ul = unit_of(bid, p.Y, p.Hkv, chunks, p.xcd_generic);
  const int kvc = ul.kvc;
  const int kvh = kvc / chunks;
  const int head0 = kvh * p.G + (kvc - kvh * chunks) * p.Gw;
  const int head = head0 + hg;
  const int item0 = ul.lane;
  const int item_step = ngrid / p.Y;

  const int q_row_bytes = 2 * p.q_stride_t, kv_row_bytes = KVB * p.kv_stride_t;

  // ---- LDS-DMA issue: wave-uniform bases + per-lane constant byte offsets ----
  // Issuing wave w fetches PW pieces of every K/V entry, all of them rows of ONE 16-key half of the tile (one
  // page of a 16-key-page cache), and all QP pieces of its own Q region: sixteen rows of one tensor, or, with the
  // V conversion on (V16L), eight rows of K and the same eight rows of V, so that every issuing wave has the
  // same share of it.
  const int piece_row = lane / CPR;  // row of this lane inside a 1-KiB piece
  const int piece_cc = lane % CPR;   // its 16-byte chunk
  const bool issuer = p.issue_hi ? (wave >= kWaves - kIssuers) : (wave < kIssuers);
  const int iw = wave % kIssuers;
  // piece x: tensor (false K, true V), first row inside the tile, byte offset inside the ring entry
  bool pc_v[PW];
  int pc_row[PW];
  unsigned pc_dst[PW];
#pragma unroll
  for (int x = 0; x < PW; ++x) {
    if constexpr (SPLIT) {
      pc_v[x] = x >= PW / 2;
      pc_row[x] = 8 * iw + KRPP * (x % (PW / 2));
    } else {
      pc_v[x] = iw * PW + x >= PT;
      pc_row[x] = ((iw * PW + x) % PT) * KRPP;
    }
    pc_dst[x] = (unsigned)(((pc_v[x] ? PT : 0) + pc_row[x] / KRPP) * 1024);
  }
  const bool iss_second = pc_row[0] >= 16;  // keys 16..31 of the tile
  const char* const k_iss = reinterpret_cast<const char*>(p.k) + (long long)KVB * kvh * p.kv_stride_h;
  const char* const v_iss = reinterpret_cast<const char*>(p.v) + (long long)KVB * kvh * p.kv_stride_h;
  // (two sources: the same tensors of the step's own rows)
  const int new_row_bytes = 2 * p.new_stride_t;
  const char* const kn_iss = reinterpret_cast<const char*>(p.k_new) + 2ll * kvh * p.new_stride_h;
  const char* const vn_iss = reinterpret_cast<const char*>(p.v_new) + 2ll * kvh * p.new_stride_h;
  const char* const q_own = reinterpret_cast<const char*>(p.q) + 2ll * head * p.q_stride_h;
  const unsigned q_lds = lds_base + NS * ENTRY + IMG + wave * QREG;
  int kv_voff[PW], q_voff[QP];
#pragma unroll
  for (int x = 0; x < PW; ++x) {
    const int row = pc_row[x] + lane / KCPR;
    // (the 16-bit images are chunk-swizzled on their way in; fp8 tiles land as they are)
    const int swz = FP8 ? 0 : (pc_v[x] ? v_chunk_swizzle<D>(row) : k_chunk_swizzle<D>(row));
    kv_voff[x] = row * kv_row_bytes + 16 * ((lane % KCPR) ^ swz);
  }
#pragma unroll
  for (int x = 0; x < QP; ++x) {
    const int r16 = x * RPP + piece_row;
    q_voff[x] = r16 * q_row_bytes + 16 * (piece_cc ^ k_chunk_swizzle<D>(r16));
  }

  // ------------------------------------------------------------------ the workgroup's items
  // item fields: a0 = (qb, col_start, ntiles, slot), a1 = (b, q_tok0, pos0, n)
  // ---- static schedule of this workgroup ("lane" item0 of item_step lanes per K/V head chunk) ----
  // The list starts with num_heavy pieces of split q-blocks (chunk_tiles tiles each), the light unsplit
  // q-blocks follow.  Pieces go round the lanes; when they do not divide evenly the lanes that got one piece
  // fewer take the first sched_e_total light items among themselves (sched_w per lane), and only what is
  // left goes round all lanes: every lane ends up with about the same amount of work.
  int sched_h = 0, sched_e = 0, sched_short = 1, sched_p1_base = 0, sched_p2_base = 0;
  if (item_step > 0) {
    const int P = imin(p.num_heavy, p.num_items);
    const int r = P % item_step;
    int e_total = 0;
    sched_h = P / item_step + (item0 < r ? 1 : 0);
    if (r > 0 && p.sched_w > 0) {
      sched_short = item_step - r;
      const long long want = (long long)sched_short * p.sched_w;
      e_total = (int)(want < (long long)(p.num_items - P) ? want : (long long)(p.num_items - P));
      if (item0 >= r) sched_e = imax(0, (e_total - (item0 - r) + sched_short - 1) / sched_short);
    }
    sched_p1_base = P + item0 - r;
    sched_p2_base = P + e_total + item0 - sched_h * item_step;  // + k * item_step for the k-th item overall
  }
  // The schedule's next item that is this workgroup's to compute: with the band path on, items of
  // q-blocks in canonical position belong to the band workgroups and are passed over here.
  const BandRule band_rule{p.sl, p.last, p.mode, p.tri_min_len};
  int sched_k = 0;
  auto fetch_item = [&](i32x4& a0, i32x4& a1) -> int {
#pragma unroll 1
    for (;;) {
      // 32-bit on purpose: compared and indexed on the scalar unit
      const int k = sched_k++;
      int item;
      if (k < sched_h) item = item0 + k * item_step;
      else if (k - sched_h < sched_e) item = sched_p1_base + (k - sched_h) * sched_short;
      else item = sched_p2_base + (k - sched_e) * item_step;
      const int ok = (item < p.num_items) && (k == 0 || item_step > 0);
      a0 = a1 = i32x4{0, 0, 0, 1};
      if (!ok) return 0;
      a0 = sload(reinterpret_cast<const i32x4*>(p.items) + 2ll * item);
      a1 = sload(reinterpret_cast<const i32x4*>(p.items) + 2ll * item + 1);
      if (p.dense_on | p.decode_on) {
        const int q_len = sload(p.cu_q + a1.x + 1) - sload(p.cu_q + a1.x);
        // causal launch with the dense kernel beside it: only the requests with few query rows are computed here
        if (p.dense_on && q_len >= kDenseMinRows) continue;
        if (p.decode_on && q_len == 1) {
          const int nf1 = p.full_lens != nullptr ? imax(a1.w, sload(p.full_lens + a1.x)) : a1.w;
          if (decode_row(p.mode, p.last, p.tri_min_len, q_len, a1.z, nf1)) continue;  // the decode kernel's
        }
        if (p.dense_on) return 1;
      }
      if (!p.band_on || a0.w >= 0) return 1;  // (pieces of split q-blocks never take the band path)
      const int nf = p.full_lens != nullptr ? imax(a1.w, sload(p.full_lens + a1.x)) : a1.w;
      if (!band_qblock(band_rule, a1.z, a1.w, nf)) return 1;
    }
  };
  // Walk order of a piece of a split q-block (the trailing dense rows): front to back (s = 0), or in 1 << s
  // interleaved parts -- entry e is tile (e >> s) of part (e & mask).  A piece covers the keys of consecutive band
  // runs, which the band workgroups stream at the same time, each from its own start; walked in as many parts the
  // piece stays within a fraction of the launch of every band run it shares keys with (an attempt to have
  // whoever comes second find the keys in the Infinity Cache: no measurable effect, see launch_attn; s = 0 is
  // shipped).  Online softmax does not care about the order; both sides of the kernel (fetching, arithmetic) use
  // the same function.
  auto walk_il = [&](const i32x4& a0) -> int {
    const int s = p.walk_shift;
    return (s > 0 && a0.w >= 0 && (a0.z & ((1 << s) - 1)) == 0) ? s : 0;
  };
  auto walk_entry = [&](int e, int cnt, int s) -> int { return (e >> s) + (cnt >> s) * (e & ((1 << s) - 1)); };
  // the item being computed and the one after it
  i32x4 ca0, ca1, cb0, cb1;
  int c_valid = fetch_item(ca0, ca1);
  int cb_valid = fetch_item(cb0, cb1);

  // ------------------------------------------------------------------ issue side (all scalar)
  // The issue cursor runs ahead of the compute side through the tile lists of the current item and of
  // the next one, never further (an item shorter than the ring is deep leaves the ring less full):
  // i_src = 0 / 1 names the item the cursor is in, 2 = nothing to issue until the compute side moves on.
  int i_src = 2;
  int i_pos = 0;    // first entry of the cursor's item in col_idx
  int i_e = 0;      // entries of the item issued so far
  int i_cnt = 0;    // its tile count
  int i_il = 0;     // its walk order: log2 of the interleaved parts (walk_entry)
  int i_left = 0;   // tiles of the cursor's item not yet issued
  int i_n = 1;      // its request: sequence length, index (block table)
  int i_b = 0;
  int gi = 0;       // entries issued so far
  int gi_slot = 0;  // gi mod NS
  // descriptor of the next K/V entry to issue, fetched one step ahead
  int d_tile = 0;
  long long d_off0 = 0, d_off1 = 0;  // byte offsets of the pages holding the tile's keys 0..15 / 16..31
  long long i_krow0 = 0;             // contiguous K/V: first row of the cursor's request
  int i_ctx = 0x7fffffff;            // two sources: keys from this position on come from the step's own rows ...
  int i_newrow0 = 0;                 // ... key j is row i_newrow0 + j of them

  auto cursor_to = [&](const i32x4& a0, const i32x4& a1, int src) {
    i_src = src;
    i_pos = a0.y;
    i_e = 0;
    i_cnt = a0.z;
    i_il = walk_il(a0);
    i_left = a0.z;
    i_n = a1.w;
    i_b = a1.x;
    if constexpr (!PAGED) i_krow0 = sload(p.cu_k + a1.x);
    if constexpr (PAGED && !FP8) {
      if (p.k_new != nullptr) {
        i_ctx = a1.w - (sload(p.cu_q + a1.x + 1) - sload(p.cu_q + a1.x));  // seq_len - q_len
        i_newrow0 = a1.y - a1.z;                                          // first query row - its position
      }
    }
  };
  // leave exhausted lists behind
  auto cursor_settle = [&]() {
    if (i_src == 0 && i_left <= 0) {
      if (cb_valid) cursor_to(cb0, cb1, 1);
      else i_src = 2;
    }
    if (i_src == 1 && i_left <= 0) i_src = 2;
  };

  auto prefetch_desc = [&]() {
    if (i_src == 2) return;
    const int pos = i_pos + walk_entry(i_e, i_cnt, i_il);
    d_tile = sload(p.col_idx + pos);
    if constexpr (PAGED) {
      if (p.tile_pages != nullptr) {
        d_off0 = (long long)KVB * sload(p.tile_pages + 2 * pos) * p.kv_stride_page;
        d_off1 = (long long)KVB * sload(p.tile_pages + 2 * pos + 1) * p.kv_stride_page;
      } else {
        const int32_t* bt = p.block_table + (long long)i_b * p.block_table_stride;
        const int pg_first = (d_tile * kBlockN) >> p.page_shift;
        d_off0 = (long long)KVB * sload(bt + pg_first) * p.kv_stride_page;
        d_off1 = (long long)KVB * sload(bt + (imin(d_tile * kBlockN + 16, i_n - 1) >> p.page_shift)) * p.kv_stride_page;
      }
    }
  };

  auto issue_one = [&]() {
    const unsigned dst = lds_base + (unsigned)gi_slot * ENTRY;
    const int n = i_n;
    {
      const int j0 = d_tile * kBlockN;
      // two sources: a tile from the step's own rows, or one that holds the last cached key AND the first new one
      bool from_new = false, straddle = false;
      if constexpr (PAGED && !FP8) {
        from_new = j0 >= i_ctx;
        straddle = j0 < i_ctx && j0 + kBlockN > i_ctx;
      }
      if (j0 + kBlockN <= n && !straddle) {
        // whole tile inside the sequence: one wave-uniform base per 16-key half
        if (from_new) {
          // rows of the step's key / value tensors: same chunk swizzle, their own row stride
          const long long bb = (long long)(i_newrow0 + j0) * new_row_bytes;
          const int lrow = opaque(lane) / KCPR;
#pragma unroll
          for (int x = 0; x < PW; ++x)
            glds16_s((pc_v[x] ? vn_iss : kn_iss) + bb, kv_voff[x] + (pc_row[x] + lrow) * (new_row_bytes - kv_row_bytes),
                     dst + pc_dst[x]);
        } else {
          long long bb;
          if constexpr (PAGED) {
            const long long b0 = d_off0 + (long long)(j0 & (p.page_size - 1)) * kv_row_bytes;
            bb = (iss_second && p.page_size == 16) ? d_off1 - 16ll * kv_row_bytes : b0;
          } else {
            bb = (i_krow0 + j0) * kv_row_bytes;
          }
#pragma unroll
          for (int x = 0; x < PW; ++x) glds16_s((pc_v[x] ? v_iss : k_iss) + bb, kv_voff[x], dst + pc_dst[x]);
        }
      } else {
        // last, partial tile: keys past the end are masked later; clamp their source to the last key
        const int l2 = opaque(lane);
#pragma unroll
        for (int x = 0; x < PW; ++x) {
          const int row = pc_row[x] + l2 / KCPR;
          const int j = imin(j0 + row, n - 1);
          long long off;
          if constexpr (PAGED) {
            off = (((j - j0) >= 16 && p.page_size == 16) ? d_off1 : d_off0) +
                  (long long)(j & (p.page_size - 1)) * kv_row_bytes;
          } else {
            off = (i_krow0 + j) * kv_row_bytes;
          }
          const int swz = FP8 ? 0 : (pc_v[x] ? v_chunk_swizzle<D>(row) : k_chunk_swizzle<D>(row));
          const char* src = (pc_v[x] ? v_iss : k_iss) + off;
          if constexpr (PAGED && !FP8) {
            if (j >= i_ctx) src = (pc_v[x] ? vn_iss : kn_iss) + (long long)(i_newrow0 + j) * new_row_bytes;
          }
          glds16(src + 16 * ((l2 % KCPR) ^ swz), dst + pc_dst[x]);
        }
      }
    }
    ++i_e;
    --i_left;
    ++gi;
    gi_slot = (gi_slot + 1 == NS) ? 0 : gi_slot + 1;
    cursor_settle();
    prefetch_desc();
  };

  if (c_valid) cursor_to(ca0, ca1, 0);
  cursor_settle();
  prefetch_desc();

  // ------------------------------------------------------------------ compute side
  // per-lane LDS read offsets inside an entry: K / Q row qi, chunk 4 * ks + g is at k_base ^ (64 * ks)
  const int k_base = qi * ROWB + 16 * (g ^ k_chunk_swizzle<D>(qi));
  // V transposed read: lane 4 * q' + pp of each 16-lane group addresses key row 4 * g + q' (+16 for the
  // second half of the tile), columns 16 * db + 4 * pp .. +3, at v_base ^ (32 * db)
  const int v_row = 4 * g + ((lane & 15) >> 2);
  const int v_base = HALF + v_row * ROWB + 8 * (lane & 3) + 32 * (v_chunk_swizzle<D>(v_row) >> 1);
  const int koff_lane = 4 * g;  // this lane's first key inside a 16-key half tile

  int gs = 0;          // entries consumed so far
  int gs_slot = 0;     // gs mod NS
  int store_mark = 0;  // entries below store_mark were issued before the last stores
  int store_ops = 0;   // vector-memory instructions of those stores (this wave)
  int q_mark = 0;      // entries below q_mark were issued before the Q transfers in flight
  int q_ops = 0;       // LDS-DMA instructions of those Q transfers (this wave)
  int q_stores = 0;    // stores this wave issued after them

#ifdef TMX_STAMPS
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long ts0, ts1, ts2, ts3, ts4;
  TMX_STAMP(ts4);
  const unsigned long long t_begin = ts4;
#endif

  // V16L: the V pieces an issuing wave fetched are turned into fp16 in place by that wave (a lane converts the 16
  // bytes it fetched; round toward zero: in-range values are exact either way, larger ones saturate), one step
  // ahead of the tile's use -- the piece has had two steps to land, and the barrier that opens the next step
  // shows the result to the other waves.  conv_next: the next ring entry whose V this wave has not converted.
  int conv_next = 0;
  // this wave's pieces of ring entry e have landed when at most this many of its later operations are in flight
  auto wait_entry = [&](int e) {
    wait_vmcnt_levels<PW>(PW * (gi - e - 1) + (e < store_mark ? store_ops : 0) + (e < q_mark ? q_ops : 0));
  };
  auto convert_v_pieces = [&](int slot) {
    const unsigned eb = lds_base + (unsigned)slot * ENTRY + 16u * (unsigned)lane_now();
#pragma unroll
    for (int x = PW / 2; x < PW; ++x) {
      const unsigned a = eb + pc_dst[x];
      const uint4 raw = lds_read_b128<uint4>(a);
      const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
      uint32_t o[4];
#pragma unroll
      for (int y = 0; y < 4; ++y)
        o[y] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(rw[y] << 16), __uint_as_float(rw[y] & 0xffff0000u)));
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      *(__attribute__((address_space(3))) u32x4*)(uintptr_t)a = u32x4{o[0], o[1], o[2], o[3]};
    }
  };

  // Wait for ring entries gs .. gs + cnt - 1, make them visible to the workgroup, then refill the
  // slots the previous step freed (entries below gs are consumed once every wave is past the barrier).
  auto step_sync = [&](int cnt) {
    TMX_STAMP(ts0);
    if (issuer) {
      wait_entry(gs + cnt - 1);
      if constexpr (V16L) {
        // (only the first entry of a run of items, or an entry issued too late to be converted a step ahead)
        if (conv_next == gs) {
          convert_v_pieces(gs_slot);
          ++conv_next;
        }
      }
    }
    TMX_STAMP(ts1);
#ifndef TMX_ABLATE_BARRIER
    // every LDS read of the entries about to be overwritten has returned before the barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
    TMX_STAMP(ts2);
    if (issuer) {
#pragma unroll 1
      while (i_src != 2 && gi - gs < NS) issue_one();
    }
    if constexpr (FP8) {
      // Conversion stage: wave w turns piece w of the entry (8 rows of K for w < 4, of V for w >= 4; a lane
      // 16 values of one row) into the 16-bit image, chunk-swizzled as the arithmetic reads it.  Every wave
      // is past the barrier above, so nobody reads the previous tile's image any more.
      const uint4 raw = lds_read_b128<uint4>(lds_base + (unsigned)gs_slot * ENTRY + wave * 1024 + lane * 16);
      const bool cv = wave >= 4;
      const int crow = 8 * (wave & 3) + (lane >> 3), cc = lane & 7;
      const float sc = cv ? p.v_scale : p.k_scale;
      const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
      uint32_t o[8];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)rw[x], false);
        const auto hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)rw[x], true);
        o[2 * x] = TR::cvt2(lo[0] * sc, lo[1] * sc);
        o[2 * x + 1] = TR::cvt2(hi[0] * sc, hi[1] * sc);
      }
      const int swz = cv ? v_chunk_swizzle<D>(crow) : k_chunk_swizzle<D>(crow);
      const unsigned img = lds_base + NS * ENTRY + (cv ? HALF : 0) + crow * ROWB;
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(img + 16 * ((2 * cc) ^ swz)) = u32x4{o[0], o[1], o[2], o[3]};
      *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(img + 16 * ((2 * cc + 1) ^ swz)) = u32x4{o[4], o[5], o[6], o[7]};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    TMX_STAMP(ts3);
    TMX_SEG(0, ts0, ts1);
    TMX_SEG(1, ts1, ts2);
    TMX_SEG(2, ts2, ts3);
  };
  auto slot_after = [&](int s, int k) { return s + k >= NS ? s + k - NS : s + k; };

  // Fetch this wave's 16 query rows of the item (a0, a1) into its private region.
  auto issue_q = [&](const i32x4& a1) {
    const int rows_valid = imin(p.block_m, a1.w - a1.z);
    const int r0 = 16 * rb16;
    q_mark = gi;
    q_stores = 0;
    q_ops = 0;
    if (r0 >= rows_valid) return;  // no rows of this item for the wave
    q_ops = QP;
    if (r0 + 16 <= rows_valid) {
      const char* sb = q_own + (long long)(a1.y + r0) * q_row_bytes;
#pragma unroll
      for (int x = 0; x < QP; ++x) glds16_s(sb, q_voff[x], q_lds + x * 1024);
    } else {
      // ragged last block: rows past the end of the request are never stored; read a valid row instead
      const int l2 = opaque(lane);  // recomputed here: nothing of the rare path stays live in the main loop
#pragma unroll
      for (int x = 0; x < QP; ++x) {
        const int r16 = x * RPP + l2 / CPR;
        const int r = imin(r0 + r16, rows_valid - 1);
        glds16(q_own + (long long)(a1.y + r) * q_row_bytes + 16 * ((l2 % CPR) ^ k_chunk_swizzle<D>(r16)),
               q_lds + x * 1024);
      }
    }
  };
  // Wait for them: everything this wave issued later (tile pieces, stores) may stay in flight.
  auto wait_q = [&]() {
    const int n = (issuer ? PW * (gi - q_mark) : 0) + q_stores;
    if (n >= 12) wait_vmcnt<12>();
    else if (n >= 8) wait_vmcnt<8>();
    else if (n >= 6) wait_vmcnt<6>();
    else if (n >= 4) wait_vmcnt<4>();
    else if (n >= 2) wait_vmcnt<2>();
    else wait_vmcnt<0>();
    q_ops = 0;
  };

  // prologue: the first item's query rows, then the ring
  if (c_valid) issue_q(ca1);
#pragma unroll 1
  for (int x = 0; x < NS; ++x)
    if (issuer && i_src != 2) issue_one();

  const float mult = HAS_BIAS ? 1.0f : p.scale_log2;  // factor between the accumulated scores and the exponent

  while (c_valid) {
    const int col_start = ca0.y, ntiles = ca0.z, slot = ca0.w;
    const int q_tok0 = ca1.y, pos0 = ca1.z, n = ca1.w;
    const int rows_valid = imin(p.block_m, n - pos0);
    // length of the whole sequence the pattern is meant for (chunked prefill: the complete prompt)
    const int nf = p.full_lens != nullptr ? imax(n, sload(p.full_lens + ca1.x)) : n;
    const int mode = effective_mode(p.mode, nf, p.tri_min_len);
    // the bias of this request and head (a mask broadcast over requests and heads has zero strides there)
    const long long bias_base = HAS_BIAS ? (long long)ca1.x * p.bias_stride_b + (long long)head * p.bias_stride_h : 0;

    // tile numbers of the first group (clamped reads stay inside the item's list)
    const int c_il = walk_il(ca0);
    int nt[kGroupMax];
#pragma unroll
    for (int u = 0; u < kGroupMax; ++u)
      nt[u] = ntiles > 0 ? sload(p.col_idx + col_start + walk_entry(imin(u, ntiles - 1), ntiles, c_il)) : 0;

    // ---- Q fragments of this wave's 16 rows (B operand of S^T), from the wave's private region ----
    frag qf[KS];
    wait_q();
    {
      const char* qs = lds + NS * ENTRY + IMG + wave * QREG;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        qf[ks] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(qs + (k_base ^ (64 * ks))));
    }
    // the region is free again once the fragments are in registers: fetch the next item's rows
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qf[0]), "+v"(qf[KS - 1])::"memory");
    if (cb_valid) issue_q(cb1);

    // ---- wave-uniform row range of the wave's 16 rows, per-lane form of the predicate ----
    // key j is attended by this lane's row iff j <= key_hi and not (hole_lo < j < hole_lo + 1 + hole_width)
    const int w_row0 = 16 * rb16;
    const int w_row1 = imin(w_row0 + 16, rows_valid);
    const bool w_has_rows = w_row0 < rows_valid;
    const int w_imin = pos0 + w_row0, w_imax = pos0 + w_row1 - 1;
    const bool w_all_last = (nf - w_imin) < p.last;
    const int hole_lo = p.si < -1 ? -1 : p.si;
    const int my_row = w_row0 + qi;
    const bool row_valid = my_row < rows_valid;
    const int pos_i = pos0 + my_row;
    int key_hi = -1;
    unsigned hole_width = 0;
    if (row_valid) {
      key_hi = (mode == TMX_MODE_NONE) ? n - 1 : imin(pos_i, n - 1);
      if (mode == TMX_MODE_TRIANGLE && !((nf - pos_i) < p.last)) {
        // keys si < j < pos_i - sl are the excluded middle section
        const long long w = (long long)pos_i - p.sl - (long long)hole_lo - 1;
        hole_width = w > 0 ? (unsigned)w : 0u;
      }
    }
    // wave-uniform bounds: keys up to w_key_hi_min are below every row's key_hi; keys from
    // w_hole_end_max on are past every row's hole
    const int w_key_hi_min = (mode == TMX_MODE_NONE) ? n - 1 : imin(w_imin, n - 1);
    const bool w_any_hole = mode == TMX_MODE_TRIANGLE && !w_all_last && (w_imax - p.sl > hole_lo + 1);
    const int w_hole_end_max = w_imax - p.sl;  // exclusive end of the widest hole among the wave's rows
    // ... and for skipping 16-key half tiles no row of this wave attends to: a half [a, a + 15] is
    // needed iff a <= w_key_max and, when every row of the wave has an excluded middle section,
    // (a <= hole_lo  or  a + 15 >= w_window_min)
    const int w_key_max = (mode == TMX_MODE_NONE) ? n - 1 : imin(w_imax, n - 1);
    const bool w_all_holes = mode == TMX_MODE_TRIANGLE && !((nf - w_imax) < p.last);
    const int w_window_min = w_imin - p.sl;

    f32x4 acc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) acc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = kMinusBig, l_run = 0.f;
    f32x4 lacc = f32x4{0.f, 0.f, 0.f, 0.f};  // V16L: the row sums, accumulated by the matrix cores (every lane: its row's total)

    // One softmax step over CNT key tiles (ring entries gs .. gs + CNT - 1).
    // One softmax step over ONE 16-key half (kb) of the tile: the sink tile and the window's edge
    // tiles of the sparse rows have a half that no row of the wave attends to.  Same arithmetic as the
    // full step on half the keys; the second product uses the 16-key matrix instruction.
    auto process_half = [&](int tile_no, int kb, bool first) {
      const unsigned eb = (FP8 ? lds_base + NS * ENTRY : lds_base + (unsigned)gs_slot * ENTRY) + (unsigned)kb * 16 * ROWB;
      const unsigned kl = eb + k_base, vl = eb + v_base;
      f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const frag kf = lds_read_b128<frag>(kl ^ (64 * ks));
        sc = TR::mfma(kf, qf[ks], sc);
      }
      const int ja = tile_no * kBlockN + 16 * kb, jb = ja + 15;
      const int jl = ja + koff_lane;
      if constexpr (HAS_BIAS) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int j = jl + c;
          float b = 0.f;
          if (row_valid && j < p.bias_cols && pos_i < p.bias_rows && j <= key_hi)
            b = kLog2e * load_bias(p.bias, p.bias_dtype, bias_base + (long long)pos_i * p.bias_stride + j);
          sc[c] = sc[c] * p.scale_log2 + b;
        }
      }
      if (jb > w_key_hi_min) {
        const int d_hi = key_hi - jl;
#pragma unroll
        for (int c = 0; c < 4; ++c) sc[c] = (c <= d_hi) ? sc[c] : -__builtin_inff();
      }
      if (w_any_hole && jb > hole_lo && ja < w_hole_end_max) {
        const unsigned d_ho = (unsigned)(jl - hole_lo - 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) sc[c] = (d_ho + (unsigned)c < hole_width) ? -__builtin_inff() : sc[c];
      }
      float tmax = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
      // The running maximum only moves when some score exceeds it by more than kMaxSlack (log2 units): the
      // test is per lane and wave-uniform, so the usual step has no cross-lane exchange and no rescaling.
      float alpha = 1.f;
      const bool moved = __builtin_amdgcn_ballot_w64(tmax * mult > m_run + kMaxSlack) != 0;
      if (moved) {
        const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
        const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
        const float m_tile = tmax * mult;
        const float m_new = (m_tile > m_run + kMaxSlack) ? m_tile : m_run;
        alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
        if constexpr (V16L) lacc *= alpha;
        if (!first) {
#pragma unroll
          for (int db = 0; db < DB; ++db) acc[db] *= alpha;
        }
      }
      const float m_new = m_run;
      uint2 ph, pl2;
      float psum = 0.f;
      {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[0], mult, -m_new));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[1], mult, -m_new));
        const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[2], mult, -m_new));
        const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[3], mult, -m_new));
        if constexpr (V16L) {
          ph.x = F16Traits::cvt2(p0, p1);
          ph.y = F16Traits::cvt2(p2, p3);
        } else {
          psum = (p0 + p1) + (p2 + p3);
          TR::pack2(p0, p1, ph.x, pl2.x);
          TR::pack2(p2, p3, ph.y, pl2.y);
        }
      }
      l_run += psum;
      if constexpr (V16L) {
        const s16x4 ones4 = __builtin_bit_cast(s16x4, make_uint2(0x3C003C00u, 0x3C003C00u));
        lacc = F16Traits::mfma16(ones4, ph, lacc);
      }
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const s16x4 v4 = lds_read_tr16_b64(vl ^ (32 * db));
        if constexpr (V16L) acc[db] = F16Traits::mfma16(v4, ph, acc[db]);
        else acc[db] = TR::mfma16(v4, ph, acc[db]);
        if constexpr (PRECISE) acc[db] = TR::mfma16(v4, pl2, acc[db]);
      }
    };

    auto process = [&](auto cnt_c, const int (&tl)[kGroupMax], bool first) {
      constexpr int CNT = decltype(cnt_c)::value;
      TMX_STAMP(ts0);
      if constexpr (CNT == 1) {
        const int a0 = tl[0] * kBlockN, a1 = a0 + 16;
        const bool need0 = a0 <= w_key_max && (!w_all_holes || a0 <= hole_lo || a0 + 15 >= w_window_min);
        const bool need1 = a1 <= w_key_max && (!w_all_holes || a1 <= hole_lo || a1 + 15 >= w_window_min);
        if (!need0 && !need1) return;
        if (need0 != need1) {
          process_half(tl[0], need1 ? 1 : 0, first);
          return;
        }
      }
      unsigned kl[CNT], vl[CNT];
#pragma unroll
      for (int u = 0; u < CNT; ++u) {
        const unsigned eb = FP8 ? lds_base + NS * ENTRY : lds_base + (unsigned)slot_after(gs_slot, u) * ENTRY;
        kl[u] = eb + k_base;
        vl[u] = eb + v_base;
      }

      // S^T = K Q^T for every tile of the group: 2 * CNT independent accumulator chains
      f32x4 s[CNT][2];
#pragma unroll
      for (int u = 0; u < CNT; ++u) s[u][0] = s[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int u = 0; u < CNT; ++u)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            const frag kf = lds_read_b128<frag>((kl[u] ^ (64 * ks)) + kb * 16 * ROWB);
            s[u][kb] = TR::mfma(kf, qf[ks], s[u][kb]);
          }

      float tmax = -__builtin_inff();
#pragma unroll
      for (int u = 0; u < CNT; ++u) {
        const int j0 = tl[u] * kBlockN, j1 = j0 + kBlockN - 1;
        const int jl = j0 + koff_lane;
        if constexpr (HAS_BIAS) {
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const int j = jl + 16 * (c >> 2) + (c & 3);
            float b = 0.f;
            if (row_valid && j < p.bias_cols && pos_i < p.bias_rows && j <= key_hi)
              b = kLog2e * load_bias(p.bias, p.bias_dtype, bias_base + (long long)pos_i * p.bias_stride + j);
            s[u][c >> 2][c & 3] = s[u][c >> 2][c & 3] * p.scale_log2 + b;
          }
        }
        // Two wave-uniform tests decide which parts of the predicate this tile needs at all.
        if (j1 > w_key_hi_min) {  // some key may lie above some row's last key
          const int d_hi = key_hi - jl;
#pragma unroll
          for (int c = 0; c < 8; ++c)
            s[u][c >> 2][c & 3] = (16 * (c >> 2) + (c & 3) <= d_hi) ? s[u][c >> 2][c & 3] : -__builtin_inff();
        }
        if (w_any_hole && j1 > hole_lo && j0 < w_hole_end_max) {  // the tile meets some row's excluded section
          const unsigned d_ho = (unsigned)(jl - hole_lo - 1);
#pragma unroll
          for (int c = 0; c < 8; ++c)
            s[u][c >> 2][c & 3] =
                (d_ho + (unsigned)(16 * (c >> 2) + (c & 3)) < hole_width) ? -__builtin_inff() : s[u][c >> 2][c & 3];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) tmax = fmaxf(tmax, s[u][c >> 2][c & 3]);
      }
      // The running maximum only moves when some score exceeds it by more than kMaxSlack (in log2
      // units): probabilities may then reach 2^kMaxSlack instead of 1, which costs nothing in range
      // or precision.  The test is per lane and wave-uniform, so the usual step has no cross-lane
      // exchange and no rescaling pass over the accumulator.
      const bool moved = __builtin_amdgcn_ballot_w64(tmax * mult > m_run + kMaxSlack) != 0;  // mult > 0
      if (moved) {
        // maximum over the 4 lanes that share a row (lanes l, l ^ 16, l ^ 32, l ^ 48)
        const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
        const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
        const float m_tile = tmax * mult;
        const float m_next = (m_tile > m_run + kMaxSlack) ? m_tile : m_run;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_next);
        m_run = m_next;
        l_run *= alpha;
        if constexpr (V16L) lacc *= alpha;
        if (!first) {
#pragma unroll
          for (int db = 0; db < DB; ++db) acc[db] *= alpha;
        }
      }
      const float m_new = m_run;
      frag pf[CNT], pl[CNT];
      f32x2 psum2 = f32x2{0.f, 0.f};  // packed arithmetic: two scores per instruction
      const f32x2 mult2 = f32x2{mult, mult}, mneg2 = f32x2{-m_new, -m_new};
#pragma unroll
      for (int u = 0; u < CNT; ++u) {
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
          const f32x2 e = f32x2{s[u][c >> 2][c & 3], s[u][c >> 2][(c & 3) + 1]} * mult2 + mneg2;
          const float p0 = __builtin_amdgcn_exp2f(e[0]);
          const float p1 = __builtin_amdgcn_exp2f(e[1]);
          if constexpr (V16L) {
            hi[c >> 1] = F16Traits::cvt2(p0, p1);
          } else {
            psum2 += f32x2{p0, p1};
            TR::pack2(p0, p1, hi[c >> 1], lo[c >> 1]);
          }
        }
        pf[u] = __builtin_bit_cast(frag, make_uint4(hi[0], hi[1], hi[2], hi[3]));
        if constexpr (PRECISE) pl[u] = __builtin_bit_cast(frag, make_uint4(lo[0], lo[1], lo[2], lo[3]));
        if constexpr (V16L) lacc = F16Traits::mfma(F16Traits::ones(), __builtin_bit_cast(f16x8, pf[u]), lacc);
      }
      l_run += psum2[0] + psum2[1];

      // O^T += V^T P^T (V fragments transposed by the LDS read)
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
          const s16x4 lo4 = lds_read_tr16_b64(vl[u] ^ (32 * db));
          const s16x4 hi4 = lds_read_tr16_b64((vl[u] ^ (32 * db)) + 16 * ROWB);
          const frag vf = __builtin_bit_cast(frag, (s16x8)__builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
          if constexpr (V16L) acc[db] = F16Traits::mfma(__builtin_bit_cast(f16x8, vf), __builtin_bit_cast(f16x8, pf[u]), acc[db]);
          else acc[db] = TR::mfma(vf, pf[u], acc[db]);
          if constexpr (PRECISE) acc[db] = TR::mfma(vf, pl[u], acc[db]);
        }
      TMX_STAMP(ts1);
      TMX_SEG(3, ts0, ts1);
    };

#pragma unroll 1
    for (int t = 0; t < ntiles;) {
      const int cnt = imin(kGroupMax, ntiles - t);
      step_sync(cnt);
      int tl[kGroupMax];
#pragma unroll
      for (int u = 0; u < kGroupMax; ++u) tl[u] = nt[u];
      const int t_next = t + cnt;
      if (t_next < ntiles) {
#pragma unroll
        for (int u = 0; u < kGroupMax; ++u)
          nt[u] = sload(p.col_idx + col_start + walk_entry(imin(t_next + u, ntiles - 1), ntiles, c_il));
      }
#ifdef TMX_ABLATE_COMPUTE
      const bool compute = w_has_rows && p.scale_log2 == 12345.f;
#else
      const bool compute = w_has_rows;
#endif
      if (compute) {
        if (cnt == 2) process(std::integral_constant<int, 2>{}, tl, t == 0);
        else process(std::integral_constant<int, 1>{}, tl, t == 0);
      }
      gs += cnt;
      gs_slot = slot_after(gs_slot, cnt);
      t = t_next;
      if constexpr (V16L) {
        // the next entry (of this item or of the one after it), issued two steps ago
        if (issuer && conv_next == gs && conv_next < gi) {
          wait_entry(conv_next);
          convert_v_pieces(gs_slot);
          ++conv_next;
        }
      }
    }
    TMX_STAMP(ts0);

    // ---- epilogue: lane (qi, g) holds columns 16 * db + 4 * g .. +3 of its row for every db ----
    int stores = 0;
    if (w_has_rows) {
      float l_tot;  // sum over the 4 lanes that share a row
      if constexpr (V16L) {
        l_tot = lacc[0];
      } else {
        const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_tot = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
        const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
        l_tot = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
      }
      if (slot >= 0) {
        // piece of a split q-block: unnormalised fp32 accumulator + (max, sum) for the combine pass
        const long long prow = ((long long)slot * p.H + head) * p.block_m + my_row;
        float* po = p.partial_o + prow * D + 4 * g;
        if (row_valid) {
#pragma unroll
          for (int db = 0; db < DB; ++db) {
#if TMX_PARTIAL_NT
            store_stream16(po + 16 * db, make_float4(acc[db][0], acc[db][1], acc[db][2], acc[db][3]));
#else
            *reinterpret_cast<float4*>(po + 16 * db) = make_float4(acc[db][0], acc[db][1], acc[db][2], acc[db][3]);
#endif
          }
          if (g == 0) *reinterpret_cast<float2*>(p.partial_ml + prow * 2) = make_float2(m_run, l_tot);
        }
        stores = DB;  // the (max, sum) store may be skipped by lanes; counting fewer only waits longer
      } else {
        const float inv = 1.0f / l_tot;  // a fully masked row gives 0 * inf = NaN, as softmax over all -inf does
        const long long orow = (long long)(q_tok0 + my_row) * p.out_stride_t + (long long)head * p.out_stride_h;
        if (p.out_dtype == TMX_F32) {
          float* po = reinterpret_cast<float*>(p.out) + orow + 4 * g;
          if (row_valid) {
#pragma unroll
            for (int db = 0; db < DB; ++db)
              store_stream16(po + 16 * db, make_float4(acc[db][0] * inv, acc[db][1] * inv, acc[db][2] * inv, acc[db][3] * inv));
          }
          stores = DB;
        } else {
          // 16-bit output: lanes l and l ^ 16 exchange one 8-byte piece per pair of output blocks so
          // that every lane stores 16 contiguous bytes (8 columns)
          uint16_t* po = reinterpret_cast<uint16_t*>(p.out) + orow + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g);
#pragma unroll
          for (int c = 0; c < DB / 2; ++c) {
            // packed conversions: two columns per instruction
            const f32x4 oa = acc[2 * c] * inv, ob = acc[2 * c + 1] * inv;
            const uint2 au = make_uint2(TR::cvt2(oa[0], oa[1]), TR::cvt2(oa[2], oa[3]));
            const uint2 bu = make_uint2(TR::cvt2(ob[0], ob[1]), TR::cvt2(ob[2], ob[3]));
            const auto r0 = __builtin_amdgcn_permlane16_swap(au.x, bu.x, false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(au.y, bu.y, false, false);
            // even 16-lane rows: [own block 2c | partner's block 2c]; odd rows: [partner's block 2c+1 | own]
            if (row_valid) store_stream16(po + 32 * c, make_uint4(r0[0], r1[0], r0[1], r1[1]));
          }
          stores = DB / 2;
        }
      }
    }
    store_mark = gi;
    store_ops = stores;
    q_stores += stores;
    TMX_STAMP(ts1);
    TMX_SEG(4, ts0, ts1);

    ca0 = cb0;
    ca1 = cb1;
    c_valid = cb_valid;
    cb_valid = fetch_item(cb0, cb1);
    // the cursor follows: what was the next item is the current one now; a cursor that had nothing left
    // (the whole list of the new current item is issued) goes on into the new next item
    if (!issuer) {
    } else if (i_src == 1) {
      i_src = 0;
    } else if (cb_valid) {
      cursor_to(cb0, cb1, 1);
      cursor_settle();
      prefetch_desc();
    }
    TMX_STAMP(ts2);
    TMX_SEG(5, ts1, ts2);
  }
#ifdef TMX_STAMPS
  TMX_STAMP(ts4);
  seg[6] = ts4 - t_begin;
  seg[7] = (unsigned long long)gs;
  if (p.stamps != nullptr && lane == 0) {
    unsigned long long* dst = p.stamps + ((size_t)blockIdx.x * kWaves + wave) * 8;
    for (int x = 0; x < 8; ++x) dst[x] = seg[x];
  }
#endif
  // Tuning builds only (TMX_BAND_RLS): the list is done, the workgroup goes on with a share of the q-blocks the
  // band workgroups left.  Measured slower for every share (see launch_attn), so the shipped kernel has no such role.
#ifdef TMX_TUNING
  if constexpr (!HAS_BIAS && !FP8) {
    if (p.band_rls > 0) {
      const int qb0 = p.band_tail0 + item0 * p.band_rls;
      if (qb0 < p.num_qblocks) {
        // every wave is done with the ring before the band's window overwrites it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        band_body<TR, D, PAGED>(p, lds_base, kvc, qb0, imin(p.num_qblocks, qb0 + p.band_rls), false);
      }
    }
  }
#endif
}

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
  if (p.dense_on && q_len >= kDenseMinRows) return;  // the dense causal kernel wrote this request's rows
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


// Persistent grid: as many workgroups as stay resident (LDS and registers allow a fixed number per
// CU), rounded down to a multiple of the units per item so that a workgroup keeps one K/V head.
// The resident count of an instantiation is asked of the runtime once per device and kept.
constexpr int kMaxDevices = 64;

template <typename K>
static int resident_workgroups(K kernel, std::atomic<unsigned>* cache, long long* out, int* cus_out) {
  int dev = 0;
  TMX_CHECK_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < kMaxDevices) {
    const unsigned c = cache[dev].load(std::memory_order_relaxed);
    if (c != 0) {
      *out = c & 0xFFFFF;
      *cus_out = (int)(c >> 20);
      return TMX_OK;
    }
  }
  int cus = 0, per_cu = 0;
  TMX_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  TMX_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64 * kWaves, 0));
  if (per_cu < 1) per_cu = 1;
  *out = (long long)cus * per_cu;
  *cus_out = cus;
  // (resident workgroups, CUs) packed into one word: same value from every thread (host threads of different
  // GPUs / streams may race here; the slot is atomic and every writer stores the same word)
  if (dev >= 0 && dev < kMaxDevices && cus < 4096 && *out < (1 << 20))
    cache[dev].store(((unsigned)cus << 20) | (unsigned)*out, std::memory_order_relaxed);
  return TMX_OK;
}

// Grid of the persistent kernel for `resident` workgroup slots: every workgroup keeps ONE of the Y
// (K/V head, head chunk) units, so the grid is a multiple of Y, or exactly the work there is.
static int persistent_grid(long long resident, const AttnParams& p, unsigned* grid_out) {
  long long w = resident < p.units_total ? resident : p.units_total;
  TMX_CHECK_SUPPORTED(w >= p.Y || w == p.units_total,
                      "tmx_attention: %d K/V head chunks need at least as many resident workgroups (%lld)", p.Y,
                      resident);
  if (w > p.Y) w -= w % p.Y;
  *grid_out = (unsigned)w;
  return TMX_OK;
}

constexpr int kRingSlotsFp8 = 4;  // fp8 K/V: entries of half the size, one more of them

template <typename TR, int D, bool PAGED, bool HAS_BIAS, bool FP8 = false>
static int launch_attn(const AttnParams& p_in, hipStream_t s) {
  auto kernel = attn_kernel<TR, D, FP8 ? kRingSlotsFp8 : kRingSlots, PAGED, HAS_BIAS, FP8>;
  static std::atomic<unsigned> resident_cache[kMaxDevices];  // per device: resident workgroups of this instantiation (0: not asked yet)
  long long resident = 0;
  int cus = 0;
  int rc = resident_workgroups(kernel, resident_cache, &resident, &cus);
  if (rc != TMX_OK) return rc;
  AttnParams p = p_in;
  unsigned grid = 0;
  if (HAS_BIAS) p.band_on = 0;
  if (p.band_on) {
    // Work-list workgroups: one per unit of the split q-blocks' pieces (they also take the few unsplit
    // q-blocks outside the canonical positions); band workgroups: runs of band_rl q-blocks per unit, about
    // one run per CU -- with one piece per CU (the piece length the library advises) every workgroup of the
    // launch is resident from the start, and a band workgroup streams beside a piece for its whole life
    // (measured at 32k tokens, Qwen3-8B heads: runs of 32 q-blocks 166 us, 16: 172, 8: 188).
    TMX_CHECK_SUPPORTED(resident >= p.Y, "tmx_attention: %d K/V head chunks need at least as many resident workgroups (%lld)",
                        p.Y, resident);
    int rl = (int)(((long long)p.num_qblocks * p.Y + cus - 1) / cus);
    rl = rl < 4 ? 4 : (rl > 64 ? 64 : rl);
#ifdef TMX_TUNING
    if (const char* env = getenv("TMX_BAND_RL")) rl = atoi(env);
    if (const char* env = getenv("TMX_EXP")) p.exp_mode = atoi(env);
#endif
    p.band_rl = rl;
    long long runs = ((long long)p.num_qblocks + rl - 1) / rl;
    // Work-list workgroups: one per unit of the split q-blocks' pieces, and at least the slots the band
    // workgroups leave free: a step may hold requests the band does not take (prompts below the length gate,
    // chunks whose context is no multiple of 32, rows of a speculative decode), whose q-blocks are all on the
    // list (a 2048-token prompt beside 128 decode rows: 799 us on 32 workgroups, the pieces' count, before).
    // Workgroups that find nothing on the list return at once.
    long long want = (long long)(p.num_heavy > 4 ? p.num_heavy : 4) * p.Y;
    {
      const long long left = resident - runs * p.Y;
      if (left > want) want = left - left % p.Y;
      if (want > p.units_total && p.units_total >= p.Y) want = p.units_total - p.units_total % p.Y;
    }
    if (want > resident) want = resident - resident % p.Y;
    // chunks of a K/V head on one XCD (unit_of): both roles' workgroup counts become multiples of 8 x chunks
    const int chunks = p.G / p.Gw;
    const bool xcd = chunks > 1 && p.Hkv <= 8 && 8 % p.Hkv == 0;
    const int unit = 8 * chunks;
    if (xcd) {
      want = (want + unit - 1) / unit * unit;
      if (want > resident) want -= unit;
    }
    p.xcd_generic = p.xcd_band = (xcd && want >= unit) ? 1 : 0;
    p.n_generic = (int)want;
    if (p.xcd_band) runs = (runs + 8 / p.Hkv - 1) / (8 / p.Hkv) * (8 / p.Hkv);  // (surplus runs find no q-block)
#ifdef TMX_TUNING
    if (const char* env = getenv("TMX_BAND_MAP")) {
      if (atoi(env) == 2 && !p.xcd_band) {
        p.xcd_band = 2;
        runs = (runs + 7) / 8 * 8;
      }
    }
#endif
    // Tail share (tuning builds, TMX_BAND_RLS = q-blocks per work-list workgroup).  A band workgroup works for the
    // whole launch, a work-list workgroup with one piece of the trailing dense rows for about four fifths of it,
    // so the last q-blocks were left to the work-list workgroups for when their list is done, the band runs
    // shortened to match.  Measured at 32k tokens, Qwen3-8B heads, one box: 162.3 us without, 167.9 / 171.9 /
    // 174.9 / 177.6 us with 2 / 3 / 4 / 5 q-blocks per work-list workgroup -- a second band run on a CU starts
    // with an empty pipeline and fetches 64 keys again, and the launch is bound by the memory system, which
    // a further stream per CU does not speed up (band workgroups alone: 137.9 us with one per CU, 144.7 with two).
    p.band_tail0 = p.num_qblocks;
    p.band_rls = 0;
#ifdef TMX_TUNING
    if (const char* env = getenv("TMX_BAND_RLS")) {
      const long long S = want / p.Y, R = runs, rls = atoi(env);
      if (p.batch == 1 && want >= p.Y && R > 0 && rls >= 1 && rls <= 64 && S * rls < p.num_qblocks) {
        p.band_rls = (int)rls;
        p.band_tail0 = (int)(p.num_qblocks - S * rls);
        p.band_rl = (int)((p.band_tail0 + R - 1) / R);
      }
    }
#endif
    // Interleaved walk of the pieces (tuning builds, TMX_WALK_SHIFT = log2 of the parts; as many parts as band runs
    // share a piece's keys would be 1 at 32k tokens).  Measured at 32k tokens, Qwen3-8B heads, one box, three
    // alternating rounds: 154.3-155.5 us front to back, 155.1-156.0 in two parts, 155.0-155.3 in four -- no
    // difference, so either the second read of K/V is not what the launch waits for or those lines do not stay in
    // the Infinity Cache beside the query / output streams; the counters (L2 fills, which count cache hits too) cannot tell.
    p.walk_shift = 0;
#ifdef TMX_TUNING
    if (const char* env = getenv("TMX_WALK_SHIFT")) p.walk_shift = atoi(env);
#endif
    const long long total = want + runs * p.Y;
    TMX_CHECK_SUPPORTED(total < (1ll << 31), "tmx_attention: too many workgroups");
    grid = (unsigned)total;
  } else {
    rc = persistent_grid(resident, p, &grid);
    if (rc != TMX_OK) return rc;
    const int chunks = p.G / p.Gw;
    const int unit = 8 * chunks;
    if (chunks > 1 && p.Hkv <= 8 && 8 % p.Hkv == 0 && grid >= (unsigned)unit) {
      grid -= grid % unit;
      p.xcd_generic = 1;
    }
    p.n_generic = (int)grid;
  }
  kernel<<<grid, 64 * kWaves, 0, s>>>(p);
  TMX_CHECK_HIP(hipGetLastError());
  return TMX_OK;
}

template <typename TR, int D>
static int launch(const AttnParams& p, bool paged, bool has_bias, bool fp8, int64_t num_split, uint32_t flags, hipStream_t s) {
  if (p.units_total > 0 && !(flags & TMX_ATTN_ONLY_COMBINE)) {
    int rc;
    if (fp8) {
      if constexpr (D == 128) rc = launch_attn<TR, 128, true, false, true>(p, s);
      else rc = TMX_ERR_UNSUPPORTED;
    } else if (paged) rc = has_bias ? launch_attn<TR, D, true, true>(p, s) : launch_attn<TR, D, true, false>(p, s);
    else rc = has_bias ? launch_attn<TR, D, false, true>(p, s) : launch_attn<TR, D, false, false>(p, s);
    if (rc != TMX_OK) return rc;
  }
  if (num_split > 0 && !(flags & TMX_ATTN_SKIP_COMBINE)) {
    const unsigned groups = (unsigned)((p.block_m + 7) / 8);
    combine_kernel<TR, D><<<dim3((unsigned)num_split * groups, (unsigned)p.H), 8 * (D / 4), 0, s>>>(p);
    TMX_CHECK_HIP(hipGetLastError());
  }
  return TMX_OK;
}

// The schedule's exchange rate: how many light (unsplit) items keep a workgroup busy as long as one piece
// of a split q-block does.  A piece is chunk_tiles arithmetic-bound tile steps; a light item costs its tile
// steps plus a fixed part (query rows in, output rows out, set-up) worth about six steps when it runs beside
// a piece (measured on MI355X: 64-tile pieces against 3-tile items balance at 6..8 items per piece).
// TMX_SCHED_W overrides it in diagnostic builds (tuning runs).
static int sched_light_per_piece(const tmx_attention_args_t* a, const tmx_index_t* ix) {
#ifdef TMX_STAMPS
  static const char* env = getenv("TMX_SCHED_W");
  if (env != nullptr) return atoi(env);
#endif
  const long long pieces = a->num_partials, light = a->num_items - a->num_partials;
  if (pieces <= 0 || light <= 0) return 0;
  // listed tiles: the count the caller passes (tmx_index_sizes), else the buffer's capacity as an upper bound
  const long long nnz = a->nnz_tiles > 0 ? a->nnz_tiles : ix->cap_nnz;
  const long long light_tiles = nnz - pieces * ix->chunk_tiles;  // lower bound: last pieces are shorter
  const double avg = light_tiles > light ? (double)light_tiles / (double)light : 1.0;
  const double w = (double)ix->chunk_tiles / (6.0 + avg);
  return w > 64.0 ? 64 : (int)(w + 0.5);
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
  TMX_CHECK_ARG(a->scale > 0.f && a->scale < 3.0e38f, "tmx_attention: scale %g must be positive and finite",
                (double)a->scale);
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
  TMX_CHECK_SUPPORTED(a->num_items < (1ll << 30), "tmx_attention: more than 2^30 work items");
  TMX_CHECK_ARG(a->num_items <= ix->cap_items && a->num_split_qblocks <= ix->cap_split,
                "tmx_attention: num_items / num_split_qblocks exceed the index capacities");
  // (every partial slot belongs to a piece, and every piece is an item)
  TMX_CHECK_ARG(a->num_partials <= a->num_items, "tmx_attention: num_partials %lld exceeds num_items %lld",
                (long long)a->num_partials, (long long)a->num_items);
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
  const bool fp8 = a->kv_dtype == TMX_FP8_E4M3;
  TMX_CHECK_SUPPORTED(a->kv_dtype == 0 || a->kv_dtype == a->io_dtype || fp8,
                      "tmx_attention: kv_dtype %d not supported (0 / the q dtype, or TMX_FP8_E4M3)", a->kv_dtype);
  if (fp8) {
    TMX_CHECK_SUPPORTED(paged && !has_bias && a->head_dim == 128,
                        "tmx_attention: an fp8 K/V cache needs the paged layout, head_dim 128 and no bias "
                        "(other cases: tmx_gather_dequant_kv_fp8 ahead of the call)");
    TMX_CHECK_ARG(a->k_scale > 0.f && a->v_scale > 0.f && a->k_scale < 3.0e38f && a->v_scale < 3.0e38f,
                  "tmx_attention: k_scale / v_scale must be positive and finite");
    TMX_CHECK_ARG(a->kv_stride_t % 16 == 0 && a->kv_stride_h % 16 == 0 && a->kv_stride_page % 16 == 0,
                  "tmx_attention: fp8 cache strides must be multiples of 16 bytes");
  }
  const bool two_src = a->k_new != nullptr || a->v_new != nullptr;
  if (two_src) {
    TMX_CHECK_ARG(a->k_new != nullptr && a->v_new != nullptr, "tmx_attention: k_new and v_new go together");
    TMX_CHECK_SUPPORTED(paged && !fp8, "tmx_attention: k_new / v_new need a 16-bit paged cache (an fp8 cache holds other "
                                       "values than the step's rows)");
    TMX_CHECK_ARG(aligned16(a->k_new) && aligned16(a->v_new) && a->new_stride_t % 8 == 0 && a->new_stride_h % 8 == 0 &&
                      a->new_stride_t >= 0 && a->new_stride_h >= 0 && a->new_stride_t < (1ll << 24) &&
                      a->new_stride_h <= 0x7FFFFFFF,
                  "tmx_attention: k_new / v_new must be 16-byte aligned with strides that are multiples of 8 elements");
  }
  if (a->cache_slots != nullptr) {
    TMX_CHECK_ARG(two_src, "tmx_attention: cache_slots needs k_new / v_new (the rows to store)");
    TMX_CHECK_SUPPORTED((long long)a->total_q * a->num_kv_heads * (a->head_dim / 8) < (1ll << 31),
                        "tmx_attention: too many rows for the fused cache write");
  }
  if (has_bias)
    TMX_CHECK_ARG((a->bias_dtype == TMX_F32 || a->bias_dtype == TMX_BF16 || a->bias_dtype == TMX_F16) &&
                      a->bias_rows > 0 && a->bias_cols > 0 && a->bias_stride >= a->bias_cols && a->bias_stride_b >= 0 &&
                      a->bias_stride_h >= 0,
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
  // diagnostic build: the stamps go to the second half of a (max, sum) buffer of at least 2^21 floats, which only
  // tools/stamps.py provides -- it says so with TMX_STAMPS_BUFFER=1; any other caller's buffer is left alone
  if (const char* env = getenv("TMX_STAMPS_BUFFER"))
    if (env[0] == '1' && a->partial_ml != nullptr) p.stamps = reinterpret_cast<unsigned long long*>(a->partial_ml + (1 << 19));
#endif
  const int64_t lim = 0x7FFFFFFF;
  TMX_CHECK_SUPPORTED(a->q_stride_t <= lim && a->q_stride_h <= lim && a->out_stride_t <= lim &&
                          a->out_stride_h <= lim && a->kv_stride_page <= lim && a->kv_stride_t <= lim &&
                          a->kv_stride_h <= lim && a->block_table_stride <= lim && a->q_stride_t >= 0 &&
                          a->q_stride_h >= 0 && a->out_stride_t >= 0 && a->out_stride_h >= 0 &&
                          a->kv_stride_page >= 0 && a->kv_stride_t >= 0 && a->kv_stride_h >= 0 &&
                          a->block_table_stride >= 0,
                      "tmx_attention: strides must be non-negative and below 2^31 elements");
  // per-lane byte offsets inside a tile (32 rows) / a wave's query block (16 rows) are 32-bit
  TMX_CHECK_SUPPORTED(a->kv_stride_t < (1ll << 24) && a->q_stride_t < (1ll << 24),
                      "tmx_attention: row strides of q and k/v must be below 2^24 elements");
  p.q_stride_t = (int)a->q_stride_t;
  p.q_stride_h = (int)a->q_stride_h;
  p.out_stride_t = (int)a->out_stride_t;
  p.out_stride_h = (int)a->out_stride_h;
  p.kv_stride_page = (int)a->kv_stride_page;
  p.kv_stride_t = (int)a->kv_stride_t;
  p.kv_stride_h = (int)a->kv_stride_h;
  p.block_table_stride = (int)a->block_table_stride;
  p.tile_pages = nullptr;
  p.full_lens = ix->full_lens;
  p.k_new = a->k_new;
  p.v_new = a->v_new;
  p.new_stride_t = (int)a->new_stride_t;
  p.new_stride_h = (int)a->new_stride_h;
  p.cache_slots = reinterpret_cast<const long long*>(a->cache_slots);
  p.total_q = a->total_q;
  if (paged && ix->tile_pages != nullptr && ix->pages_block_table == a->block_table &&
      ix->pages_page_size == a->page_size && ix->pages_block_table_stride == a->block_table_stride)
    p.tile_pages = ix->tile_pages;
  p.bias_stride = a->bias_stride;
  p.bias_stride_b = a->bias_stride_b;
  p.bias_stride_h = a->bias_stride_h;
  p.page_size = paged ? a->page_size : 1;
  p.page_shift = paged ? __builtin_ctz(a->page_size) : 0;
  p.H = a->num_heads;
  p.Hkv = a->num_kv_heads;
  p.G = G;
  p.Gw = heads_per_wg(G);
  p.gw_shift = __builtin_ctz(p.Gw);
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
  p.k_scale = fp8 ? a->k_scale : 1.f;
  p.v_scale = fp8 ? a->v_scale : 1.f;

  p.Y = a->num_kv_heads * (G / p.Gw);
  p.num_items = (int)a->num_items;
  p.issue_hi = (a->flags & TMX_ATTN_SHORT_QUERIES) ? 1 : 0;
  p.num_heavy = (int)(a->num_partials < a->num_items ? a->num_partials : a->num_items);
  p.sched_w = sched_light_per_piece(a, ix);
  p.units_total = (long long)a->num_items * p.Y;
  // The band path (sparse q-blocks in canonical position on workgroups of their own) covers the default
  // pattern's shape: a window of 17..32 keys below the row, at most 16 sink columns, four query heads per
  // workgroup, 32-row q-blocks that the index does not split.
  p.num_qblocks = (int)(a->num_items - a->num_partials + a->num_split_qblocks);
  p.band_on = (a->mode == TMX_MODE_TRIANGLE && !has_bias && p.Gw == 4 && p.block_m == 32 && a->sl >= 17 && a->sl <= 32 &&
               a->si >= -1 && a->si <= 15 && a->last >= 0 && ix->chunk_tiles >= 3 && p.num_qblocks >= 8 &&
               !(a->flags & (TMX_ATTN_SHORT_QUERIES | TMX_ATTN_NO_BAND)))
                  ? 1
                  : 0;
  if (p.band_on) p.sched_w = 0;
  // Dense causal kernel: long requests of a causal launch (the layers outside the triangle set) go to the
  // kernel built for the matrix cores; this launch keeps the requests with few query rows.
  p.dense_on = (a->mode == TMX_MODE_CAUSAL && !has_bias && G % 4 == 0 && a->head_dim == 128 && a->batch <= kDenseMaxBatch &&
                a->total_q >= kDenseMinRows && !(a->flags & (TMX_ATTN_SHORT_QUERIES | TMX_ATTN_NO_DENSE | TMX_ATTN_ONLY_COMBINE)))
                   ? 1
                   : 0;
  p.band_rl = 0;
  p.band_tail0 = 0;
  p.band_rls = 0;
  p.batch = a->batch;
  p.chunk_tiles = ix->chunk_tiles;
  p.walk_shift = 0;
  p.n_generic = 0;
  p.exp_mode = 0;
  p.xcd_generic = p.xcd_band = 0;
  hipStream_t s = (hipStream_t)stream;
  if (p.dense_on) {
    DenseParams d;
    d.q = a->q;
    d.out = a->out;
    d.k = a->k;
    d.v = a->v;
    d.block_table = a->block_table;
    d.cu_k = a->cu_seqlens_k;
    d.cu_q = a->cu_seqlens_q;
    d.seq_lens = a->seq_lens;
    d.k_new = a->k_new;
    d.v_new = a->v_new;
    d.new_stride_t = p.new_stride_t;
    d.new_stride_h = p.new_stride_h;
    d.cache_slots = p.cache_slots;  // the dense kernel, launched first, stores the step's rows
    p.cache_slots = nullptr;
    d.q_stride_t = p.q_stride_t;
    d.q_stride_h = p.q_stride_h;
    d.out_stride_t = p.out_stride_t;
    d.out_stride_h = p.out_stride_h;
    d.kv_stride_page = p.kv_stride_page;
    d.kv_stride_t = p.kv_stride_t;
    d.kv_stride_h = p.kv_stride_h;
    d.block_table_stride = p.block_table_stride;
    d.page_size = p.page_size;
    d.page_shift = p.page_shift;
    d.H = p.H;
    d.Hkv = p.Hkv;
    d.G = G;
    d.Y = a->num_kv_heads * (G / 4);
    d.batch = a->batch;
    d.total_q = a->total_q;
    d.out_dtype = a->out_dtype;
    d.scale_log2 = p.scale_log2;
    d.kv_fp8 = fp8 ? 1 : 0;
    d.k_scale = p.k_scale;
    d.v_scale = p.v_scale;
    const int rc = launch_dense_causal(d, a->io_dtype == TMX_BF16, paged, s);
    if (rc != TMX_OK) return rc;
  }
  // Decode kernel: one-row requests whose row is dense (every decode step of the default pattern, every row of
  // a causal layer) stream their K/V straight into registers there.  A step of one-row requests only is all
  // the kernel's; in a mixed step its grid also covers the other requests' items (workgroups that return at
  // once), which pays only while that grid is small.
  p.decode_on = 0;
  uint32_t flags = a->flags;
  {
    const int hw_d = G % 8 == 0 ? 8 : 4;  // query heads per decode workgroup (all of them share the K/V loads)
    const int Yd = a->num_kv_heads * (G / hw_d > 0 ? G / hw_d : 1);
    const bool shape_ok = !has_bias && G % 4 == 0 && a->head_dim == 128 && (!paged || a->page_size >= 16) &&
                          (a->mode != TMX_MODE_TRIANGLE || a->last >= 2) &&
                          !(a->flags & (TMX_ATTN_NO_DECODE | TMX_ATTN_ONLY_COMBINE));
    const bool pure = a->total_q == a->batch;
    if (shape_ok && (pure || (a->batch >= 2 && (long long)a->num_items * Yd <= kDecodeMaxMixedGrid))) p.decode_on = 1;
    if (p.decode_on) {
      DecodeParams d;
      d.q = a->q;
      d.out = a->out;
      d.k = a->k;
      d.v = a->v;
      d.block_table = a->block_table;
      d.cu_k = a->cu_seqlens_k;
      d.cu_q = a->cu_seqlens_q;
      d.full_lens = p.full_lens;
      d.items = p.items;
      d.col_idx = p.col_idx;
      d.partial_o = p.partial_o;
      d.partial_ml = p.partial_ml;
      d.k_new = a->k_new;
      d.v_new = a->v_new;
      d.cache_slots = p.cache_slots;  // launched first, the decode kernel stores the step's rows
      p.cache_slots = nullptr;
      d.new_stride_t = p.new_stride_t;
      d.new_stride_h = p.new_stride_h;
      d.total_q = a->total_q;
      d.q_stride_t = p.q_stride_t;
      d.q_stride_h = p.q_stride_h;
      d.out_stride_t = p.out_stride_t;
      d.out_stride_h = p.out_stride_h;
      d.kv_stride_page = p.kv_stride_page;
      d.kv_stride_t = p.kv_stride_t;
      d.kv_stride_h = p.kv_stride_h;
      d.block_table_stride = p.block_table_stride;
      d.page_size = p.page_size;
      d.page_shift = p.page_shift;
      d.H = p.H;
      d.Hkv = p.Hkv;
      d.G = G;
      d.hw = hw_d;
      d.valu_sum = (a->flags & TMX_ATTN_DECODE_VALU) ? 1 : 0;
      d.Y = Yd;
      d.block_m = p.block_m;
      d.num_items = p.num_items;
      d.mode = p.mode;
      d.last = p.last;
      d.tri_min_len = p.tri_min_len;
      d.out_dtype = a->out_dtype;
      d.scale_log2 = p.scale_log2;
      d.kv_fp8 = fp8 ? 1 : 0;
      d.k_scale = p.k_scale;
      d.v_scale = p.v_scale;
      const int rc = launch_decode(d, a->io_dtype == TMX_BF16, paged, s);
      if (rc != TMX_OK) return rc;
      // every request has one row and every such row is dense: nothing is left for the work-list kernel
      const bool all_dense_rows = a->mode != TMX_MODE_TRIANGLE || (a->last >= 2 && p.full_lens == nullptr);
      if ((a->flags & TMX_ATTN_DECODE_ONLY) && pure && all_dense_rows) flags |= TMX_ATTN_ONLY_COMBINE;
    }
  }
  if (a->io_dtype == TMX_BF16) {
    if (a->head_dim == 128) return launch<Bf16Traits, 128>(p, paged, has_bias, fp8, a->num_split_qblocks, flags, s);
    return launch<Bf16Traits, 64>(p, paged, has_bias, fp8, a->num_split_qblocks, flags, s);
  }
  if (a->head_dim == 128) return launch<F16Traits, 128>(p, paged, has_bias, fp8, a->num_split_qblocks, flags, s);
  return launch<F16Traits, 64>(p, paged, has_bias, fp8, a->num_split_qblocks, flags, s);
}

// ---------------------------------------------------------------------------------------------------
// The same call for callers that may not allocate: the work area of the split q-blocks lives in the
// library's own device memory.  Every device the library runs on holds one copy of the array (at its own
// address) until the process ends.
namespace tmx {

constexpr size_t kStaticWsBytes = (size_t)TMX_STATIC_WORKSPACE_BYTES;
static __device__ __align__(256) unsigned char static_ws[kStaticWsBytes];

// Device address of this device's copy of static_ws.  The array's name is only a handle on the host:
// the address comes from hipGetSymbolAddress, asked once per device with that device current.
static int static_ws_base(unsigned char** out) {
  static std::atomic<unsigned char*> base[kMaxDevices];
  int dev = 0;
  TMX_CHECK_HIP(hipGetDevice(&dev));
  TMX_CHECK_SUPPORTED(dev >= 0 && dev < kMaxDevices, "tmx_attention_static_ws: device ordinal %d out of range", dev);
  unsigned char* ptr = base[dev].load(std::memory_order_acquire);
  if (ptr == nullptr) {
    TMX_CHECK_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(static_ws)));
    TMX_CHECK_ARG(ptr != nullptr, "tmx_attention_static_ws: no device address for the static work area");
    base[dev].store(ptr, std::memory_order_release);  // every thread gets the same answer for a device
  }
  *out = ptr;
  return TMX_OK;
}

}  // namespace tmx

extern "C" size_t tmx_static_workspace_bytes(void) { return tmx::kStaticWsBytes; }

extern "C" int tmx_static_workspace_warmup(void) {
  unsigned char* base = nullptr;
  return tmx::static_ws_base(&base);
}

extern "C" int tmx_attention_static_ws(const tmx_attention_args_t* a, void* stream) {
  using namespace tmx;
  TMX_CHECK_ARG(a != nullptr, "tmx_attention_static_ws: args is NULL");
  TMX_CHECK_ARG(a->struct_size == sizeof(tmx_attention_args_t),
                "tmx_attention_static_ws: struct_size %u != %zu (ABI mismatch)", a->struct_size,
                sizeof(tmx_attention_args_t));
  tmx_attention_args_t b = *a;
  b.partial_o = nullptr;
  b.partial_ml = nullptr;
  if (a->num_partials > 0) {
    TMX_CHECK_ARG(a->index != nullptr && a->num_heads > 0 && a->head_dim > 0 && a->index->block_m > 0,
                  "tmx_attention_static_ws: bad sizes");
    // bounds ahead of the products below, so that none of them can wrap (a slot is a piece, a piece an item)
    TMX_CHECK_ARG(a->num_partials <= a->num_items && a->num_items < (1ll << 30) && a->num_heads < (1 << 16) &&
                      a->head_dim <= 1024 && a->index->block_m <= 1024,
                  "tmx_attention_static_ws: num_partials %lld / num_heads %d / head_dim %d out of range",
                  (long long)a->num_partials, a->num_heads, a->head_dim);
    // partial_o [P][H][block_m][D] floats, then partial_ml [P][H][block_m][2] floats on the next 256-byte boundary
    const unsigned long long rows = (unsigned long long)a->num_partials * (unsigned long long)a->num_heads *
                                    (unsigned long long)a->index->block_m;
    const unsigned long long o_bytes = (rows * (unsigned long long)a->head_dim * 4ull + 255ull) & ~255ull;
    const unsigned long long need = o_bytes + rows * 8ull;
    if (need > kStaticWsBytes) {
      set_error("tmx_attention_static_ws: the split q-blocks of this step need %llu bytes of work area, the library "
                "holds %zu (use tmx_attention with a workspace of your own, or longer pieces)", need, kStaticWsBytes);
      return TMX_ERR_CAPACITY;
    }
    unsigned char* base = nullptr;
    const int rc = static_ws_base(&base);
    if (rc != TMX_OK) return rc;
    b.partial_o = reinterpret_cast<float*>(base);
    b.partial_ml = reinterpret_cast<float*>(base + o_bytes);
  }
  return tmx_attention(&b, stream);
}
