// Decode rows for gfx950 (MI355X): a request with ONE query row whose row is dense -- every decode step of the
// reference's pattern is (the last row always lies among the trailing rows, N - i < last,
// vllm/attention/trianglemix_attention.py:126-129) and so is every row of a causal layer.  Such a row reads all
// of its K/V once and does almost no arithmetic: the work-list kernel's LDS ring, barriers and 16-row tiles
// buy nothing there.  This kernel streams K/V straight from memory into registers:
//
//   workgroup = 4 waves = one work item (a piece of a request's key range) x 4 query heads of one K/V head (8
//   when the group size is a multiple of 8: the K/V of a head are then still read once);
//   the waves take the piece's 32-key tiles in turn and never synchronise until the end.
//   Per 16 keys and wave: K rows arrive in the A layout of v_mfma_f32_16x16x32 (lane = key, 16 bytes of the
//   row per k-step), the 4 query heads are 4 of the 16 B columns, so 4 matrix instructions give all scores;
//   V rows arrive as 16 bytes per lane (lane group g of 16 lanes = key 4 g + i of step i, lane = 8 head-dim
//   columns).  The weighted sum runs on the matrix cores as well (the default): the V rows of 16 keys go
//   through a 4-KiB LDS stage of the wave's own and come back transposed (ds_read_b64_tr_b16), the probabilities
//   stay in registers as the second operand (two bf16 terms, one fp16 term).  TMX_ATTN_DECODE_VALU keeps the
//   earlier form for comparison: the probabilities reach the V lanes by ds_bpermute (16 per 16 keys) and the sum
//   runs on the vector units in fp32.  The next 16 keys' 8 loads per lane are in flight while a group is computed.
//   An fp8 (e4m3fn) cache is read at half the bytes and descaled in registers: value = fp8 x scale, rounded
//   once to the 16-bit type (vllm/v1/attention/backends/rocm_attn.py:421-422).
//   At the end the 16 partial results of a workgroup (4 waves x 4 lane groups) merge through LDS into the
//   output row, or into the piece's partial slot for the combine pass of the work-list kernel's launch.
#include <atomic>
#include <type_traits>

#include "tmx_common.h"
#include "tmx_decode.h"

namespace tmx {

typedef __attribute__((ext_vector_type(4))) float dc_f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 dc_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 dc_f16x8;
typedef __attribute__((ext_vector_type(4))) int dc_i32x4;
typedef __attribute__((ext_vector_type(2))) float dc_f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned dc_u32x4;
typedef __attribute__((ext_vector_type(4))) short dc_s16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 dc_f16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 dc_bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 dc_f16x2;

// Cache policy of the K/V stream (every byte is read once; 2 = non-temporal, 0 = default).  A V load of a wave
// covers whole cache lines (4 keys x the head's row): non-temporal.  A K load covers half a line per key, the
// next k-step's load the other half: those lines must stay in the CU's cache, default policy.  Measured on one
// box, 64 x 8192 decode, 16-bit cache: both default 370 us, both non-temporal 363, K default + V non-temporal
// 338 (6.35 TB/s); fp8 cache: 191 / 215 / 185 us.
#ifndef TMX_DC_POLICY16K
#define TMX_DC_POLICY16K 0
#endif
#ifndef TMX_DC_POLICY16V
#define TMX_DC_POLICY16V 2
#endif
#ifndef TMX_DC_POLICY8K
#define TMX_DC_POLICY8K 0
#endif
#ifndef TMX_DC_POLICY8V
#define TMX_DC_POLICY8V 2
#endif
constexpr float kDcMinusBig = -1.0e30f;
constexpr float kDcSlack = 8.0f;

template <typename T>
__device__ __forceinline__ T dcs(const T* ptr) {
  return *reinterpret_cast<const __attribute__((address_space(4))) T*>((const __attribute__((address_space(4))) void*)ptr);
}

template <bool BF16>
__device__ __forceinline__ uint32_t dc_pack(float a, float b) {
  if constexpr (BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(dc_f32x2{a, b}, dc_bf16x2));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(dc_f32x2{a, b}, dc_f16x2));
}
// the two 16-bit values of a dword as floats
template <bool BF16>
__device__ __forceinline__ void dc_unpack(uint32_t w, float& lo, float& hi) {
  if constexpr (BF16) {
    lo = __uint_as_float(w << 16);
    hi = __uint_as_float(w & 0xffff0000u);
  } else {
    const dc_f16x2 h = __builtin_bit_cast(dc_f16x2, w);
    lo = (float)h[0];
    hi = (float)h[1];
  }
}

// ---- the fixed-grid form's plan (see DecodeParams): computed by every workgroup for itself ----
// Sum of v over the NT threads of the workgroup (sm: NT / 64 ints of LDS, free again on return).
template <int NT>
__device__ __forceinline__ int dc_block_sum(int v, int* sm, int tid) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if ((tid & 63) == 0) sm[tid >> 6] = v;
  __syncthreads();
  int tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) tot += sm[w];
  __syncthreads();
  return tot;
}
__device__ __forceinline__ int dc_tiles_of(int n) { return n > 0 ? (n + kBlockN - 1) / kBlockN : 0; }
// Piece length of the step: one piece per workgroup slot of the chip, at least 8 tiles.
__device__ __forceinline__ int dc_piece_len(long long total_tiles, int units, int slots) {
  const long long c = (total_tiles * units + slots - 1) / slots;
  return c < 8 ? 8 : (int)c;
}
struct DcPiece {
  int b, n, tile0, ntiles, pieces;  // request (-1: no such piece), its keys, the piece's tiles, the request's piece count
};
// The request that owns piece w of the step, and which of its tiles the piece covers.  All NT threads call it.
template <int NT>
__device__ __forceinline__ DcPiece dc_find_piece(const int32_t* seq_lens, int batch, int units, int slots, int w, int* sm, int tid) {
  const int per = (batch + NT - 1) / NT;  // requests per thread, a contiguous run
  const int j0 = imin(batch, tid * per), j1 = imin(batch, j0 + per);
  int mine = 0;
  for (int j = j0; j < j1; ++j) mine += dc_tiles_of(seq_lens[j]);
  const int total = dc_block_sum<NT>(mine, sm, tid);
  const int P = dc_piece_len(total, units, slots);
  int cnt = 0;
  for (int j = j0; j < j1; ++j) cnt += (dc_tiles_of(seq_lens[j]) + P - 1) / P;
  // exclusive prefix of cnt over the threads: inside the wave by shuffles, across waves through LDS
  int inc = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(inc, d, 64);
    if ((tid & 63) >= d) inc += up;
  }
  if ((tid & 63) == 63) sm[tid >> 6] = inc;
  if (tid == 0) sm[NT / 64] = -1;  // the answer's slot: (request, first piece of the request)
  __syncthreads();
  int start = inc - cnt;
  for (int x = 0; x < (tid >> 6); ++x) start += sm[x];
  if (w >= start && w < start + cnt) {
    int first = start;
    for (int j = j0; j < j1; ++j) {
      const int pj = (dc_tiles_of(seq_lens[j]) + P - 1) / P;
      if (w < first + pj) {
        sm[NT / 64] = j;
        sm[NT / 64 + 1] = first;
        break;
      }
      first += pj;
    }
  }
  __syncthreads();
  DcPiece r;
  r.b = sm[NT / 64];
  if (r.b < 0) return r;
  r.n = seq_lens[r.b];
  const int tiles = dc_tiles_of(r.n);
  r.pieces = (tiles + P - 1) / P;
  r.tile0 = (w - sm[NT / 64 + 1]) * P;
  r.ntiles = imin(P, tiles - r.tile0);
  return r;
}

// FIXED: the grid does not come from an index.  Workgroup (request, piece, K/V head chunk) derives its key range
// from seq_lens on the device, so the launch has the same shape whatever the requests' lengths are (a step
// captured in a HIP graph replays with other lengths and block tables).
template <bool BF16, bool PAGED, bool KV8, bool UNIT, int HW, bool MPV, bool FIXED = false>
__global__ __launch_bounds__(256, 2) void decode_kernel(const DecodeParams p) {
  constexpr int D = 128;
  constexpr int KVB = KV8 ? 1 : 2;
  static_assert(HW == 4 || HW == 8, "4 or 8 query heads per workgroup");
  // MPV: the weighted sum runs on the matrix cores too (V through a 4-KiB LDS stage of the wave's own, read
  // back transposed); the accumulators then hold a wave's sum over all its keys
  __shared__ float part_o[MPV ? 4 : 16][HW][D];   // [wave (* 4 + lane group)][head][column]
  __shared__ float part_ml[16][HW][2];             // [wave * 4 + lane group][head][(max, sum)]
  __shared__ __attribute__((aligned(1024))) char v_stage[MPV ? 4 * 4096 : 16];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g = lane >> 4;

  if (p.cache_slots != nullptr) {
    // The cache write of the step (the reference's reshape_and_cache_flash, csrc/cache_kernels.cu:282-355),
    // shared out over the launch's threads.  Nothing in this launch reads those slots: the step's own keys
    // come from k_new / v_new.
    constexpr int CPR = D / 8;  // 16-byte chunks per row of one head
    const int units = p.total_q * p.Hkv * CPR;
    const int stride = (int)gridDim.x * 256;
    const uint16_t* kn = reinterpret_cast<const uint16_t*>(p.k_new);
    const uint16_t* vn = reinterpret_cast<const uint16_t*>(p.v_new);
    for (int u = (int)blockIdx.x * 256 + tid; u < units; u += stride) {
      const int c = u % CPR, th = u / CPR;
      const int h = th % p.Hkv, t = th / p.Hkv;
      const long long slot = p.cache_slots[t];
      if (slot < 0) continue;
      const long long src = (long long)t * p.new_stride_t + (long long)h * p.new_stride_h + 8 * c;
      // (cache strides: elements of the cache's type -- bytes for an fp8 cache)
      const long long dst = (slot >> p.page_shift) * p.kv_stride_page + (slot & (p.page_size - 1)) * (long long)p.kv_stride_t +
                            (long long)h * p.kv_stride_h + 8 * c;
      const uint4 kx = *reinterpret_cast<const uint4*>(kn + src);
      const uint4 vx = *reinterpret_cast<const uint4*>(vn + src);
      if constexpr (KV8) {
        // an fp8 cache: the quantising write (the reference's reshape_and_cache_flash with kv_cache_dtype "fp8"),
        // the same bytes as tmx_reshape_and_cache_fp8 stores
        *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(const_cast<void*>(p.k)) + dst) = quant8_fp8(kx, BF16, p.k_scale);
        *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(const_cast<void*>(p.v)) + dst) = quant8_fp8(vx, BF16, p.v_scale);
      } else {
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(const_cast<void*>(p.k)) + dst) = kx;
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(const_cast<void*>(p.v)) + dst) = vx;
      }
    }
  }

  int kvc, b, q_tok, pos, n, ntiles, slot, tile0;
  if constexpr (FIXED) {
    __shared__ int plan_sm[4 + 2];
    const int w = (int)blockIdx.x / p.Y;
    kvc = (int)blockIdx.x - w * p.Y;
    const DcPiece dp = dc_find_piece<256>(p.seq_lens, p.total_q, p.plan_units, p.plan_slots, w, plan_sm, tid);
    if (dp.b < 0) return;  // fewer pieces in this step than the grid provides for
    b = dp.b;
    n = dp.n;
    tile0 = dp.tile0;
    ntiles = dp.ntiles;
    slot = dp.pieces > 1 ? w : -1;  // one piece: the output row is written straight away
    q_tok = b;
    pos = n - 1;
  } else {
    const int item = (int)blockIdx.x / p.Y;
    kvc = (int)blockIdx.x % p.Y;
    const dc_i32x4 a0 = dcs(reinterpret_cast<const dc_i32x4*>(p.items) + 2ll * item);
    const dc_i32x4 a1 = dcs(reinterpret_cast<const dc_i32x4*>(p.items) + 2ll * item + 1);
    // a0 = (q-block, first entry of col_idx, tiles, partial slot), a1 = (request, query row, position, keys)
    b = a1.x, q_tok = a1.y, pos = a1.z, n = a1.w;
    {
      const int q_len = dcs(p.cu_q + b + 1) - dcs(p.cu_q + b);
      const int nf = p.full_lens != nullptr ? imax(n, dcs(p.full_lens + b)) : n;
      if (!decode_row(p.mode, p.last, p.tri_min_len, q_len, pos, nf)) return;  // the work-list kernel computes it
    }
    ntiles = a0.z, slot = a0.w;
    tile0 = ntiles > 0 ? dcs(p.col_idx + a0.y) : 0;  // a dense row's tiles are consecutive
  }
  const int chunks = p.G / HW;
  const int kvh = kvc / chunks;
  const int head0 = kvh * p.G + (kvc - kvh * chunks) * HW;
  const int key_end = imin(n, pos + 1);  // keys 0 .. key_end - 1 are attended (all n of them when nothing is masked)
  const int n_att = p.mode == TMX_MODE_NONE ? n : key_end;

  // ---- the query rows of the 4 heads: B operand of S^T = K Q^T, column c16 = head (columns 4..15 zero).
  // 16-bit K: k-step ks covers columns 32 ks + 8 g .. + 7 of lane group g.  fp8 K is fetched 16 bytes (16
  // columns) at a time, whose halves feed k-steps 2 j and 2 j + 1: those cover columns 64 j + 16 g + 8 h .. ----
  uint4 qf[4];
  {
    const char* qrow = reinterpret_cast<const char*>(p.q) + 2ll * ((long long)q_tok * p.q_stride_t + (long long)(head0 + (c16 & (HW - 1))) * p.q_stride_h);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int col = KV8 ? 64 * (ks >> 1) + 16 * g + 8 * (ks & 1) : 32 * ks + 8 * g;
      qf[ks] = *reinterpret_cast<const uint4*>(qrow + 2 * col);
      if (c16 >= HW) qf[ks] = make_uint4(0, 0, 0, 0);
    }
  }

  const int kv_row_bytes = KVB * p.kv_stride_t;
  const char* const kbase = reinterpret_cast<const char*>(p.k) + (long long)KVB * kvh * p.kv_stride_h;
  const char* const vbase = reinterpret_cast<const char*>(p.v) + (long long)KVB * kvh * p.kv_stride_h;
  const long long krow0 = PAGED ? 0 : (long long)dcs(p.cu_k + b);
  const int32_t* const bt = PAGED ? p.block_table + (long long)b * p.block_table_stride : nullptr;
  // two sources: keys from position ctx on (the step's own row) come from k_new / v_new
  // (out of an fp8 cache the step's own key is read as the cache will hold it: its row quantised on the way in)
  const bool two = PAGED && p.k_new != nullptr;
  const int ctx = n - 1;
  const long long new_off = two ? 2ll * ((long long)q_tok * p.new_stride_t + (long long)kvh * p.new_stride_h) : 0;

  // Raw loads of one 16-key group.  K in the matrix layout: lane = key c16, 16 bytes per load at byte 16 g of
  // every 64 of the row (16-bit: 4 loads = 4 k-steps; fp8: 2 loads).  V: lane group g holds key 4 g + i in step
  // i, lane c16 its columns 8 c16 .. + 7 (16 bytes, fp8: 8).
  constexpr int KL = KV8 ? 2 : 4;
  using VRaw = std::conditional_t<KV8, uint2, uint4>;
  struct Raw {
    uint4 k[KL];
    VRaw v[4];
  };
  // lane offsets inside a group whose 16 keys are consecutive rows
  const unsigned k_lane = (unsigned)c16 * (unsigned)kv_row_bytes + 16u * g;
  const unsigned v_lane = (unsigned)(4 * g) * (unsigned)kv_row_bytes + (unsigned)(KVB * 8 * c16);
  const int cache_keys = two ? ctx : n;  // keys below this position are rows of the cache
  // The usual group: 16 rows of one page (16-aligned groups, pages of 16 keys or more).  Buffer loads: the
  // descriptor (scalar registers) holds the group's base, the per-lane offsets never change.
  auto fetch = [&](int jb, Raw& r) {
    long long off;
    if constexpr (PAGED) off = (long long)KVB * dcs(bt + (jb >> p.page_shift)) * p.kv_stride_page + (long long)(jb & (p.page_size - 1)) * kv_row_bytes;
    else off = (krow0 + jb) * kv_row_bytes;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kbase + off), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vbase + off), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int x = 0; x < KL; ++x) r.k[x] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rk, k_lane, 64 * x, KV8 ? TMX_DC_POLICY8K : TMX_DC_POLICY16K));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (KV8) r.v[i] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rv, v_lane, i * kv_row_bytes, TMX_DC_POLICY8V));
      else r.v[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rv, v_lane, i * kv_row_bytes, TMX_DC_POLICY16V));
    }
  };
  // A group of the sequence's last tile: keys past the end read the last key (masked in the scores), the
  // step's own key comes from its row.
  auto fetch_tail = [&](int jb, Raw& r) {
    long long page_off = 0;
    if constexpr (PAGED) page_off = (long long)KVB * dcs(bt + (imin(jb, n - 1) >> p.page_shift)) * p.kv_stride_page;
    auto row_ptr = [&](const char* base, const void* fresh, int key) -> const char* {
      const int kk = imin(key, n - 1);
      if (two && kk >= ctx) return reinterpret_cast<const char*>(fresh) + new_off;
      if constexpr (PAGED) return base + page_off + (long long)(kk & (p.page_size - 1)) * kv_row_bytes;
      else return base + (krow0 + kk) * kv_row_bytes;
    };
    if constexpr (KV8) {
      // the step's own row is 16-bit: its fp8 image, byte for byte what the cache write stores (16 columns of K
      // per load, 8 columns of V)
      const bool k_fresh = two && imin(jb + c16, n - 1) >= ctx;
      const char* kr = row_ptr(kbase, p.k_new, jb + c16);
#pragma unroll
      for (int x = 0; x < KL; ++x) {
        if (k_fresh) {
          const char* src = kr + 2 * (16 * g + 64 * x);
          const uint2 q0 = quant8_fp8(*reinterpret_cast<const uint4*>(src), BF16, p.k_scale);
          const uint2 q1 = quant8_fp8(*reinterpret_cast<const uint4*>(src + 16), BF16, p.k_scale);
          r.k[x] = make_uint4(q0.x, q0.y, q1.x, q1.y);
        } else {
          r.k[x] = *reinterpret_cast<const uint4*>(kr + 16 * g + 64 * x);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool v_fresh = two && imin(jb + 4 * g + i, n - 1) >= ctx;
        const char* vr = row_ptr(vbase, p.v_new, jb + 4 * g + i);
        if (v_fresh) r.v[i] = quant8_fp8(*reinterpret_cast<const uint4*>(vr + 2 * 8 * c16), BF16, p.v_scale);
        else r.v[i] = *reinterpret_cast<const VRaw*>(vr + 8 * c16);
      }
    } else {
      const char* kr = row_ptr(kbase, p.k_new, jb + c16) + 16 * g;
#pragma unroll
      for (int x = 0; x < KL; ++x) r.k[x] = *reinterpret_cast<const uint4*>(kr + 64 * x);
#pragma unroll
      for (int i = 0; i < 4; ++i) r.v[i] = *reinterpret_cast<const VRaw*>(row_ptr(vbase, p.v_new, jb + 4 * g + i) + KVB * 8 * c16);
    }
  };

  float m_run = kDcMinusBig, l_run = 0.f;  // of head c16 over the keys of lane group g (lanes c16 < HW)
  float o[HW][8];                           // of the HW heads, columns 8 c16 .. + 7, over the keys of lane group g
#pragma unroll
  for (int hh = 0; hh < HW; ++hh)
#pragma unroll
    for (int x = 0; x < 8; ++x) o[hh][x] = 0.f;
  int src_lane[HW];  // ds_bpermute byte address of lane (head hh, group g)
#pragma unroll
  for (int hh = 0; hh < HW; ++hh) src_lane[hh] = 4 * (16 * g + hh);
  // MPV: O^T[column][head]: register c of acc[db] is column 16 db + 4 g + c of head c16, summed over all keys
  dc_f32x4 acc[8];
#pragma unroll
  for (int db = 0; db < 8; ++db) acc[db] = dc_f32x4{0.f, 0.f, 0.f, 0.f};
  // the wave's V stage: 16 key rows of 256 bytes, 32-byte chunks XOR-swizzled by the row (rows 0..7 of a
  // transposed read then hit different banks).  Write side: this lane's 16 bytes of key row 4 g + i; read side
  // (ds_read_b64_tr_b16): lane 4 q + pp of a 16-lane group addresses key row 4 g + q, columns 16 db + 4 pp ..
  const unsigned vst = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)v_stage) + (MPV ? 4096u * wave : 0u);
  unsigned vst_w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * g + i;
    vst_w[i] = vst + row * 256 + 32 * ((c16 >> 1) ^ (row & 7)) + 16 * (c16 & 1);
  }
  const int tr_row = 4 * g + (c16 >> 2);
  const unsigned vst_r = vst + tr_row * 256 + 8 * (c16 & 3);

  // 8 fp8 values (two dwords) as 8 values of the 16-bit type: value = fp8 x scale rounded once; a unit scale
  // converts exactly (every e4m3 value is a bf16 and an fp16 value)
  auto k_from_fp8 = [&](uint32_t w0, uint32_t w1, float scale) -> uint4 {
    if constexpr (UNIT) {
      if constexpr (BF16)
        return make_uint4(__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w0, 1.0f, false)),
                          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w0, 1.0f, true)),
                          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w1, 1.0f, false)),
                          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w1, 1.0f, true)));
      else
        return make_uint4(__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w0, 1.0f, false)),
                          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w0, 1.0f, true)),
                          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w1, 1.0f, false)),
                          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w1, 1.0f, true)));
    } else {
      const auto f0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w0, false), f1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w0, true);
      const auto f2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w1, false), f3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w1, true);
      return make_uint4(dc_pack<BF16>(f0[0] * scale, f0[1] * scale), dc_pack<BF16>(f1[0] * scale, f1[1] * scale),
                        dc_pack<BF16>(f2[0] * scale, f2[1] * scale), dc_pack<BF16>(f3[0] * scale, f3[1] * scale));
    }
  };

  auto compute = [&](int jb, const Raw& r) {
    // ---- scores of the 16 keys: register c of lane (head, g) is key jb + 4 g + c ----
    dc_f32x4 s = dc_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 kf;
      if constexpr (KV8) kf = (ks & 1) ? k_from_fp8(r.k[ks >> 1].z, r.k[ks >> 1].w, p.k_scale) : k_from_fp8(r.k[ks >> 1].x, r.k[ks >> 1].y, p.k_scale);
      else kf = r.k[ks];
      if constexpr (BF16) s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dc_bf16x8, kf), __builtin_bit_cast(dc_bf16x8, qf[ks]), s, 0, 0, 0);
      else s = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dc_f16x8, kf), __builtin_bit_cast(dc_f16x8, qf[ks]), s, 0, 0, 0);
    }
    const int d_hi = n_att - 1 - (jb + 4 * g);  // key offset c is attended iff c <= d_hi
    float sc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) sc[c] = (c <= d_hi) ? s[c] * p.scale_log2 : -__builtin_inff();
    float tmax = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
    if constexpr (MPV) {
      // the 4 lanes of a head (one per lane group) share one running maximum; it moves only when a group
      // exceeds it by more than the slack (rare)
      if (__builtin_amdgcn_ballot_w64(tmax > m_run + kDcSlack) != 0) {
        const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
        const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
        if (tmax > m_run + kDcSlack) {
          const float alpha = __builtin_amdgcn_exp2f(m_run - tmax);
          m_run = tmax;
          l_run *= alpha;
#pragma unroll
          for (int db = 0; db < 8; ++db) acc[db] *= alpha;
        }
      }
      float pr[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) pr[c] = __builtin_amdgcn_exp2f(sc[c] - m_run);
      l_run += (pr[0] + pr[1]) + (pr[2] + pr[3]);
      // probabilities of keys 4 g .. 4 g + 3 for head c16: the B operand of O^T += V^T P^T as they stand.  bf16 has
      // 8 significant bits: two terms (truncated high part + rounded rest), fp16 one
      uint2 p_hi, p_lo = make_uint2(0, 0);
      if constexpr (BF16) {
        const uint32_t b0 = __float_as_uint(pr[0]) & 0xffff0000u, b1 = __float_as_uint(pr[1]) & 0xffff0000u;
        const uint32_t b2 = __float_as_uint(pr[2]) & 0xffff0000u, b3 = __float_as_uint(pr[3]) & 0xffff0000u;
        p_hi = make_uint2(b1 | (b0 >> 16), b3 | (b2 >> 16));
        p_lo = make_uint2(dc_pack<true>(pr[0] - __uint_as_float(b0), pr[1] - __uint_as_float(b1)),
                          dc_pack<true>(pr[2] - __uint_as_float(b2), pr[3] - __uint_as_float(b3)));
      } else {
        p_hi = make_uint2(dc_pack<false>(pr[0], pr[1]), dc_pack<false>(pr[2], pr[3]));
      }
      // V rows to the stage as 16-bit values (an fp8 cache descaled on the way)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint4 v16;
        if constexpr (KV8) v16 = k_from_fp8(r.v[i].x, r.v[i].y, p.v_scale);
        else v16 = r.v[i];
        *(__attribute__((address_space(3))) dc_u32x4*)(uintptr_t)vst_w[i] = __builtin_bit_cast(dc_u32x4, v16);
      }
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        const dc_s16x4 vf = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) dc_s16x4*)(uintptr_t)(vst_r + 32 * (db ^ (tr_row & 7))));
        if constexpr (BF16) {
          acc[db] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf, __builtin_bit_cast(dc_s16x4, p_hi), acc[db], 0, 0, 0);
          acc[db] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf, __builtin_bit_cast(dc_s16x4, p_lo), acc[db], 0, 0, 0);
        } else {
          acc[db] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(dc_f16x4, vf), __builtin_bit_cast(dc_f16x4, p_hi), acc[db], 0, 0, 0);
        }
      }
      return;
    }
    // the running maximum moves only when a group exceeds it by more than the slack (rare): then the sums of
    // this head in the lanes that hold them (the whole lane group) are rescaled
    float alpha = 1.f;
    if (tmax > m_run + kDcSlack) {
      alpha = __builtin_amdgcn_exp2f(m_run - tmax);
      m_run = tmax;
      l_run *= alpha;
    }
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
      for (int hh = 0; hh < HW; ++hh) {
        const float ah = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane[hh], __float_as_int(alpha)));
#pragma unroll
        for (int x = 0; x < 8; ++x) o[hh][x] *= ah;
      }
    }
    float pr[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) pr[c] = __builtin_amdgcn_exp2f(sc[c] - m_run);
    l_run += (pr[0] + pr[1]) + (pr[2] + pr[3]);
    // ---- weighted sum: step i is key jb + 4 g + i, whose probability for head hh is register i of lane (hh, g) ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float vf[8];
      if constexpr (KV8) {
        const auto f0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)r.v[i].x, false), f1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)r.v[i].x, true);
        const auto f2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)r.v[i].y, false), f3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)r.v[i].y, true);
        const float raw8[8] = {f0[0], f0[1], f1[0], f1[1], f2[0], f2[1], f3[0], f3[1]};
#pragma unroll
        for (int x = 0; x < 8; x += 2) {
          if constexpr (UNIT) {
            vf[x] = raw8[x];
            vf[x + 1] = raw8[x + 1];
          } else {
            dc_unpack<BF16>(dc_pack<BF16>(raw8[x] * p.v_scale, raw8[x + 1] * p.v_scale), vf[x], vf[x + 1]);
          }
        }
      } else {
        dc_unpack<BF16>(r.v[i].x, vf[0], vf[1]);
        dc_unpack<BF16>(r.v[i].y, vf[2], vf[3]);
        dc_unpack<BF16>(r.v[i].z, vf[4], vf[5]);
        dc_unpack<BF16>(r.v[i].w, vf[6], vf[7]);
      }
#pragma unroll
      for (int hh = 0; hh < HW; ++hh) {
        const float ph = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane[hh], __float_as_int(pr[i])));
#pragma unroll
        for (int x = 0; x < 8; ++x) o[hh][x] = __builtin_fmaf(ph, vf[x], o[hh][x]);
      }
    }
  };

  // ---- this wave's tiles: tile0 + wave, + 4, ... of the piece, two 16-key groups each; the loads of a group
  // are in flight while the group before it is computed.  Only the sequence's last tile can hold keys that are
  // not plain cache rows (the end of the sequence, the step's own key): it is taken after the loop. ----
  const int my_tiles = imax(0, (ntiles - wave + 3) / 4);
  const int first_tile = tile0 + wave;
  // tiles t with 32 (first_tile + 4 t + 1) <= cache_keys
  const int full_tiles = cache_keys >> 5;  // tiles 0 .. full_tiles - 1 are whole and in the cache
  const int nt_fast = imin(my_tiles, full_tiles > first_tile ? (full_tiles - first_tile + 3) / 4 : 0);
  // NB - 1 groups are in flight while one is computed: the 16-bit cache keeps two 8-KiB groups per wave on
  // their way, the fp8 cache (half the bytes per group) four
  constexpr int NB = KV8 ? 4 : 2;
  Raw r[NB];
  Raw& ra = r[0];
  const int ng = 2 * nt_fast;
  auto gkey = [&](int i) { return 32 * (first_tile + 4 * (i >> 1)) + 16 * (i & 1); };
#pragma unroll
  for (int x = 0; x < NB - 1; ++x)
    if (x < ng) fetch(gkey(x), r[x]);
#pragma unroll 1
  for (int i = 0; i < ng; i += NB) {
#pragma unroll
    for (int x = 0; x < NB; ++x) {
      if (i + x + NB - 1 < ng) fetch(gkey(i + x + NB - 1), r[(x + NB - 1) % NB]);
      if (i + x < ng) compute(gkey(i + x), r[x]);
    }
  }
  if (nt_fast < my_tiles) {
    const int jb = 32 * (first_tile + 4 * nt_fast);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      if (jb + 16 * h < n_att || h == 0) {
        fetch_tail(jb + 16 * h, ra);
        compute(jb + 16 * h, ra);
      }
    }
  }

  // ---- merge the 16 partial results of the workgroup ----
  const int pi = wave * 4 + g;
  if (c16 < HW) {
    part_ml[pi][c16][0] = m_run;
    part_ml[pi][c16][1] = l_run;
  }
  if constexpr (MPV) {
    if (c16 < HW) {
#pragma unroll
      for (int db = 0; db < 8; ++db)
        *reinterpret_cast<float4*>(&part_o[wave][c16][16 * db + 4 * g]) = make_float4(acc[db][0], acc[db][1], acc[db][2], acc[db][3]);
    }
  } else {
#pragma unroll
    for (int hh = 0; hh < HW; ++hh)
#pragma unroll
      for (int x = 0; x < 8; ++x) part_o[pi][hh][8 * c16 + x] = o[hh][x];
  }
  __syncthreads();
#pragma unroll
  for (int hh = tid >> 6; hh < HW; hh += 4) {  // head(s) of this thread
    const int d0 = 2 * (tid & 63);     // its two columns
    float M = kDcMinusBig;
#pragma unroll
    for (int x = 0; x < 16; ++x) M = fmaxf(M, part_ml[x][hh][0]);
    float L = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const float w = __builtin_amdgcn_exp2f(part_ml[x][hh][0] - M);
      L += w * part_ml[x][hh][1];
      if constexpr (MPV) {
        if ((x & 3) == 0) {  // one accumulator per wave (its four lane groups share the maximum)
          o0 += w * part_o[x >> 2][hh][d0];
          o1 += w * part_o[x >> 2][hh][d0 + 1];
        }
      } else {
        o0 += w * part_o[x][hh][d0];
        o1 += w * part_o[x][hh][d0 + 1];
      }
    }
    const int head = head0 + hh;
    if (slot >= 0) {
      // piece of a split q-block: row 0 of the q-block's partial slot (merged by the combine pass)
      const long long prow = ((long long)slot * p.H + head) * p.block_m;
      *reinterpret_cast<float2*>(p.partial_o + prow * D + d0) = make_float2(o0, o1);
      if (d0 == 0) *reinterpret_cast<float2*>(p.partial_ml + prow * 2) = make_float2(M, L);
    } else {
      const float inv = 1.0f / L;
      const long long off = (long long)q_tok * p.out_stride_t + (long long)head * p.out_stride_h + d0;
      if (p.out_dtype == TMX_F32) *reinterpret_cast<float2*>(reinterpret_cast<float*>(p.out) + off) = make_float2(o0 * inv, o1 * inv);
      else if (p.out_dtype == TMX_BF16) *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(p.out) + off) = dc_pack<true>(o0 * inv, o1 * inv);
      else *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(p.out) + off) = dc_pack<false>(o0 * inv, o1 * inv);
    }
  }
}

// Merge launch of the fixed-grid form: one workgroup per (request, head), one thread per column.  A request with
// one piece wrote its output row itself; the others left (max, sum) and an unnormalised row per piece.
template <int D>
__global__ __launch_bounds__(D) void decode_merge_kernel(const DecodeParams p) {
  __shared__ int sm[D / 64];
  const int b = (int)blockIdx.x, head = (int)blockIdx.y, d = (int)threadIdx.x;
  const int batch = p.total_q;
  const int per = (batch + D - 1) / D;
  const int j0 = imin(batch, d * per), j1 = imin(batch, j0 + per);
  int mine = 0;
  for (int j = j0; j < j1; ++j) mine += dc_tiles_of(p.seq_lens[j]);
  const int total = dc_block_sum<D>(mine, sm, d);
  const int P = dc_piece_len(total, p.plan_units, p.plan_slots);
  const int np = (dc_tiles_of(p.seq_lens[b]) + P - 1) / P;
  if (np <= 1) return;  // (uniform over the workgroup)
  int before = 0;
  for (int j = j0; j < imin(j1, b); ++j) before += (dc_tiles_of(p.seq_lens[j]) + P - 1) / P;
  const int first = dc_block_sum<D>(before, sm, d);
  const long long row0 = (long long)first * p.H + head;  // piece x of the request: row0 + x * H
  float M = kDcMinusBig;
  for (int x = 0; x < np; ++x) M = fmaxf(M, p.partial_ml[2 * (row0 + (long long)x * p.H)]);
  float L = 0.f, o = 0.f;
  for (int x = 0; x < np; ++x) {
    const long long row = row0 + (long long)x * p.H;
    const float wgt = __builtin_amdgcn_exp2f(p.partial_ml[2 * row] - M);
    L += wgt * p.partial_ml[2 * row + 1];
    o += wgt * p.partial_o[row * D + d];
  }
  const float r = o / L;
  const long long off = (long long)b * p.out_stride_t + (long long)head * p.out_stride_h + d;
  if (p.out_dtype == TMX_F32) reinterpret_cast<float*>(p.out)[off] = r;
  else if (p.out_dtype == TMX_BF16) reinterpret_cast<uint16_t*>(p.out)[off] = (uint16_t)(dc_pack<true>(r, 0.f) & 0xffffu);
  else reinterpret_cast<uint16_t*>(p.out)[off] = (uint16_t)(dc_pack<false>(r, 0.f) & 0xffffu);
}

int launch_decode_fixed(const DecodeParams& p, bool bf16, int batch, hipStream_t s) {
  const long long grid = (long long)p.pieces_cap * p.Y;
  if (grid <= 0 || batch <= 0) return TMX_OK;
  TMX_CHECK_SUPPORTED(grid < (1ll << 31), "decode kernel: too many workgroups");
  const unsigned gsz = (unsigned)grid;
#define TMX_DECODE_FIXED(BF, K8, UN)                                                        \
  do {                                                                                      \
    if (p.hw == 8) decode_kernel<BF, true, K8, UN, 8, true, true><<<gsz, 256, 0, s>>>(p);   \
    else decode_kernel<BF, true, K8, UN, 4, true, true><<<gsz, 256, 0, s>>>(p);             \
  } while (0)
  if (p.kv_fp8) {
    const bool unit = p.k_scale == 1.0f && p.v_scale == 1.0f;
    if (unit) {
      if (bf16) TMX_DECODE_FIXED(true, true, true);
      else TMX_DECODE_FIXED(false, true, true);
    } else {
      if (bf16) TMX_DECODE_FIXED(true, true, false);
      else TMX_DECODE_FIXED(false, true, false);
    }
  } else if (bf16) {
    TMX_DECODE_FIXED(true, false, false);
  } else {
    TMX_DECODE_FIXED(false, false, false);
  }
#undef TMX_DECODE_FIXED
  TMX_CHECK_HIP(hipGetLastError());
  // (a step whose requests all fit one piece finds nothing to merge: its workgroups return after the plan)
  decode_merge_kernel<128><<<dim3((unsigned)batch, (unsigned)p.H), 128, 0, s>>>(p);
  TMX_CHECK_HIP(hipGetLastError());
  return TMX_OK;
}

int launch_decode(const DecodeParams& p, bool bf16, bool paged, hipStream_t s) {
  const long long grid = (long long)p.num_items * p.Y;
  if (grid <= 0) return TMX_OK;
  TMX_CHECK_SUPPORTED(grid < (1ll << 31), "decode kernel: too many work items");
  const unsigned gsz = (unsigned)grid;
  // (instantiations: cache type x scales x heads per workgroup x where the weighted sum runs)
  // The weighted sum on the matrix cores is the default: 64 x 8192 decode, 16 / 2 heads: 95 -> 83 us (16-bit
  // cache), 92 -> 77 us (fp8); 32 / 8 heads: 337 / 185 us either way.  TMX_ATTN_DECODE_VALU selects the vector
  // units (kept for comparison and as the second implementation the parity tests run against).
  const bool mpv = p.valu_sum == 0;
#define TMX_DECODE_LAUNCH(BF, PG, K8, UN)                                             \
  do {                                                                               \
    if (p.hw == 8 && mpv) decode_kernel<BF, PG, K8, UN, 8, true><<<gsz, 256, 0, s>>>(p);   \
    else if (p.hw == 8) decode_kernel<BF, PG, K8, UN, 8, false><<<gsz, 256, 0, s>>>(p);     \
    else if (mpv) decode_kernel<BF, PG, K8, UN, 4, true><<<gsz, 256, 0, s>>>(p);            \
    else decode_kernel<BF, PG, K8, UN, 4, false><<<gsz, 256, 0, s>>>(p);                    \
  } while (0)
  if (p.kv_fp8) {
    const bool unit = p.k_scale == 1.0f && p.v_scale == 1.0f;
    if (unit) {
      if (bf16) TMX_DECODE_LAUNCH(true, true, true, true);
      else TMX_DECODE_LAUNCH(false, true, true, true);
    } else {
      if (bf16) TMX_DECODE_LAUNCH(true, true, true, false);
      else TMX_DECODE_LAUNCH(false, true, true, false);
    }
  } else if (bf16) {
    if (paged) TMX_DECODE_LAUNCH(true, true, false, false);
    else TMX_DECODE_LAUNCH(true, false, false, false);
  } else {
    if (paged) TMX_DECODE_LAUNCH(false, true, false, false);
    else TMX_DECODE_LAUNCH(false, false, false, false);
  }
#undef TMX_DECODE_LAUNCH
  TMX_CHECK_HIP(hipGetLastError());
  return TMX_OK;
}

}  // namespace tmx

// ---------------------------------------------------------------------------------------------------
// tmx_decode_attention: a step of one-row requests with no index and nothing sized on the host from the
// requests' lengths (see include/trianglemix_hip.h).
namespace tmx {

static int decode_cus(int* out) {
  static std::atomic<int> cache[64];
  int dev = 0;
  TMX_CHECK_HIP(hipGetDevice(&dev));
  int cus = (dev >= 0 && dev < 64) ? cache[dev].load(std::memory_order_relaxed) : 0;
  if (cus == 0) {
    TMX_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (dev >= 0 && dev < 64) cache[dev].store(cus, std::memory_order_relaxed);
  }
  *out = cus;
  return TMX_OK;
}

static bool dc_aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; }

}  // namespace tmx

namespace tmx {
// (K/V head chunks the plan counts per piece, workgroup slots of the chip): 4 query heads per unit out of a 16-bit
// cache; 8 when the group allows out of an fp8 cache (half the bytes per key, so twice the pieces keep the chip busy
// -- the split measured fastest for index-driven decode steps, 64 x 8192: 16 / 2 heads 54 us in 64-tile pieces against
// 78 us in 128-tile ones)
static void decode_plan_shape(int num_heads, int num_kv_heads, bool fp8, int num_cus, int* units, int* slots) {
  const int G = num_heads / num_kv_heads;
  const int hw = (fp8 && G % 8 == 0) ? 8 : 4;
  *units = num_kv_heads * (G / hw > 0 ? G / hw : 1);
  *slots = 2 * num_cus;  // two workgroups are resident per CU
}
}  // namespace tmx

extern "C" int64_t tmx_decode_workspace_rows(int batch, int num_heads, int num_kv_heads, int kv_dtype, int num_cus) {
  using namespace tmx;
  if (batch <= 0 || num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads != 0) return 0;
  if (num_cus <= 0 && decode_cus(&num_cus) != TMX_OK) return -1;
  int units = 1, slots = 1;
  decode_plan_shape(num_heads, num_kv_heads, kv_dtype == TMX_FP8_E4M3, num_cus, &units, &slots);
  const long long pieces = (slots + units - 1) / units + batch;  // bound of the step's pieces (see DecodeParams)
  return (int64_t)pieces * num_heads;
}

extern "C" int tmx_decode_attention(const tmx_decode_args_t* a, void* stream) {
  using namespace tmx;
  TMX_CHECK_ARG(a != nullptr, "tmx_decode_attention: args is NULL");
  TMX_CHECK_ARG(a->struct_size == sizeof(tmx_decode_args_t), "tmx_decode_attention: struct_size %u != %zu (ABI mismatch)",
                a->struct_size, sizeof(tmx_decode_args_t));
  TMX_CHECK_SUPPORTED(a->io_dtype == TMX_BF16 || a->io_dtype == TMX_F16,
                      "tmx_decode_attention: q dtype %d not supported (bf16 and fp16 are)", a->io_dtype);
  TMX_CHECK_SUPPORTED(a->out_dtype == a->io_dtype || a->out_dtype == TMX_F32,
                      "tmx_decode_attention: out dtype %d must equal the input dtype or be fp32", a->out_dtype);
  TMX_CHECK_ARG(a->num_heads > 0 && a->num_kv_heads > 0 && a->num_heads % a->num_kv_heads == 0,
                "tmx_decode_attention: num_heads %d must be a positive multiple of num_kv_heads %d", a->num_heads,
                a->num_kv_heads);
  const int G = a->num_heads / a->num_kv_heads;
  TMX_CHECK_SUPPORTED(a->head_dim == 128 && G % 4 == 0,
                      "tmx_decode_attention: head_dim %d / %d query heads per K/V head not covered (head size 128, "
                      "groups that are multiples of 4; other shapes: tmx_attention with an index)", a->head_dim, G);
  TMX_CHECK_ARG(a->scale > 0.f && a->scale < 3.0e38f, "tmx_decode_attention: scale %g must be positive and finite",
                (double)a->scale);
  TMX_CHECK_ARG(a->batch >= 0, "tmx_decode_attention: negative batch");
  if (a->batch == 0) return TMX_OK;
  TMX_CHECK_ARG(a->q && a->out && a->k && a->v && a->seq_lens && a->block_table, "tmx_decode_attention: NULL tensor");
  TMX_CHECK_SUPPORTED(a->page_size >= 16 && (a->page_size & (a->page_size - 1)) == 0,
                      "tmx_decode_attention: page_size %d not supported (powers of two >= 16 are)", a->page_size);
  TMX_CHECK_ARG(dc_aligned16(a->q) && dc_aligned16(a->k) && dc_aligned16(a->v) && dc_aligned16(a->out),
                "tmx_decode_attention: q / k / v / out must be 16-byte aligned");
  const int64_t lim = 0x7FFFFFFF;
  TMX_CHECK_SUPPORTED(a->q_stride_t >= 0 && a->q_stride_t <= lim && a->q_stride_h >= 0 && a->q_stride_h <= lim &&
                          a->out_stride_t >= 0 && a->out_stride_t <= lim && a->out_stride_h >= 0 && a->out_stride_h <= lim &&
                          a->kv_stride_page >= 0 && a->kv_stride_page <= lim && a->kv_stride_t >= 0 &&
                          a->kv_stride_t < (1ll << 24) && a->kv_stride_h >= 0 && a->kv_stride_h <= lim &&
                          a->block_table_stride >= 0 && a->block_table_stride <= lim,
                      "tmx_decode_attention: strides must be non-negative and below 2^31 elements (K/V rows: 2^24)");
  TMX_CHECK_ARG(a->q_stride_t % 8 == 0 && a->q_stride_h % 8 == 0 && a->out_stride_t % 2 == 0 && a->out_stride_h % 2 == 0 &&
                    a->kv_stride_t % 8 == 0 && a->kv_stride_h % 8 == 0 && a->kv_stride_page % 8 == 0,
                "tmx_decode_attention: strides must be multiples of 8 elements");
  const bool fp8 = a->kv_dtype == TMX_FP8_E4M3;
  TMX_CHECK_SUPPORTED(a->kv_dtype == 0 || a->kv_dtype == a->io_dtype || fp8,
                      "tmx_decode_attention: kv_dtype %d not supported (0 / the q dtype, or TMX_FP8_E4M3)", a->kv_dtype);
  if (fp8) {
    TMX_CHECK_ARG(a->k_scale > 0.f && a->v_scale > 0.f && a->k_scale < 3.0e38f && a->v_scale < 3.0e38f,
                  "tmx_decode_attention: k_scale / v_scale must be positive and finite");
    TMX_CHECK_ARG(a->kv_stride_t % 16 == 0 && a->kv_stride_h % 16 == 0 && a->kv_stride_page % 16 == 0,
                  "tmx_decode_attention: fp8 cache strides must be multiples of 16 bytes");
  }
  const bool two_src = a->k_new != nullptr || a->v_new != nullptr;
  if (two_src) {
    TMX_CHECK_ARG(a->k_new != nullptr && a->v_new != nullptr, "tmx_decode_attention: k_new and v_new go together");
    TMX_CHECK_ARG(dc_aligned16(a->k_new) && dc_aligned16(a->v_new) && a->new_stride_t % 8 == 0 && a->new_stride_h % 8 == 0 &&
                      a->new_stride_t >= 0 && a->new_stride_h >= 0 && a->new_stride_t < (1ll << 24) && a->new_stride_h <= lim,
                  "tmx_decode_attention: k_new / v_new must be 16-byte aligned with strides that are multiples of 8 elements");
  }
  TMX_CHECK_ARG(a->cache_slots == nullptr || two_src, "tmx_decode_attention: cache_slots needs k_new / v_new (the rows to store)");
  int cus = 0;
  {
    const int rc = decode_cus(&cus);
    if (rc != TMX_OK) return rc;
  }
  int units = 1, slots = 1;
  decode_plan_shape(a->num_heads, a->num_kv_heads, fp8, cus, &units, &slots);
  const long long pieces = (slots + units - 1) / units + a->batch;
  TMX_CHECK_ARG(a->partial_o != nullptr && a->partial_ml != nullptr && dc_aligned16(a->partial_o),
                "tmx_decode_attention: the work area is missing (tmx_decode_workspace_rows)");
  TMX_CHECK_ARG(a->workspace_rows >= pieces * a->num_heads,
                "tmx_decode_attention: the work area holds %lld rows, this batch size needs %lld (tmx_decode_workspace_rows)",
                (long long)a->workspace_rows, pieces * a->num_heads);
  DecodeParams d{};
  d.q = a->q;
  d.out = a->out;
  d.k = a->k;
  d.v = a->v;
  d.block_table = a->block_table;
  d.partial_o = a->partial_o;
  d.partial_ml = a->partial_ml;
  d.k_new = a->k_new;
  d.v_new = a->v_new;
  d.cache_slots = reinterpret_cast<const long long*>(a->cache_slots);
  d.new_stride_t = (int)a->new_stride_t;
  d.new_stride_h = (int)a->new_stride_h;
  d.total_q = a->batch;
  d.q_stride_t = (int)a->q_stride_t;
  d.q_stride_h = (int)a->q_stride_h;
  d.out_stride_t = (int)a->out_stride_t;
  d.out_stride_h = (int)a->out_stride_h;
  d.kv_stride_page = (int)a->kv_stride_page;
  d.kv_stride_t = (int)a->kv_stride_t;
  d.kv_stride_h = (int)a->kv_stride_h;
  d.block_table_stride = (int)a->block_table_stride;
  d.page_size = a->page_size;
  d.page_shift = __builtin_ctz(a->page_size);
  d.H = a->num_heads;
  d.Hkv = a->num_kv_heads;
  d.G = G;
  d.valu_sum = 0;
  d.hw = G % 8 == 0 ? 8 : 4;  // query heads per workgroup (all of them share the K/V loads), as in tmx_attention
  d.Y = a->num_kv_heads * (G / d.hw);
  d.block_m = 1;
  d.num_items = 0;
  d.mode = TMX_MODE_CAUSAL;  // the row sits at position n - 1: every key of its sequence is attended
  d.last = 0;
  d.tri_min_len = 0;
  d.out_dtype = a->out_dtype;
  d.scale_log2 = a->scale * kLog2e;
  d.kv_fp8 = fp8 ? 1 : 0;
  d.k_scale = fp8 ? a->k_scale : 1.f;
  d.v_scale = fp8 ? a->v_scale : 1.f;
  d.seq_lens = a->seq_lens;
  d.pieces_cap = (int)pieces;
  d.plan_units = units;
  d.plan_slots = slots;
  return launch_decode_fixed(d, a->io_dtype == TMX_BF16, a->batch, (hipStream_t)stream);
}
