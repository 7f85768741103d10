// Shared host/device definitions of the TriangleMix HIP library (gfx950).
//
// The closed forms below are the single statement of the mask pattern used by
// every kernel and by the host-side sizing code.  They follow the reference
// predicate (vllm/attention/trianglemix_attention.py:82-89, 126-129, 192-196):
//   allowed(i, j) = j <= i && (j <= si || i - j <= sl || N - i < last)
// N is the length of the whole sequence the mask is meant for: the number of keys n, or, while a
// prompt still arrives in chunks, the length it will have (n_full >= n).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trianglemix_hip.h"

#define TMX_HD __host__ __device__ __forceinline__

namespace tmx {

constexpr int kBlockN = 32;     // keys per tile of the attention kernel
constexpr int kWavesPerWG = 4;  // waves of one attention workgroup
constexpr int kRowBlocksPerWave = 2;  // 16-row blocks of one query head each wave computes
constexpr float kLog2e = 1.4426950408889634f;

TMX_HD int ceil_div(int a, int b) { return (a + b - 1) / b; }
TMX_HD int imin(int a, int b) { return a < b ? a : b; }
TMX_HD int imax(int a, int b) { return a > b ? a : b; }

TMX_HD int gcd_int(int a, int b) {
  while (b) {
    int t = a % b;
    a = b;
    b = t;
  }
  return a;
}

// Heads of one K/V group handled by one workgroup, and 16-row blocks per workgroup.
TMX_HD int heads_per_wg(int group) { return gcd_int(group, kWavesPerWG); }
TMX_HD int block_m_for_group(int group) { return 16 * kRowBlocksPerWave * (kWavesPerWG / heads_per_wg(group)); }

// The triangle pattern applies to requests longer than the gate only.
TMX_HD int effective_mode(int mode, int n, int tri_min_len) {
  return (mode == TMX_MODE_TRIANGLE && n <= tri_min_len) ? TMX_MODE_CAUSAL : mode;
}

TMX_HD bool allowed(int mode, int i, int j, int n_full, int si, int sl, int last) {
  if (mode == TMX_MODE_NONE) return true;
  if (j > i) return false;
  if (mode == TMX_MODE_CAUSAL) return true;
  return j <= si || (i - j) <= sl || (n_full - i) < last;
}

// Key-tile set of the query positions i_min..i_max (a q-block, or one wave's
// rows) of a sequence of n keys: at most two inclusive, ascending, disjoint
// tile ranges [a0, a1] and [b0, b1]; a range with hi < lo is empty.
struct TileRanges {
  int a0, a1, b0, b1;
  TMX_HD int count() const { return imax(0, a1 - a0 + 1) + imax(0, b1 - b0 + 1); }
  TMX_HD int tile_at(int e) const {
    const int la = imax(0, a1 - a0 + 1);
    return e < la ? a0 + e : b0 + (e - la);
  }
  TMX_HD bool contains(int t) const { return (t >= a0 && t <= a1) || (t >= b0 && t <= b1); }
};

// bn_shift: log2(bn) when bn is a power of two (every quotient below then is a shift of a non-negative
// number: a general integer division costs the GPU some forty instructions), -1 otherwise.
TMX_HD TileRanges tile_ranges(int mode, int i_min, int i_max, int n, int n_full, int si, int sl, int last, int bn,
                              int bn_shift = -1) {
  TileRanges r{0, -1, 0, -1};
  auto tile_of = [&](int key) { return bn_shift >= 0 ? key >> bn_shift : key / bn; };  // key >= 0
  const int nt = n > 0 ? tile_of(n - 1) + 1 : 0;
  if (nt <= 0 || i_max < i_min) return r;
  if (mode == TMX_MODE_NONE) {
    r.a1 = nt - 1;
    return r;
  }
  const int t_diag = tile_of(i_max);
  if (mode == TMX_MODE_CAUSAL || (n_full - i_max) < last) {
    r.a1 = t_diag;
    return r;
  }
  const int a_hi = imin(i_max, si);
  const bool has_sink = a_hi >= 0;
  if (has_sink) r.a1 = tile_of(a_hi);
  if (sl >= 0) {
    const int b_lo = tile_of(imax(0, i_min - sl));
    if (has_sink && b_lo <= r.a1 + 1) {
      r.a1 = imax(r.a1, t_diag);
    } else if (has_sink) {
      r.b0 = b_lo;
      r.b1 = t_diag;
    } else {
      r.a0 = b_lo;
      r.a1 = t_diag;
    }
  }
  return r;
}

// Host-side sizing walks a request q-block by q-block; inside a triangle request that would be thousands of
// identical answers.  Number (>= 1) of consecutive q-blocks, from q-block k on, that list exactly as many tiles
// as q-block k: past the first few q-blocks (whose window still reaches the sink tiles) and before the trailing
// dense rows every full q-block lists sink tiles + window tiles, and with block_m a multiple of block_n both
// ends of the window move by the same whole number of tiles from one q-block to the next.
TMX_HD int same_count_run(int mode, int k, int ctx, int q_len, int n_full, int si, int sl, int last, int bm, int bn) {
  if (mode != TMX_MODE_TRIANGLE || bm % bn != 0) return 1;
  const int full_blocks = q_len / bm;  // q-blocks 0 .. full_blocks - 1 have all bm rows
  if (k >= full_blocks) return 1;
  const int i_min = ctx + k * bm, i_max = i_min + bm - 1;
  if ((n_full - i_max) < last) return 1;    // trailing dense rows
  if (si >= 0 && i_max < si) return 1;      // still inside the sink columns
  if (sl >= 0) {
    if (i_min - sl < 0) return 1;           // window clamped at key 0
    if (si >= 0 && (i_min - sl) / bn <= si / bn + 1) return 1;  // window tiles touch the sink tiles
  }
  // the last q-block with i_max <= n_full - last, i.e. (k' + 1) * bm <= n_full - last - ctx + 1
  const long long lim = ((long long)n_full - last - ctx + 1) / bm - 1;
  const long long k_end = lim < full_blocks - 1 ? lim : full_blocks - 1;
  return k_end >= k ? (int)(k_end - k + 1) : 1;
}

#if defined(__HIPCC__)
// Eight 16-bit values (bf16 or fp16, 16 bytes) as eight fp8 (e4m3fn) bytes of an fp8 K/V cache: fp8(x / scale), round
// to nearest even, finite values saturating at +-448, NaN staying NaN (the reference's scaled conversion,
// csrc/quantization/fp8/amd/quant_utils.cuh, as reshape_and_cache_flash applies it, csrc/cache_kernels.cu:282-355).
// ONE definition for the stand-alone cache write and for the kernels that write the cache themselves: their bytes
// must be the same.
__device__ __forceinline__ uint2 quant8_fp8(uint4 in, bool bf16, float scale) {
  const uint32_t w[4] = {in.x, in.y, in.z, in.w};
  float f[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (bf16) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    } else {
      typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
      const h2_t h = __builtin_bit_cast(h2_t, w[i]);
      f[2 * i] = (float)h[0];
      f[2 * i + 1] = (float)h[1];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = fminf(fmaxf(f[i] / scale, -448.f), 448.f);
  int a = 0, b = 0;
  a = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], a, false);
  a = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], a, true);
  b = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], b, false);
  b = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], b, true);
  return make_uint2((uint32_t)a, (uint32_t)b);
}
#endif

void set_error(const char* fmt, ...);

#define TMX_CHECK_ARG(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      tmx::set_error(__VA_ARGS__);    \
      return TMX_ERR_INVALID_ARG;     \
    }                                 \
  } while (0)

#define TMX_CHECK_SUPPORTED(cond, ...) \
  do {                                 \
    if (!(cond)) {                     \
      tmx::set_error(__VA_ARGS__);     \
      return TMX_ERR_UNSUPPORTED;      \
    }                                  \
  } while (0)

#define TMX_CHECK_HIP(expr)                                                        \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      tmx::set_error("%s failed: %s", #expr, hipGetErrorString(e_));               \
      return TMX_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

}  // namespace tmx
