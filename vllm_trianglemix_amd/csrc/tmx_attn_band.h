// Band workgroups of the attention kernel: the q-blocks of a long triangle request in canonical position (see
// band_qblock in tmx_attn.h), a run of consecutive q-blocks per workgroup over a rolling window of keys in LDS.
// A device function template: it is inlined into attn_kernel (tmx_attn_kernel.hip), whose launch carries both roles.
#pragma once
#include "tmx_attn.h"

namespace tmx {

// bf16 launches: the V rows are turned into fp16 inside LDS by the waves that fetched them (every value of the
// bf16 range that fp16 holds converts exactly; larger ones saturate at +-65504, smaller ones than 2^-24 vanish)
// and multiplied by ONE fp16 term of the probabilities (11 bits) instead of two bf16 terms (band workgroups: 34
// matrix instructions per q-block and wave instead of 50; work list: 24 per tile instead of 40), and the row
// sums come from the matrix cores as well (a ones operand in V's place: they add up the same rounded
// probabilities as the weighted sum, so that rounding a probability only shifts weight between keys).

// kvc: the workgroup's K/V head chunk; [qb, qb_end): its run of q-blocks.
// FP8: K and V are an fp8 (e4m3fn) paged cache.  The wave that fetches eight rows of a half (1 KiB of fp8 bytes)
// lands them in the first half of those rows' own place in the window and, one q-block ahead of their use like the
// V conversion of bf16 launches, expands them there into the 16-bit image the arithmetic reads: value x scale
// rounded once to the 16-bit type of q (the descale of the reference's attention kernels,
// vllm/v1/attention/backends/rocm_attn.py:421-422); V of a bf16 launch goes on from that bf16 value to fp16 (exact).
// BUILD: the index is being built by workgroup 0 of this launch meanwhile: nothing of it is read here -- the
// request and the first row of a q-block (qb_meta in the index) are derived from the batch description instead,
// request by request (the launch takes this form for batches of a few requests only).
template <typename TR, int D, bool PAGED, bool FP8 = false, bool BUILD = false>
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
    int b, row0;
    if constexpr (BUILD) {
      int before = 0;  // q-blocks of the requests ahead of b (32 rows each on this path)
      b = 0;
      for (;;) {
        const int nqb = (sload(p.cu_q + b + 1) - sload(p.cu_q + b) + 31) >> 5;
        if (qb < before + nqb || b >= p.batch - 1) break;
        before += nqb;
        ++b;
      }
      row0 = (qb - before) * 32;
    } else {
      b = sload(p.qb_meta + 2 * qb);
      row0 = sload(p.qb_meta + 2 * qb + 1);
    }
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
      const long long off2 = has2 ? half_offset(16 * (h0 + 6 + hf)) : 0;
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
      if (p.out_dtype == TMX_F32) {
        // (wave-uniform base + a 32-bit lane offset computed here: nothing of it stays live across the loop)
        const int l = opaque(lane);
        char* const ob = reinterpret_cast<char*>(p.out) + 4 * orows;
        const unsigned lo = 4u * (unsigned)((l & 15) * p.out_stride_t + 4 * (l >> 4));
#pragma unroll
        for (int db = 0; db < DB; ++db)
          store_stream16(ob + lo + 64 * db, make_float4(acc[db][0] * inv, acc[db][1] * inv, acc[db][2] * inv, acc[db][3] * inv));
      } else if (ROWST) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        // two passes, one per half of the row (ROWB / 2 bytes = DB / 4 pieces per lane), each in two rounds of
        // 2 RPP rows: the stage holds 2 RPP half rows; a store instruction covers 2 RPP rows x one half row
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

}  // namespace tmx
