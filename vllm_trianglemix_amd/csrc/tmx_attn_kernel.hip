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
#include "tmx_attn.h"
#include "tmx_attn_band.h"

namespace tmx {

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
// BUILD: the launch also builds the index (TMX_ATTN_BUILD_INDEX).  Workgroup 0 -- dispatched first, and the whole
// grid is resident, so nobody it waits for and nobody waiting for it can be left out -- runs the single-workgroup
// build before its own work and publishes an epoch; the other work-list workgroups sleep until they read that epoch.
template <typename TR, int D, int NS, bool PAGED, bool HAS_BIAS, bool FP8, bool BUILD = false>
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

  if constexpr (BUILD) {
    // (ahead of everything else: nothing computed for the attention is live across the build)
    static_assert(sizeof(IndexScanShared) <= sizeof(lds), "the build's LDS must fit the attention's");
    if (blockIdx.x == 0) {
      index_scan_body<64 * kWaves>(p.ix, *reinterpret_cast<IndexScanShared*>(lds));
      // The barrier waits for every wave's stores to have reached this XCD's L2; ONE thread then writes that L2
      // back (the release of its device-scope store) and publishes the epoch.  (A device-scope fence per thread
      // would write back and invalidate the whole L2 over and over under the band workgroups' feet.)
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(p.ix_flag, p.ix_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else if ((int)blockIdx.x < p.n_generic) {
      // (a bounded wait, about two seconds: a workgroup that never sees the epoch leaves without touching anything
      // rather than keeping the GPU; the build takes ten to twenty microseconds)
      int* const seen = reinterpret_cast<int*>(lds);
      if (threadIdx.x == 0) {
        int ok = 0;
        for (int spin = 0; spin < (1 << 22); ++spin) {
          if (__hip_atomic_load(p.ix_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.ix_epoch) {
            ok = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(16);
        }
        *seen = ok;
      }
      __syncthreads();
      const int ok = *seen;
      __syncthreads();  // (the LDS word is free again)
      if (!ok) return;
      // No cache invalidation on this side: nothing of the index has been read by this launch before the epoch is
      // seen (the band workgroups of this form do not read it), and a launch starts with the caches clean of other launches'
      // lines -- so no cache on this workgroup's path can hold an older copy of what workgroup 0 wrote.
    }
    __builtin_amdgcn_sched_barrier(0);
  }

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
      band_body<TR, D, PAGED, FP8, BUILD>(p, lds_base, bl.kvc, qb0, imin(p.band_tail0, qb0 + p.band_rl), true);
      return;
    }
  }
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
    // every LDS read of the entries about to be overwritten has returned before the barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
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
      const bool compute = w_has_rows;
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
            *reinterpret_cast<float4*>(po + 16 * db) = make_float4(acc[db][0], acc[db][1], acc[db][2], acc[db][3]);
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
}

// Merge the pieces of every split q-block: out = sum_p O_p 2^(m_p - M) / sum_p l_p 2^(m_p - M).
// One workgroup per (split q-block, head, 8 rows): a thread owns 4 output columns of one row.  The pass is a chain
// of dependent trips to memory, so it is kept short: the (max, sum) pairs AND the partial rows of up to 16 pieces
// are asked for in one go, together with the request's bounds (the partial buffers hold block_m rows per slot, so

// Persistent grid: as many workgroups as stay resident (LDS and registers allow a fixed number per
// CU), rounded down to a multiple of the units per item so that a workgroup keeps one K/V head.
// The resident count of an instantiation is asked of the runtime once per device and kept.
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

template <typename TR, int D, bool PAGED, bool HAS_BIAS, bool FP8 = false, bool BUILD = false>
static int launch_attn(const AttnParams& p_in, hipStream_t s) {
  auto kernel = attn_kernel<TR, D, FP8 ? kRingSlotsFp8 : kRingSlots, PAGED, HAS_BIAS, FP8, BUILD>;
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
    p.band_tail0 = p.num_qblocks;  // (the band runs cover every q-block; the work-list workgroups take none of them)
    p.band_rls = 0;
    p.walk_shift = 0;              // (pieces walk their tile list front to back)
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
static int launch_for(const AttnParams& p, bool paged, bool has_bias, bool fp8, hipStream_t s) {
  if (p.ix_build) {
    // (the entry point asks attention_kernel_builds_index first)
    if constexpr (D == 128) {
      if (has_bias || fp8) return TMX_ERR_UNSUPPORTED;
      return paged ? launch_attn<TR, 128, true, false, false, true>(p, s) : launch_attn<TR, 128, false, false, false, true>(p, s);
    } else {
      return TMX_ERR_UNSUPPORTED;
    }
  }
  if (fp8) {
    if constexpr (D == 128) return launch_attn<TR, 128, true, false, true>(p, s);
    else return TMX_ERR_UNSUPPORTED;
  }
  if (paged) return has_bias ? launch_attn<TR, D, true, true>(p, s) : launch_attn<TR, D, true, false>(p, s);
  return has_bias ? launch_attn<TR, D, false, true>(p, s) : launch_attn<TR, D, false, false>(p, s);
}

bool attention_kernel_builds_index(int head_dim, bool has_bias, bool fp8) { return head_dim == 128 && !has_bias && !fp8; }

int launch_attention_kernel(const AttnParams& p, bool bf16, int head_dim, bool paged, bool has_bias, bool fp8, hipStream_t s) {
  if (bf16) return head_dim == 64 ? launch_for<Bf16Traits, 64>(p, paged, has_bias, fp8, s) : launch_for<Bf16Traits, 128>(p, paged, has_bias, fp8, s);
  return head_dim == 64 ? launch_for<F16Traits, 64>(p, paged, has_bias, fp8, s) : launch_for<F16Traits, 128>(p, paged, has_bias, fp8, s);
}

}  // namespace tmx
