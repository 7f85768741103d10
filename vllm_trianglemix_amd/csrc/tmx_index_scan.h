// The single-workgroup form of the index build (up to 1024 q-blocks) as a device function, so that it can run as a
// launch of its own (index_scan_kernel, 1024 threads) or as the first thing workgroup 0 of the attention launch does
// (512 threads), and the pieces both share with the blocked form (tmx_index.hip).
#pragma once
#include "tmx_common.h"

#ifdef TMX_INDEX_STAMPS
__device__ long long g_idx_stamps[16];
#define ISTAMP(i) do { if (threadIdx.x == 0) g_idx_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ISTAMP(i)
#endif

namespace tmx {

struct Int4 {
  int x, y, z, w;
};
__device__ __forceinline__ Int4 operator+(const Int4& a, const Int4& b) {
  return Int4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
}

// Inclusive scan over the 64 lanes of a wave by data-parallel-primitive moves (no LDS, no lane arithmetic): four
// shifts inside each row of 16 lanes (lanes without a source read 0), then the last lane of rows 0 and 2 is added to
// rows 1 and 3, then lane 31 to lanes 32..63.  (Through __shfl_up, which goes through the LDS crossbar with an address
// per lane, the two scans of the index build took 6,000 cycles each with 16 waves at it.)
__device__ __forceinline__ int wave_inclusive_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
  return x;
}


// Exclusive scan of one Int4 per thread over a workgroup of NT threads.
// Returns the exclusive prefix; `total` receives the workgroup sum.
template <int NT>
__device__ __forceinline__ Int4 block_exclusive_scan(Int4 v, Int4* wave_sums /* LDS [NT / 64] */, Int4& total) {
  static_assert(NT % 64 == 0 && NT / 64 <= 16, "at most 16 waves: their sums are scanned inside one row of 16 lanes");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const Int4 inc{wave_inclusive_scan(v.x), wave_inclusive_scan(v.y), wave_inclusive_scan(v.z), wave_inclusive_scan(v.w)};
  if (lane == 63) wave_sums[wave] = inc;
  __syncthreads();
  // Every wave scans the waves' sums itself: lane i < NT / 64 holds wave i's sum, four shifts inside the first row of
  // 16 lanes give the inclusive prefixes, and two lane reads pick this wave's base and the total (one LDS read and a
  // dozen lane moves where a loop over the 16 sums took 16 LDS reads and 64 additions per thread).
  Int4 ws{0, 0, 0, 0};
  if (lane < NT / 64) ws = wave_sums[lane];
  auto row_scan = [](int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);  // row_shr:8
    return x;
  };
  const Int4 pre{row_scan(ws.x), row_scan(ws.y), row_scan(ws.z), row_scan(ws.w)};  // lane i: sums of waves 0 .. i
  const int prev = wave > 0 ? wave - 1 : 0;
  Int4 base{__builtin_amdgcn_readlane(pre.x, prev), __builtin_amdgcn_readlane(pre.y, prev),
            __builtin_amdgcn_readlane(pre.z, prev), __builtin_amdgcn_readlane(pre.w, prev)};
  if (wave == 0) base = Int4{0, 0, 0, 0};
  total = Int4{__builtin_amdgcn_readlane(pre.x, NT / 64 - 1), __builtin_amdgcn_readlane(pre.y, NT / 64 - 1),
               __builtin_amdgcn_readlane(pre.z, NT / 64 - 1), __builtin_amdgcn_readlane(pre.w, NT / 64 - 1)};
  __syncthreads();
  Int4 exc{inc.x - v.x, inc.y - v.y, inc.z - v.z, inc.w - v.w};
  return base + exc;
}

struct IndexParams {
  const int32_t* cu_q;
  const int32_t* seq_lens;
  const int32_t* full_lens;  // [batch] or NULL: length each sequence will have once its prompt is complete
  int32_t* row_ptr;
  int32_t* col_idx;
  int32_t* qb_meta;
  int32_t* part_ptr;
  int32_t* items;
  int32_t* split_qb;
  int32_t* totals;
  int32_t* tile_pages;        // [nnz][2] or NULL
  const int32_t* block_table;  // for tile_pages
  long long block_table_stride;
  int page_shift;
  int32_t* seq_qb_ptr;  // scratch [batch + 1]
  int32_t* piece_ptr;   // scratch [R + 1]
  int32_t* split_pos;   // scratch [R + 1]: blocked pass 1, position of a split q-block among its block's
  int32_t* blk_tot;     // scratch [8 x blocks]: blocked pass 1, per block (tiles, pieces, partials, split q-blocks, longest list, R)
  int blocked;          // pass 1 ran per block of 1024 q-blocks (row_ptr / piece_ptr / part_ptr hold prefixes inside the block)
  int batch, block_m, block_n, mode, chunk, si, sl, last, tri_min_len;
  int cap_qblocks, cap_nnz, cap_items, cap_split;
  int fused;  // pass 2 inside the scan launch (host: cap_qblocks <= 1024)
  int bn_shift, bm_shift;  // log2 of block_n / block_m when powers of two, else -1
  int dense_rows_from;     // > 0: requests with at least this many query rows own no q-blocks (tmx_index_t.dense_rows_from)
};

// Position of a q-block's first work item.  The list starts with the pieces of all split q-blocks (the
// long jobs: q-blocks descending, pieces ascending) and goes on with the unsplit q-blocks, descending:
// whatever the batch, every workgroup that gets a long piece gets it first.
// total_pieces / total_parts: items and partial slots of the whole index; piece_prefix / part_prefix: of the
// q-blocks before this one; pieces: its own.
__device__ __forceinline__ int first_item_of(int total_pieces, int total_parts, int piece_prefix, int part_prefix,
                                             int pieces) {
  if (pieces > 1) return total_parts - part_prefix - pieces;
  const int unsplit_before = piece_prefix - part_prefix;
  return total_pieces - unsplit_before - 1;  // = total_parts + (unsplit q-blocks after this one)
}

// Pass 2 for one q-block: entries e0, e0 + estep, ... of its tile list and pieces pc0, pc0 + estep, ...
// The physical pages of a tile's two 16-key halves (keys past the end share the last page).
__device__ __forceinline__ int2 tile_pages_of(const IndexParams& p, int b, int n, int tile) {
  const int32_t* bt = p.block_table + (long long)b * p.block_table_stride;
  return make_int2(bt[(tile * p.block_n) >> p.page_shift], bt[imin(tile * p.block_n + 16, n - 1) >> p.page_shift]);
}

// `pre`: the pages of entries e0, e0 + estep, e0 + 2 estep, e0 + 3 estep, looked up by the caller earlier (or NULL).
__device__ __forceinline__ void fill_qblock(const IndexParams& p, int qb, int b, int r0, int n, int ctx, int q_row0,
                                            const TileRanges& tr, int cnt, int start, int first_item, int slot0,
                                            int e0, int estep, const int2* pre = nullptr) {
  // four entries at a time: their block-table lookups are all in flight before the first store (one trip to
  // memory per group instead of one per entry)
  for (int e = e0; e < cnt; e += 4 * estep) {
    int tile[4];
    int2 pg[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ee = e + u * estep;
      ok[u] = ee < cnt && start + ee < p.cap_nnz;
      tile[u] = ok[u] ? tr.tile_at(ee) : 0;
      pg[u] = make_int2(0, 0);
    }
    if (p.tile_pages != nullptr && pre != nullptr && e == e0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) pg[u] = pre[u];
    } else if (p.tile_pages != nullptr) {
      // physical pages of the tile's two 16-key halves (keys past the end share the last page)
      const int32_t* bt = p.block_table + (long long)b * p.block_table_stride;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) {
          pg[u].x = bt[(tile[u] * p.block_n) >> p.page_shift];
          pg[u].y = bt[imin(tile[u] * p.block_n + 16, n - 1) >> p.page_shift];
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (ok[u]) {
        const int ee = e + u * estep;
        p.col_idx[start + ee] = tile[u];
        if (p.tile_pages != nullptr) reinterpret_cast<int2*>(p.tile_pages)[start + ee] = pg[u];
      }
  }
  // work items: pieces of one q-block ascending from first_item
  const int pieces = cnt <= p.chunk ? 1 : ceil_div(cnt, p.chunk);
  for (int pc = e0; pc < pieces; pc += estep) {
    const int it = first_item + pc;
    if (it < p.cap_items) {
      const int s = start + pc * p.chunk;
      int4 rec;
      rec.x = qb;
      rec.y = s;
      rec.z = imax(0, imin(p.chunk, start + cnt - s));
      rec.w = pieces > 1 ? slot0 + pc : -1;
      int4 geo;  // everything the attention kernel needs to place the q-block, so it reads nothing else
      geo.x = b;
      geo.y = q_row0;    // first row of the q-block in the packed q / out tensors
      geo.z = ctx + r0;  // its position inside the sequence
      geo.w = n;
      reinterpret_cast<int4*>(p.items)[2 * it] = rec;
      reinterpret_cast<int4*>(p.items)[2 * it + 1] = geo;
    }
  }
}

// LDS of the single-workgroup build (22 KB)
struct IndexScanShared {
  Int4 wave_sums[16];
  int num_qblocks, num_long;
  // q-blocks whose lists the whole workgroup writes: q-block, request, first row, tile / piece / partial prefix
  int long_list[64][6];
  // up to 1024 requests the batch description is read from memory once and kept here: every later
  // step looks it up without another trip to memory
  int qbp[1025];
  int4 req[1024];  // first query row, query rows, keys, full length
};

// The whole build on ONE workgroup of NT threads (1024 or 512), for batches of up to 1024 q-blocks: q-blocks per
// request, the q-blocks' counts and their four scans in rounds of NT q-blocks, and -- once the totals are known --
// pass 2 for every q-block a thread counted (it still holds their numbers): short tile lists are written by the
// owning thread, long ones (the dense trailing rows) by the whole workgroup.
template <int NT>
__device__ __forceinline__ void index_scan_body(const IndexParams& p, IndexScanShared& sh) {
  constexpr int kLongList = 16;        // tile lists longer than this are written by the whole workgroup
  constexpr int kRounds = 1024 / NT;   // rounds a thread keeps its q-blocks' numbers for (pass 2 needs R <= 1024)
  const bool cached = p.batch <= 1024;
  const int tid = threadIdx.x;
  if (tid == 0) sh.num_long = 0;
  ISTAMP(0);

  // (a) q-blocks per request -> seq_qb_ptr
  int carry = 0;
  for (int base = 0; base < p.batch; base += NT) {
    const int b = base + tid;
    int nqb = 0;
    if (b < p.batch) {
      const int q0 = p.cu_q[b], q1 = p.cu_q[b + 1];
      int rows = imax(0, q1 - q0);
      if (p.dense_rows_from > 0 && rows >= p.dense_rows_from) rows = 0;
      nqb = p.bm_shift >= 0 ? (rows + p.block_m - 1) >> p.bm_shift : ceil_div(rows, p.block_m);
      if (cached) {
        const int n = p.seq_lens[b];
        sh.req[b] = make_int4(q0, q1 - q0, n, p.full_lens ? imax(n, p.full_lens[b]) : n);
      }
    }
    Int4 tot;
    Int4 e = block_exclusive_scan<NT>(Int4{nqb, 0, 0, 0}, sh.wave_sums, tot);
    if (b < p.batch) {
      p.seq_qb_ptr[b] = carry + e.x;
      if (cached) sh.qbp[b] = carry + e.x;
    }
    carry += tot.x;
  }
  if (tid == 0) {
    p.seq_qb_ptr[p.batch] = carry;
    sh.qbp[p.batch <= 1024 ? p.batch : 0] = carry;
    sh.num_qblocks = carry;
  }
  __syncthreads();
  ISTAMP(1);
  const int R = sh.num_qblocks;
  // request b: (first query row, query rows, keys, full length)
  auto request = [&](int b) -> int4 {
    if (cached) return sh.req[b];
    const int q0 = p.cu_q[b], n = p.seq_lens[b];
    return make_int4(q0, p.cu_q[b + 1] - q0, n, p.full_lens ? imax(n, p.full_lens[b]) : n);
  };

  // (b) per q-block counts -> row_ptr / piece_ptr / part_ptr / split positions
  Int4 run{0, 0, 0, 0};
  int max_tiles = 0;
  // what a thread keeps of the q-block it counted in round k
  Int4 kv[kRounds], ke[kRounds];
  int kb[kRounds], kr0[kRounds];
  TileRanges ktr[kRounds];
  // ... and the pages of the first four entries of its tile list (a sparse q-block lists three tiles): asked for
  // ahead of the scan, so that the trip to the block table is over when the list is written
  int2 kpg[kRounds][4];
#pragma unroll
  for (int k = 0; k < kRounds; ++k) {
    kv[k] = Int4{0, 0, 0, 0};
    ke[k] = Int4{0, 0, 0, 0};
    kb[k] = 0;
    kr0[k] = 0;
    ktr[k] = TileRanges{0, -1, 0, -1};
#pragma unroll
    for (int u = 0; u < 4; ++u) kpg[k][u] = make_int2(0, 0);
  }
  const bool one_pass = p.fused && R <= 1024;
  auto count_round = [&](int base, Int4& v, Int4& e, int& b, int& r0, TileRanges& tr_own, int2 (&pg)[4]) {
    const int qb = base + tid;
    v = Int4{0, 0, 0, 0};
    b = 0;
    r0 = 0;
    tr_own = TileRanges{0, -1, 0, -1};
    if (qb < R) {
      int lo = 0, hi = p.batch;  // last b with seq_qb_ptr[b] <= qb
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((cached ? sh.qbp[mid] : p.seq_qb_ptr[mid]) <= qb) lo = mid; else hi = mid;
      }
      b = lo;
      r0 = (qb - (cached ? sh.qbp[b] : p.seq_qb_ptr[b])) * p.block_m;
      const int4 rq = request(b);
      const int ql = rq.y, n = rq.z, nf = rq.w;
      const int ctx = n - ql;
      const int r1 = imin(r0 + p.block_m, ql);
      tr_own = tile_ranges(effective_mode(p.mode, nf, p.tri_min_len), ctx + r0, ctx + r1 - 1, n, nf, p.si, p.sl, p.last, p.block_n, p.bn_shift);
      const int cnt = tr_own.count();
      const int pieces = cnt <= p.chunk ? 1 : ceil_div(cnt, p.chunk);
      v = Int4{cnt, pieces, pieces > 1 ? pieces : 0, pieces > 1 ? 1 : 0};
      max_tiles = imax(max_tiles, cnt);
      if (one_pass && p.tile_pages != nullptr && cnt <= kLongList) {
#pragma unroll
        for (int u = 0; u < 4; ++u) pg[u] = u < cnt ? tile_pages_of(p, b, n, tr_own.tile_at(u)) : make_int2(0, 0);
      }
    }
    ISTAMP(7);
    Int4 tot;
    e = block_exclusive_scan<NT>(v, sh.wave_sums, tot);
    e = e + run;
    ISTAMP(8);
    if (qb < R && qb < p.cap_qblocks) {
      p.row_ptr[qb] = e.x;
      p.piece_ptr[qb] = e.y;
      p.part_ptr[qb] = e.z;
      p.qb_meta[2 * qb] = b;
      p.qb_meta[2 * qb + 1] = r0;
      if (v.w && e.w < p.cap_split) p.split_qb[e.w] = qb;
    }
    run = run + tot;
    ISTAMP(9);
  };
  if (one_pass) {
#pragma unroll
    for (int k = 0; k < kRounds; ++k)
      if (k * NT < R) count_round(k * NT, kv[k], ke[k], kb[k], kr0[k], ktr[k], kpg[k]);
  } else {
    for (int base = 0; base < R; base += NT) count_round(base, kv[0], ke[0], kb[0], kr0[0], ktr[0], kpg[0]);
  }
  if (one_pass) {
    // run holds the totals, this thread still holds its q-blocks' numbers
    bool is_long[kRounds];
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
      const int qb = k * NT + tid;
      is_long[k] = false;
      if (qb < R && qb < p.cap_qblocks) {
        const int4 rq = request(kb[k]);
        const int ql = rq.y, n = rq.z;
        const int ctx = n - ql;
        is_long[k] = kv[k].x > kLongList;
        if (!is_long[k])
          fill_qblock(p, qb, kb[k], kr0[k], n, ctx, rq.x + kr0[k], ktr[k], kv[k].x, ke[k].x,
                      first_item_of(run.y, run.z, ke[k].y, ke[k].z, kv[k].y), ke[k].z, 0, 1, kpg[k]);
      }
      ISTAMP(10);
      if (is_long[k]) {
        const int at = atomicAdd(&sh.num_long, 1);
        if (at < 64) {
          sh.long_list[at][0] = qb;
          sh.long_list[at][1] = kb[k];
          sh.long_list[at][2] = kr0[k];
          sh.long_list[at][3] = ke[k].x;
          sh.long_list[at][4] = ke[k].y;
          sh.long_list[at][5] = ke[k].z;
        }
      }
    }
    ISTAMP(11);
    // (a barrier for the LDS table only: the short lists' stores drain while the long lists are written)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ISTAMP(3);
    const int num_long = sh.num_long;
    // The long lists are written side by side: the threads are dealt round the lists (a power of two of them at a
    // time, so that NT divides evenly), every thread takes entries of ONE list -- the lookups of all lists make one
    // trip to the block table instead of one trip per list.
    const int listed_long = imin(num_long, 64);
    int side = 1;
    while (side < listed_long) side <<= 1;
    for (int k = tid & (side - 1); k < listed_long; k += side) {
      const int lq = sh.long_list[k][0], lb = sh.long_list[k][1], lr0 = sh.long_list[k][2];
      const int4 rq = request(lb);
      const int ql = rq.y, n = rq.z, nf = rq.w;
      const int ctx = n - ql;
      const int r1 = imin(lr0 + p.block_m, ql);
      const TileRanges tr = tile_ranges(effective_mode(p.mode, nf, p.tri_min_len), ctx + lr0, ctx + r1 - 1, n, nf, p.si,
                                        p.sl, p.last, p.block_n, p.bn_shift);
      const int cnt = tr.count();
      const int pieces = imax(1, ceil_div(cnt, p.chunk));
      fill_qblock(p, lq, lb, lr0, n, ctx, rq.x + lr0, tr, cnt, sh.long_list[k][3],
                  first_item_of(run.y, run.z, sh.long_list[k][4], sh.long_list[k][5], pieces), sh.long_list[k][5],
                  tid / side, NT / side);
    }
    if (num_long > 64) {
      // more long lists than the table holds: the remaining ones are written by their own threads
#pragma unroll
      for (int k = 0; k < kRounds; ++k) {
        if (!is_long[k]) continue;
        const int qb = k * NT + tid;
        bool listed = false;
#pragma unroll 1
        for (int j = 0; j < 64; ++j) listed = listed || sh.long_list[j][0] == qb;
        if (!listed) {
          const int4 rq = request(kb[k]);
          const int ql = rq.y, n = rq.z;
          const int ctx = n - ql;
          fill_qblock(p, qb, kb[k], kr0[k], n, ctx, rq.x + kr0[k], ktr[k], kv[k].x, ke[k].x,
                      first_item_of(run.y, run.z, ke[k].y, ke[k].z, kv[k].y), ke[k].z, 0, 1);
        }
      }
    }
  }
  // workgroup max of max_tiles (reuse the scan scratch)
  __syncthreads();
  ISTAMP(5);
  int m = max_tiles;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = imax(m, __shfl_xor(m, d, 64));
  if ((tid & 63) == 0) sh.wave_sums[tid >> 6].x = m;
  __syncthreads();
  ISTAMP(6);
  if (tid == 0) {
    for (int w = 1; w < NT / 64; ++w) m = imax(m, sh.wave_sums[w].x);
    if (R <= p.cap_qblocks) {
      p.row_ptr[R] = run.x;
      p.piece_ptr[R] = run.y;
      p.part_ptr[R] = run.z;
    }
    p.totals[0] = R;
    p.totals[1] = run.x;
    p.totals[2] = run.y;
    p.totals[3] = run.z;
    p.totals[4] = run.w;
    p.totals[5] = m;
    p.totals[6] = 0;
    p.totals[7] = 0;
  }
}

// Fills the kernel parameters of a build from the caller's index description (argument checks included).
int index_params_from(const tmx_index_t* idx, const int32_t* cu_q_dev, const int32_t* seq_lens_dev, IndexParams* out);

}  // namespace tmx
