// Packed block-sparse index builder (integer work, coalesced int32 stores).
//
// Replaces the O(N^2) Python mask loops of the reference
// (vllm/attention/trianglemix_attention.py:82-89, 126-129) with the closed
// form of the pattern evaluated per q-block: every q-block's key-tile set is
// at most two tile ranges, so the CSR index, the per-q-block metadata and the
// work list come out of one scan pass and one fill pass.
//
//   pass 1 (one workgroup, 1024 threads): q-blocks per request -> request of
//           each q-block -> tile / piece / partial / split counts -> four
//           exclusive scans (row_ptr, piece_ptr, part_ptr, split position).
//   pass 2 (one wave per q-block): lanes write consecutive col_idx entries and
//           the q-block's work items.
// Up to 1024 q-blocks (a 32k-token prompt at block_m 32) the scan needs a single
// round, every thread still holds its q-block's numbers, and pass 2 runs inside
// the same launch: short tile lists are written by the owning thread, long ones
// (the dense trailing rows) by the whole workgroup.
// Beyond 1024 q-blocks pass 1 runs on one workgroup per 1024 q-blocks
// (index_block_kernel: prefixes inside the block, the block's totals) and pass 2
// adds the totals of the blocks before its q-block's: no workgroup waits for
// another, and the time no longer grows with the batch (8 prompts of 64k tokens:
// 16 blocks side by side instead of 16 rounds on one workgroup).
#include "tmx_common.h"

#ifdef TMX_INDEX_STAMPS
__device__ long long g_idx_stamps[16];
extern "C" int tmx_index_read_stamps(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_idx_stamps), sizeof(long long) * 16); }
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

// Exclusive scan of one Int4 per thread over a 1024-thread workgroup.
// Returns the exclusive prefix; `total` receives the workgroup sum.
__device__ Int4 block_exclusive_scan(Int4 v, Int4* wave_sums /* LDS [16] */, Int4& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const Int4 inc{wave_inclusive_scan(v.x), wave_inclusive_scan(v.y), wave_inclusive_scan(v.z), wave_inclusive_scan(v.w)};
  if (lane == 63) wave_sums[wave] = inc;
  __syncthreads();
  Int4 base{0, 0, 0, 0};
  Int4 tot{0, 0, 0, 0};
  for (int w = 0; w < 16; ++w) {
    if (w < wave) base = base + wave_sums[w];
    tot = tot + wave_sums[w];
  }
  __syncthreads();
  total = tot;
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
__device__ __forceinline__ void fill_qblock(const IndexParams& p, int qb, int b, int r0, int n, int ctx, int q_row0,
                                            const TileRanges& tr, int cnt, int start, int first_item, int slot0,
                                            int e0, int estep) {
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
    if (p.tile_pages != nullptr) {
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

__global__ __launch_bounds__(1024) void index_scan_kernel(IndexParams p) {
  constexpr int kLongList = 16;  // tile lists longer than this are written by the whole workgroup
  __shared__ Int4 wave_sums[16];
  __shared__ int s_num_qblocks;
  // q-blocks whose lists the whole workgroup writes: q-block, request, first row, tile / piece / partial prefix
  __shared__ int s_long[64][6], s_num_long;
  // up to 1024 requests the batch description is read from memory once and kept here: every later
  // step looks it up without another trip to memory
  __shared__ int s_qbp[1025];
  __shared__ int4 s_req[1024];  // first query row, query rows, keys, full length
  const bool cached = p.batch <= 1024;
  const int tid = threadIdx.x;
  if (tid == 0) s_num_long = 0;
  ISTAMP(0);

  // (a) q-blocks per request -> seq_qb_ptr
  int carry = 0;
  for (int base = 0; base < p.batch; base += 1024) {
    const int b = base + tid;
    int nqb = 0;
    if (b < p.batch) {
      const int q0 = p.cu_q[b], q1 = p.cu_q[b + 1];
      const int rows = imax(0, q1 - q0);
      nqb = p.bm_shift >= 0 ? (rows + p.block_m - 1) >> p.bm_shift : ceil_div(rows, p.block_m);
      if (cached) {
        const int n = p.seq_lens[b];
        s_req[b] = make_int4(q0, q1 - q0, n, p.full_lens ? imax(n, p.full_lens[b]) : n);
      }
    }
    Int4 tot;
    Int4 e = block_exclusive_scan(Int4{nqb, 0, 0, 0}, wave_sums, tot);
    if (b < p.batch) {
      p.seq_qb_ptr[b] = carry + e.x;
      if (cached) s_qbp[b] = carry + e.x;
    }
    carry += tot.x;
  }
  if (tid == 0) {
    p.seq_qb_ptr[p.batch] = carry;
    s_qbp[p.batch <= 1024 ? p.batch : 0] = carry;
    s_num_qblocks = carry;
  }
  __syncthreads();
  ISTAMP(1);
  const int R = s_num_qblocks;
  // request b: (first query row, query rows, keys, full length)
  auto request = [&](int b) -> int4 {
    if (cached) return s_req[b];
    const int q0 = p.cu_q[b], n = p.seq_lens[b];
    return make_int4(q0, p.cu_q[b + 1] - q0, n, p.full_lens ? imax(n, p.full_lens[b]) : n);
  };

  // (b) per q-block counts -> row_ptr / piece_ptr / part_ptr / split positions
  Int4 run{0, 0, 0, 0};
  int max_tiles = 0;
  for (int base = 0; base < R; base += 1024) {
    const int qb = base + tid;
    Int4 v{0, 0, 0, 0};
    int b = 0, r0 = 0;
    TileRanges tr_own{0, -1, 0, -1};  // this thread's q-block (kept for pass 2)
    if (qb < R) {
      int lo = 0, hi = p.batch;  // last b with seq_qb_ptr[b] <= qb
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((cached ? s_qbp[mid] : p.seq_qb_ptr[mid]) <= qb) lo = mid; else hi = mid;
      }
      b = lo;
      r0 = (qb - (cached ? s_qbp[b] : p.seq_qb_ptr[b])) * p.block_m;
      const int4 rq = request(b);
      const int ql = rq.y, n = rq.z, nf = rq.w;
      const int ctx = n - ql;
      const int r1 = imin(r0 + p.block_m, ql);
      tr_own = tile_ranges(effective_mode(p.mode, nf, p.tri_min_len), ctx + r0, ctx + r1 - 1, n, nf, p.si, p.sl, p.last, p.block_n, p.bn_shift);
      const int cnt = tr_own.count();
      const int pieces = cnt <= p.chunk ? 1 : ceil_div(cnt, p.chunk);
      v = Int4{cnt, pieces, pieces > 1 ? pieces : 0, pieces > 1 ? 1 : 0};
      max_tiles = imax(max_tiles, cnt);
    }
    ISTAMP(7);
    Int4 tot;
    Int4 e = block_exclusive_scan(v, wave_sums, tot);
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
    if (p.fused && R <= 1024) {
      // single round: run holds the totals, this thread still holds its q-block's numbers
      bool is_long = false;
      if (qb < R && qb < p.cap_qblocks) {
        const int4 rq = request(b);
        const int ql = rq.y, n = rq.z;
        const int ctx = n - ql;
        is_long = v.x > kLongList;
        if (!is_long)
          fill_qblock(p, qb, b, r0, n, ctx, rq.x + r0, tr_own, v.x, e.x, first_item_of(run.y, run.z, e.y, e.z, v.y), e.z, 0, 1);
      }
      if (is_long) {
        const int at = atomicAdd(&s_num_long, 1);
        if (at < 64) {
          s_long[at][0] = qb;
          s_long[at][1] = b;
          s_long[at][2] = r0;
          s_long[at][3] = e.x;
          s_long[at][4] = e.y;
          s_long[at][5] = e.z;
        }
      }
      __syncthreads();
      ISTAMP(3);
      const int num_long = s_num_long;
      for (int k = 0; k < imin(num_long, 64); ++k) {
        const int lq = s_long[k][0], lb = s_long[k][1], lr0 = s_long[k][2];
        const int4 rq = request(lb);
        const int ql = rq.y, n = rq.z, nf = rq.w;
        const int ctx = n - ql;
        const int r1 = imin(lr0 + p.block_m, ql);
        const TileRanges tr = tile_ranges(effective_mode(p.mode, nf, p.tri_min_len), ctx + lr0, ctx + r1 - 1, n, nf, p.si,
                                          p.sl, p.last, p.block_n, p.bn_shift);
        const int cnt = tr.count();
        const int pieces = imax(1, ceil_div(cnt, p.chunk));
        fill_qblock(p, lq, lb, lr0, n, ctx, rq.x + lr0, tr, cnt, s_long[k][3],
                    first_item_of(run.y, run.z, s_long[k][4], s_long[k][5], pieces), s_long[k][5], tid, 1024);
      }
      if (num_long > 64) {
        // more long lists than the table holds: the remaining ones are written by their own threads
        if (is_long) {
          bool listed = false;
          for (int k = 0; k < 64; ++k) listed = listed || s_long[k][0] == qb;
          if (!listed) {
            const int4 rq = request(b);
            const int ql = rq.y, n = rq.z, nf = rq.w;
            const int ctx = n - ql;
            const int r1 = imin(r0 + p.block_m, ql);
            const TileRanges tr = tile_ranges(effective_mode(p.mode, nf, p.tri_min_len), ctx + r0, ctx + r1 - 1, n, nf,
                                              p.si, p.sl, p.last, p.block_n, p.bn_shift);
            fill_qblock(p, qb, b, r0, n, ctx, rq.x + r0, tr, v.x, e.x, first_item_of(run.y, run.z, e.y, e.z, v.y), e.z, 0, 1);
          }
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
  if ((tid & 63) == 0) wave_sums[tid >> 6].x = m;
  __syncthreads();
  ISTAMP(6);
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) m = imax(m, wave_sums[w].x);
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

// Pass 1 for more than 1024 q-blocks: workgroup k takes q-blocks 1024 k .. 1024 k + 1023.  Every workgroup
// derives the q-blocks per request itself (the batch description is small), then writes the prefixes INSIDE its
// block and the block's totals; index_fill_kernel adds the totals of the blocks before.
__global__ __launch_bounds__(1024) void index_block_kernel(IndexParams p) {
  __shared__ Int4 wave_sums[16];
  __shared__ int s_num_qblocks;
  __shared__ int s_qbp[1025];
  __shared__ int4 s_req[1024];  // first query row, query rows, keys, full length
  const bool cached = p.batch <= 1024;
  const int tid = threadIdx.x;
  // (a) q-blocks per request (every workgroup writes the same seq_qb_ptr values)
  int carry = 0;
  for (int base = 0; base < p.batch; base += 1024) {
    const int b = base + tid;
    int nqb = 0;
    if (b < p.batch) {
      const int q0 = p.cu_q[b], q1 = p.cu_q[b + 1];
      const int rows = imax(0, q1 - q0);
      nqb = p.bm_shift >= 0 ? (rows + p.block_m - 1) >> p.bm_shift : ceil_div(rows, p.block_m);
      if (cached) {
        const int n = p.seq_lens[b];
        s_req[b] = make_int4(q0, q1 - q0, n, p.full_lens ? imax(n, p.full_lens[b]) : n);
      }
    }
    Int4 tot;
    Int4 e = block_exclusive_scan(Int4{nqb, 0, 0, 0}, wave_sums, tot);
    if (b < p.batch) {
      p.seq_qb_ptr[b] = carry + e.x;
      if (cached) s_qbp[b] = carry + e.x;
    }
    carry += tot.x;
  }
  if (tid == 0) {
    p.seq_qb_ptr[p.batch] = carry;
    s_qbp[p.batch <= 1024 ? p.batch : 0] = carry;
    s_num_qblocks = carry;
  }
  __syncthreads();
  const int R = s_num_qblocks;
  // (b) this block's q-blocks
  const int qb = (int)blockIdx.x * 1024 + tid;
  Int4 v{0, 0, 0, 0};
  int b = 0, r0 = 0, cnt = 0;
  if (qb < R) {
    int lo = 0, hi = p.batch;  // last b with seq_qb_ptr[b] <= qb
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((cached ? s_qbp[mid] : p.seq_qb_ptr[mid]) <= qb) lo = mid; else hi = mid;
    }
    b = lo;
    r0 = (qb - (cached ? s_qbp[b] : p.seq_qb_ptr[b])) * p.block_m;
    int4 rq;
    if (cached) rq = s_req[b];
    else {
      const int q0 = p.cu_q[b], n = p.seq_lens[b];
      rq = make_int4(q0, p.cu_q[b + 1] - q0, n, p.full_lens ? imax(n, p.full_lens[b]) : n);
    }
    const int ql = rq.y, n = rq.z, nf = rq.w;
    const int ctx = n - ql;
    const int r1 = imin(r0 + p.block_m, ql);
    const TileRanges tr = tile_ranges(effective_mode(p.mode, nf, p.tri_min_len), ctx + r0, ctx + r1 - 1, n, nf, p.si, p.sl, p.last, p.block_n, p.bn_shift);
    cnt = tr.count();
    const int pieces = cnt <= p.chunk ? 1 : ceil_div(cnt, p.chunk);
    v = Int4{cnt, pieces, pieces > 1 ? pieces : 0, pieces > 1 ? 1 : 0};
  }
  Int4 tot;
  const Int4 e = block_exclusive_scan(v, wave_sums, tot);
  if (qb < R && qb < p.cap_qblocks) {
    p.row_ptr[qb] = e.x;
    p.piece_ptr[qb] = e.y;
    p.part_ptr[qb] = e.z;
    p.split_pos[qb] = e.w;
    p.qb_meta[2 * qb] = b;
    p.qb_meta[2 * qb + 1] = r0;
  }
  // longest tile list of the block (the scan's scratch is free again)
  int m = cnt;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = imax(m, __shfl_xor(m, d, 64));
  if ((tid & 63) == 0) wave_sums[tid >> 6].x = m;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) m = imax(m, wave_sums[w].x);
    int32_t* t = p.blk_tot + 8 * (int)blockIdx.x;
    t[0] = tot.x;
    t[1] = tot.y;
    t[2] = tot.z;
    t[3] = tot.w;
    t[4] = m;
    t[5] = R;
    t[6] = 0;
    t[7] = 0;
  }
}

__global__ __launch_bounds__(256) void index_fill_kernel(IndexParams p) {
  const int lane = threadIdx.x & 63;
  const int qb = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p.blocked) {
    // prefixes inside the q-block's block + the totals of the blocks before it; the grand totals
    const int num_qblocks = p.blk_tot[5];
    // only the blocks that were launched (and have a place in the work area) wrote their totals: the device's
    // count may exceed the capacity the caller sized for, which is exactly the case the check below refuses
    const int nblk = imin((num_qblocks + 1023) >> 10, (p.cap_qblocks + 1023) >> 10), myblk = qb >> 10;
    Int4 off{0, 0, 0, 0}, all{0, 0, 0, 0};
    int maxt = 0;
    for (int k = lane; k < nblk; k += 64) {
      const int32_t* t = p.blk_tot + 8 * k;
      const Int4 bt{t[0], t[1], t[2], t[3]};
      all = all + bt;
      if (k < myblk) off = off + bt;
      maxt = imax(maxt, t[4]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      off = off + Int4{__shfl_xor(off.x, d, 64), __shfl_xor(off.y, d, 64), __shfl_xor(off.z, d, 64), __shfl_xor(off.w, d, 64)};
      all = all + Int4{__shfl_xor(all.x, d, 64), __shfl_xor(all.y, d, 64), __shfl_xor(all.z, d, 64), __shfl_xor(all.w, d, 64)};
      maxt = imax(maxt, __shfl_xor(maxt, d, 64));
    }
    if (qb == 0 && lane == 0) {
      if (num_qblocks <= p.cap_qblocks) {
        p.row_ptr[num_qblocks] = all.x;
        p.part_ptr[num_qblocks] = all.z;
      }
      p.totals[0] = num_qblocks;
      p.totals[1] = all.x;
      p.totals[2] = all.y;
      p.totals[3] = all.z;
      p.totals[4] = all.w;
      p.totals[5] = maxt;
      p.totals[6] = 0;
      p.totals[7] = 0;
    }
    if (num_qblocks > p.cap_qblocks || qb >= num_qblocks) return;
    // (this q-block's own entries: read by this wave alone, then replaced by the prefixes over the whole index)
    const int start = p.row_ptr[qb] + off.x, piece_prefix = p.piece_ptr[qb] + off.y, part_prefix = p.part_ptr[qb] + off.z;
    const int split_at = p.split_pos[qb] + off.w;
    const int b = p.qb_meta[2 * qb], r0 = p.qb_meta[2 * qb + 1];
    const int ql = p.cu_q[b + 1] - p.cu_q[b];
    const int n = p.seq_lens[b];
    const int nf = p.full_lens ? imax(n, p.full_lens[b]) : n;
    const int ctx = n - ql;
    const int r1 = imin(r0 + p.block_m, ql);
    const TileRanges tr = tile_ranges(effective_mode(p.mode, nf, p.tri_min_len), ctx + r0, ctx + r1 - 1, n, nf, p.si, p.sl, p.last, p.block_n, p.bn_shift);
    const int cnt = tr.count();
    const int pieces = imax(1, ceil_div(cnt, p.chunk));
    fill_qblock(p, qb, b, r0, n, ctx, p.cu_q[b] + r0, tr, cnt, start, first_item_of(all.y, all.z, piece_prefix, part_prefix, pieces),
                part_prefix, lane, 64);
    if (lane == 0) {
      p.row_ptr[qb] = start;
      p.part_ptr[qb] = part_prefix;
      if (pieces > 1 && split_at < p.cap_split) p.split_qb[split_at] = qb;
    }
    return;
  }
  const int num_qblocks = p.totals[0];  // counted by index_scan_kernel earlier on this stream
  if (num_qblocks > p.cap_qblocks || qb >= num_qblocks) return;
  const int b = p.qb_meta[2 * qb], r0 = p.qb_meta[2 * qb + 1];
  const int ql = p.cu_q[b + 1] - p.cu_q[b];
  const int n = p.seq_lens[b];
  const int nf = p.full_lens ? imax(n, p.full_lens[b]) : n;
  const int ctx = n - ql;
  const int r1 = imin(r0 + p.block_m, ql);
  const TileRanges tr = tile_ranges(effective_mode(p.mode, nf, p.tri_min_len), ctx + r0, ctx + r1 - 1, n, nf, p.si, p.sl, p.last, p.block_n, p.bn_shift);
  const int cnt = tr.count();
  const int pieces = imax(1, ceil_div(cnt, p.chunk));
  fill_qblock(p, qb, b, r0, n, ctx, p.cu_q[b] + r0, tr, cnt, p.row_ptr[qb],
              first_item_of(p.piece_ptr[num_qblocks], p.part_ptr[num_qblocks], p.piece_ptr[qb], p.part_ptr[qb], pieces),
              p.part_ptr[qb], lane, 64);
}

}  // namespace tmx

extern "C" int tmx_build_block_index(tmx_index_t* idx, const int32_t* cu_q_dev, const int32_t* seq_lens_dev,
                                     const int32_t* cu_q_host, const int32_t* seq_lens_host, void* stream) {
  TMX_CHECK_ARG(idx != nullptr, "tmx_build_block_index: idx is NULL");
  TMX_CHECK_ARG(idx->batch >= 0, "tmx_build_block_index: batch < 0");
  TMX_CHECK_ARG(idx->block_m > 0 && idx->block_n > 0 && idx->chunk_tiles > 0,
                "tmx_build_block_index: block_m / block_n / chunk_tiles must be positive");
  TMX_CHECK_ARG(idx->mode >= TMX_MODE_NONE && idx->mode <= TMX_MODE_TRIANGLE, "tmx_build_block_index: bad mode %d",
                idx->mode);
  TMX_CHECK_ARG(idx->totals && idx->scratch && idx->row_ptr && idx->part_ptr,
                "tmx_build_block_index: NULL totals / scratch / row_ptr / part_ptr");
  TMX_CHECK_ARG(idx->batch == 0 || (cu_q_dev && seq_lens_dev), "tmx_build_block_index: NULL device batch arrays");
  TMX_CHECK_ARG(idx->cap_qblocks >= 0 && idx->cap_nnz >= 0 && idx->cap_items >= 0 && idx->cap_split >= 0,
                "tmx_build_block_index: negative capacity");
  TMX_CHECK_ARG((idx->cap_qblocks == 0 || idx->qb_meta) && (idx->cap_nnz == 0 || idx->col_idx) &&
                    (idx->cap_items == 0 || idx->items) && (idx->cap_split == 0 || idx->split_qb),
                "tmx_build_block_index: NULL buffer with non-zero capacity");
  TMX_CHECK_ARG((reinterpret_cast<uintptr_t>(idx->items) & 15) == 0, "tmx_build_block_index: items must be 16-byte aligned");
  // (full_lens without its host mirror: the sizes cannot be recomputed here, the check is skipped)
  if (cu_q_host && seq_lens_host && (idx->full_lens == nullptr || idx->full_lens_host != nullptr)) {
    tmx_index_sizes_t sz;
    int rc = tmx_index_sizes(cu_q_host, seq_lens_host, idx->full_lens ? idx->full_lens_host : nullptr, idx->batch, idx->si, idx->sl, idx->last, idx->block_m,
                             idx->block_n, idx->mode, idx->tri_min_len, idx->chunk_tiles, &sz);
    if (rc != TMX_OK) return rc;
    if (sz.num_qblocks > idx->cap_qblocks || sz.nnz_tiles > idx->cap_nnz || sz.num_items > idx->cap_items ||
        sz.num_split_qblocks > idx->cap_split) {
      tmx::set_error("tmx_build_block_index: capacities (%lld q-blocks, %lld tiles, %lld items, %lld split) below "
                     "the required (%lld, %lld, %lld, %lld)",
                     (long long)idx->cap_qblocks, (long long)idx->cap_nnz, (long long)idx->cap_items,
                     (long long)idx->cap_split, (long long)sz.num_qblocks, (long long)sz.nnz_tiles,
                     (long long)sz.num_items, (long long)sz.num_split_qblocks);
      return TMX_ERR_CAPACITY;
    }
  }
  tmx::IndexParams p;
  p.cu_q = cu_q_dev;
  p.seq_lens = seq_lens_dev;
  p.full_lens = idx->full_lens;
  p.row_ptr = idx->row_ptr;
  p.col_idx = idx->col_idx;
  p.qb_meta = idx->qb_meta;
  p.part_ptr = idx->part_ptr;
  p.items = idx->items;
  p.split_qb = idx->split_qb;
  p.totals = idx->totals;
  p.tile_pages = nullptr;
  p.block_table = nullptr;
  p.block_table_stride = 0;
  p.page_shift = 0;
  if (idx->tile_pages != nullptr && idx->pages_block_table != nullptr) {
    TMX_CHECK_SUPPORTED(idx->block_n == 32 && idx->pages_page_size >= 16 &&
                            (idx->pages_page_size & (idx->pages_page_size - 1)) == 0,
                        "tmx_build_block_index: page resolution needs block_n 32 and a power-of-two page size >= 16");
    TMX_CHECK_ARG((reinterpret_cast<uintptr_t>(idx->tile_pages) & 7) == 0, "tmx_build_block_index: tile_pages must be 8-byte aligned");
    p.tile_pages = idx->tile_pages;
    p.block_table = idx->pages_block_table;
    p.block_table_stride = idx->pages_block_table_stride;
    p.page_shift = __builtin_ctz(idx->pages_page_size);
  }
  p.seq_qb_ptr = idx->scratch;
  p.piece_ptr = idx->scratch + idx->batch + 1;
  p.split_pos = p.piece_ptr + idx->cap_qblocks + 1;
  p.blk_tot = p.split_pos + idx->cap_qblocks + 1;
  p.batch = idx->batch;
  p.block_m = idx->block_m;
  p.block_n = idx->block_n;
  p.mode = idx->mode;
  p.chunk = idx->chunk_tiles;
  p.si = idx->si;
  p.sl = idx->sl;
  p.last = idx->last;
  p.tri_min_len = idx->tri_min_len;
  p.cap_qblocks = (int)idx->cap_qblocks;
  p.cap_nnz = (int)idx->cap_nnz;
  p.cap_items = (int)idx->cap_items;
  p.cap_split = (int)idx->cap_split;
  p.fused = idx->cap_qblocks <= 1024 ? 1 : 0;
  p.blocked = p.fused ? 0 : 1;
  p.bn_shift = (idx->block_n & (idx->block_n - 1)) == 0 ? __builtin_ctz(idx->block_n) : -1;
  p.bm_shift = (idx->block_m & (idx->block_m - 1)) == 0 ? __builtin_ctz(idx->block_m) : -1;
  hipStream_t s = (hipStream_t)stream;
  if (p.fused) {
    tmx::index_scan_kernel<<<1, 1024, 0, s>>>(p);
    TMX_CHECK_HIP(hipGetLastError());
  } else {
    const int nq = (int)idx->cap_qblocks;  // upper bound of the q-block count; surplus threads / waves find nothing
    tmx::index_block_kernel<<<tmx::ceil_div(nq, 1024), 1024, 0, s>>>(p);
    TMX_CHECK_HIP(hipGetLastError());
    tmx::index_fill_kernel<<<tmx::ceil_div(nq, 4), 256, 0, s>>>(p);
    TMX_CHECK_HIP(hipGetLastError());
  }
  return TMX_OK;
}
