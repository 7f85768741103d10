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
#include "tmx_index_scan.h"

#ifdef TMX_INDEX_STAMPS
extern "C" int tmx_index_read_stamps(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_idx_stamps), sizeof(long long) * 16); }
#endif
namespace tmx {

__global__ __launch_bounds__(1024) void index_scan_kernel(IndexParams p) {
  __shared__ IndexScanShared sh;
  index_scan_body<1024>(p, sh);
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
      int rows = imax(0, q1 - q0);
      if (p.dense_rows_from > 0 && rows >= p.dense_rows_from) rows = 0;
      nqb = p.bm_shift >= 0 ? (rows + p.block_m - 1) >> p.bm_shift : ceil_div(rows, p.block_m);
      if (cached) {
        const int n = p.seq_lens[b];
        s_req[b] = make_int4(q0, q1 - q0, n, p.full_lens ? imax(n, p.full_lens[b]) : n);
      }
    }
    Int4 tot;
    Int4 e = block_exclusive_scan<1024>(Int4{nqb, 0, 0, 0}, wave_sums, tot);
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
  const Int4 e = block_exclusive_scan<1024>(v, wave_sums, tot);
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

namespace tmx {

int index_params_from(const tmx_index_t* idx, const int32_t* cu_q_dev, const int32_t* seq_lens_dev, IndexParams* out) {
  TMX_CHECK_ARG(idx != nullptr, "tmx_build_block_index: idx is NULL");
  TMX_CHECK_ARG(idx->batch >= 0, "tmx_build_block_index: batch < 0");
  TMX_CHECK_ARG(idx->block_m > 0 && idx->block_n > 0 && idx->chunk_tiles > 0,
                "tmx_build_block_index: block_m / block_n / chunk_tiles must be positive");
  TMX_CHECK_ARG(idx->mode >= TMX_MODE_NONE && idx->mode <= TMX_MODE_TRIANGLE, "tmx_build_block_index: bad mode %d",
                idx->mode);
  TMX_CHECK_ARG(idx->dense_rows_from == 0 || (idx->dense_rows_from == TMX_DENSE_MIN_ROWS && idx->mode == TMX_MODE_CAUSAL),
                "tmx_build_block_index: dense_rows_from is 0 or TMX_DENSE_MIN_ROWS (%d), the latter with TMX_MODE_CAUSAL only",
                TMX_DENSE_MIN_ROWS);
  TMX_CHECK_ARG(idx->totals && idx->scratch && idx->row_ptr && idx->part_ptr,
                "tmx_build_block_index: NULL totals / scratch / row_ptr / part_ptr");
  TMX_CHECK_ARG(idx->batch == 0 || (cu_q_dev && seq_lens_dev), "tmx_build_block_index: NULL device batch arrays");
  TMX_CHECK_ARG(idx->cap_qblocks >= 0 && idx->cap_nnz >= 0 && idx->cap_items >= 0 && idx->cap_split >= 0,
                "tmx_build_block_index: negative capacity");
  TMX_CHECK_ARG((idx->cap_qblocks == 0 || idx->qb_meta) && (idx->cap_nnz == 0 || idx->col_idx) &&
                    (idx->cap_items == 0 || idx->items) && (idx->cap_split == 0 || idx->split_qb),
                "tmx_build_block_index: NULL buffer with non-zero capacity");
  TMX_CHECK_ARG((reinterpret_cast<uintptr_t>(idx->items) & 15) == 0, "tmx_build_block_index: items must be 16-byte aligned");
  IndexParams p;
  p.cu_q = cu_q_dev;
  p.seq_lens = seq_lens_dev;
  p.full_lens = idx->full_lens;
  p.dense_rows_from = idx->dense_rows_from;
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
  *out = p;
  return TMX_OK;
}

}  // namespace tmx

extern "C" int tmx_build_block_index(tmx_index_t* idx, const int32_t* cu_q_dev, const int32_t* seq_lens_dev,
                                     const int32_t* cu_q_host, const int32_t* seq_lens_host, void* stream) {
  tmx::IndexParams p;
  int rc0 = tmx::index_params_from(idx, cu_q_dev, seq_lens_dev, &p);
  if (rc0 != TMX_OK) return rc0;
  // (full_lens without its host mirror: the sizes cannot be recomputed here, the check is skipped)
  if (cu_q_host && seq_lens_host && (idx->full_lens == nullptr || idx->full_lens_host != nullptr)) {
    tmx_index_sizes_t sz;
    int rc = tmx_index_sizes_without_dense(cu_q_host, seq_lens_host, idx->full_lens ? idx->full_lens_host : nullptr, idx->batch, idx->si,
                                           idx->sl, idx->last, idx->block_m, idx->block_n, idx->mode, idx->tri_min_len, idx->chunk_tiles,
                                           idx->dense_rows_from, &sz);
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