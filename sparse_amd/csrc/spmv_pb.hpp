// spmv_pb.hpp -- two-pass SpMV for gather-bound records (power-law graphs: BASELINE configs[2]).
// OPT-IN ("spmv_pb" = 1): measured slower than the one-pass path on MI355X, kept with its tests as the record
// of the experiment.
//
// On an RMAT matrix the one-pass kernel (spmv.hpp) is bound by the rate at which a CU turns over x
// gathers that miss its L1 (~140-150 G gathers/s chip-wide, measured with every gather hitting L2), not by
// HBM: 3e8 entries cost 2 ms where their bytes would stream in 0.35 ms.  Here no gather leaves the CU.
// At its first product such a record gets a second copy of its entries cut into PIECES = (row panel) x
// (column block), a panel being <= R consecutive rows holding <= ~E entries and a block W columns, R and W
// sized so that a slice of y, or of x, is 128 KB of LDS.  A product is then
//   pass 1  (workgroup = column block, its slice of x in LDS): every entry's product val * x[col], the gather
//           served by LDS, streamed out in piece order;
//   pass 2  (workgroup = row panel, its slice of y in LDS): the products of the panel, contiguous, added
//           into the slice by LDS atomics (runs of equal rows are first folded inside the wavefront), then the
//           slice is written to y.
// Entries are laid out [panel][block][row][column], so pass 2 streams one contiguous range and pass 1 walks
// one piece per panel.  Columns are dealt to blocks by popularity (32-column lines ranked by entry count,
// line of rank q -> block q mod NB): every block, hence every piece, holds about the same number of entries
// whatever the skew, and x is brought into that order once per product by copying whole lines.
// Bytes per entry: 4+2 read, 4 written (pass 1), 4+2 read (pass 2) = 16 against 8 algorithmic (f32), all of
// it streaming.  The sum of a row is formed in piece order by atomics, so it is NOT the oracle's left-to-right
// order and not bit-reproducible run to run for floating point (within the 1e-5 / 1e-12 bars; exact for i32).
//
// Measured on C3 (RMAT scale 24, f32, 2.96e8 entries; rocprofv3 kernel stats, gpurun_out/pb_lab_*): x into block
// order 0.04 ms, pass 1 0.80 ms (0.55 without the product store: walking ~540-entry pieces is latency-bound),
// pass 2 1.99 ms of which 0.47 is the streaming and 1.5 the LDS accumulation -- ds_add_f32 retires about one
// lane every 3-4 cycles per CU, with or without the in-wavefront fold (2.35 ms with plain atomics).  Total
// 2.8 ms against 2.08 ms for the one-pass kernel with hot-column renumbering.  What would have to change: pass 2
// without LDS atomics (each wavefront owning a row range of the panel and walking its ~34-entry fragment of
// every piece with plain read-modify-writes), estimated 0.7 ms issue-bound, i.e. ~1.5 ms in all: not built.
#pragma once

#include <vector>

#include "spmv.hpp"

namespace samd {

constexpr int kPbThreads = 1024;  // one workgroup per CU: the slice takes most of the CU's 160 KB of LDS
constexpr int kPbWaves = kPbThreads / 64;
constexpr int kPbLine = 32;                         // columns per line (the unit dealt to blocks)
constexpr int kPbSliceBytes = 128 << 10;            // slice of x (pass 1) or y (pass 2) in LDS
constexpr int64_t kPbPanelEntries = 512 << 10;      // a panel is closed once it holds this many entries
constexpr int64_t kPbMinPiece = 192;                // auto mode: mean entries per piece below this -> not worth it

__global__ __launch_bounds__(kBlock) void pb_line_count_kernel(const int32_t *__restrict__ col, int64_t nnz,
                                                               uint32_t *__restrict__ counts) {
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * kBlock)
    atomicAdd(&counts[col[e] / kPbLine], 1u);
}

__global__ __launch_bounds__(kBlock) void pb_rank_kernel(const uint32_t *__restrict__ line_of_rank, int64_t nl,
                                                         uint32_t *__restrict__ rank_of_line) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q < nl) rank_of_line[line_of_rank[q]] = (uint32_t)q;
}

// key of entry e = panel * nb + block
__global__ __launch_bounds__(kBlock) void pb_key_kernel(const int64_t *__restrict__ row_ptr,
                                                        const int32_t *__restrict__ col, int64_t nnz,
                                                        const int64_t *__restrict__ tile_lb, int64_t tile,
                                                        const int64_t *__restrict__ panel_row, int64_t np,
                                                        const uint32_t *__restrict__ rank_of_line, int64_t nb,
                                                        uint32_t *__restrict__ keys) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= nnz) return;
  const int64_t r = find_row(row_ptr, tile_lb, tile, e);
  int64_t lo = 0, hi = np;  // largest p with panel_row[p] <= r
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (panel_row[mid] <= r) lo = mid; else hi = mid;
  }
  const int64_t q = rank_of_line[col[e] / kPbLine];
  keys[e] = (uint32_t)(lo * nb + q % nb);
}

// piece_ptr[k] = first position whose key is >= k, k = 0 .. pieces
__global__ __launch_bounds__(kBlock) void pb_piece_ptr_kernel(const uint32_t *__restrict__ sorted_keys, int64_t nnz,
                                                              int64_t pieces, int64_t *__restrict__ piece_ptr) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k > pieces) return;
  int64_t lo = 0, hi = nnz;
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    if ((int64_t)sorted_keys[mid] < k) lo = mid + 1; else hi = mid;
  }
  piece_ptr[k] = lo;
}

template <class T>
__global__ __launch_bounds__(kBlock) void pb_build_kernel(const int64_t *__restrict__ row_ptr,
                                                          const int32_t *__restrict__ col, const T *__restrict__ val,
                                                          int64_t nnz, const int64_t *__restrict__ tile_lb,
                                                          int64_t tile, const int64_t *__restrict__ panel_row,
                                                          const uint32_t *__restrict__ rank_of_line, int64_t nb,
                                                          const uint32_t *__restrict__ sorted_keys,
                                                          const uint32_t *__restrict__ perm,
                                                          uint16_t *__restrict__ col16, uint16_t *__restrict__ row16,
                                                          T *__restrict__ pval) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nnz) return;
  const int64_t src = perm ? (int64_t)perm[i] : i;  // no sort pass ran (one piece): identity
  const int64_t p = (int64_t)sorted_keys[i] / nb;
  const int64_t r = find_row(row_ptr, tile_lb, tile, src);
  const int32_t c = col[src];
  const int64_t q = rank_of_line[c / kPbLine];
  col16[i] = (uint16_t)((q / nb) * kPbLine + (c % kPbLine));
  row16[i] = (uint16_t)(r - panel_row[p]);
  pval[i] = val[src];
}

// x in block order: line of rank q -> block q mod nb, slot q / nb; w = columns per block
template <class T>
__global__ __launch_bounds__(kBlock) void pb_xperm_kernel(const T *__restrict__ x, int64_t m,
                                                          const uint32_t *__restrict__ line_of_rank, int64_t nl,
                                                          int64_t nb, int64_t w, T *__restrict__ xperm) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t q = t / kPbLine, j = t % kPbLine;
  if (q >= nl) return;
  const int64_t c = (int64_t)line_of_rank[q] * kPbLine + j;
  xperm[(q % nb) * w + (q / nb) * kPbLine + j] = c < m ? x[c] : (T)0;
}

template <class T, int U>
__global__ __launch_bounds__(kPbThreads) void pb_pass1_kernel(const int64_t *__restrict__ piece_ptr,
                                                              const uint16_t *__restrict__ col16,
                                                              const T *__restrict__ pval, const T *__restrict__ xperm,
                                                              T *__restrict__ prod, int64_t np, int64_t nb, int w,
                                                              int lab) {
  __shared__ __attribute__((aligned(16))) T xs[kPbSliceBytes / sizeof(T)];
  const int64_t b = blockIdx.x;
  {
    constexpr int PER = 16 / (int)sizeof(T);  // w is a multiple of 32 columns: whole 128-bit pieces
    const raw16 *src = reinterpret_cast<const raw16 *>(xperm + b * (int64_t)w);
    for (int j = threadIdx.x; j < w / PER; j += kPbThreads) reinterpret_cast<raw16 *>(xs)[j] = src[j];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t stride = (int64_t)gridDim.y * kPbWaves;
  int64_t p = (int64_t)blockIdx.y * kPbWaves + wave;
  int64_t s = 0, e = 0;
  if (p < np) { s = piece_ptr[p * nb + b]; e = piece_ptr[p * nb + b + 1]; }
  while (p < np) {
    const int64_t pn = p + stride;
    int64_t sn = 0, en = 0;
    if (pn < np) { sn = piece_ptr[pn * nb + b]; en = piece_ptr[pn * nb + b + 1]; }  // in flight during this piece
    for (int64_t i = s + lane; i < e; i += 64 * U) {
      uint16_t c[U];
      T v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int64_t ii = i + 64 * u;
        const bool ok = ii < e;
        c[u] = ok ? ntload(&col16[ii]) : (uint16_t)0;
        v[u] = ok ? ntload(&pval[ii]) : (T)0;
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int64_t ii = i + 64 * u;
        if (ii < e) {
          const T pr = lab == 2 ? v[u] : (T)(v[u] * xs[c[u]]);  // lab 2: no LDS gather
          if (lab != 1 || pr == (T)12345) __builtin_nontemporal_store(pr, &prod[ii]);  // lab 1: no store
        }
      }
    }
    p = pn; s = sn; e = en;
  }
}

// Adds v into ys[row] for the lanes of one wavefront whose rows come in runs (entries of a piece are sorted by
// row): a run is folded inside the wavefront and only its last lane touches LDS, so a heavy row costs one
// atomic per 64 entries instead of 64 colliding ones.  row < 0: lane holds nothing.
template <class T>
__device__ __forceinline__ void pb_seg_add(T *ys, int row, T v, int lane) {
  const int rprev = wshfl_up(row, 1);
  const bool head = lane == 0 || rprev != row;
  const unsigned long long heads = __ballot(head);
  if (heads != ~0ull) {  // some lane continues its neighbour's row: segmented inclusive scan over the wavefront
    const unsigned long long below = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    const int head_pos = 63 - __builtin_clzll(below);  // lane 0 is always a head, so below != 0
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const T up = wshfl_up(v, d);
      if (lane - d >= head_pos) v = (T)(v + up);
    }
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    if (row >= 0 && tail) atomicAdd(&ys[row], v);
  } else if (row >= 0) {
    atomicAdd(&ys[row], v);
  }
}

template <class T, int U>
__global__ __launch_bounds__(kPbThreads) void pb_pass2_kernel(const int64_t *__restrict__ piece_ptr,
                                                              const int64_t *__restrict__ panel_row,
                                                              const uint16_t *__restrict__ row16,
                                                              const T *__restrict__ prod, T *__restrict__ y,
                                                              int64_t nb, int lab) {
  __shared__ T ys[kPbSliceBytes / sizeof(T)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t p = blockIdx.x;
  const int64_t r0 = panel_row[p];
  const int rows = (int)(panel_row[p + 1] - r0);
  const int64_t s = piece_ptr[p * nb], e = piece_ptr[(p + 1) * nb];
  for (int j = tid; j < rows; j += kPbThreads) ys[j] = (T)0;
  __syncthreads();
  for (int64_t i0 = s; i0 < e; i0 += (int64_t)kPbThreads * U) {
    int r[U];
    T v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t ii = i0 + (int64_t)u * kPbThreads + tid;
      const bool ok = ii < e;
      r[u] = ok ? (int)ntload(&row16[ii]) : -1;
      v[u] = ok ? ntload(&prod[ii]) : (T)0;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (lab == 0) pb_seg_add<T>(ys, r[u], v[u], lane);
      else if (lab == 1) { if (r[u] == 12345 && v[u] == (T)77) ys[0] = v[u]; }  // lab: stream only
      else if (r[u] >= 0) atomicAdd(&ys[r[u]], v[u]);                            // lab: plain atomics
    }
  }
  __syncthreads();
  for (int j = tid; j < rows; j += kPbThreads) y[r0 + j] = ys[j];
}

// Decides and, if worthwhile, builds the piece copy.  One-time host synchronisations.
template <class T>
int pb_prepare(futhark_context *ctx, const CsrBody<T> *a) {
  a->pb_state = -1;
  if (ctx->opt_spmv_pb <= 0 || !a->col32 || a->nnz == 0 || a->n == 0 || a->tile <= 0) return 0;
  const bool force = ctx->opt_spmv_pb > 1;  // tests: any record (2); 1: records that qualify by size and shape
  // slice sizes (overridable for tests so that small records have many panels and blocks)
  const int64_t slice = kPbSliceBytes / (int64_t)sizeof(T);
  int64_t R = ctx->opt_spmv_pb_rows > 0 ? ctx->opt_spmv_pb_rows : slice;
  int64_t lpb = ctx->opt_spmv_pb_lines > 0 ? ctx->opt_spmv_pb_lines : slice / kPbLine;  // lines per block
  int64_t E = ctx->opt_spmv_pb_entries > 0 ? ctx->opt_spmv_pb_entries : kPbPanelEntries;
  if (R > slice) R = slice;
  if (R > 65536) R = 65536;  // 16-bit row offsets
  if (lpb * kPbLine > slice) lpb = slice / kPbLine;
  if (lpb * kPbLine > 65536) lpb = 65536 / kPbLine;
  if (E < a->tile) E = a->tile;
  const int64_t nl = cdiv(a->m, kPbLine);
  const int64_t nb = cdiv(nl, lpb);
  const int64_t w = cdiv(nl, nb) * kPbLine;  // columns per block actually used (<= lpb * 32)
  if (!force) {
    // the gather has to be the problem: 4-byte values (16 B moved per entry against 8; for 8-byte values the
    // one-pass paths are on par), x beyond the L2s, most tiles spanning a wide column range
    if (sizeof(T) != 4 || a->m * (int64_t)sizeof(T) < kReorderMinBytes) return 0;
    const int64_t np_est = cdiv(a->n, R) + cdiv(a->nnz, E);
    if (a->nnz / (np_est * nb) < kPbMinPiece) return 0;
    unsigned long long *stat = reinterpret_cast<unsigned long long *>(ctx->d_small);
    SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
    hipLaunchKernelGGL(reorder_wide_tiles_kernel, dim3((unsigned)cdiv(a->ntiles, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const int32_t *)a->col32, a->nnz, a->ntiles, a->tile, stat);
    SAMD_LAUNCH_CHECK(ctx);
    int64_t wide = 0;
    SAMD_TRY(read_small(ctx, ctx->d_small, 1, &wide));
    if (wide * 2 < a->ntiles) return 0;  // banded / stencil-like: the gather is already local
  }
  // ---- row panels: cut where E more entries have started and at every R-th row (host: a few thousand values)
  std::vector<int64_t> lb((size_t)a->ntiles + 1);
  SAMD_HIP(ctx, hipMemcpyAsync(lb.data(), a->tile_lb, sizeof(int64_t) * lb.size(), hipMemcpyDeviceToHost, ctx->stream));
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<int64_t> cuts;
  cuts.push_back(0);
  {
    const int64_t et = E / a->tile > 0 ? E / a->tile : 1;
    int64_t next_entry_cut = et;  // in tiles
    int64_t cur = 0;
    while (cur < a->n) {
      int64_t nxt = cur + R < a->n ? cur + R : a->n;  // at most R rows
      while (next_entry_cut <= a->ntiles && lb[(size_t)next_entry_cut] <= cur) next_entry_cut += et;
      if (next_entry_cut <= a->ntiles && lb[(size_t)next_entry_cut] < nxt) nxt = lb[(size_t)next_entry_cut];
      cuts.push_back(nxt);
      cur = nxt;
    }
  }
  const int64_t np = (int64_t)cuts.size() - 1;
  const int64_t pieces = np * nb;
  if (pieces >= (1ll << 31) || a->nnz >= (int64_t)0xffff0000LL) return 0;
  if (!force && a->nnz / pieces < kPbMinPiece) return 0;

  Tmp<int64_t> panel_row(ctx), piece_ptr(ctx);
  Tmp<uint32_t> line_of_rank(ctx);
  Tmp<uint16_t> col16(ctx), row16(ctx);
  Tmp<T> pval(ctx), prod(ctx), xperm(ctx);
  SAMD_TRY(panel_row.get(np + 1));
  SAMD_TRY(piece_ptr.get(pieces + 1));
  SAMD_TRY(line_of_rank.get(nl));
  SAMD_HIP(ctx, hipMemcpyAsync(panel_row.p, cuts.data(), sizeof(int64_t) * (size_t)(np + 1), hipMemcpyHostToDevice,
                               ctx->stream));
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));  // cuts is host memory that goes away
  int *sort_err = reinterpret_cast<int *>(ctx->d_small + 2);
  SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small + 2, 0, sizeof(int64_t), ctx->stream));
  {
    // ---- lines ranked by popularity
    Tmp<uint32_t> counts(ctx), keys_b(ctx), pay_a(ctx), pay_b(ctx), rank_of_line(ctx);
    SAMD_TRY(counts.get(nl));
    SAMD_TRY(keys_b.get(nl));
    SAMD_TRY(pay_a.get(nl));
    SAMD_TRY(pay_b.get(nl));
    SAMD_TRY(rank_of_line.get(nl));
    SAMD_HIP(ctx, hipMemsetAsync(counts.p, 0, sizeof(uint32_t) * (size_t)nl, ctx->stream));
    {
      const int64_t want = cdiv(a->nnz, kBlock), cap = (int64_t)ctx->num_cu * 32;
      hipLaunchKernelGGL(pb_line_count_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(kBlock), 0, ctx->stream,
                         (const int32_t *)a->col32, a->nnz, counts.p);
      SAMD_LAUNCH_CHECK(ctx);
    }
    const unsigned lgrid = (unsigned)cdiv(nl, kBlock);
    hipLaunchKernelGGL(reorder_keys_kernel, dim3(lgrid), dim3(kBlock), 0, ctx->stream, (const uint32_t *)counts.p, nl,
                       counts.p);  // in place: ~count, ascending = most popular first, ties by line number
    SAMD_LAUNCH_CHECK(ctx);
    uint32_t *ks = nullptr;
    const uint32_t *perm = nullptr;
    SAMD_TRY((radix_sort_pairs<uint32_t, uint32_t>(ctx, counts.p, keys_b.p, (const uint32_t *)nullptr, pay_a.p, pay_b.p,
                                                    nl, 0, 32, &ks, &perm, sort_err)));
    SAMD_HIP(ctx, hipMemcpyAsync(line_of_rank.p, perm, sizeof(uint32_t) * (size_t)nl, hipMemcpyDeviceToDevice,
                                 ctx->stream));
    hipLaunchKernelGGL(pb_rank_kernel, dim3(lgrid), dim3(kBlock), 0, ctx->stream, (const uint32_t *)line_of_rank.p, nl,
                       rank_of_line.p);
    SAMD_LAUNCH_CHECK(ctx);
    // ---- entries sorted by piece (stable: a piece keeps the record's (row, column) order)
    Tmp<uint32_t> ek_a(ctx), ek_b(ctx), ep_a(ctx), ep_b(ctx);
    SAMD_TRY(ek_a.get(a->nnz));
    SAMD_TRY(ek_b.get(a->nnz));
    SAMD_TRY(ep_a.get(a->nnz));
    SAMD_TRY(ep_b.get(a->nnz));
    const unsigned egrid = (unsigned)cdiv(a->nnz, kBlock);
    hipLaunchKernelGGL(pb_key_kernel, dim3(egrid), dim3(kBlock), 0, ctx->stream, (const int64_t *)a->row_ptr,
                       (const int32_t *)a->col32, a->nnz, (const int64_t *)a->tile_lb, a->tile,
                       (const int64_t *)panel_row.p, np, (const uint32_t *)rank_of_line.p, nb, ek_a.p);
    SAMD_LAUNCH_CHECK(ctx);
    uint32_t *eks = nullptr;
    const uint32_t *eperm = nullptr;
    SAMD_TRY((radix_sort_pairs<uint32_t, uint32_t>(ctx, ek_a.p, ek_b.p, (const uint32_t *)nullptr, ep_a.p, ep_b.p, a->nnz,
                                                    0, bits_for(pieces), &eks, &eperm, sort_err)));
    hipLaunchKernelGGL(pb_piece_ptr_kernel, dim3((unsigned)cdiv(pieces + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const uint32_t *)eks, a->nnz, pieces, piece_ptr.p);
    SAMD_LAUNCH_CHECK(ctx);
    SAMD_TRY(col16.get(a->nnz));
    SAMD_TRY(row16.get(a->nnz));
    SAMD_TRY(pval.get(a->nnz));
    hipLaunchKernelGGL((pb_build_kernel<T>), dim3(egrid), dim3(kBlock), 0, ctx->stream, (const int64_t *)a->row_ptr,
                       (const int32_t *)a->col32, (const T *)a->vals, a->nnz, (const int64_t *)a->tile_lb, a->tile,
                       (const int64_t *)panel_row.p, (const uint32_t *)rank_of_line.p, nb, (const uint32_t *)eks, eperm,
                       col16.p, row16.p, pval.p);
    SAMD_LAUNCH_CHECK(ctx);
    int64_t flag[3];
    SAMD_TRY(read_small(ctx, ctx->d_small, 3, flag));  // also orders the temporaries' release after their use
    if ((int)(flag[2] & 0xffffffff) != 0) return set_error(ctx, "internal: radix sort look-back timed out");
  }
  SAMD_TRY(prod.get(a->nnz));
  SAMD_TRY(xperm.get(nb * w));
  a->pb_np = np;
  a->pb_nb = nb;
  a->pb_w = w;
  a->pb_nl = nl;
  a->pb_panel_row = panel_row.release();
  a->pb_piece_ptr = piece_ptr.release();
  a->pb_line_of_rank = line_of_rank.release();
  a->pb_col16 = col16.release();
  a->pb_row16 = row16.release();
  a->pb_val = pval.release();
  a->pb_prod = prod.release();
  a->pb_x = xperm.release();
  a->pb_state = 1;
  return 0;
}

template <class T>
int spmv_pb(futhark_context *ctx, const CsrBody<T> *a, const T *x, T *y) {
  hipLaunchKernelGGL((pb_xperm_kernel<T>), dim3((unsigned)cdiv(a->pb_nl * kPbLine, kBlock)), dim3(kBlock), 0, ctx->stream,
                     x, a->m, (const uint32_t *)a->pb_line_of_rank, a->pb_nl, a->pb_nb, a->pb_w, a->pb_x);
  SAMD_LAUNCH_CHECK(ctx);
  // pass 1: at least ~4 workgroups per CU's worth of work units so that the tail is short
  int64_t split = cdiv((int64_t)ctx->num_cu * 4, a->pb_nb);
  if (split < 1) split = 1;
  if (split > cdiv(a->pb_np, kPbWaves)) split = cdiv(a->pb_np, kPbWaves);
  if (split > 65535) split = 65535;
  hipLaunchKernelGGL((pb_pass1_kernel<T, 4>), dim3((unsigned)a->pb_nb, (unsigned)split), dim3(kPbThreads), 0, ctx->stream,
                     (const int64_t *)a->pb_piece_ptr, (const uint16_t *)a->pb_col16, (const T *)a->pb_val,
                     (const T *)a->pb_x, a->pb_prod, a->pb_np, a->pb_nb, (int)a->pb_w, (int)((ctx->opt_ablate >> 2) & 3));
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL((pb_pass2_kernel<T, 8>), dim3((unsigned)a->pb_np), dim3(kPbThreads), 0, ctx->stream,
                     (const int64_t *)a->pb_piece_ptr, (const int64_t *)a->pb_panel_row, (const uint16_t *)a->pb_row16,
                     (const T *)a->pb_prod, y, a->pb_nb, (int)(ctx->opt_ablate & 3));
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

}  // namespace samd
