// spmv_2d_lab.hpp -- LAB MATERIAL, not compiled into the product since round 3 (it was sparse_amd/csrc/spmv_2d.hpp,
// option spmv_2d, path 4): measured 2.27-2.45 ms on C3 against 0.7 ms for the two-pass product
// (profiles/r02_2d_blocked_lab_c3.txt) and never chosen by itself.  Kept for the record of what was tried.
//
// one-pass SpMV for gather-bound power-law records with BOTH slices in LDS
// (BASELINE configs[2]; sr.smvm, compressed.fut:134-146).
//
// The two-pass product (spmv_pb.hpp) keeps every gather inside the CU but pays for it with a product stream:
// 16 bytes move per entry against 8 algorithmic ones, and its second pass is bound by LDS atomics.  A power-law
// matrix offers something better: once the COLUMNS are renumbered by popularity (32-column lines ranked by
// sampled entry count, as in spmv_pb.hpp, but hot lines kept TOGETHER), most entries fall into a small number
// of heavily populated rectangles.  Cut the matrix into PIECES = (panel of R = 8192 consecutive rows) x (block
// of W = 16384 renumbered columns): a piece's slice of x (64 KB of f32) and the panel's slice of y (64 KB of
// f64 accumulators) fit one CU's LDS together.  For RMAT scale 24 about 89 % of the entries sit in pieces of
// >= 512 entries ("dense" pieces, ~43 K of the 2 M pieces).  A workgroup owns a UNIT = <= 256 K consecutive
// entries of one panel's copy, stored [panel][dense pieces by block][the rest by block]:
//   * per dense piece: the block's x slice -> LDS (64 KB from L2 / Infinity Cache: x is only 64 MB), then the
//     piece's entries are streamed (2 B column + 2 B row + 4 B value), gathered from LDS, added into the y slice
//     by LDS atomics (runs of equal rows folded inside the wavefront first, as in the two-pass product);
//   * the rest (entries of thinly populated pieces, ~11 %): gathered from the renumbered x in global memory;
//   * the y slice leaves once: stored (a panel with one unit) or added to a small f64 buffer by hardware
//     atomics (the few heavy panels that are cut into several units; `tb_finish_kernel` converts those).
// Rows are NOT renumbered: y needs no permutation, a panel's rows are consecutive.  Per product: one pass over
// x (renumbering, 64 MB), the unit kernel, the finish kernel.  8 bytes per entry + the slice loads.
// The row sum is formed in f64 in another order than the oracle's left-to-right f32: within the 1e-5 bar
// (closer to the exact sum), exact for i32.  The order in which LDS / global atomics land is not fixed, so the
// last bit of an f32 result can differ between two runs (as with the two-pass product).
#pragma once

#include <vector>

#include "spmv_pb.hpp"

namespace samd {

constexpr int kTbThreads = 1024;                 // one workgroup per CU (the slices take 128 KB of its LDS)
constexpr int kTbWaves = kTbThreads / 64;
constexpr int kTbXBytes = 64 << 10;              // slice of x
constexpr int kTbRows = 8192;                    // rows per panel: 64 KB of 8-byte accumulators
constexpr int64_t kTbUnitEntries = 256 << 10;    // a panel is cut into units of about this many entries
constexpr int64_t kTbDense = 512;                // a piece estimated to hold at least this many entries gets its x slice in LDS
constexpr int kTbSample = 8;                     // the estimate counts every 8th entry
constexpr int kTbMaxDesc = 1024;                 // dense-piece descriptors staged in LDS at a time
constexpr int64_t kTbMinDensity = 32;            // auto mode: mean entries per piece below this -> the pieces are all thin

struct TbPlan {
  int64_t np = 0, nb = 0, W = 0, R = 0, nl = 0, units = 0, nslots = 0, ndesc = 0;
  int64_t *panel_ptr = nullptr;    // [np + 1] entry range of every panel in the copy
  int64_t *sparse_ptr = nullptr;   // [np]     where the thin pieces' entries start inside the panel
  int64_t *desc_ptr = nullptr;     // [np + 1] dense pieces of every panel
  int32_t *d_blk = nullptr;        // [ndesc]
  int64_t *d_s = nullptr, *d_e = nullptr;  // [ndesc] entry range of a dense piece
  int64_t *unit_ptr = nullptr;     // [np + 1] first unit of every panel
  int32_t *slot = nullptr;         // [np] index into `part` for a panel of several units, else -1
  int32_t *panel_of_slot = nullptr;  // [nslots]
  uint32_t *line_of_rank = nullptr;  // [nl]
  uint16_t *col16 = nullptr, *row16 = nullptr, *blk16 = nullptr;  // [nnz]
  void *val = nullptr;             // [nnz] T
  void *xperm = nullptr;           // [nb * W] T, rewritten by every product
  void *part = nullptr;            // [nslots * R] accumulators, zero between products
};

inline void tb_free(futhark_context *ctx, void *p) {
  TbPlan *t = reinterpret_cast<TbPlan *>(p);
  if (!t) return;
  dfree(ctx, t->panel_ptr); dfree(ctx, t->sparse_ptr); dfree(ctx, t->desc_ptr); dfree(ctx, t->d_blk);
  dfree(ctx, t->d_s); dfree(ctx, t->d_e); dfree(ctx, t->unit_ptr); dfree(ctx, t->slot); dfree(ctx, t->panel_of_slot);
  dfree(ctx, t->line_of_rank); dfree(ctx, t->col16); dfree(ctx, t->row16); dfree(ctx, t->blk16);
  dfree(ctx, t->val); dfree(ctx, t->xperm); dfree(ctx, t->part);
  delete t;
}

// ---- set-up kernels
// estimated entries per piece from every S-th entry (classification only: a piece near the threshold may go
// either way, which costs a little speed, never correctness)
__global__ __launch_bounds__(kBlock) void tb_est_kernel(const int64_t *__restrict__ row_ptr,
                                                        const int32_t *__restrict__ col, int64_t nnz,
                                                        const int64_t *__restrict__ tile_lb, int64_t tile, int64_t stride,
                                                        const uint32_t *__restrict__ rank_of_line, int64_t R, int64_t lpb,
                                                        int64_t nb, uint32_t *__restrict__ est) {
  const int64_t samples = (nnz + stride - 1) / stride;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < samples; i += (int64_t)gridDim.x * kBlock) {
    const int64_t e = i * stride;
    const int64_t r = find_row(row_ptr, tile_lb, tile, e);
    const int64_t q = rank_of_line[col[e] / kPbLine];
    atomicAdd(&est[(r / R) * nb + q / lpb], 1u);
  }
}

// key = (panel * 2 + thin) * nb + block: a panel's dense pieces first, by block, then its thin pieces, by block
__global__ __launch_bounds__(kBlock) void tb_key_kernel(const int64_t *__restrict__ row_ptr,
                                                        const int32_t *__restrict__ col, int64_t nnz,
                                                        const int64_t *__restrict__ tile_lb, int64_t tile,
                                                        const uint32_t *__restrict__ rank_of_line, int64_t R, int64_t lpb,
                                                        int64_t nb, const uint32_t *__restrict__ est, uint32_t dense_min,
                                                        uint32_t *__restrict__ keys) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= nnz) return;
  const int64_t r = find_row(row_ptr, tile_lb, tile, e);
  const int64_t p = r / R, b = (int64_t)rank_of_line[col[e] / kPbLine] / lpb;
  const int64_t thin = est[p * nb + b] < dense_min ? 1 : 0;
  keys[e] = (uint32_t)((p * 2 + thin) * nb + b);
}

template <class T>
__global__ __launch_bounds__(kBlock) void tb_build_kernel(const int64_t *__restrict__ row_ptr,
                                                          const int32_t *__restrict__ col, const T *__restrict__ val,
                                                          int64_t nnz, const int64_t *__restrict__ tile_lb, int64_t tile,
                                                          const uint32_t *__restrict__ rank_of_line, int64_t R, int64_t W,
                                                          const uint32_t *__restrict__ perm,
                                                          uint16_t *__restrict__ col16, uint16_t *__restrict__ row16,
                                                          uint16_t *__restrict__ blk16, T *__restrict__ pval) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nnz) return;
  const int64_t src = perm ? (int64_t)perm[i] : i;  // no sort pass ran (one piece): identity
  const int64_t r = find_row(row_ptr, tile_lb, tile, src);
  const int32_t c = col[src];
  const int64_t cnew = (int64_t)rank_of_line[c / kPbLine] * kPbLine + (c % kPbLine);
  col16[i] = (uint16_t)(cnew % W);
  blk16[i] = (uint16_t)(cnew / W);
  row16[i] = (uint16_t)(r % R);
  pval[i] = val[src];
}

// x in rank order: line of rank q occupies elements [32 q, 32 q + 32), so block b = columns [b W, (b + 1) W)
template <class T>
__global__ __launch_bounds__(kBlock) void tb_xperm_kernel(const T *__restrict__ x, int64_t m,
                                                          const uint32_t *__restrict__ line_of_rank, int64_t nl,
                                                          T *__restrict__ xperm) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t q = t / kPbLine, j = t % kPbLine;
  if (q >= nl) return;
  const int64_t c = (int64_t)line_of_rank[q] * kPbLine + j;
  xperm[t] = c < m ? x[c] : (T)0;
}

__device__ __forceinline__ void tb_global_add(double *p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void tb_global_add(int32_t *p, int32_t v) { atomicAdd(p, v); }

// ---- the product
template <class T, int U>
__global__ __launch_bounds__(kTbThreads) void tb_kernel(
    const int64_t *__restrict__ panel_ptr, const int64_t *__restrict__ sparse_ptr, const int64_t *__restrict__ desc_ptr,
    const int32_t *__restrict__ d_blk, const int64_t *__restrict__ d_s, const int64_t *__restrict__ d_e,
    const int64_t *__restrict__ unit_ptr, const int32_t *__restrict__ slot, const uint16_t *__restrict__ col16,
    const uint16_t *__restrict__ row16, const uint16_t *__restrict__ blk16, const T *__restrict__ val,
    const T *__restrict__ xperm, T *__restrict__ y, typename PbAcc<T>::type *__restrict__ part, int64_t np, int64_t n,
    int W, int R, int fold_max, int lab) {
  typedef typename PbAcc<T>::type acc_t;
  __shared__ __attribute__((aligned(16))) T xs[kTbXBytes / sizeof(T)];
  __shared__ acc_t ys[kTbRows];
  __shared__ int s_blk[kTbMaxDesc], s_s[kTbMaxDesc], s_e[kTbMaxDesc];  // dense pieces clipped to this unit, relative to us
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // unit -> panel: the largest p with unit_ptr[p] <= unit (every panel has at least one unit)
  const int64_t unit = blockIdx.x;
  int64_t lo = 0, hi = np;
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (unit_ptr[mid] <= unit) lo = mid; else hi = mid;
  }
  const int64_t p = lo;
  const int64_t nu = unit_ptr[p + 1] - unit_ptr[p], k = unit - unit_ptr[p];
  const int64_t ps = panel_ptr[p], pe = panel_ptr[p + 1];
  int64_t per = ((pe - ps + nu - 1) / nu + 63) & ~(int64_t)63;
  if (per < 64) per = 64;
  int64_t us = ps + k * per, ue = us + per;
  if (us > pe) us = pe;
  if (ue > pe) ue = pe;
  const int rows = (int)(n - p * R < (int64_t)R ? n - p * R : (int64_t)R);

  for (int j = tid; j < R; j += kTbThreads) ys[j] = (acc_t)0;

  // one chunk = 64 U consecutive entries of one wavefront; loads are never predicated (clamped offsets instead)
  auto run = [&](int64_t s, int64_t e, bool dense) {
    for (int64_t c0 = s + (int64_t)wave * (64 * U); c0 < e; c0 += (int64_t)kTbWaves * (64 * U)) {
      const int cnt = (int)(e - c0 < (int64_t)(64 * U) ? e - c0 : (int64_t)(64 * U));
      int c[U], r[U], b[U];
      T v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int o = lane + 64 * u;
        const int oc = o < cnt ? o : cnt - 1;
        c[u] = (int)ntload(col16 + c0 + oc);
        r[u] = (int)ntload(row16 + c0 + oc);
        v[u] = ntload(val + c0 + oc);
        b[u] = dense ? 0 : (int)ntload(blk16 + c0 + oc);
      }
      if (dense) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          if (64 * u >= cnt) break;  // uniform
          const T pr = (lab & 4) ? v[u] : (T)(v[u] * xs[c[u]]);                       // lab 4: no LDS gather
          if (lab & 8) { if (pr == (T)12345 && r[u] == 77) ys[0] = (acc_t)pr; }        // lab 8: no accumulation
          else if (lab & 16) { if (lane + 64 * u < cnt) atomicAdd(&ys[r[u]], (acc_t)pr); }  // lab 16: plain atomics, no fold
          else pb_seg_add<T>(ys, lane + 64 * u < cnt ? r[u] : -1, pr, lane, fold_max);
        }
      } else {
        T xv[U];
#pragma unroll
        for (int u = 0; u < U; u++) xv[u] = xperm[(int64_t)b[u] * W + c[u]];  // clamped lanes repeat a valid address
#pragma unroll
        for (int u = 0; u < U; u++) {
          if (64 * u >= cnt) break;
          if (lane + 64 * u < cnt) atomicAdd(&ys[r[u]], (acc_t)(T)(v[u] * xv[u]));
        }
      }
    }
  };

  // ---- dense pieces of the panel that overlap [us, ue)
  const int64_t dlo = desc_ptr[p], dhi = desc_ptr[p + 1];
  bool first = true;
  for (int64_t d0 = dlo; d0 < dhi; d0 += kTbMaxDesc) {
    const int nd = (int)(dhi - d0 < (int64_t)kTbMaxDesc ? dhi - d0 : (int64_t)kTbMaxDesc);
    __syncthreads();  // the previous batch's readers are done (first batch: ys is zeroed)
    if (tid < nd) {
      int64_t s = d_s[d0 + tid], e = d_e[d0 + tid];
      s = s > us ? s : us;
      e = e < ue ? e : ue;
      const bool ok = e > s;
      s_blk[tid] = d_blk[d0 + tid];
      s_s[tid] = ok ? (int)(s - us) : 0;
      s_e[tid] = ok ? (int)(e - us) : 0;
    }
    __syncthreads();
    for (int i = 0; i < nd; i++) {
      const int s = s_s[i], e = s_e[i];
      if (e <= s || (lab & 2)) continue;  // uniform: the piece lies outside this unit (lab 2: no dense pieces)
      if (!first) __syncthreads();  // everyone has finished gathering from the previous slice
      first = false;
      {
        const raw16 *src = reinterpret_cast<const raw16 *>(xperm + (int64_t)s_blk[i] * W);
        const int pieces = W * (int)sizeof(T) / 16;
        if (!(lab & 32))  // lab 32: no slice load
          for (int j = tid; j < pieces; j += kTbThreads) reinterpret_cast<raw16 *>(xs)[j] = src[j];
      }
      __syncthreads();
      run(us + s, us + e, true);
    }
  }
  if (first) __syncthreads();  // no dense piece ran: ys must be zeroed before the adds below

  // ---- the thin pieces' entries of this unit
  {
    const int64_t sp = sparse_ptr[p];
    const int64_t s = sp > us ? sp : us;
    if (s < ue && !(lab & 1)) run(s, ue, false);  // lab 1: no thin pieces
  }
  __syncthreads();

  // ---- the slice of y leaves
  const int sl = slot[p];
  if (sl < 0) {
    for (int j = tid; j < rows; j += kTbThreads) __builtin_nontemporal_store((T)ys[j], &y[p * R + j]);
  } else {
    acc_t *dst = part + (int64_t)sl * R;
    for (int j = tid; j < rows; j += kTbThreads) {
      const acc_t v = ys[j];
      if (v != (acc_t)0) tb_global_add(&dst[j], v);
    }
  }
}

// panels cut into several units: their sums are in `part`; convert, store, and leave zeros for the next product
template <class T>
__global__ __launch_bounds__(kBlock) void tb_finish_kernel(typename PbAcc<T>::type *__restrict__ part,
                                                           const int32_t *__restrict__ panel_of_slot, int64_t n, int R,
                                                           T *__restrict__ y) {
  typedef typename PbAcc<T>::type acc_t;
  const int64_t base = (int64_t)blockIdx.x * R, r0 = (int64_t)panel_of_slot[blockIdx.x] * R;
  for (int j = threadIdx.x; j < R; j += kBlock) {
    if (r0 + j >= n) break;
    const acc_t v = part[base + j];
    part[base + j] = (acc_t)0;
    y[r0 + j] = (T)v;
  }
}

template <class V>
int tb_upload(futhark_context *ctx, V **dst, const std::vector<V> &src) {
  V *d = nullptr;
  SAMD_TRY(dalloc(ctx, &d, (int64_t)src.size()));
  *dst = d;  // owned by the plan from here on
  if (!src.empty())
    SAMD_HIP(ctx, hipMemcpyAsync(d, src.data(), sizeof(V) * src.size(), hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

// Decides and, if worthwhile, builds the copy.  One-time host synchronisations.  `vetted` as in pb_prepare.
template <class T>
int tb_prepare(futhark_context *ctx, const CsrBody<T> *a, bool vetted) {
  typedef typename PbAcc<T>::type acc_t;
  a->tb_state = -1;
  // Not chosen by itself: measured on C3 it loses to the two-pass product (2.27 against 1.55 ms; DESIGN 3.1 has the
  // breakdown).  Option 1 lets it take the records reorder_prepare has vetted, 2 forces it (tests, labs).
  if (ctx->opt_spmv_2d <= 0 || (ctx->opt_spmv_2d == 1 && !vetted)) return 0;
  if (!a->col32 || a->nnz == 0 || a->n == 0 || a->tile <= 0) return 0;
  const bool force = ctx->opt_spmv_2d > 1;
  int64_t W = kTbXBytes / (int64_t)sizeof(T), R = kTbRows, E = kTbUnitEntries, dense_min = kTbDense;
  if (ctx->opt_spmv_2d_cols > 0 && ctx->opt_spmv_2d_cols < W) W = cdiv(ctx->opt_spmv_2d_cols, kPbLine) * kPbLine;  // tests
  if (ctx->opt_spmv_2d_rows > 0 && ctx->opt_spmv_2d_rows < R) R = ctx->opt_spmv_2d_rows;
  if (ctx->opt_spmv_2d_entries > 0) E = ctx->opt_spmv_2d_entries;
  if (ctx->opt_spmv_2d_dense > 0) dense_min = ctx->opt_spmv_2d_dense;
  if (W > 65536) W = 65536;  // 16-bit columns inside a block
  if (E < 64) E = 64;
  const int64_t nl = cdiv(a->m, kPbLine), lpb = W / kPbLine;
  const int64_t nb = cdiv(nl, lpb), np = cdiv(a->n, R);
  if (nb > 65536 || np * 2 * nb >= (1ll << 31) || a->nnz >= (int64_t)0xffff0000LL) return 0;  // 16-bit blocks, 32-bit keys
  if (!force) {
    if (sizeof(T) != 4 || a->m * (int64_t)sizeof(T) < kPbMinBytes) return 0;
    if (a->nnz / (np * nb) < kTbMinDensity) return 0;
  }
  if (!force && !vetted) {
    unsigned long long *stat = reinterpret_cast<unsigned long long *>(ctx->d_small);
    SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
    hipLaunchKernelGGL(reorder_wide_tiles_kernel, dim3((unsigned)cdiv(a->ntiles, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const int32_t *)a->col32, a->nnz, a->ntiles, a->tile, stat);
    SAMD_LAUNCH_CHECK(ctx);
    int64_t wide = 0;
    SAMD_TRY(read_small(ctx, ctx->d_small, 1, &wide));
    if (wide * 2 < a->ntiles) return 0;  // banded / stencil-like: the gather is already local
  }
  const int64_t pieces2 = np * 2 * nb;

  Tmp<uint32_t> line_of_rank(ctx);
  Tmp<uint16_t> col16(ctx), row16(ctx), blk16(ctx);
  Tmp<T> pval(ctx), xperm(ctx);
  Tmp<int64_t> piece_ptr(ctx);
  SAMD_TRY(line_of_rank.get(nl));
  SAMD_TRY(piece_ptr.get(pieces2 + 1));
  int *sort_err = reinterpret_cast<int *>(ctx->d_small + 2);
  SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small + 2, 0, sizeof(int64_t), ctx->stream));
  {
    // ---- lines ranked by popularity (as in pb_prepare)
    Tmp<uint32_t> counts(ctx), keys_b(ctx), pay_a(ctx), pay_b(ctx), rank_of_line(ctx), est(ctx);
    SAMD_TRY(counts.get(nl));
    SAMD_TRY(keys_b.get(nl));
    SAMD_TRY(pay_a.get(nl));
    SAMD_TRY(pay_b.get(nl));
    SAMD_TRY(rank_of_line.get(nl));
    SAMD_TRY(est.get(np * nb));
    SAMD_HIP(ctx, hipMemsetAsync(counts.p, 0, sizeof(uint32_t) * (size_t)nl, ctx->stream));
    SAMD_HIP(ctx, hipMemsetAsync(est.p, 0, sizeof(uint32_t) * (size_t)(np * nb), ctx->stream));
    const int64_t cap = (int64_t)ctx->num_cu * 32;
    {
      int64_t stride = a->nnz >> 24;  // about 2^24 samples
      if (stride < 1) stride = 1;
      if (stride % 2 == 0) stride++;
      const int64_t want = cdiv(cdiv(a->nnz, stride), kBlock);
      hipLaunchKernelGGL(pb_line_count_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(kBlock), 0, ctx->stream,
                         (const int32_t *)a->col32, a->nnz, stride, counts.p);
      SAMD_LAUNCH_CHECK(ctx);
    }
    const unsigned lgrid = (unsigned)cdiv(nl, kBlock);
    hipLaunchKernelGGL(reorder_keys_kernel, dim3(lgrid), dim3(kBlock), 0, ctx->stream, (const uint32_t *)counts.p, nl,
                       counts.p);
    SAMD_LAUNCH_CHECK(ctx);
    uint32_t *ks = nullptr;
    const uint32_t *perm = nullptr;
    SAMD_TRY((radix_sort_pairs<uint32_t, uint32_t>(ctx, counts.p, keys_b.p, (const uint32_t *)nullptr, pay_a.p, pay_b.p,
                                                    nl, 0, 32, &ks, &perm, sort_err)));
    SAMD_HIP(ctx, hipMemcpyAsync(line_of_rank.p, perm, sizeof(uint32_t) * (size_t)nl, hipMemcpyDeviceToDevice,
                                 ctx->stream));
    hipLaunchKernelGGL(pb_rank_kernel, dim3(lgrid), dim3(kBlock), 0, ctx->stream, (const uint32_t *)perm, nl, (int64_t)1,
                       nl, 0, rank_of_line.p, line_of_rank.p);  // dealt: place = rank
    SAMD_LAUNCH_CHECK(ctx);
    // ---- estimated entries per piece, then the entries sorted by (panel, thin, block); stable: a piece keeps
    // the record's (row, column) order, so the entries of one row are consecutive inside a piece
    const int64_t sample = a->nnz >= (int64_t)kTbSample * 1024 ? kTbSample : 1;
    {
      const int64_t want = cdiv(cdiv(a->nnz, sample), kBlock);
      hipLaunchKernelGGL(tb_est_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(kBlock), 0, ctx->stream,
                         (const int64_t *)a->row_ptr, (const int32_t *)a->col32, a->nnz, (const int64_t *)a->tile_lb,
                         a->tile, sample, (const uint32_t *)rank_of_line.p, R, lpb, nb, est.p);
      SAMD_LAUNCH_CHECK(ctx);
    }
    Tmp<uint32_t> ek_a(ctx), ek_b(ctx), ep_a(ctx), ep_b(ctx);
    SAMD_TRY(ek_a.get(a->nnz));
    SAMD_TRY(ek_b.get(a->nnz));
    SAMD_TRY(ep_a.get(a->nnz));
    SAMD_TRY(ep_b.get(a->nnz));
    const unsigned egrid = (unsigned)cdiv(a->nnz, kBlock);
    const uint32_t est_min = (uint32_t)cdiv(dense_min, sample);
    hipLaunchKernelGGL(tb_key_kernel, dim3(egrid), dim3(kBlock), 0, ctx->stream, (const int64_t *)a->row_ptr,
                       (const int32_t *)a->col32, a->nnz, (const int64_t *)a->tile_lb, a->tile,
                       (const uint32_t *)rank_of_line.p, R, lpb, nb, (const uint32_t *)est.p, est_min, ek_a.p);
    SAMD_LAUNCH_CHECK(ctx);
    uint32_t *eks = nullptr;
    const uint32_t *eperm = nullptr;
    SAMD_TRY((radix_sort_pairs<uint32_t, uint32_t>(ctx, ek_a.p, ek_b.p, (const uint32_t *)nullptr, ep_a.p, ep_b.p, a->nnz,
                                                    0, bits_for(pieces2), &eks, &eperm, sort_err)));
    hipLaunchKernelGGL(pb_piece_ptr_kernel, dim3((unsigned)cdiv(pieces2 + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const uint32_t *)eks, a->nnz, pieces2, piece_ptr.p, 0);
    SAMD_LAUNCH_CHECK(ctx);
    SAMD_TRY(col16.get(a->nnz));
    SAMD_TRY(row16.get(a->nnz));
    SAMD_TRY(blk16.get(a->nnz));
    SAMD_TRY(pval.get(a->nnz));
    hipLaunchKernelGGL((tb_build_kernel<T>), dim3(egrid), dim3(kBlock), 0, ctx->stream, (const int64_t *)a->row_ptr,
                       (const int32_t *)a->col32, (const T *)a->vals, a->nnz, (const int64_t *)a->tile_lb, a->tile,
                       (const uint32_t *)rank_of_line.p, R, W, eperm, col16.p, row16.p, blk16.p, pval.p);
    SAMD_LAUNCH_CHECK(ctx);
    int64_t flag[3];
    SAMD_TRY(read_small(ctx, ctx->d_small, 3, flag));  // also orders the temporaries' release after their use
    if ((int)(flag[2] & 0xffffffff) != 0) return set_error(ctx, "internal: radix sort look-back timed out");
  }
  // ---- descriptors and units on the host (a few thousand panels; one read of the piece bounds)
  std::vector<int64_t> pp((size_t)pieces2 + 1);
  SAMD_HIP(ctx, hipMemcpyAsync(pp.data(), piece_ptr.p, sizeof(int64_t) * pp.size(), hipMemcpyDeviceToHost, ctx->stream));
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<int64_t> h_panel((size_t)np + 1), h_sparse((size_t)np), h_desc((size_t)np + 1), h_unit((size_t)np + 1), h_ds, h_de;
  std::vector<int32_t> h_blk, h_slot((size_t)np), h_pos;
  int64_t units = 0, dense_entries = 0;
  for (int64_t p = 0; p < np; p++) {
    const int64_t b0 = p * 2 * nb;
    h_panel[(size_t)p] = pp[(size_t)b0];
    h_sparse[(size_t)p] = pp[(size_t)(b0 + nb)];
    h_desc[(size_t)p] = (int64_t)h_blk.size();
    for (int64_t b = 0; b < nb; b++) {
      const int64_t s = pp[(size_t)(b0 + b)], e = pp[(size_t)(b0 + b + 1)];
      if (e > s) { h_blk.push_back((int32_t)b); h_ds.push_back(s); h_de.push_back(e); dense_entries += e - s; }
    }
    const int64_t cnt = pp[(size_t)(b0 + 2 * nb)] - pp[(size_t)b0];
    const int64_t nu = cnt > E ? cdiv(cnt, E) : 1;
    h_unit[(size_t)p] = units;
    units += nu;
    h_slot[(size_t)p] = nu > 1 ? (int32_t)h_pos.size() : -1;
    if (nu > 1) h_pos.push_back((int32_t)p);
  }
  h_panel[(size_t)np] = pp[(size_t)pieces2];
  h_desc[(size_t)np] = (int64_t)h_blk.size();
  h_unit[(size_t)np] = units;
  if (units >= (1ll << 31)) return 0;
  // auto mode: the x slices must pay for themselves -- most entries in dense pieces
  if (!force && dense_entries * 2 < a->nnz) return 0;

  TbPlan *t = new TbPlan();
  struct Guard {
    futhark_context *ctx; TbPlan *t;
    ~Guard() { if (t) tb_free(ctx, t); }
  } guard{ctx, t};
  t->np = np; t->nb = nb; t->W = W; t->R = R; t->nl = nl; t->units = units;
  t->nslots = (int64_t)h_pos.size(); t->ndesc = (int64_t)h_blk.size();
  SAMD_TRY(tb_upload(ctx, &t->panel_ptr, h_panel));
  SAMD_TRY(tb_upload(ctx, &t->sparse_ptr, h_sparse));
  SAMD_TRY(tb_upload(ctx, &t->desc_ptr, h_desc));
  SAMD_TRY(tb_upload(ctx, &t->d_blk, h_blk));
  SAMD_TRY(tb_upload(ctx, &t->d_s, h_ds));
  SAMD_TRY(tb_upload(ctx, &t->d_e, h_de));
  SAMD_TRY(tb_upload(ctx, &t->unit_ptr, h_unit));
  SAMD_TRY(tb_upload(ctx, &t->slot, h_slot));
  SAMD_TRY(tb_upload(ctx, &t->panel_of_slot, h_pos));
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the vectors go away
  SAMD_TRY(xperm.get(nb * W));
  SAMD_HIP(ctx, hipMemsetAsync(xperm.p, 0, sizeof(T) * (size_t)(nb * W), ctx->stream));  // the tail beyond nl lines stays zero
  acc_t *part = nullptr;
  SAMD_TRY(dalloc(ctx, &part, t->nslots * R));
  t->part = part;
  if (t->nslots > 0) SAMD_HIP(ctx, hipMemsetAsync(part, 0, sizeof(acc_t) * (size_t)(t->nslots * R), ctx->stream));
  t->line_of_rank = line_of_rank.release();
  t->col16 = col16.release();
  t->row16 = row16.release();
  t->blk16 = blk16.release();
  t->val = pval.release();
  t->xperm = xperm.release();
  a->tb_ext = t;
  a->tb_ext_free = tb_free;
  guard.t = nullptr;
  a->tb_state = 1;
  if (ctx->logging)
    fprintf(stderr, "[sparse_amd] 2-D SpMV: %lld panels x %lld blocks of %lld columns, %lld dense pieces holding %.1f %% of the "
            "entries, %lld units, %lld panels in several units\n", (long long)np, (long long)nb, (long long)W,
            (long long)t->ndesc, 100.0 * (double)dense_entries / (double)a->nnz, (long long)units, (long long)t->nslots);
  return 0;
}

template <class T>
int spmv_2d(futhark_context *ctx, const CsrBody<T> *a, const T *x, T *y) {
  typedef typename PbAcc<T>::type acc_t;
  const TbPlan *t = reinterpret_cast<const TbPlan *>(a->tb_ext);
  hipLaunchKernelGGL((tb_xperm_kernel<T>), dim3((unsigned)cdiv(t->nl * kPbLine, kBlock)), dim3(kBlock), 0, ctx->stream, x,
                     a->m, (const uint32_t *)t->line_of_rank, t->nl, reinterpret_cast<T *>(t->xperm));
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL((tb_kernel<T, 4>), dim3((unsigned)t->units), dim3(kTbThreads), 0, ctx->stream,
                     (const int64_t *)t->panel_ptr, (const int64_t *)t->sparse_ptr, (const int64_t *)t->desc_ptr,
                     (const int32_t *)t->d_blk, (const int64_t *)t->d_s, (const int64_t *)t->d_e,
                     (const int64_t *)t->unit_ptr, (const int32_t *)t->slot, (const uint16_t *)t->col16,
                     (const uint16_t *)t->row16, (const uint16_t *)t->blk16, reinterpret_cast<const T *>(t->val),
                     reinterpret_cast<const T *>(t->xperm), y, reinterpret_cast<acc_t *>(t->part), t->np, a->n, (int)t->W,
                     (int)t->R, kPbFoldMax, (int)ctx->opt_ablate);
  SAMD_LAUNCH_CHECK(ctx);
  if (t->nslots > 0) {
    hipLaunchKernelGGL((tb_finish_kernel<T>), dim3((unsigned)t->nslots), dim3(kBlock), 0, ctx->stream,
                       reinterpret_cast<acc_t *>(t->part), (const int32_t *)t->panel_of_slot, a->n, (int)t->R, y);
    SAMD_LAUNCH_CHECK(ctx);
  }
  return 0;
}

}  // namespace samd
