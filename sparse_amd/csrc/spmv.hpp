// spmv.hpp -- y = A x for a CSR record (sr.smvm, compressed.fut:134-146; sc.vsmm
// :223-224 is the same kernel on the relabelled record).
//
// The reference flattens the product (expand_outer_reduce): it materialises an
// nnz-long row-id array, an nnz-long iota and an nnz-long product array and
// runs segmented scans over them.  Here the product array never leaves the CU:
//
// spmv_stream (default; nnz-balanced, "merge-path"-style partition on entries)
//   * tile t owns entries [t*TILE, (t+1)*TILE): col_idx/vals are streamed with
//     128-bit coalesced loads whatever the row lengths are; x is gathered
//     through L1/L2; products land in LDS.
//   * rows are then reduced out of LDS: one lane per row fragment when the
//     fragment is short, a whole wavefront (64 lanes + cross-lane reduction)
//     for fragments longer than kLongFrag.
//   * a row that starts in tile t is written by tile t (plain store).  The part
//     of a row that spills into later tiles is written to head[t'] by those
//     tiles and added in tile order by spmv_fixup: no atomics, so y is
//     bit-reproducible run to run.
//   * blockIdx -> tile is XCD-aware: the 8 XCDs (blocks b, b+8, ... share one)
//     each take a contiguous eighth of the tiles, so the x window a row block
//     gathers is cached in one XCD's L2 instead of all eight.
// spmv_vector<LANES> (2..64 lanes per row; classic wavefront / sub-wavefront per
//   row) is kept for matrices whose rows are uniformly long, where staging
//   through LDS buys nothing.
//
// Algorithmic bytes per entry (f64, col_idx held as i32): 12 streamed + the row
// terms 8 (row_ptr) + 8 (y) per row + 8 per x element.  HBM-bound; no MFMA.
#pragma once

#include "assembly.hpp"

namespace samd {

constexpr int kSpmvTile = (int)kPlanTile;  // entries per block
constexpr int kLongFrag = 32;              // row fragments longer than this are reduced by a wavefront

template <class T>
struct alignas(sizeof(T) * 4) Vec4 {
  T v[4];
};

__device__ __forceinline__ int64_t xcd_remap(int64_t bid, int64_t nblocks) {
  const int64_t xcd = bid & 7, q = nblocks >> 3, r = nblocks & 7;
  const int64_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + (bid >> 3);
}

template <class T, class I, int TILE>
__global__ __launch_bounds__(kBlock) void spmv_stream_kernel(const int64_t *__restrict__ row_ptr,
                                                             const I *__restrict__ col, const T *__restrict__ val,
                                                             const T *__restrict__ x, T *__restrict__ y,
                                                             const int64_t *__restrict__ tile_lb,
                                                             T *__restrict__ head, int64_t nnz, int64_t ntiles) {
  __shared__ T prod[TILE];
  __shared__ int longrows[TILE / kLongFrag + 2];
  __shared__ int nlong;

  const int tid = threadIdx.x;
  const int64_t t = xcd_remap(blockIdx.x, ntiles);
  const int64_t base = t * TILE;
  const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);

  // ---- phase 1: stream the tile, gather x, products to LDS
  constexpr int ITERS = TILE / (kBlock * 4);
  const I *cbase = col + base;
  const T *vbase = val + base;
  Vec4<I> c[ITERS];
  Vec4<T> a[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; it++) {
    const int e = (it * kBlock + tid) * 4;
    if (e + 4 <= cnt) {
      c[it] = *reinterpret_cast<const Vec4<I> *>(cbase + e);
      a[it] = *reinterpret_cast<const Vec4<T> *>(vbase + e);
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const bool ok = e + q < cnt;
        c[it].v[q] = ok ? cbase[e + q] : (I)0;
        a[it].v[q] = ok ? vbase[e + q] : (T)0;
      }
    }
  }
  if (tid == 0) nlong = 0;
#pragma unroll
  for (int it = 0; it < ITERS; it++) {
    const int e = (it * kBlock + tid) * 4;
    Vec4<T> p;
#pragma unroll
    for (int q = 0; q < 4; q++) p.v[q] = (e + q < cnt) ? (T)(a[it].v[q] * x[c[it].v[q]]) : (T)0;
    *reinterpret_cast<Vec4<T> *>(&prod[e]) = p;
  }
  __syncthreads();

  // ---- phase 2: one lane per row fragment
  const int64_t lb0 = tile_lb[t], lb1 = tile_lb[t + 1];
  const bool has_head = lb0 > 0 && row_ptr[lb0] > base;
  const int64_t rfirst = lb0 - (has_head ? 1 : 0);
  const int nrows = (int)(lb1 - rfirst);
  for (int j = tid; j < nrows; j += kBlock) {
    const int64_t r = rfirst + j;
    const int64_t rs = row_ptr[r], re = row_ptr[r + 1];
    const int s = (int)((rs > base ? rs : base) - base);
    const int64_t lim = base + cnt;
    const int e = (int)((re < lim ? re : lim) - base);
    if (e - s > kLongFrag) {
      const int slot = atomicAdd(&nlong, 1);
      longrows[slot] = j;
      continue;
    }
    T acc = (T)0;
    for (int q = s; q < e; q++) acc = (T)(acc + prod[q]);
    if (rs >= base) y[r] = acc;
    else head[t] = acc;
  }
  __syncthreads();

  // ---- phase 3: one wavefront per long fragment
  const int nl = nlong;
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = wave; i < nl; i += kWaves) {
    const int64_t r = rfirst + longrows[i];
    const int64_t rs = row_ptr[r], re = row_ptr[r + 1];
    const int s = (int)((rs > base ? rs : base) - base);
    const int64_t lim = base + cnt;
    const int e = (int)((re < lim ? re : lim) - base);
    T acc = (T)0;
    for (int q = s + lane; q < e; q += 64) acc = (T)(acc + prod[q]);
    acc = wave_reduce(acc, OpAdd());
    if (lane == 0) {
      if (rs >= base) y[r] = acc;
      else head[t] = acc;
    }
  }
}

// Adds the spilled fragments: for tile t whose head row started in tile t-1
// (the first spill of that row), fold head[t], head[t+1], ... in tile order.
template <class T, int TILE>
__global__ __launch_bounds__(kBlock) void spmv_fixup_kernel(const int64_t *__restrict__ row_ptr,
                                                            const int64_t *__restrict__ tile_lb,
                                                            const T *__restrict__ head, T *__restrict__ y,
                                                            int64_t ntiles) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x + 1;
  if (t >= ntiles) return;
  const int64_t lb = tile_lb[t];
  if (lb == 0) return;
  const int64_t base = t * TILE;
  const int64_t r = lb - 1;
  const int64_t re = row_ptr[lb];
  if (re <= base) return;                  // no fragment enters this tile
  if (row_ptr[r] < base - TILE) return;    // an earlier tile is this row's first spill
  T acc = y[r];
  for (int64_t u = t; u < ntiles && u * TILE < re; u++) acc = (T)(acc + head[u]);
  y[r] = acc;
}

// LANES lanes per row, rows dealt to lane groups in order.
template <class T, class I, int LANES>
__global__ __launch_bounds__(kBlock) void spmv_vector_kernel(const int64_t *__restrict__ row_ptr,
                                                             const I *__restrict__ col, const T *__restrict__ val,
                                                             const T *__restrict__ x, T *__restrict__ y, int64_t n) {
  const int sub = threadIdx.x & (LANES - 1);
  const int64_t group = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
  const int64_t ngroups = (int64_t)gridDim.x * kBlock / LANES;
  for (int64_t r = group; r < n; r += ngroups) {
    const int64_t rs = row_ptr[r], re = row_ptr[r + 1];
    T acc = (T)0;
    for (int64_t q = rs + sub; q < re; q += LANES) acc = (T)(acc + (T)(val[q] * x[col[q]]));
#pragma unroll
    for (int d = LANES / 2; d >= 1; d >>= 1) acc = (T)(acc + wshfl_xor(acc, d));
    if (sub == 0) y[r] = acc;
  }
}

template <class T, class I>
int spmv_typed(futhark_context *ctx, const CsrBody<T> *a, const I *col, const T *x, T *y) {
  int64_t algo = ctx->opt_spmv_algo;
  if (algo == 0) algo = 1;
  if (algo == 1) {
    if (a->tile != kSpmvTile) return set_error(ctx, "spmv: plan tile mismatch");
    hipLaunchKernelGGL((spmv_stream_kernel<T, I, kSpmvTile>), dim3((unsigned)a->ntiles), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)a->row_ptr, col, (const T *)a->vals, x, y, (const int64_t *)a->tile_lb,
                       a->head, a->nnz, a->ntiles);
    SAMD_LAUNCH_CHECK(ctx);
    if (a->ntiles > 1) {
      hipLaunchKernelGGL((spmv_fixup_kernel<T, kSpmvTile>), dim3((unsigned)cdiv(a->ntiles - 1, kBlock)), dim3(kBlock),
                         0, ctx->stream, (const int64_t *)a->row_ptr, (const int64_t *)a->tile_lb,
                         (const T *)a->head, y, a->ntiles);
      SAMD_LAUNCH_CHECK(ctx);
    }
    return 0;
  }
  int lanes = algo == 2 ? 64 : (int)ctx->opt_spmv_lanes;
  if (lanes <= 0) lanes = 8;
  int64_t want = cdiv(a->n * lanes, kBlock);
  const int64_t cap = (int64_t)ctx->num_cu * 32;
  const unsigned grid = (unsigned)(want < cap ? (want > 0 ? want : 1) : cap);
#define SAMD_VEC(L)                                                                                              \
  hipLaunchKernelGGL((spmv_vector_kernel<T, I, L>), dim3(grid), dim3(kBlock), 0, ctx->stream,                    \
                     (const int64_t *)a->row_ptr, col, (const T *)a->vals, x, y, a->n)
  switch (lanes) {
    case 2: SAMD_VEC(2); break;
    case 4: SAMD_VEC(4); break;
    case 8: SAMD_VEC(8); break;
    case 16: SAMD_VEC(16); break;
    case 32: SAMD_VEC(32); break;
    case 64: SAMD_VEC(64); break;
    default: return set_error(ctx, "spmv: spmv_lanes must be a power of two in 2..64");
  }
#undef SAMD_VEC
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

// y[n] = A[n][m] x[m].  x and y are device pointers of the right lengths.
template <class T>
int spmv(futhark_context *ctx, const CsrBody<T> *a, const T *x, T *y) {
  if (a->n == 0) return 0;
  if (a->nnz == 0) {
    SAMD_HIP(ctx, hipMemsetAsync(y, 0, sizeof(T) * (size_t)a->n, ctx->stream));  // empty row -> T.i64 0
    return 0;
  }
  if (a->col32) return spmv_typed<T, int32_t>(ctx, a, a->col32, x, y);
  return spmv_typed<T, int64_t>(ctx, a, a->col64, x, y);
}

}  // namespace samd
