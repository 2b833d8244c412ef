// spmv.hpp -- y = A x for a CSR record (sr.smvm, compressed.fut:134-146; sc.vsmm
// :223-224 is the same kernel on the relabelled record).
//
// The reference flattens the product (expand_outer_reduce): it materialises an
// nnz-long row-id array, an nnz-long iota and an nnz-long product array and
// runs segmented scans over them.  Here nothing but y leaves the CU:
//
// spmv_stream (default; nnz-balanced, "merge-path"-style partition on entries)
//   * tile t owns entries [t*TILE, (t+1)*TILE).  col_idx/vals are streamed into LDS
//     with 128-bit loads in which every wavefront instruction covers one contiguous
//     KiB, marked non-temporal: the matrix is read once, so it must not displace x
//     (or the y lines being written) from L2.  No x access happens while streaming.
//   * rows are then walked out of LDS in ROW-major order: at step k every lane
//     handles entry k of its own row.  For banded / stencil matrices the lanes of one
//     gather instruction therefore read consecutive x addresses, which is what the
//     per-CU L1 can serve at rate (an entry-major gather costs one L1 tag look-up
//     per lane; measured on MI355X that look-up rate, not HBM, was the bound).
//   * lanes per row L in {1,2,..,64} is chosen per tile from its mean fragment
//     length (sub-wavefront-per-row), fragments longer than 16 L are handed to a
//     whole wavefront (wavefront-per-row) with a cross-lane reduction.
//   * a row that starts in tile t is written by tile t (plain/non-temporal store).
//     The part of a row that spills into later tiles is written to head[t'] by those
//     tiles and added in tile order by spmv_fixup: no atomics, so y is
//     bit-reproducible run to run.  Records of at most kInlineMaxTiles tiles (launch-bound)
//     use the INLINE variant: the owner fetches a spill of <= kSpillInline entries with its
//     tile and finishes the row, and the fix-up launch is skipped when no row spills further.
//   * blockIdx -> tile is XCD-aware: the 8 XCDs (blocks b, b+8, ... share one)
//     each take a contiguous eighth of the tiles, so the x window a row block
//     gathers is cached in one XCD's L2 instead of all eight.
// spmv_vector<LANES> (2..64 lanes per row straight from global memory; classic
//   wavefront / sub-wavefront per row) is kept as a selectable alternative.
//
// Algorithmic bytes per entry (f64, col_idx held as i32): 12 streamed + the row
// terms 8 (row_ptr) + 8 (y) per row + 8 per x element.  HBM-bound; no MFMA.
#pragma once

#include "assembly.hpp"

namespace samd {

constexpr int kSpmvTile = (int)kPlanTile;  // entries per block
constexpr int kSpillInline = 64;           // INLINE: a row running at most this far past its tile is finished by its owner
constexpr int64_t kInlineMaxTiles = 4096;  // INLINE is used below this many tiles (launch-bound sizes)
constexpr int64_t kDenseRowsMax = 2048;    // tiles in which more rows than this start take the many-empty-rows path

typedef unsigned int raw16 __attribute__((ext_vector_type(4)));  // one 128-bit load/store

template <class V>
__device__ __forceinline__ V ntload(const V *p) { return __builtin_nontemporal_load(p); }

template <class T>
struct alignas(sizeof(T) * 4) Vec4 {
  T v[4];
};

__device__ __forceinline__ int64_t xcd_remap(int64_t bid, int64_t nblocks) {
  const int64_t xcd = bid & 7, q = nblocks >> 3, r = nblocks & 7;
  const int64_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + (bid >> 3);
}

// lane `sub` of L: sum of entries s+sub, s+sub+L, ... below e, U gathers in flight
template <class T, class I, int U>
__device__ __forceinline__ T row_partial(const T *sval, const I *scol, const T *__restrict__ x, int s, int e, int sub,
                                         int L) {
  T acc = (T)0;
  for (int q = s + sub; q < e; q += U * L) {
    T xv[U], av[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int qq = q + u * L;
      const bool ok = qq < e;
      const I cc = scol[ok ? qq : s];
      av[u] = ok ? sval[qq] : (T)0;
      xv[u] = ok ? x[cc] : (T)0;
    }
#pragma unroll
    for (int u = 0; u < U; u++) acc = (T)(acc + (T)(av[u] * xv[u]));
  }
  return acc;
}

template <class T, class I, int TILE, bool INLINE, int U>
__global__ __launch_bounds__(kBlock) void spmv_stream_kernel(const int64_t *__restrict__ row_ptr,
                                                             const I *__restrict__ col, const T *__restrict__ val,
                                                             const T *__restrict__ x, T *__restrict__ y,
                                                             const int64_t *__restrict__ tile_lb,
                                                             T *__restrict__ head, int64_t nnz, int64_t ntiles,
                                                             int force_lanes, const uint16_t *__restrict__ row_rel,
                                                             const int32_t *__restrict__ tile_c0) {
  __shared__ __attribute__((aligned(16))) T sval[TILE + (INLINE ? kSpillInline : 0)];
  __shared__ __attribute__((aligned(16))) I scol[TILE + (INLINE ? kSpillInline : 0)];
  __shared__ int longrows[TILE / 16 + 6];
  __shared__ int nlong;

  const int tid = threadIdx.x;
  const int64_t t = xcd_remap(blockIdx.x, ntiles);
  const int64_t base = t * TILE;
  const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);
  const int64_t lb0 = tile_lb[t], lb1 = tile_lb[t + 1];
  // 16-bit columns are relative to the tile's smallest column (tile_c0): the tile gathers from x + that
  const T *__restrict__ xt = sizeof(I) == 2 ? x + tile_c0[t] : x;

  // ---- phase 1: stream the tile into LDS (line-dense, non-temporal)
  constexpr int CW = TILE * (int)sizeof(I) / 16;  // 128-bit pieces of the tile's columns
  constexpr int CV = CW / kBlock, CR = CW % kBlock;  // whole rounds of the block + a partial one (16-bit columns)
  constexpr int VV = TILE * (int)sizeof(T) / (16 * kBlock);
  static_assert(CW >= 1 && VV >= 1, "tile too small for 128-bit loads");
  if (cnt == TILE) {
    raw16 c[CV + 1], a[VV];
    const raw16 *cp = reinterpret_cast<const raw16 *>(col + base);
    const raw16 *vp = reinterpret_cast<const raw16 *>(val + base);
#pragma unroll
    for (int it = 0; it < CV; it++) c[it] = ntload(cp + it * kBlock + tid);
    if (CR) c[CV] = ntload(cp + CV * kBlock + (tid < CR ? tid : 0));  // unpredicated: clamped piece, stored by tid < CR only
#pragma unroll
    for (int it = 0; it < VV; it++) a[it] = ntload(vp + it * kBlock + tid);
#pragma unroll
    for (int it = 0; it < CV; it++) reinterpret_cast<raw16 *>(scol)[it * kBlock + tid] = c[it];
    if (CR && tid < CR) reinterpret_cast<raw16 *>(scol)[CV * kBlock + tid] = c[CV];
#pragma unroll
    for (int it = 0; it < VV; it++) reinterpret_cast<raw16 *>(sval)[it * kBlock + tid] = a[it];
  } else {
    for (int e = tid; e < cnt; e += kBlock) {
      scol[e] = col[base + e];
      sval[e] = val[base + e];
    }
  }

  // ---- INLINE: a short spill of the last row that starts here is fetched along with the tile
  const int64_t lim = base + cnt;
  int64_t lim2 = lim;
  if (INLINE && lb1 > lb0) {
    const int64_t sp = row_ptr[lb1] - lim;  // uniform over the block
    if (sp > 0 && sp <= kSpillInline) {
      lim2 = lim + sp;
      if (tid < sp) {
        scol[TILE + tid] = col[lim + tid];
        sval[TILE + tid] = val[lim + tid];
      }
    }
  }

  // ---- rows of this tile: the candidate head row (lb0-1) and the rows that start here
  const int64_t rfirst = lb0 > 0 ? lb0 - 1 : 0;
  const int64_t nrows64 = lb1 - rfirst;
  const int nrows = (int)(nrows64 <= kDenseRowsMax ? nrows64 : kDenseRowsMax + 1);
  int L = 1;  // lanes per row, from the mean fragment length (uniform over the block)
  {
    const int avg4 = cnt / (nrows > 0 ? nrows : 1) / 4;
    while (L < 64 && L * 2 <= avg4) L *= 2;
    if (force_lanes > 0) L = force_lanes;  // test / tuning override (power of two <= 64)
  }

  if (tid == 0) nlong = 0;
  const int grp = tid / L, sub = tid & (L - 1), ngrp = kBlock / L;
  // Rows are dealt to the lane groups so that group g holds a row with (row & 15) == (g & 15): the y stores of a
  // wavefront then cover whole 128-byte lines instead of straddling them (ngrp is a multiple of 16 for L <= 16).
  const int shift = ngrp >= 16 ? (int)(rfirst & 15) : 0;
  const int j0 = grp - shift;
  int64_t rs0 = 0, re0 = 0;  // first row's bounds, requested before the barrier
  int a0 = 0, b0 = 0;        // ... or its 16-bit tile-relative form (large records, see row_rel_kernel)
  if (INLINE) {
    if (j0 >= 0 && j0 < nrows) {
      rs0 = ntload(&row_ptr[rfirst + j0]);
      re0 = ntload(&row_ptr[rfirst + j0 + 1]);
    }
  } else {
    // never behind a branch (a 16-bit load behind a branch is waited for at once): clamped index, masked later
    const int jc = j0 < 0 ? 0 : (j0 < nrows ? j0 : (nrows > 0 ? nrows - 1 : 0));
    a0 = (int)ntload(&row_rel[rfirst + jc]);
    b0 = (int)ntload(&row_rel[rfirst + jc + 1]);
  }
  __syncthreads();

  // ---- more rows than kDenseRowsMax start in this tile: all but <= TILE of them are empty (a run of
  // empty rows shares one row_ptr value, so whichever tile holds that position gets the whole run).
  // The host has zero-filled y; here every entry looks up its row (binary search over the tile's rows)
  // and the entry that opens a row fragment sums it.  Work is bounded by the tile's entries, not its rows.
  if (nrows64 > kDenseRowsMax) {  // uniform over the block
    for (int p = tid; p < cnt; p += kBlock) {
      const int64_t gp = base + p;
      int64_t lo = rfirst, hi = lb1;  // largest r in [rfirst, lb1) with row_ptr[r] <= gp
      while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (row_ptr[mid] <= gp) lo = mid; else hi = mid;
      }
      const int64_t r = lo, rs = row_ptr[r];
      if (gp != rs && p != 0) continue;  // not the first entry of its row's fragment
      const int64_t re = row_ptr[r + 1];
      if (INLINE && r < lb0 && re - (rs / TILE + 1) * TILE <= kSpillInline) continue;
      const int e = (int)((re < lim2 ? re : lim2) - base);
      const T acc = row_partial<T, I, 4>(sval, scol, xt, p, e, 0, 1);
      if (rs >= base) y[r] = acc;
      else head[t] = acc;
    }
    return;
  }

  // ---- phase 2: L lanes per row fragment, row-major gather of x
  for (int j = j0; j < nrows; j += ngrp) {
    if (j < 0) continue;
    const int64_t r = rfirst + j;
    int s, e;
    bool to_y;
    if (INLINE) {
      int64_t rs, re;
      if (j == j0) { rs = rs0; re = re0; }
      else { rs = ntload(&row_ptr[r]); re = ntload(&row_ptr[r + 1]); }
      if (r < lb0) {
        if (re <= base) continue;                                    // candidate head row does not reach this tile
        if (re - (rs / TILE + 1) * TILE <= kSpillInline) continue;   // ... or was finished by its owner
      }
      s = (int)((rs > base ? rs : base) - base);
      e = (int)((re < lim2 ? re : lim2) - base);
      to_y = rs >= base;
    } else {
      // Tile-relative bounds from row_rel (2 bytes per row instead of 8): a row that starts here begins at its own
      // offset; any row ends where the next row starting here begins, or runs to the end of the tile.
      int a, b;
      if (j == j0) { a = a0; b = b0; }
      else { a = (int)ntload(&row_rel[r]); b = (int)ntload(&row_rel[r + 1]); }
      to_y = r >= lb0;
      s = to_y ? a : 0;
      e = r + 1 < lb1 ? b : cnt;
      if (!to_y && e <= 0) continue;  // candidate head row does not reach this tile
    }
    if (e - s > 16 * L) {
      if (sub == 0) {
        const int slot = atomicAdd(&nlong, 1);
        longrows[slot] = j;
      }
      continue;
    }
    // entries in flight per lane: as many as a typical fragment needs, no more (predicated-off slots
    // still cost issue slots); U is chosen per record on the host from its mean row length
    T acc = row_partial<T, I, U>(sval, scol, xt, s, e, sub, L);
    for (int d = L >> 1; d >= 1; d >>= 1) acc = (T)(acc + wshfl_xor(acc, d));
    if (sub == 0) {
      if (to_y) __builtin_nontemporal_store(acc, &y[r]);
      else head[t] = acc;
    }
  }
  __syncthreads();

  // ---- phase 3: one wavefront per long fragment
  const int nl = nlong;
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = wave; i < nl; i += kWaves) {
    const int64_t r = rfirst + longrows[i];
    const int64_t rs = row_ptr[r], re = row_ptr[r + 1];
    const int s = (int)((rs > base ? rs : base) - base);
    const int e = (int)((re < lim2 ? re : lim2) - base);
    T acc = (T)0;
    for (int q = s + lane; q < e; q += 4 * 64) {
      T xv[4], av[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int qq = q + u * 64;
        const bool ok = qq < e;
        const I cc = scol[ok ? qq : s];
        av[u] = ok ? sval[qq] : (T)0;
        xv[u] = ok ? xt[cc] : (T)0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) acc = (T)(acc + (T)(av[u] * xv[u]));
    }
    acc = wave_reduce(acc, OpAdd());
    if (lane == 0) {
      if (rs >= base) y[r] = acc;
      else head[t] = acc;
    }
  }
}

// row_rel[r] = start of row r relative to the tile it is listed in (0 .. TILE; TILE only for empty rows at the very
// end of a record whose entry count is a multiple of TILE).  Built at a large record's first product: the stream
// kernel then reads 2 bytes per row instead of 8 (declared in the algorithmic bytes: rb = 2).
__global__ __launch_bounds__(kBlock) void row_rel_kernel(const int64_t *__restrict__ row_ptr, int64_t n, int64_t tile,
                                                         int64_t ntiles, uint16_t *__restrict__ row_rel) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r > n) return;
  const int64_t rs = row_ptr[r];
  int64_t t = rs / tile;
  if (t > ntiles - 1) t = ntiles - 1;
  row_rel[r] = (uint16_t)(rs - t * tile);
}

// Adds the spilled fragments: for tile t whose head row started in tile t-1
// (the first spill of that row), fold head[t], head[t+1], ... in tile order.
template <class T, int TILE>
__global__ __launch_bounds__(kBlock) void spmv_fixup_kernel(const int64_t *__restrict__ row_ptr,
                                                            const int64_t *__restrict__ tile_lb,
                                                            const T *__restrict__ head, T *__restrict__ y,
                                                            int64_t ntiles, bool inline_spills) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x + 1;
  if (t >= ntiles) return;
  const int64_t lb = tile_lb[t];
  if (lb == 0) return;
  const int64_t base = t * TILE;
  const int64_t r = lb - 1;
  const int64_t re = row_ptr[lb];
  if (re <= base) return;                  // no fragment enters this tile
  if (row_ptr[r] < base - TILE) return;    // an earlier tile is this row's first spill
  if (inline_spills && re - base <= kSpillInline) return;  // finished by its owner inside spmv_stream
  T acc = y[r];
  for (int64_t u = t; u < ntiles && u * TILE < re; u++) acc = (T)(acc + head[u]);
  y[r] = acc;
}

// The same fold from a list made once per record (fixup_list_kernel): entry i = a row whose first spill enters tile
// fix_tile[i] and runs through fix_span[i] tiles.  Two dependent hops (list -> y / head) instead of the four of
// spmv_fixup_kernel (tile_lb -> row_ptr -> y / head): the launch sits between two SpMV steps of a solver, so its
// latency is all it costs.
template <class T>
__global__ __launch_bounds__(kBlock) void spmv_fixup_list_kernel(const int32_t *__restrict__ fix_row_lo,
                                                                 const int32_t *__restrict__ fix_row_hi,
                                                                 const int32_t *__restrict__ fix_tile,
                                                                 const int32_t *__restrict__ fix_span, int64_t count,
                                                                 const T *__restrict__ head, T *__restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= count) return;
  const int64_t r = ((int64_t)fix_row_hi[i] << 31) | (int64_t)fix_row_lo[i];
  const int64_t t = fix_tile[i];
  const int span = fix_span[i];
  T acc = y[r];
  T h0 = head[t];
  for (int u = 1; u < span; u++) { acc = (T)(acc + h0); h0 = head[t + u]; }
  y[r] = (T)(acc + h0);
}

template <int TILE>
__global__ __launch_bounds__(kBlock) void fixup_list_kernel(const int64_t *__restrict__ row_ptr,
                                                            const int64_t *__restrict__ tile_lb, int64_t ntiles,
                                                            int32_t *__restrict__ fix_row_lo, int32_t *__restrict__ fix_row_hi,
                                                            int32_t *__restrict__ fix_tile, int32_t *__restrict__ fix_span,
                                                            unsigned long long *__restrict__ count) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x + 1;
  if (t >= ntiles) return;
  const int64_t lb = tile_lb[t];
  if (lb == 0) return;
  const int64_t base = t * TILE, r = lb - 1, re = row_ptr[lb];
  if (re <= base || row_ptr[r] < base - TILE) return;  // nothing enters here / an earlier tile is the first spill
  int64_t span = (re - base + TILE - 1) / TILE;
  if (t + span > ntiles) span = ntiles - t;
  const unsigned long long slot = atomicAdd(count, 1ull);
  fix_row_lo[slot] = (int32_t)(r & 0x7fffffff);
  fix_row_hi[slot] = (int32_t)(r >> 31);
  fix_tile[slot] = (int32_t)t;
  fix_span[slot] = (int32_t)span;
}

// LANES lanes per row, rows dealt to lane groups in order.
template <class T, class I, int LANES>
__global__ __launch_bounds__(kBlock) void spmv_vector_kernel(const int64_t *__restrict__ row_ptr,
                                                             const I *__restrict__ col, const T *__restrict__ val,
                                                             const T *__restrict__ x, T *__restrict__ y, int64_t n,
                                                             const int64_t *__restrict__ rows, int64_t ylen) {
  const int sub = threadIdx.x & (LANES - 1);
  const int64_t group = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
  const int64_t ngroups = (int64_t)gridDim.x * kBlock / LANES;
  for (int64_t r = group; r < n; r += ngroups) {
    const int64_t rs = row_ptr[r], re = row_ptr[r + 1];
    T acc = (T)0;
    for (int64_t q = rs + sub; q < re; q += LANES) acc = (T)(acc + (T)(val[q] * x[col[q]]));
#pragma unroll
    for (int d = LANES / 2; d >= 1; d >>= 1) acc = (T)(acc + wshfl_xor(acc, d));
    if (sub == 0) {
      if (!rows) y[r] = acc;
      else {  // y[rows[r]] += (A x)[r]: distinct indices (the caller's contract), so no atomics
        const int64_t d = rows[r];
        if (d >= 0 && d < ylen) y[d] = (T)(y[d] + acc);
      }
    }
  }
}

// ------------------------------------------------------------------ reorder
// Power-law matrices (C3: RMAT) are bound by the x gather, not by streaming A: measured on MI355X
// 19 % L1 and 59 % L2 hit rate, 504 cycles mean L1->L2 read latency, 7.5 GB fetched for 2.6 GB of
// algorithmic bytes -- a 4-byte x value costs a 64-byte line and hot columns are scattered over
// lines.  Renumbering the columns by popularity packs the hot x entries into the same lines (16x the
// effective cache capacity for them at every level): C3 2.62 -> 1.9 ms.  Built once per record at its
// first product: column histogram -> stable sort by descending count -> rank of every entry's column
// (col_hot).  Each product then gathers x into rank order (m elements) and runs the same kernel on
// (col_hot, x_hot).  Entry order is untouched, so y is bit-identical to the plain path.
// Auto mode keeps it only where it can pay: x larger than the L2s, most tiles spanning a wide column
// range, and the columns that fit in 2 MB holding at least half of the accesses.
constexpr int64_t kReorderMinBytes = 8 << 20;
constexpr int64_t kReorderHotBytes = 2 << 20;

__global__ __launch_bounds__(kBlock) void reorder_wide_tiles_kernel(const int32_t *__restrict__ col, int64_t nnz,
                                                                    int64_t ntiles, int64_t tile,
                                                                    unsigned long long *__restrict__ wide) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= ntiles) return;
  const int64_t a = t * tile, b = (a + tile - 1 < nnz ? a + tile - 1 : nnz - 1);
  const int64_t d = (int64_t)col[b] - (int64_t)col[a];
  if (d > 65536 || d < -65536) atomicAdd(wide, 1ull);
}

// every stride-th entry: the ranking and the hot-mass test need the popularity profile, not exact counts, and
// counting all 3e8 entries of C3 took 28 ms of atomics on the hottest counters
__global__ __launch_bounds__(kBlock) void reorder_count_kernel(const int32_t *__restrict__ col, int64_t nnz,
                                                               int64_t stride, uint32_t *__restrict__ counts) {
  const int64_t samples = (nnz + stride - 1) / stride;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < samples; i += (int64_t)gridDim.x * kBlock)
    atomicAdd(&counts[col[i * stride]], 1u);
}

// key = ~count: ascending key order = descending popularity; ties keep ascending column (stable sort)
__global__ __launch_bounds__(kBlock) void reorder_keys_kernel(const uint32_t *__restrict__ counts, int64_t m,
                                                              uint32_t *__restrict__ keys) {
  const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (c < m) keys[c] = ~counts[c];
}

__global__ __launch_bounds__(kBlock) void reorder_rank_kernel(const uint32_t *__restrict__ perm,
                                                              const uint32_t *__restrict__ sorted_keys, int64_t m,
                                                              int64_t hot, uint32_t *__restrict__ rank_of,
                                                              unsigned long long *__restrict__ hot_mass) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  unsigned long long mine = 0;
  if (r < m) {
    rank_of[perm[r]] = (uint32_t)r;
    if (r < hot) mine = (unsigned long long)(~sorted_keys[r]);
  }
  mine = wave_reduce(mine, OpAdd());
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(hot_mass, mine);
}

__global__ __launch_bounds__(kBlock) void reorder_remap_kernel(const int32_t *__restrict__ col,
                                                               const uint32_t *__restrict__ rank_of, int64_t nnz,
                                                               int32_t *__restrict__ col_hot) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e < nnz) col_hot[e] = (int32_t)rank_of[col[e]];
}

template <class T>
__global__ __launch_bounds__(kBlock) void reorder_gather_x_kernel(const T *__restrict__ x,
                                                                  const uint32_t *__restrict__ perm, int64_t m,
                                                                  T *__restrict__ x_hot) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r < m) x_hot[r] = x[perm[r]];
}

// two-pass path for gather-bound power-law records (spmv_pb.hpp)
template <class T>
int pb_prepare(futhark_context *ctx, const CsrBody<T> *a, bool vetted, bool flat = false);
template <class T>
int spmv_pb(futhark_context *ctx, const CsrBody<T> *a, const T *x, T *y);

// Decides and, if worthwhile, builds the renumbering.  One-time host synchronisations (two small reads).
template <class T>
int reorder_prepare(futhark_context *ctx, const CsrBody<T> *a) {
  a->reorder_state = -1;
  if (ctx->opt_spmv_reorder < 0 || !a->col32 || a->nnz == 0) return 0;
  const bool force = ctx->opt_spmv_reorder > 0;  // tests: any size
  if ((!force && a->m * (int64_t)sizeof(T) < kReorderMinBytes) || a->m >= 0x7fffffffLL) return 0;
  unsigned long long *stat = reinterpret_cast<unsigned long long *>(ctx->d_small);
  int64_t host = 0;
  if (!force) {
    SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
    hipLaunchKernelGGL(reorder_wide_tiles_kernel, dim3((unsigned)cdiv(a->ntiles, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const int32_t *)a->col32, a->nnz, a->ntiles, a->tile, stat);
    SAMD_LAUNCH_CHECK(ctx);
    SAMD_TRY(read_small(ctx, ctx->d_small, 1, &host));
    if (host * 2 < a->ntiles) return 0;  // banded / stencil-like: the gather is already local
  }
  int64_t stride = a->nnz >> 25;  // about 2^25 samples (all entries below that)
  if (stride < 1) stride = 1;
  if (stride % 2 == 0) stride++;
  const int64_t samples = cdiv(a->nnz, stride);
  Tmp<uint32_t> counts(ctx), keys_b(ctx), pay_a(ctx), pay_b(ctx);
  SAMD_TRY(counts.get(a->m));
  SAMD_TRY(keys_b.get(a->m));
  SAMD_TRY(pay_a.get(a->m));
  SAMD_TRY(pay_b.get(a->m));
  SAMD_HIP(ctx, hipMemsetAsync(counts.p, 0, sizeof(uint32_t) * (size_t)a->m, ctx->stream));
  {
    const int64_t want = cdiv(cdiv(a->nnz, stride), kBlock);
    const int64_t cap = (int64_t)ctx->num_cu * 32;
    hipLaunchKernelGGL(reorder_count_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(kBlock), 0, ctx->stream,
                       (const int32_t *)a->col32, a->nnz, stride, counts.p);
    SAMD_LAUNCH_CHECK(ctx);
  }
  const unsigned mgrid = (unsigned)cdiv(a->m, kBlock);
  hipLaunchKernelGGL(reorder_keys_kernel, dim3(mgrid), dim3(kBlock), 0, ctx->stream, (const uint32_t *)counts.p, a->m,
                     counts.p);  // in place: counts now holds the keys
  SAMD_LAUNCH_CHECK(ctx);
  uint32_t *ks = nullptr;
  const uint32_t *perm = nullptr;
  SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small + 2, 0, sizeof(int64_t), ctx->stream));
  SAMD_TRY((radix_sort_pairs<uint32_t, uint32_t>(ctx, counts.p, keys_b.p, (const uint32_t *)nullptr, pay_a.p, pay_b.p,
                                                  a->m, 0, 32, &ks, &perm, (int *)(ctx->d_small + 2))));
  // perm[rank] = column; keep it, and turn it into rank_of for the remap
  uint32_t *hot_perm = nullptr;
  SAMD_TRY(dalloc(ctx, &hot_perm, a->m));
  SAMD_HIP(ctx, hipMemcpyAsync(hot_perm, perm, sizeof(uint32_t) * (size_t)a->m, hipMemcpyDeviceToDevice, ctx->stream));
  uint32_t *rank_of = (perm == pay_a.p) ? pay_b.p : pay_a.p;  // the payload buffer the sort did not end in
  const int64_t hot = kReorderHotBytes / (int64_t)sizeof(T) < a->m ? kReorderHotBytes / (int64_t)sizeof(T) : a->m;
  SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
  hipLaunchKernelGGL(reorder_rank_kernel, dim3(mgrid), dim3(kBlock), 0, ctx->stream, (const uint32_t *)hot_perm,
                     (const uint32_t *)ks, a->m, hot, rank_of, stat);
  SAMD_LAUNCH_CHECK(ctx);
  int64_t h2[3];
  if (read_small(ctx, ctx->d_small, 3, h2) != 0) { dfree(ctx, hot_perm); return 1; }
  if ((int)(h2[2] & 0xffffffff) != 0) { dfree(ctx, hot_perm); return set_error(ctx, "internal: radix sort look-back timed out"); }
  if (!force && h2[0] * 2 < samples) {  // flat popularity: nothing to pack -- the two-pass product, else column blocks
    dfree(ctx, hot_perm);
    a->gather_flat = true;
    if (a->pb_state == 0) SAMD_TRY(pb_prepare<T>(ctx, a, true, true));
    return 0;
  }
  // Large, wide, skewed: for 4-byte values the two-pass product with the x and y slices in LDS is faster than any
  // renumbering (C3: 1.25 against 2.08 ms) -- if the record qualifies it takes over and nothing more is built here.
  if (!force && a->pb_state == 0) {
    if (pb_prepare<T>(ctx, a, true) != 0) { dfree(ctx, hot_perm); return 1; }
    if (a->pb_state == 1) { dfree(ctx, hot_perm); return 0; }
  }
  int32_t *col_hot = nullptr;
  T *x_hot = nullptr;
  if (dalloc(ctx, &col_hot, a->nnz) != 0 || dalloc(ctx, &x_hot, a->m) != 0) {
    dfree(ctx, hot_perm); dfree(ctx, col_hot);
    return 1;
  }
  hipLaunchKernelGGL(reorder_remap_kernel, dim3((unsigned)cdiv(a->nnz, kBlock)), dim3(kBlock), 0, ctx->stream,
                     (const int32_t *)a->col32, (const uint32_t *)rank_of, a->nnz, col_hot);
  SAMD_LAUNCH_CHECK(ctx);
  a->hot_perm = hot_perm;
  a->col_hot = col_hot;
  a->x_hot = x_hot;
  a->reorder_state = 1;
  return 0;
}

// ------------------------------------------------------------- column blocks
// Wide matrices with flat column popularity (uniform random: BASELINE configs[4]) are bound by the x gather
// as well, but no renumbering helps: every gather misses the L2s and each miss moves a whole line over the
// fabric (measured: 1e9 f64 gathers run at the rate of ~128 B each).  Such a record gets, at its first
// product, a second copy of its entries cut into PIECES: (panel of kCblkRows rows) x (block of columns whose
// slice of x, kCblkBytes, stays in every XCD's L2), entries of a piece contiguous and in (row, column) order,
// the pieces of a column block consecutive.  A product is one launch per column block, in ascending order,
// workgroup = piece, y += A_b x: all gathers of a launch fall into one slice and hit L2; a row's entries are
// added strictly left to right (the oracle's order); no atomics, no fix-up.  Per piece and row there is one
// 16-bit end offset (cblk_end) and y is read and written once per block that holds entries of the row,
// which is why auto mode wants about one entry per row and block on average.
// Measured (f64, uniform random, m = 1e7, 100 entries per row): 1.25e6 rows 2.37 -> 1.04 ms, 1e7 rows
// (1e9 entries) 19.0 -> 6.3 ms.  What remains is the rate at which a CU turns over L1 misses that hit L2
// (~140 G gathers/s chip-wide), not HBM.
constexpr int64_t kCblkBytes = 2 << 20;
constexpr int kCblkRpt = 4;                   // rows per thread
constexpr int kCblkRows = kBlock * kCblkRpt;  // rows per panel (one workgroup)
constexpr int kCblkCap = 4096;                // products staged in LDS at a time

__device__ __forceinline__ int64_t cblk_lower_bound(const int32_t *__restrict__ col, int64_t lo, int64_t hi, int64_t c) {
  while (lo < hi) {  // first position in [lo, hi) whose column is >= c
    const int64_t mid = lo + (hi - lo) / 2;
    if ((int64_t)col[mid] < c) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// grid = pieces (block-major), thread = kCblkRpt consecutive rows of the panel.  COPY=false: per-row end
// offsets inside the piece and the piece's size; COPY=true: the entries move to their place.
template <class T, bool COPY>
__global__ __launch_bounds__(kBlock) void cblk_build_kernel(const int64_t *__restrict__ row_ptr,
                                                            const int32_t *__restrict__ col, const T *__restrict__ val,
                                                            int64_t n, int64_t m, int64_t nb, int64_t width,
                                                            uint16_t *__restrict__ row_end,
                                                            int64_t *__restrict__ piece_ptr /* COPY: starts; else [1 + piece] sizes */,
                                                            int32_t *__restrict__ pc_col, T *__restrict__ pc_val,
                                                            int *__restrict__ too_big) {
  __shared__ int64_t sh[kWaves];
  const int64_t piece = blockIdx.x, panels = gridDim.x / nb;  // block-major: a launch of the product reads one run
  const int64_t b = piece / panels, p = piece - b * panels;
  const int64_t r0 = p * kCblkRows + (int64_t)threadIdx.x * kCblkRpt;
  const int64_t clo = b * width, chi = (b + 1) * width < m ? (b + 1) * width : m;
  int64_t s[kCblkRpt], cnt[kCblkRpt], mine = 0;
#pragma unroll
  for (int j = 0; j < kCblkRpt; j++) {
    s[j] = 0;
    cnt[j] = 0;
    if (r0 + j < n) {
      const int64_t rs = row_ptr[r0 + j], re = row_ptr[r0 + j + 1];
      s[j] = cblk_lower_bound(col, rs, re, clo);
      cnt[j] = cblk_lower_bound(col, s[j], re, chi) - s[j];
    }
    mine += cnt[j];
  }
  int64_t v[1] = {mine}, total;
  block_scan_striped<int64_t, OpAdd, 1>(v, (int64_t)0, OpAdd(), sh, total);
  int64_t off = v[0] - mine;  // entries of the piece before this thread's rows
  if (!COPY) {
    if (total > 65535 && threadIdx.x == 0) *too_big = 1;
#pragma unroll
    for (int j = 0; j < kCblkRpt; j++) {
      off += cnt[j];
      row_end[piece * kCblkRows + threadIdx.x * kCblkRpt + j] = (uint16_t)(off > 65535 ? 65535 : off);
    }
    if (threadIdx.x == 0) piece_ptr[1 + piece] = total;
  } else {
    int64_t dst = piece_ptr[piece] + off;
#pragma unroll
    for (int j = 0; j < kCblkRpt; j++)
      for (int64_t q = 0; q < cnt[j]; q++, dst++) {
        pc_col[dst] = col[s[j] + q];
        pc_val[dst] = val[s[j] + q];
      }
  }
}

// One launch per column block (the launch boundary is what keeps all workgroups on the same slice of x:
// a single persistent sweep drifts apart by more than a block within the dispatch skew alone and then
// every gather misses L2 again -- measured).  Workgroup = piece.  Entry-parallel gather: every thread fetches
// a few entries of the piece, gathers their x values (all gathers of the piece in flight at once, whatever
// the row lengths) and leaves the products in LDS; after the barrier thread = row adds its products from
// the left and adds the sum to y (first block: stores it).  Rows without entries in the block are skipped.
template <class T>
__global__ __launch_bounds__(kBlock) void spmv_cblk_kernel(const int64_t *__restrict__ piece_ptr,
                                                           const uint16_t *__restrict__ row_end,
                                                           const int32_t *__restrict__ pc_col,
                                                           const T *__restrict__ pc_val, const T *__restrict__ x,
                                                           T *__restrict__ y, int64_t n, int64_t panels, int64_t b) {
  constexpr int E = kCblkCap / kBlock;  // entries per thread and chunk
  __shared__ T sprod[kCblkCap];
  const int tid = threadIdx.x;
  const int64_t piece = b * panels + blockIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * kCblkRows + (int64_t)tid * kCblkRpt;
  const int64_t ps = piece_ptr[piece];
  const int cnt = (int)(piece_ptr[piece + 1] - ps);
  int ends[kCblkRpt + 1];  // ends[j]: where row j of this thread starts, ends[j + 1]: where it ends
  {
    const uint16_t *re = row_end + piece * kCblkRows + (int64_t)tid * kCblkRpt;
    ends[0] = tid ? (int)re[-1] : 0;
#pragma unroll
    for (int j = 0; j < kCblkRpt; j++) ends[j + 1] = (int)re[j];
  }
  T acc[kCblkRpt];
#pragma unroll
  for (int j = 0; j < kCblkRpt; j++) {
    acc[j] = (T)0;
    if (b > 0 && ends[j + 1] > ends[j]) acc[j] = y[r0 + j];  // requested now, needed after the gather
  }
  for (int c0 = 0; c0 < cnt; c0 += kCblkCap) {
    const int cc = cnt - c0 < kCblkCap ? cnt - c0 : kCblkCap;
    int32_t c[E];
    T v[E], xv[E];
#pragma unroll
    for (int i = 0; i < E; i++) {
      const int e = tid + i * kBlock;
      const bool ok = e < cc;
      c[i] = ok ? ntload(&pc_col[ps + c0 + e]) : -1;
      v[i] = ok ? ntload(&pc_val[ps + c0 + e]) : (T)0;
    }
#pragma unroll
    for (int i = 0; i < E; i++) xv[i] = c[i] >= 0 ? x[c[i]] : (T)0;
    if (c0) __syncthreads();  // the previous chunk's readers are done
#pragma unroll
    for (int i = 0; i < E; i++) sprod[tid + i * kBlock] = (T)(v[i] * xv[i]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kCblkRpt; j++) {
      const int qs = (ends[j] > c0 ? ends[j] : c0) - c0, qe = (ends[j + 1] < c0 + cc ? ends[j + 1] : c0 + cc) - c0;
      for (int q = qs; q < qe; q++) acc[j] = (T)(acc[j] + sprod[q]);
    }
  }
#pragma unroll
  for (int j = 0; j < kCblkRpt; j++)
    if (r0 + j < n && (b == 0 || ends[j + 1] > ends[j])) y[r0 + j] = acc[j];
}

template <class T>
int cblk_prepare(futhark_context *ctx, const CsrBody<T> *a) {
  a->cblk_state = -1;
  if (ctx->opt_spmv_cblk < 0 || !a->col32 || a->nnz == 0 || a->n == 0) return 0;
  const bool force = ctx->opt_spmv_cblk > 0;  // tests: any size, block width in columns given by the option
  int64_t width = force ? ctx->opt_spmv_cblk : kCblkBytes / (int64_t)sizeof(T);
  const int64_t nb = cdiv(a->m, width);
  // measured break-even near 0.4 entries per row and block (n = m = 1e7: 12.5 per row, 39 blocks: 2.61 vs 2.41 ms)
  if (!force && (!a->gather_flat || nb < 4 || 4 * a->nnz < 3 * a->n * nb)) return 0;
  width = cdiv(a->m, nb);
  const int64_t panels = cdiv(a->n, kCblkRows), pieces = panels * nb;
  if (pieces >= 0x7fffffffLL) return 0;
  Tmp<int64_t> piece_ptr(ctx);
  Tmp<uint16_t> row_end(ctx);
  Tmp<int32_t> pc_col(ctx);
  Tmp<T> pc_val(ctx);
  SAMD_TRY(piece_ptr.get(pieces + 1));
  SAMD_TRY(row_end.get(pieces * kCblkRows));
  SAMD_TRY(pc_col.get(a->nnz));
  SAMD_TRY(pc_val.get(a->nnz));
  int *flag = reinterpret_cast<int *>(ctx->d_small);
  SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
  SAMD_HIP(ctx, hipMemsetAsync(piece_ptr.p, 0, sizeof(int64_t), ctx->stream));
  hipLaunchKernelGGL((cblk_build_kernel<T, false>), dim3((unsigned)pieces), dim3(kBlock), 0, ctx->stream,
                     (const int64_t *)a->row_ptr, (const int32_t *)a->col32, (const T *)a->vals, a->n, a->m, nb, width,
                     row_end.p, piece_ptr.p, pc_col.p, pc_val.p, flag);
  SAMD_LAUNCH_CHECK(ctx);
  int64_t host = 0;
  SAMD_TRY(read_small(ctx, ctx->d_small, 1, &host));
  if ((int)(host & 0xffffffff) != 0) return 0;  // a piece too large for 16-bit offsets: rows far too uneven for this path
  SAMD_TRY((device_scan<int64_t, OpAdd, PtrIn<int64_t>, false>(ctx, PtrIn<int64_t>{piece_ptr.p + 1}, pieces, (int64_t)0,
                                                                OpAdd(), piece_ptr.p + 1, (int64_t *)nullptr)));
  hipLaunchKernelGGL((cblk_build_kernel<T, true>), dim3((unsigned)pieces), dim3(kBlock), 0, ctx->stream,
                     (const int64_t *)a->row_ptr, (const int32_t *)a->col32, (const T *)a->vals, a->n, a->m, nb, width,
                     row_end.p, piece_ptr.p, pc_col.p, pc_val.p, flag);
  SAMD_LAUNCH_CHECK(ctx);
  a->cblk_nb = nb;
  a->cblk_piece_ptr = piece_ptr.release();
  a->cblk_end = row_end.release();
  a->cblk_col = pc_col.release();
  a->cblk_val = pc_val.release();
  a->cblk_state = 1;
  return 0;
}

// ------------------------------------------------------------ 16-bit columns
// One block per tile: the tile's smallest column goes to tile_c0, every column is stored as its 16-bit distance
// from it; a tile spanning 65536 columns or more raises the flag (the record then keeps its 32-bit columns).
__global__ __launch_bounds__(kBlock) void col16_build_kernel(const int32_t *__restrict__ col, int64_t nnz, int64_t tile,
                                                             int32_t *__restrict__ tile_c0,
                                                             uint16_t *__restrict__ col16, int *__restrict__ too_wide) {
  __shared__ int smin[kWaves], smax[kWaves];
  const int64_t base = (int64_t)blockIdx.x * tile;
  const int per = (int)(tile / kBlock);
  int mn = 0x7fffffff, mx = -1;
  for (int k = 0; k < per; k++) {
    const int64_t e = base + (int64_t)k * kBlock + threadIdx.x;
    if (e < nnz) {
      const int c = col[e];
      mn = c < mn ? c : mn;
      mx = c > mx ? c : mx;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int a = wshfl_xor(mn, d), b = wshfl_xor(mx, d);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < kWaves; w++) {
    mn = smin[w] < mn ? smin[w] : mn;
    mx = smax[w] > mx ? smax[w] : mx;
  }
  if (threadIdx.x == 0) {
    tile_c0[blockIdx.x] = mn;
    if ((int64_t)mx - (int64_t)mn > 65535) *too_wide = 1;
  }
  for (int k = 0; k < per; k++) {
    const int64_t e = base + (int64_t)k * kBlock + threadIdx.x;
    if (e < nnz) col16[e] = (uint16_t)(col[e] - mn);
  }
}

// Decides and, if every tile qualifies, builds the 16-bit columns.  One-time host synchronisation.
template <class T>
int col16_prepare(futhark_context *ctx, const CsrBody<T> *a) {
  a->col16_state = -1;
  if (ctx->opt_spmv_col16 < 0 || !a->col32 || a->ntiles <= kInlineMaxTiles || a->tile != kSpmvTile) return 0;
  Tmp<uint16_t> c16(ctx);
  Tmp<int32_t> c0(ctx);
  SAMD_TRY(c16.get(a->nnz));
  SAMD_TRY(c0.get(a->ntiles));
  int *flag = reinterpret_cast<int *>(ctx->d_small);
  SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
  hipLaunchKernelGGL(col16_build_kernel, dim3((unsigned)a->ntiles), dim3(kBlock), 0, ctx->stream,
                     (const int32_t *)a->col32, a->nnz, a->tile, c0.p, c16.p, flag);
  SAMD_LAUNCH_CHECK(ctx);
  int64_t host = 0;
  SAMD_TRY(read_small(ctx, ctx->d_small, 1, &host));
  if ((int)(host & 0xffffffff) != 0) return 0;  // some tile is too wide (random, power-law): 32-bit columns stay
  a->col16 = c16.release();
  a->tile_c0 = c0.release();
  a->col16_state = 1;
  return 0;
}

// the plan's two statistics: read back by the first product with a record, then cached
template <class T>
int plan_stats(futhark_context *ctx, const CsrBody<T> *a) {
  if (a->max_tile_rows >= 0) return 0;
  int64_t stats[2];
  SAMD_TRY(read_small(ctx, a->tile_lb + a->ntiles + 1, 2, stats));
  a->spill_max = stats[0];
  a->max_tile_rows = stats[1];
  return 0;
}

// y = 0 by a kernel of the library's own.  The allocation-free entry uses it instead of hipMemsetAsync: captured into
// a graph, that memset came back as something else than zeros on replay (ROCm 7.0 runtime under torch.cuda.graph:
// the zero-filled rows of tests/test_zz_no_alloc.py read 4.3e-310), and a kernel node is what the rest of the call is.
template <class T>
__global__ __launch_bounds__(kBlock) void zero_fill_kernel(T *__restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) y[i] = (T)0;
}

template <class T>
int zero_fill(futhark_context *ctx, T *y, int64_t n) {
  if (n <= 0) return 0;
  const int64_t want = cdiv(n, kBlock), cap = (int64_t)ctx->num_cu * 16;
  hipLaunchKernelGGL((zero_fill_kernel<T>), dim3((unsigned)(want < cap ? want : cap)), dim3(kBlock), 0, ctx->stream, y, n);
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

// 16-bit tile-relative row offsets and the list of rows the fix-up launch folds, for the stream kernel on large
// records (one-off, one host synchronisation for the list's length)
template <class T>
int row_rel_prepare(futhark_context *ctx, const CsrBody<T> *a) {
  if (a->row_rel) return 0;
  uint16_t *rel = nullptr;
  SAMD_TRY(dalloc(ctx, &rel, a->n + 1));
  hipLaunchKernelGGL(row_rel_kernel, dim3((unsigned)cdiv(a->n + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                     (const int64_t *)a->row_ptr, a->n, a->tile, a->ntiles, rel);
  SAMD_LAUNCH_CHECK(ctx);
  a->row_rel = rel;
  if (a->ntiles > 1 && a->ntiles < 0x7fffffffLL) {
    Tmp<int32_t> fl(ctx);
    SAMD_TRY(fl.get(4 * a->ntiles));
    SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
    hipLaunchKernelGGL((fixup_list_kernel<kSpmvTile>), dim3((unsigned)cdiv(a->ntiles - 1, kBlock)), dim3(kBlock), 0,
                       ctx->stream, (const int64_t *)a->row_ptr, (const int64_t *)a->tile_lb, a->ntiles, fl.p,
                       fl.p + a->ntiles, fl.p + 2 * a->ntiles, fl.p + 3 * a->ntiles,
                       reinterpret_cast<unsigned long long *>(ctx->d_small));
    SAMD_LAUNCH_CHECK(ctx);
    int64_t cnt = 0;
    SAMD_TRY(read_small(ctx, ctx->d_small, 1, &cnt));
    a->fix_count = cnt;
    a->fix_list = fl.release();
  }
  return 0;
}

// head_override: the array that receives the tiles' head sums instead of the record's own (spmv_noalloc);
// with it nothing may be allocated here either, so a record without row_rel takes the INLINE kernels
// (they read row_ptr itself and are valid at any size, just slower on large records).
template <class T, class I>
int spmv_typed(futhark_context *ctx, const CsrBody<T> *a, const I *col, const T *x, T *y,
               const int32_t *tile_c0 = nullptr, T *head_override = nullptr, bool build_nothing = false) {
  ctx->last_spmv_col_bytes = (int64_t)sizeof(I);
  const bool noalloc = head_override != nullptr;
  T *const head = noalloc ? head_override : a->head;
  int64_t algo = ctx->opt_spmv_algo;
  if (algo == 0 || noalloc) algo = 1;
  if (algo == 1) {
    if (a->tile != kSpmvTile) return set_error(ctx, "spmv: plan tile mismatch");
    int force = (int)ctx->opt_spmv_lanes;
    if (force < 0 || force > 64 || (force & (force - 1))) return set_error(ctx, "spmv: spmv_lanes must be a power of two <= 64");
    // Small records are launch-bound: there the owner tile finishes short spills itself and the fix-up
    // launch is dropped when no row spills far.  Large records keep the fix-up launch (measured faster:
    // C2 0.238 vs 0.244 ms), so nothing is read back for them.
    // build_nothing (a record too young to have paid for its 16-bit row offsets, see spmv_may_build): as noalloc
    const bool inl = (a->ntiles > 1 && a->ntiles <= kInlineMaxTiles) || ((noalloc || build_nothing) && !a->row_rel);
    // The plan's two statistics reach the host through a read-back that waits for the stream (once per record).
    // The allocation-free entry must not wait (its callers may be capturing the stream into a graph), so on a
    // record no ordinary product has seen yet it assumes the worst instead: y zero-filled, fix-up launch kept.
    if (!noalloc) SAMD_TRY(plan_stats(ctx, a));
    const bool stats = a->max_tile_rows >= 0;
    if (!stats || a->max_tile_rows > kDenseRowsMax) {  // some tile takes the many-empty-rows path, which writes no zeros
      if (noalloc) SAMD_TRY(zero_fill(ctx, y, a->n));
      else SAMD_HIP(ctx, hipMemsetAsync(y, 0, sizeof(T) * (size_t)a->n, ctx->stream));
    }
    // gathers in flight per lane: what a typical row fragment needs (mean row length over the lanes it gets)
    int unroll = 8;
    {
      const int64_t avg = a->n > 0 ? a->nnz / a->n : 0;
      int L = 1;
      while (L < 64 && L * 2 <= avg / 4) L *= 2;
      const int64_t per_lane = (avg + L - 1) / L;
      unroll = per_lane <= 6 ? 6 : 8;
    }
    if (!inl && !a->row_rel) SAMD_TRY(row_rel_prepare(ctx, a));
#define SAMD_SPMV_LAUNCH(INL, UU)                                                                                  \
  hipLaunchKernelGGL((spmv_stream_kernel<T, I, kSpmvTile, INL, UU>), dim3((unsigned)a->ntiles), dim3(kBlock), 0,   \
                     ctx->stream, (const int64_t *)a->row_ptr, col, (const T *)a->vals, x, y,                      \
                     (const int64_t *)a->tile_lb, head, a->nnz, a->ntiles, force, (const uint16_t *)a->row_rel, tile_c0)
    if constexpr (sizeof(I) == 2) {  // 16-bit columns exist for large records only
      if (inl || !tile_c0) return set_error(ctx, "spmv: 16-bit columns on a record that has none");
      if (unroll == 6) SAMD_SPMV_LAUNCH(false, 6); else SAMD_SPMV_LAUNCH(false, 8);
    } else {
      if (inl) { if (unroll == 6) SAMD_SPMV_LAUNCH(true, 6); else SAMD_SPMV_LAUNCH(true, 8); }
      else { if (unroll == 6) SAMD_SPMV_LAUNCH(false, 6); else SAMD_SPMV_LAUNCH(false, 8); }
    }
#undef SAMD_SPMV_LAUNCH
    SAMD_LAUNCH_CHECK(ctx);
    if (!inl && a->fix_list) {
      if (a->fix_count > 0) {
        hipLaunchKernelGGL((spmv_fixup_list_kernel<T>), dim3((unsigned)cdiv(a->fix_count, kBlock)), dim3(kBlock), 0, ctx->stream,
                           (const int32_t *)a->fix_list, (const int32_t *)(a->fix_list + a->ntiles),
                           (const int32_t *)(a->fix_list + 2 * a->ntiles), (const int32_t *)(a->fix_list + 3 * a->ntiles),
                           a->fix_count, (const T *)head, y);
        SAMD_LAUNCH_CHECK(ctx);
      }
    } else if (a->ntiles > 1 && !(inl && stats && a->spill_max <= kSpillInline)) {
      hipLaunchKernelGGL((spmv_fixup_kernel<T, kSpmvTile>), dim3((unsigned)cdiv(a->ntiles - 1, kBlock)), dim3(kBlock),
                         0, ctx->stream, (const int64_t *)a->row_ptr, (const int64_t *)a->tile_lb,
                         (const T *)head, y, a->ntiles, inl);
      SAMD_LAUNCH_CHECK(ctx);
    }
    return 0;
  }
  if constexpr (sizeof(I) == 2) return set_error(ctx, "spmv: 16-bit columns are for the stream kernel only");
  else {
  int lanes = algo == 2 ? 64 : (int)ctx->opt_spmv_lanes;
  if (lanes <= 0) lanes = 8;
  int64_t want = cdiv(a->n * lanes, kBlock);
  const int64_t cap = (int64_t)ctx->num_cu * 32;
  const unsigned grid = (unsigned)(want < cap ? (want > 0 ? want : 1) : cap);
#define SAMD_VEC(L)                                                                                              \
  hipLaunchKernelGGL((spmv_vector_kernel<T, I, L>), dim3(grid), dim3(kBlock), 0, ctx->stream,                    \
                     (const int64_t *)a->row_ptr, col, (const T *)a->vals, x, y, a->n, (const int64_t *)nullptr,       \
                     (int64_t)0)
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
}

// y[rows[i]] += (A x)[i] in ONE launch, for the small records a row-sharded rank multiplies against received x
// entries (a few thousand rows): sub-wavefront-per-row straight from global memory, indirect output.
// Returns 1 if the record is not one of those (the caller then multiplies into a temporary and adds).
template <class T>
int spmv_rows_add_small(futhark_context *ctx, const CsrBody<T> *a, const T *x, T *y, const int64_t *rows,
                        int64_t ylen, bool *done) {
  *done = false;
  if (a->nnz > (4 << 20) || a->n == 0 || ctx->opt_spmv_algo != 0) return 0;
  *done = true;
  if (a->nnz == 0) return 0;
  const int64_t avg = a->nnz / a->n;
  const int lanes = avg <= 2 ? 2 : (avg <= 12 ? 8 : 32);
  const int64_t want = cdiv(a->n * lanes, kBlock), cap = (int64_t)ctx->num_cu * 32;
  const unsigned grid = (unsigned)(want < cap ? (want > 0 ? want : 1) : cap);
#define SAMD_VEC_ROWS(I, COL, L)                                                                                \
  hipLaunchKernelGGL((spmv_vector_kernel<T, I, L>), dim3(grid), dim3(kBlock), 0, ctx->stream,                   \
                     (const int64_t *)a->row_ptr, (const I *)COL, (const T *)a->vals, x, y, a->n, rows, ylen)
  if (a->col32) {
    if (lanes == 2) SAMD_VEC_ROWS(int32_t, a->col32, 2); else if (lanes == 8) SAMD_VEC_ROWS(int32_t, a->col32, 8);
    else SAMD_VEC_ROWS(int32_t, a->col32, 32);
  } else {
    if (lanes == 2) SAMD_VEC_ROWS(int64_t, a->col64, 2); else if (lanes == 8) SAMD_VEC_ROWS(int64_t, a->col64, 8);
    else SAMD_VEC_ROWS(int64_t, a->col64, 32);
  }
#undef SAMD_VEC_ROWS
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

// y = A x without allocating anything (capi.hip: sparse_amd_csr_*_smvm_into_noalloc): the one-pass stream
// kernel on whatever the record already holds, the tiles' head sums in `scratch` (the library's static device
// array) instead of the record's head[].  Nothing is built lazily here: 16-bit columns and row offsets are used
// if an earlier ordinary product made them, the gather-bound paths (renumbering, column blocks, two passes) never.
template <class T>
int spmv_noalloc(futhark_context *ctx, const CsrBody<T> *a, const T *x, T *y, void *scratch, size_t scratch_bytes) {
  if (a->n == 0) return 0;
  if (a->nnz == 0) return zero_fill(ctx, y, a->n);
  if (!scratch) return set_error(ctx, "smvm_into_noalloc: the static scratch array is not available on this device");
  if ((size_t)a->ntiles * sizeof(T) > scratch_bytes)
    return set_error(ctx, "smvm_into_noalloc: record of %lld entries needs %zu bytes of scratch, the static array holds %zu "
                     "(at most %zu entries for this element type)", (long long)a->nnz, (size_t)a->ntiles * sizeof(T),
                     scratch_bytes, scratch_bytes / sizeof(T) * (size_t)kSpmvTile);
  ctx->last_spmv_path = 0;
  T *head = reinterpret_cast<T *>(scratch);
  if (a->col16_state == 1 && a->row_rel && a->ntiles > kInlineMaxTiles)
    return spmv_typed<T, uint16_t>(ctx, a, (const uint16_t *)a->col16, x, y, (const int32_t *)a->tile_c0, head);
  if (a->col32) return spmv_typed<T, int32_t>(ctx, a, a->col32, x, y, nullptr, head);
  return spmv_typed<T, int64_t>(ctx, a, a->col64, x, y, nullptr, head);
}

// ------------------------------------------------------------ amortisation
// Everything a record builds lazily for its products costs many products' worth of time: the two-pass copy 47 ms on
// C3 (a product saves 1.9 ms over the plain kernel: 25 products to break even) and 146 ms on C5 (11 products), the
// 16-bit indices of the stream path 1.2 ms on C2 (30 products).  A caller who multiplies once must not pay that, and
// the library cannot know -- so it rents before it buys: until a record has been multiplied kBuildAfter times it is
// served by kernels that need nothing built (the stream kernel on row_ptr / col32 as assembled), and what it has
// spent by then on the slower kernel is about what the copies cost (the classic 2-competitive rule).  A caller who
// knows better says so: sparse_amd_csr_*_prepare(ctx, A, expected_products) -- 1..kBuildAfter-1 pins the light path,
// anything larger builds at once (inside prepare, so that the first product already runs at its steady state).
// An explicit kernel request (spmv_algo, spmv_pb / spmv_reorder / spmv_cblk > 0) or option spmv_eager = 1 builds at
// the first product, as every record did before round 3.
constexpr int64_t kBuildAfter = 16;

template <class T>
bool spmv_may_build(futhark_context *ctx, const CsrBody<T> *a) {
  if (ctx->opt_spmv_eager > 0 || ctx->opt_spmv_algo != 0 || ctx->opt_spmv_pb > 0 || ctx->opt_spmv_reorder > 0 ||
      ctx->opt_spmv_cblk > 0)
    return true;
  const int64_t budget = a->expected_products > 0 ? a->expected_products : a->products;
  return budget >= kBuildAfter;
}

// The per-record choices and their builds, without a product (idempotent).
template <class T>
int spmv_build(futhark_context *ctx, const CsrBody<T> *a) {
  if (a->n == 0 || a->nnz == 0) return 0;
  const int64_t algo = ctx->opt_spmv_algo;
  if (algo != 0 && algo != 1) return 0;
  if (a->pb_state == 0 && ctx->opt_spmv_pb > 0) SAMD_TRY(pb_prepare<T>(ctx, a, false));
  if (a->pb_state == 1) return 0;
  if (a->reorder_state == 0) SAMD_TRY(reorder_prepare<T>(ctx, a));  // may choose the two-pass path
  if (a->pb_state == 1 || a->reorder_state == 1) return 0;
  if (a->reorder_state == -1 && a->cblk_state == 0) SAMD_TRY(cblk_prepare<T>(ctx, a));
  if (a->cblk_state == 1) return 0;
  if (a->col16_state == 0) SAMD_TRY(col16_prepare<T>(ctx, a));
  if (a->ntiles > kInlineMaxTiles && a->tile == kSpmvTile) {
    SAMD_TRY(plan_stats(ctx, a));
    SAMD_TRY(row_rel_prepare(ctx, a));
  }
  return 0;
}

// sparse_amd_csr_*_prepare
template <class T>
int spmv_prepare(futhark_context *ctx, const CsrBody<T> *a, int64_t expected_products) {
  if (expected_products < 1) return set_error(ctx, "prepare: expected_products must be at least 1");
  a->expected_products = expected_products;
  if (spmv_may_build(ctx, a)) return spmv_build(ctx, a);
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
  a->products++;
  const int64_t algo = ctx->opt_spmv_algo;
  const bool auto_algo = algo == 0 || algo == 1;
  const bool build = spmv_may_build(ctx, a);  // whatever is built already is used either way
  if (auto_algo && a->pb_state == 0 && ctx->opt_spmv_pb > 0) SAMD_TRY(pb_prepare<T>(ctx, a, false));
  if (auto_algo && a->pb_state == 1) { ctx->last_spmv_path = 3; return spmv_pb<T>(ctx, a, x, y); }
  if (auto_algo && build && a->reorder_state == 0) SAMD_TRY(reorder_prepare<T>(ctx, a));  // may choose the two-pass path
  if (auto_algo && a->pb_state == 1) { ctx->last_spmv_path = 3; return spmv_pb<T>(ctx, a, x, y); }
  ctx->last_spmv_path = 0;
  if (auto_algo && a->reorder_state == 1) {
    ctx->last_spmv_path = 1;
    hipLaunchKernelGGL((reorder_gather_x_kernel<T>), dim3((unsigned)cdiv(a->m, kBlock)), dim3(kBlock), 0, ctx->stream,
                       x, (const uint32_t *)a->hot_perm, a->m, a->x_hot);
    SAMD_LAUNCH_CHECK(ctx);
    return spmv_typed<T, int32_t>(ctx, a, a->col_hot, (const T *)a->x_hot, y);
  }
  if (auto_algo && build && a->reorder_state == -1 && a->cblk_state == 0) SAMD_TRY(cblk_prepare<T>(ctx, a));
  if (auto_algo && a->cblk_state == 1) {
    ctx->last_spmv_path = 2;
    const int64_t panels = cdiv(a->n, kCblkRows);
    for (int64_t b = 0; b < a->cblk_nb; b++) {
      hipLaunchKernelGGL((spmv_cblk_kernel<T>), dim3((unsigned)panels), dim3(kBlock), 0, ctx->stream,
                         (const int64_t *)a->cblk_piece_ptr, (const uint16_t *)a->cblk_end,
                         (const int32_t *)a->cblk_col, (const T *)a->cblk_val, x, y, a->n, panels, b);
      SAMD_LAUNCH_CHECK(ctx);
    }
    return 0;
  }
  if (auto_algo && build && a->col16_state == 0) SAMD_TRY(col16_prepare<T>(ctx, a));
  if (auto_algo && a->col16_state == 1 && (build || a->row_rel))
    return spmv_typed<T, uint16_t>(ctx, a, (const uint16_t *)a->col16, x, y, (const int32_t *)a->tile_c0);
  if (a->col32) return spmv_typed<T, int32_t>(ctx, a, a->col32, x, y, nullptr, nullptr, !build);
  return spmv_typed<T, int64_t>(ctx, a, a->col64, x, y, nullptr, nullptr, !build);
}

}  // namespace samd
