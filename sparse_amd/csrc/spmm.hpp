// spmm.hpp -- sparse x dense panel, C[n][k] = A[n][m] B[m][k], B and C row-major.
//
// Not a member of the reference's `compressed` module; defined
// as the column-wise sr.smvm  C[:,j] = smvm A B[:,j]  (compressed.fut:134-146;
// precedent mono.sr.dmsmm, mono.fut:161-162).
//
// spmm_rowpanel: one wavefront owns one row of A and a 64-wide strip of C; lane j
// owns column j of the strip.  The row's (col, val) pairs are read once per strip
// with one coalesced load and broadcast lane by lane (readlane -> SGPR), so every
// B access is a full coalesced row segment (512 B for f64) and each A entry costs
// one FMA per lane.  HBM-bound: 2 nnz k flops over nnz (vb+ib) + (m+n) k vb bytes
// (~2.4 flop/B at k=64, below the f64 ridge).
#pragma once

#include <type_traits>
#include <vector>

#include "spmv.hpp"

namespace samd {

template <class T>
__device__ __forceinline__ T bcast_lane(T v, int src) { return __shfl(v, src, 64); }

template <class T, class I>
__global__ __launch_bounds__(kBlock) void spmm_rowpanel_kernel(const int64_t *__restrict__ row_ptr,
                                                               const I *__restrict__ col, const T *__restrict__ val,
                                                               const T *__restrict__ B, T *__restrict__ C, int64_t n,
                                                               int64_t k) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    const int64_t rs = row_ptr[r], re = row_ptr[r + 1];
    for (int64_t cb = 0; cb < k; cb += 64) {
      const int64_t j = cb + lane;
      const bool live = j < k;
      T acc = (T)0;
      for (int64_t q0 = rs; q0 < re; q0 += 64) {
        const int64_t q = q0 + lane;
        const I myc = q < re ? col[q] : (I)0;
        const T myv = q < re ? val[q] : (T)0;
        const int cnt = (int)((re - q0) < 64 ? (re - q0) : 64);
        for (int u = 0; u < cnt; u++) {
          const int64_t c = (int64_t)bcast_lane(myc, u);
          const T a = bcast_lane(myv, u);
          if (live) acc = (T)(acc + (T)(a * B[c * k + j]));
        }
      }
      if (live) C[r * k + j] = acc;
    }
  }
}

// spmm_stream: the SpMV streaming structure with a 64-wide strip of C per row.
//   * same nnz-balanced tiles, col_idx/vals staged in LDS by line-dense non-temporal loads,
//     the tile's row bounds cached in LDS;
//   * a block is 16 slots x 16 lanes: a slot owns one row at a time, each of its 16 lanes four
//     adjacent columns of the strip (one 32-B access for f64), so a wavefront instruction
//     serves 4 rows x 64 columns and the per-entry index arithmetic is paid once per 4 FMAs;
//     the 16 slots walk ADJACENT rows at the same time;
//   * LDS-tiled row panel: when the tile's columns span at most kWinRows rows of B (banded /
//     FEM matrices: the dense block of the RHS this row panel multiplies), that window of the
//     strip is staged in LDS once with coalesced loads and every entry reads B from LDS; a B
//     row is then fetched once per tile instead of once per entry (the per-CU L1 returns
//     64 B/clk, so a 512-B row segment per entry caps the direct path near 34 TB/s chip-wide);
//   * fragments longer than 64 entries are split over the 16 slots and combined by cross-lane
//     adds + LDS;
//   * rows crossing tiles: owner tile stores, later tiles store a k-wide head row, the
//     fix-up kernel adds them in tile order (no atomics).
//   * dense block on the matrix cores: when the tile is banded (column window <= kWinRows) and its rows
//     are long (mean fragment >= kMfmaMinLen entries, at most 64 rows), the rows are laid out as
//     up to four dense 16 x window blocks in LDS (zero-filled) and each wavefront multiplies them with its
//     16 columns of the strip's B window by v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32: one B fragment
//     (read once from global memory into registers) serves all row groups, and a 16-row group costs
//     2 LDS reads per MFMA instead of 2 per entry and 16 lanes.  A window slice that holds a non-finite
//     B value is multiplied entry by entry instead (0 * inf of the zero fill must not reach C).
constexpr int kWinRows = 72;    // rows of B (x 64 columns) staged per tile
constexpr int kMfmaMinLen = 16; // mean row-fragment length from which a banded tile goes to the matrix cores
constexpr int kMfmaSteps = kWinRows / 4;


template <class T>
struct Mfma { static constexpr bool ok = false; };
template <>
struct Mfma<double> {
  static constexpr bool ok = true;
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // result register i of lane l holds row (l >> 4) + 4 i, column l & 15
  static __device__ __forceinline__ int row(int lane, int i) { return (lane >> 4) + 4 * i; }
  static __device__ __forceinline__ bool nonfinite(double v) {
    return (__double_as_longlong(v) & 0x7ff0000000000000LL) == 0x7ff0000000000000LL;
  }
};
template <>
struct Mfma<float> {
  static constexpr bool ok = true;
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // result register i of lane l holds row 4 (l >> 4) + i, column l & 15
  static __device__ __forceinline__ int row(int lane, int i) { return (lane >> 4) * 4 + i; }
  static __device__ __forceinline__ bool nonfinite(float v) {
    return (__float_as_int(v) & 0x7f800000) == 0x7f800000;
  }
};
constexpr int kRowCache = 256;  // row bounds of the tile's first rows, read once into LDS

__device__ __forceinline__ int rel_clamp(int64_t v, int64_t base, int tile) {
  const int64_t d = v - base;
  return d < -1 ? -1 : (d > tile + 1 ? tile + 1 : (int)d);
}

template <class T, class I, int TILE, bool VEC>
__global__ __launch_bounds__(kBlock) void spmm_stream_kernel(const int64_t *__restrict__ row_ptr,
                                                             const I *__restrict__ col, const T *__restrict__ val,
                                                             const T *__restrict__ B, T *__restrict__ C,
                                                             const int64_t *__restrict__ tile_lb,
                                                             T *__restrict__ head, int64_t nnz, int64_t ntiles,
                                                             int64_t k, int ablate,
                                                             const int32_t *__restrict__ tile_list, int64_t ntlist) {
  __shared__ __attribute__((aligned(16))) T sval[TILE];
  __shared__ __attribute__((aligned(16))) I scol[TILE];
  __shared__ raw16 sBv[kWinRows * 64 * sizeof(T) / 16];  // B window, addressed in 16-byte units
  T *sB = reinterpret_cast<T *>(sBv);
  __shared__ __attribute__((aligned(32))) T part[kWaves][64];
  __shared__ long long swin[2 * kWaves];
  __shared__ int srel[kRowCache + 1];  // row bounds relative to the tile, clamped to [-1, TILE+1]
  __shared__ int longrows[TILE / 64 + 2];
  __shared__ int nlong, nlist;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = tid >> 4, cg = tid & 15;
  // tile_list: the tiles that are not spmm_mfma_kernel's (null: all tiles)
  const int64_t t = tile_list ? (int64_t)tile_list[xcd_remap(blockIdx.x, ntlist)] : xcd_remap(blockIdx.x, ntiles);
  const int64_t base = t * TILE;
  const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);
  const int64_t lb0 = tile_lb[t], lb1 = tile_lb[t + 1];
  constexpr int CV = TILE * (int)sizeof(I) / (16 * kBlock);
  constexpr int VV = TILE * (int)sizeof(T) / (16 * kBlock);
  if (cnt == TILE) {
    raw16 c[CV], a[VV];
    const raw16 *cp = reinterpret_cast<const raw16 *>(col + base);
    const raw16 *vp = reinterpret_cast<const raw16 *>(val + base);
#pragma unroll
    for (int it = 0; it < CV; it++) c[it] = ntload(cp + it * kBlock + tid);
#pragma unroll
    for (int it = 0; it < VV; it++) a[it] = ntload(vp + it * kBlock + tid);
#pragma unroll
    for (int it = 0; it < CV; it++) reinterpret_cast<raw16 *>(scol)[it * kBlock + tid] = c[it];
#pragma unroll
    for (int it = 0; it < VV; it++) reinterpret_cast<raw16 *>(sval)[it * kBlock + tid] = a[it];
  } else {
    for (int e = tid; e < cnt; e += kBlock) {
      scol[e] = col[base + e];
      sval[e] = val[base + e];
    }
  }
  if (tid == 0) { nlong = 0; nlist = 0; }
  const int64_t rfirst = lb0 > 0 ? lb0 - 1 : 0;
  const int64_t nrows64 = lb1 - rfirst;
  // more rows than kDenseRowsMax start here (a run of empty rows): the host has zero-filled C and only the
  // rows that own entries are visited; their list is built from the entries and lives in the window's LDS
  const bool sparse_rows = nrows64 > kDenseRowsMax;  // uniform over the block
  int nrows = sparse_rows ? 0 : (int)nrows64;
  int64_t *rowlist = reinterpret_cast<int64_t *>(sBv);
  for (int jr = tid; jr <= nrows && jr <= kRowCache; jr += kBlock) srel[jr] = rel_clamp(row_ptr[rfirst + jr], base, TILE);
  __syncthreads();
  if (sparse_rows) {
    for (int p = tid; p < cnt; p += kBlock) {
      const int64_t gp = base + p;
      int64_t lo = rfirst, hi = lb1;  // largest r in [rfirst, lb1) with row_ptr[r] <= gp
      while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (row_ptr[mid] <= gp) lo = mid; else hi = mid;
      }
      if (gp == row_ptr[lo] || p == 0) rowlist[atomicAdd(&nlist, 1)] = lo;
    }
    __syncthreads();
    nrows = nlist;
  }

  // column window of the tile
  long long cmin = 0x7fffffffffffffffLL, cmax = -1;
  for (int e = tid; e < cnt; e += kBlock) {
    const long long c = (long long)scol[e];
    cmin = c < cmin ? c : cmin;
    cmax = c > cmax ? c : cmax;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const long long a = wshfl_xor(cmin, d), b = wshfl_xor(cmax, d);
    cmin = a < cmin ? a : cmin;
    cmax = b > cmax ? b : cmax;
  }
  if (lane == 0) { swin[wave] = cmin; swin[kWaves + wave] = cmax; }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < kWaves; w++) {
    cmin = swin[w] < cmin ? swin[w] : cmin;
    cmax = swin[kWaves + w] > cmax ? swin[kWaves + w] : cmax;
  }
  const int wrows = (int)(!sparse_rows && cmax - cmin + 1 <= kWinRows ? cmax - cmin + 1 : 0);  // 0: no window

  constexpr int P = 16 / (int)sizeof(T);  // elements per 16-byte piece
  constexpr int P16 = P;
  constexpr int H = 4 / P;
  // In window mode the staged column ids become offsets of B's row inside sB, in 16-byte units.
  if (wrows > 0) {
    for (int e = tid; e < cnt; e += kBlock) scol[e] = (I)(((long long)scol[e] - cmin) * (64 / P16));
    __syncthreads();
  }

  // Lane cg of a slot owns 4 columns of the 64-wide strip as 16-byte pieces: piece h holds
  // columns h*16*P + cg*P .. +P-1 (P = 16 B / sizeof(T)), so the 16 lanes of a slot touch 256
  // contiguous bytes per access: coalesced in memory, conflict-free in LDS.
  auto loadB = [&](int64_t c, int64_t cb, bool ok, T (&b)[4]) {
#pragma unroll
    for (int h = 0; h < H; h++) {
      const int jl = h * 16 * P + cg * P;  // first column of the piece, relative to the strip
      if (wrows > 0) {
        const raw16 v = sBv[(int)c + h * 16 + cg];
        const T *pv = reinterpret_cast<const T *>(&v);
#pragma unroll
        for (int p = 0; p < P; p++) b[h * P + p] = pv[p];
      } else if (VEC) {
        raw16 v = {0u, 0u, 0u, 0u};
        if (ok && cb + jl < k) v = *reinterpret_cast<const raw16 *>(&B[c * k + cb + jl]);
        const T *pv = reinterpret_cast<const T *>(&v);
#pragma unroll
        for (int p = 0; p < P; p++) b[h * P + p] = pv[p];
      } else {
#pragma unroll
        for (int p = 0; p < P; p++) b[h * P + p] = (ok && cb + jl + p < k) ? B[c * k + cb + jl + p] : (T)0;
      }
    }
  };
  auto storeRow = [&](T *dst, int64_t cb, const T (&acc)[4], bool nt) {
#pragma unroll
    for (int h = 0; h < H; h++) {
      const int64_t j0 = cb + h * 16 * P + cg * P;
      if (VEC) {
        if (j0 < k) {
          raw16 v;
          T *pv = reinterpret_cast<T *>(&v);
#pragma unroll
          for (int p = 0; p < P; p++) pv[p] = acc[h * P + p];
          if (nt) __builtin_nontemporal_store(v, reinterpret_cast<raw16 *>(dst + j0));
          else *reinterpret_cast<raw16 *>(dst + j0) = v;
        }
      } else {
#pragma unroll
        for (int p = 0; p < P; p++)
          if (j0 + p < k) dst[j0 + p] = acc[h * P + p];
      }
    }
  };

  for (int64_t cb = 0; cb < k; cb += 64) {
    const int64_t j = cb + lane;
    if (wrows > 0) {
      if (cb > 0) __syncthreads();  // previous strip's readers are done with sB
      if (VEC) {
        // the whole window in one round of 16-byte pieces: every load is in flight at once
        constexpr int PPR = 64 / P;                                  // pieces per row of the strip
        constexpr int NP = (kWinRows * PPR + kBlock - 1) / kBlock;   // pieces per thread
        raw16 tmp[NP];
#pragma unroll
        for (int u = 0; u < NP; u++) {
          const int pc = u * kBlock + tid, w = pc / PPR, jp = (pc % PPR) * P;
          tmp[u] = raw16{0u, 0u, 0u, 0u};
          if (w < wrows && cb + jp < k && !(ablate & 4)) tmp[u] = *reinterpret_cast<const raw16 *>(&B[(cmin + w) * k + cb + jp]);
        }
#pragma unroll
        for (int u = 0; u < NP; u++) {
          const int pc = u * kBlock + tid;
          if (pc < wrows * PPR) sBv[pc] = tmp[u];
        }
      } else {
        for (int w0 = wave; w0 < wrows; w0 += kWaves * 8) {  // 8 row segments in flight per wavefront
          T tmp[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int w = w0 + u * kWaves;
            tmp[u] = (w < wrows && j < k) ? B[(cmin + w) * k + j] : (T)0;
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int w = w0 + u * kWaves;
            if (w < wrows) sB[w * 64 + lane] = tmp[u];
          }
        }
      }
      __syncthreads();
    }
    // ---- one slot (16 lanes x 4 columns) per row fragment, adjacent rows on the 16 slots
    for (int jr = slot; jr < nrows; jr += 16) {
      const int64_t r = sparse_rows ? rowlist[jr] : rfirst + jr;
      int rs, re;  // relative to the tile
      if (!sparse_rows && jr < kRowCache) { rs = srel[jr]; re = srel[jr + 1]; }
      else { rs = rel_clamp(row_ptr[r], base, TILE); re = rel_clamp(row_ptr[r + 1], base, TILE); }
      if (r < lb0 && re <= 0) continue;
      const int s = rs > 0 ? rs : 0;
      const int e = re < cnt ? re : cnt;
      if (e - s > 64) {
        if (cb == 0 && cg == 0) {
          const int sl = atomicAdd(&nlong, 1);
          longrows[sl] = jr;
        }
        continue;
      }
      T acc[4] = {(T)0, (T)0, (T)0, (T)0};
      constexpr int U = 4;  // entries in flight per slot: LDS index -> LDS window is a dependent chain
      for (int q = s; q < ((ablate & 1) ? s : e); q += U) {
        T a[U], b[U][4];
        int64_t c[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const bool ok = q + u < e;
          c[u] = (int64_t)scol[ok ? q + u : q];
          a[u] = ok ? sval[q + u] : (T)0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) loadB(c[u], cb, q + u < e, b[u]);
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (int v = 0; v < 4; v++) acc[v] = (T)(acc[v] + (T)(a[u] * b[u][v]));
      }
      if (!(ablate & 2)) storeRow(rs >= 0 ? C + r * k : head + t * k, cb, acc, rs >= 0);
    }
    __syncthreads();
    // ---- long fragments: the 16 slots split the entries; partial strips meet via shuffles + LDS
    const int nl = nlong;
    for (int i = 0; i < nl; i++) {
      const int64_t r = sparse_rows ? rowlist[longrows[i]] : rfirst + longrows[i];
      const int rs = rel_clamp(row_ptr[r], base, TILE), re = rel_clamp(row_ptr[r + 1], base, TILE);
      const int s = rs > 0 ? rs : 0;
      const int e = re < cnt ? re : cnt;
      T acc[4] = {(T)0, (T)0, (T)0, (T)0};
      for (int q = s + slot; q < e; q += 32) {
        T b0[4], b1[4];
        const bool ok1 = q + 16 < e;
        const T a0 = sval[q], a1 = ok1 ? sval[q + 16] : (T)0;
        loadB((int64_t)scol[q], cb, true, b0);
        loadB((int64_t)scol[ok1 ? q + 16 : q], cb, ok1, b1);
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] = (T)(acc[u] + (T)(a0 * b0[u]));
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] = (T)(acc[u] + (T)(a1 * b1[u]));
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {  // the 4 slots of a wavefront hold the same columns 16 lanes apart
        acc[u] = (T)(acc[u] + wshfl_xor(acc[u], 16));
        acc[u] = (T)(acc[u] + wshfl_xor(acc[u], 32));
      }
      if (lane < 16) {
#pragma unroll
        for (int u = 0; u < 4; u++) part[wave][(u / P) * 16 * P + cg * P + (u % P)] = acc[u];
      }
      __syncthreads();
      if (tid < 64 && cb + tid < k) {
        T sum = part[0][tid];
#pragma unroll
        for (int w = 1; w < kWaves; w++) sum = (T)(sum + part[w][tid]);
        if (rs >= 0) C[r * k + cb + tid] = sum;
        else head[t * k + cb + tid] = sum;
      }
      __syncthreads();
    }
  }
}

// ---- dense-block tiles on the matrix cores
// Which tiles qualify is a property of the record: decided once (first product), one byte per tile.
template <class I, int TILE>
__global__ __launch_bounds__(64) void spmm_classify_kernel(const I *__restrict__ col,
                                                           const int64_t *__restrict__ tile_lb, int64_t nnz,
                                                           int64_t ntiles, uint8_t *__restrict__ tile_class,
                                                           unsigned long long *__restrict__ count) {
  const int64_t t = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t base = t * TILE;
  const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);
  const int64_t lb0 = tile_lb[t], lb1 = tile_lb[t + 1];
  const int64_t rfirst = lb0 > 0 ? lb0 - 1 : 0;
  const int64_t nrows = lb1 - rfirst;
  bool ok = nrows > 0 && nrows <= 64 && (int64_t)cnt >= kMfmaMinLen * nrows;  // uniform
  if (ok) {
    long long cmin = 0x7fffffffffffffffLL, cmax = -1;
    for (int e = lane; e < cnt; e += 64) {
      const long long c = (long long)col[base + e];
      cmin = c < cmin ? c : cmin;
      cmax = c > cmax ? c : cmax;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const long long a = wshfl_xor(cmin, d), b = wshfl_xor(cmax, d);
      cmin = a < cmin ? a : cmin;
      cmax = b > cmax ? b : cmax;
    }
    ok = cmax - cmin + 1 <= kWinRows;
  }
  if (lane == 0) {
    tile_class[t] = ok ? 1 : 0;
    if (ok) atomicAdd(count, 1ULL);
  }
}

// One block per qualifying tile (tile_list).  The tile's rows (at most 64) are taken 16 at a time: a group
// becomes a dense [window column][16 rows] block in LDS (two buffers: group g+1 is laid out while group g
// is multiplied; a buffer is cleaned by writing zeros over exactly the entries it held, by the same
// 16 lanes that wrote them).  Wavefront w owns columns 16w..16w+15 of every 64-wide strip of C: its
// B fragments of the whole window come straight from global memory into registers (128-byte row pieces,
// all loads in flight while the first block is built) and serve every row group; A fragments come from
// LDS (64 consecutive elements per read: conflict-free).
template <class T, class I, int TILE>
__global__ __launch_bounds__(kBlock) void spmm_mfma_kernel(
    const int64_t *__restrict__ row_ptr, const I *__restrict__ col, const T *__restrict__ val,
    const T *__restrict__ B, T *__restrict__ C, const int64_t *__restrict__ tile_lb, T *__restrict__ head,
    const int32_t *__restrict__ tile_list, int64_t nlist, int64_t nnz, int64_t k, int64_t m) {
  typedef typename Mfma<T>::acc_t acc_t;
  __shared__ __attribute__((aligned(16))) T sval[TILE];
  __shared__ __attribute__((aligned(16))) I scol[TILE];
  __shared__ raw16 sAv[2 * kWinRows * 16 * sizeof(T) / 16];
  __shared__ int srel[66];
  T *Ad = reinterpret_cast<T *>(sAv);  // [buffer][window column][16 rows]
  constexpr int kBuf = kWinRows * 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = tid >> 4, cg = tid & 15;
  const int64_t t = (int64_t)tile_list[xcd_remap(blockIdx.x, nlist)];
  const int64_t base = t * TILE;
  const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);
  constexpr int CV = TILE * (int)sizeof(I) / (16 * kBlock);
  constexpr int VV = TILE * (int)sizeof(T) / (16 * kBlock);
  raw16 ec[CV], ea[VV];
  if (cnt == TILE) {  // the entries are requested first: nothing below is needed to address them
    const raw16 *cp = reinterpret_cast<const raw16 *>(col + base);
    const raw16 *vp = reinterpret_cast<const raw16 *>(val + base);
#pragma unroll
    for (int it = 0; it < CV; it++) ec[it] = ntload(cp + it * kBlock + tid);
#pragma unroll
    for (int it = 0; it < VV; it++) ea[it] = ntload(vp + it * kBlock + tid);
  }
  const int64_t lb0 = tile_lb[t], lb1 = tile_lb[t + 1];
  const int64_t rfirst = lb0 > 0 ? lb0 - 1 : 0;
  const int nrows = (int)(lb1 - rfirst);  // 1..64 (classified)
  const int ngroups = (nrows + 15) >> 4;
  for (int jr = tid; jr <= nrows; jr += kBlock) srel[jr] = rel_clamp(row_ptr[rfirst + jr], base, TILE);
  constexpr int kZ = 2 * kBuf * (int)sizeof(T) / 16;
  for (int p = tid; p < kZ; p += kBlock) sAv[p] = raw16{0u, 0u, 0u, 0u};
  if (cnt == TILE) {
#pragma unroll
    for (int it = 0; it < CV; it++) reinterpret_cast<raw16 *>(scol)[it * kBlock + tid] = ec[it];
#pragma unroll
    for (int it = 0; it < VV; it++) reinterpret_cast<raw16 *>(sval)[it * kBlock + tid] = ea[it];
  } else {
    for (int e = tid; e < cnt; e += kBlock) {
      scol[e] = col[base + e];
      sval[e] = val[base + e];
    }
  }
  __syncthreads();

  // Every wavefront works out the window by itself (lane = row): columns ascend within a row, so the
  // first / last entry of each row fragment bound its group's columns and the tile's window.
  typedef typename std::conditional<sizeof(I) == 4, int, long long>::type cw_t;  // a column, as wide as stored
  constexpr cw_t kNone = sizeof(I) == 4 ? (cw_t)0x7fffffff : (cw_t)0x7fffffffffffffffLL;
  cw_t c0 = kNone, c1 = -1;
  if (lane < nrows) {
    const int rs = srel[lane], re = srel[lane + 1];
    const int s = rs > 0 ? rs : 0;
    const int e = re < cnt ? re : cnt;
    if (e > s) { c0 = (cw_t)scol[s]; c1 = (cw_t)scol[e - 1]; }
  }
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) {
    const cw_t a = wshfl_xor(c0, d), b = wshfl_xor(c1, d);
    c0 = a < c0 ? a : c0;
    c1 = b > c1 ? b : c1;
  }
  cw_t cmin_v = c0;
#pragma unroll
  for (int d = 32; d >= 16; d >>= 1) {
    const cw_t a = wshfl_xor(cmin_v, d);
    cmin_v = a < cmin_v ? a : cmin_v;
  }
  // the same in every lane: kept in scalar registers, and everything below is relative to it in 32 bits
  int64_t cmin;
  if (sizeof(I) == 4) cmin = (int64_t)__builtin_amdgcn_readfirstlane((int)cmin_v);
  else cmin = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((long long)cmin_v >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)cmin_v));
  const int rel0 = c1 < 0 ? 0x3fffffff : (int)(c0 - cmin_v), rel1 = c1 < 0 ? -1 : (int)(c1 - cmin_v);  // of this lane's group

  const int kq = lane >> 4, jc = lane & 15;
  // B fragments of the window for this wavefront's 16 columns: rows cmin + 4 kk + kq, column j.  Rows past
  // the matrix and columns past k are clamped to valid addresses: their products meet the zero fill / are
  // never stored.  Byte offsets inside the window fit 32 bits (checked by the host).
  const char *Bwin = reinterpret_cast<const char *>(B + cmin * k);
  const uint32_t rowbytes = (uint32_t)k * (uint32_t)sizeof(T);
  const int wlast = (int)((m - cmin) < (int64_t)(kWinRows + 4) ? (m - cmin) : (int64_t)(kWinRows + 4)) - 1;
  T b[kMfmaSteps];
  auto load_b = [&](int64_t j) {
    const uint32_t jb = (uint32_t)(j < k ? j : k - 1) * (uint32_t)sizeof(T);
#pragma unroll
    for (int kk = 0; kk < kMfmaSteps; kk++) {
      const int w = kk * 4 + kq;
      const uint32_t off = (uint32_t)(w < wlast ? w : wlast) * rowbytes + jb;
      b[kk] = *reinterpret_cast<const T *>(Bwin + off);
    }
  };
  const int64_t cb0 = 16 * wave;
  const bool single = k <= 64;  // one strip: its B fragments are requested once, before the first block is built
  if (single && cb0 < k) load_b(cb0 + jc);

  // row jr = 16 g + slot of group g is laid out (or wiped) by the 16 lanes of its slot
  const int cmin32 = (int)cmin;  // only used when columns are stored in 32 bits
  auto lay = [&](int g, T *buf, bool wipe) {
    const int jr = g * 16 + slot;
    if (jr >= nrows) return;
    const int rs = srel[jr], re = srel[jr + 1];
    const int s = rs > 0 ? rs : 0;
    const int e = re < cnt ? re : cnt;
    for (int q = s + cg; q < e; q += 16) {
      const int w = sizeof(I) == 4 ? (int)scol[q] - cmin32 : (int)((int64_t)scol[q] - cmin);
      buf[w * 16 + slot] = wipe ? (T)0 : sval[q];
    }
  };
  lay(0, Ad, false);
  __syncthreads();

  constexpr int kSect = 6;  // MFMA steps per section (3 and 1 measured the same: the MFMA count is not the bound)
  const bool enters = lb0 > 0;  // row rfirst started in an earlier tile: its part here goes to head[t], if any
  const bool enters_live = enters && srel[1] > 0;
  for (int g = 0; g < ngroups; g++) {
    const T *cur = Ad + (g & 1) * kBuf;
    if (g + 1 < ngroups) {
      T *nxt = Ad + ((g + 1) & 1) * kBuf;
      if (g >= 1) lay(g - 1, nxt, true);
      lay(g + 1, nxt, false);
    }
    // MFMA steps (4 window columns each) holding entries of the group; the window's 18 steps are walked
    // in sections of kSect and a group skips the sections it has nothing in (the rest of a section is zero fill)
    const int klo = __builtin_amdgcn_readfirstlane(__shfl(rel0, g * 16, 64)) >> 2;
    const int khi = __builtin_amdgcn_readfirstlane(__shfl(rel1, g * 16, 64)) >> 2;
    // this lane's rows of the group: jr = 16 g + row(lane, i)
    T *Cg = C + (rfirst + g * 16) * k;
    for (int64_t cb = cb0; cb < k; cb += 64) {
      const int64_t j = cb + jc;
      const bool jl = j < k;
      if (!single) load_b(j);
      acc_t acc = acc_t{(T)0, (T)0, (T)0, (T)0};
#pragma unroll
      for (int h = 0; h < kMfmaSteps / kSect; h++) {
        if (klo < (h + 1) * kSect && khi >= h * kSect) {  // uniform
          T a[kSect];
#pragma unroll
          for (int i = 0; i < kSect; i++) a[i] = cur[((h * kSect + i) * 4 + kq) * 16 + jc];
#pragma unroll
          for (int i = 0; i < kSect; i++) acc = Mfma<T>::mma(a[i], b[h * kSect + i], acc);
        }
      }
      const bool bad = Mfma<T>::nonfinite(acc[0]) || Mfma<T>::nonfinite(acc[1]) || Mfma<T>::nonfinite(acc[2]) ||
                       Mfma<T>::nonfinite(acc[3]);
      if (__any(bad)) {
        // a non-finite result: it may be 0 * inf of the zero fill (a non-finite value in a row of B the
        // matrix row does not reference), which must not reach C; these 16 rows x 16 columns are multiplied
        // again entry by entry
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int jr = g * 16 + kq + 4 * i;
          if (jr >= nrows || !jl) continue;
          const int rs = srel[jr], re = srel[jr + 1];
          const int64_t r = rfirst + jr;
          if (r < lb0 && re <= 0) continue;
          const int s = rs > 0 ? rs : 0;
          const int e = re < cnt ? re : cnt;
          T sum = (T)0;
          for (int q = s; q < e; q++) sum = (T)(sum + (T)(sval[q] * B[(int64_t)scol[q] * k + j]));
          (rs >= 0 ? C + r * k : head + t * k)[j] = sum;
        }
        continue;
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int jl16 = Mfma<T>::row(lane, i);  // row inside the group
        const int jr = g * 16 + jl16;
        if (jr < nrows && jl) {
          if (jr == 0 && enters) {
            if (enters_live) head[t * k + j] = acc[i];
          } else {
            __builtin_nontemporal_store(acc[i], Cg + (uint32_t)jl16 * (uint32_t)k + j);
          }
        }
      }
    }
    if (g + 1 < ngroups) __syncthreads();
  }
}

// one block per tile t >= 1: if t is the first spill of its head row, fold the k-wide heads
template <class T, int TILE>
__global__ __launch_bounds__(64) void spmm_fixup_kernel(const int64_t *__restrict__ row_ptr,
                                                        const int64_t *__restrict__ tile_lb,
                                                        const T *__restrict__ head, T *__restrict__ C,
                                                        int64_t ntiles, int64_t k) {
  const int64_t t = (int64_t)blockIdx.x + 1;
  if (t >= ntiles) return;
  const int64_t lb = tile_lb[t];
  if (lb == 0) return;
  const int64_t base = t * TILE;
  const int64_t r = lb - 1;
  const int64_t re = row_ptr[lb];
  if (re <= base) return;
  if (row_ptr[r] < base - TILE) return;
  for (int64_t j = threadIdx.x; j < k; j += 64) {
    T acc = C[r * k + j];
    for (int64_t u = t; u < ntiles && u * TILE < re; u++) acc = (T)(acc + head[u * k + j]);
    C[r * k + j] = acc;
  }
}

// First product of a record: classify its tiles and split the tile ids into the dense-block tiles
// (first n_mfma_tiles of tile_split, ascending) and the rest.  One-off: the class bytes take a round trip
// through the host.
template <class T, class I>
int spmm_classify(futhark_context *ctx, const CsrBody<T> *a, const I *col) {
  if (a->n_mfma_tiles >= 0) return 0;
  Tmp<uint8_t> cls(ctx);
  SAMD_TRY(cls.get(a->ntiles));
  SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
  hipLaunchKernelGGL((spmm_classify_kernel<I, kSpmvTile>), dim3((unsigned)a->ntiles), dim3(64), 0, ctx->stream, col,
                     (const int64_t *)a->tile_lb, a->nnz, a->ntiles, cls.p,
                     reinterpret_cast<unsigned long long *>(ctx->d_small));
  SAMD_LAUNCH_CHECK(ctx);
  int64_t nm = 0;
  SAMD_TRY(read_small(ctx, ctx->d_small, 1, &nm));
  if (nm > 0) {
    std::vector<uint8_t> h((size_t)a->ntiles);
    SAMD_HIP(ctx, hipMemcpyAsync(h.data(), cls.p, (size_t)a->ntiles, hipMemcpyDeviceToHost, ctx->stream));
    SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int32_t> split((size_t)a->ntiles);
    size_t i0 = 0, i1 = (size_t)nm;
    for (int64_t t = 0; t < a->ntiles; t++) {
      if (h[(size_t)t]) split[i0++] = (int32_t)t;
      else split[i1++] = (int32_t)t;
    }
    SAMD_TRY(dalloc(ctx, &a->tile_split, a->ntiles));
    SAMD_HIP(ctx, hipMemcpyAsync(a->tile_split, split.data(), sizeof(int32_t) * (size_t)a->ntiles, hipMemcpyHostToDevice,
                                 ctx->stream));
    SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));  // split is a local
  }
  a->n_mfma_tiles = nm;
  return 0;
}

template <class T, class I>
int spmm_typed(futhark_context *ctx, const CsrBody<T> *a, const I *col, const T *B, int64_t k, T *C) {
  if (ctx->opt_spmm_algo == 2) {
    int64_t want = cdiv(a->n * 64, kBlock);
    const int64_t cap = (int64_t)ctx->num_cu * 64;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    hipLaunchKernelGGL((spmm_rowpanel_kernel<T, I>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)a->row_ptr, col, (const T *)a->vals, B, C, a->n, k);
    SAMD_LAUNCH_CHECK(ctx);
    return 0;
  }
  if (a->tile != kSpmvTile) return set_error(ctx, "spmm: plan tile mismatch");
  Tmp<T> head(ctx);
  SAMD_TRY(head.get(a->ntiles * k));
  SAMD_TRY(plan_stats(ctx, a));
  if (a->max_tile_rows > kDenseRowsMax)  // some tile visits only the rows that own entries: the rest must be zero
    SAMD_HIP(ctx, hipMemsetAsync(C, 0, sizeof(T) * (size_t)(a->n * k), ctx->stream));
  // 16-byte accesses need every row of B, C and of the head rows 16-byte aligned
  const bool vec = ((k * sizeof(T)) % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0);
  const int flags = (int)ctx->opt_ablate;
  const int32_t *vec_list = nullptr;
  int64_t nvec = a->ntiles;
  if constexpr (Mfma<T>::ok) {
    // spmm_algo 3: never the matrix cores; the kernel addresses a B window with 32-bit byte offsets
    if (ctx->opt_spmm_algo != 3 && a->ntiles < 0x7fffffffLL && k * (int64_t)sizeof(T) * (kWinRows + 8) < 0x7fffffffLL) {
      SAMD_TRY((spmm_classify<T, I>(ctx, a, col)));
      if (a->n_mfma_tiles > 0) {
        hipLaunchKernelGGL((spmm_mfma_kernel<T, I, kSpmvTile>), dim3((unsigned)a->n_mfma_tiles), dim3(kBlock), 0,
                           ctx->stream, (const int64_t *)a->row_ptr, col, (const T *)a->vals, B, C,
                           (const int64_t *)a->tile_lb, head.p, (const int32_t *)a->tile_split, a->n_mfma_tiles, a->nnz,
                           k, a->m);
        SAMD_LAUNCH_CHECK(ctx);
        vec_list = a->tile_split + a->n_mfma_tiles;
        nvec = a->ntiles - a->n_mfma_tiles;
      }
    }
  }
  if (nvec == 0) {
  } else if (vec)
    hipLaunchKernelGGL((spmm_stream_kernel<T, I, kSpmvTile, true>), dim3((unsigned)nvec), dim3(kBlock), 0,
                       ctx->stream, (const int64_t *)a->row_ptr, col, (const T *)a->vals, B, C,
                       (const int64_t *)a->tile_lb, head.p, a->nnz, a->ntiles, k, flags, vec_list, nvec);
  else
    hipLaunchKernelGGL((spmm_stream_kernel<T, I, kSpmvTile, false>), dim3((unsigned)nvec), dim3(kBlock), 0,
                       ctx->stream, (const int64_t *)a->row_ptr, col, (const T *)a->vals, B, C,
                       (const int64_t *)a->tile_lb, head.p, a->nnz, a->ntiles, k, flags, vec_list, nvec);
  SAMD_LAUNCH_CHECK(ctx);
  if (a->ntiles > 1) {
    hipLaunchKernelGGL((spmm_fixup_kernel<T, kSpmvTile>), dim3((unsigned)(a->ntiles - 1)), dim3(64), 0, ctx->stream,
                       (const int64_t *)a->row_ptr, (const int64_t *)a->tile_lb, (const T *)head.p, C, a->ntiles, k);
    SAMD_LAUNCH_CHECK(ctx);
  }
  return 0;
}

template <class T>
int spmm(futhark_context *ctx, const CsrBody<T> *a, const T *B, int64_t k, T *C) {
  if (a->n == 0 || k == 0) return 0;
  if (a->nnz == 0) {
    SAMD_HIP(ctx, hipMemsetAsync(C, 0, sizeof(T) * (size_t)(a->n * k), ctx->stream));
    return 0;
  }
  if (a->col32) return spmm_typed<T, int32_t>(ctx, a, a->col32, B, k, C);
  return spmm_typed<T, int64_t>(ctx, a, a->col64, B, k, C);
}

// ------------------------------------------------------------------- smsmm
// compressed.fut:268-331.  The reference expands every A entry k times and every
// B entry n times, sorts all contributions and keeps matching pairs.  The pairs
// that survive are exactly {(A[i,s], B[s,j])}; they are produced here directly
// (expand), in (i, s, j) order, and handed to the same sort/merge/row-offset
// pipeline the reference ends in (:331 sr.sparse).  Duplicate (i,j) products are
// therefore folded in increasing s, the order the reference's sort leaves them in.
template <class I>
__global__ __launch_bounds__(kBlock) void smsmm_count_kernel(const I *__restrict__ colA, int64_t nnzA,
                                                             const int64_t *__restrict__ row_ptr_b,
                                                             int64_t *__restrict__ counts) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= nnzA) return;
  const int64_t s = (int64_t)colA[e];
  counts[e] = row_ptr_b[s + 1] - row_ptr_b[s];
}

template <class K, class T, class IA, class IB>
__global__ __launch_bounds__(kBlock) void smsmm_expand_kernel(const int64_t *__restrict__ row_ptr_a,
                                                              const IA *__restrict__ colA, const T *__restrict__ valA,
                                                              int64_t nnzA, const int64_t *__restrict__ tile_lb_a,
                                                              int64_t tile_a, const int64_t *__restrict__ row_ptr_b,
                                                              const IB *__restrict__ colB, const T *__restrict__ valB,
                                                              const int64_t *__restrict__ offs /*inclusive*/,
                                                              int minor_bits, K *__restrict__ keys,
                                                              T *__restrict__ vals) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= nnzA) return;
  const int64_t i = find_row(row_ptr_a, tile_lb_a, tile_a, e);
  const int64_t s = (int64_t)colA[e];
  const T a = valA[e];
  const int64_t bs = row_ptr_b[s], be = row_ptr_b[s + 1];
  int64_t o = offs[e] - (be - bs);
  for (int64_t q = bs; q < be; q++, o++) {
    keys[o] = shl_safe<K>((K)i, minor_bits) | (K)colB[q];
    vals[o] = (T)(a * valB[q]);
  }
}

// A: record n x m.  Bt: the stored record of B : sc[m][k], i.e. k x m.
template <class T>
int smsmm(futhark_context *ctx, const CsrBody<T> *A, const CsrBody<T> *Bt, CsrBody<T> **out) {
  if (A->m != Bt->m) return set_error(ctx, "smsmm: inner dimensions differ");
  const int64_t n = A->n, k = Bt->n;
  // B as rows: physical conversion of the stored k x m record to m x k
  CsrBody<T> *Brow = nullptr;
  SAMD_TRY(transpose_physical(ctx, Bt, &Brow));
  CsrGuard<T> gb(ctx, Brow);
  Tmp<int64_t> offs(ctx);
  SAMD_TRY(offs.get(A->nnz));
  int64_t total = 0;
  if (A->nnz > 0) {
    const unsigned grid = (unsigned)cdiv(A->nnz, kBlock);
    if (A->col32)
      hipLaunchKernelGGL((smsmm_count_kernel<int32_t>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         (const int32_t *)A->col32, A->nnz, (const int64_t *)Brow->row_ptr, offs.p);
    else
      hipLaunchKernelGGL((smsmm_count_kernel<int64_t>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         (const int64_t *)A->col64, A->nnz, (const int64_t *)Brow->row_ptr, offs.p);
    SAMD_LAUNCH_CHECK(ctx);
    SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
    SAMD_TRY((device_scan<int64_t, OpAdd, PtrIn<int64_t>, false>(ctx, PtrIn<int64_t>{offs.p}, A->nnz, (int64_t)0,
                                                                  OpAdd(), offs.p, ctx->d_small)));
    SAMD_TRY(read_small(ctx, ctx->d_small, 1, &total));
  }
  const int rbits = bits_for(n), cbits = bits_for(k);
  if (rbits + cbits > 64) return set_error(ctx, "smsmm: n*k exceeds 2^64 key space");
  Tmp<T> vals(ctx);
  SAMD_TRY(vals.get(total));
  const unsigned grid = (unsigned)cdiv(A->nnz > 0 ? A->nnz : 1, kBlock);
#define SAMD_EXPAND(K, IA, CA, IB, CB)                                                                          \
  hipLaunchKernelGGL((smsmm_expand_kernel<K, T, IA, IB>), dim3(grid), dim3(kBlock), 0, ctx->stream,             \
                     (const int64_t *)A->row_ptr, (const IA *)CA, (const T *)A->vals, A->nnz,                   \
                     (const int64_t *)A->tile_lb, A->tile, (const int64_t *)Brow->row_ptr, (const IB *)CB,      \
                     (const T *)Brow->vals, (const int64_t *)offs.p, cbits, keys.p, vals.p)
#define SAMD_EXPAND_ALL(K)                                                                                      \
  do {                                                                                                          \
    if (A->nnz > 0 && total > 0) {                                                                              \
      if (A->col32 && Brow->col32) SAMD_EXPAND(K, int32_t, A->col32, int32_t, Brow->col32);                     \
      else if (A->col32) SAMD_EXPAND(K, int32_t, A->col32, int64_t, Brow->col64);                               \
      else if (Brow->col32) SAMD_EXPAND(K, int64_t, A->col64, int32_t, Brow->col32);                            \
      else SAMD_EXPAND(K, int64_t, A->col64, int64_t, Brow->col64);                                             \
      SAMD_LAUNCH_CHECK(ctx);                                                                                   \
    }                                                                                                           \
  } while (0)
  if (rbits + cbits <= 32) {
    Tmp<uint32_t> keys(ctx);
    SAMD_TRY(keys.get(total));
    SAMD_EXPAND_ALL(uint32_t);
    return assemble_core<uint32_t, T>(ctx, keys, total, 0, rbits + cbits, cbits, vals.p, n, k, nullptr, out);
  }
  Tmp<uint64_t> keys(ctx);
  SAMD_TRY(keys.get(total));
  SAMD_EXPAND_ALL(uint64_t);
  return assemble_core<uint64_t, T>(ctx, keys, total, 0, rbits + cbits, cbits, vals.p, n, k, nullptr, out);
#undef SAMD_EXPAND_ALL
#undef SAMD_EXPAND
}

}  // namespace samd
