// spmv_lab.hip -- kernel laboratory for the C2 workload (5-point Laplacian, g^2 grid, f64).
//
// Builds the CSR record analytically on the device, then times SpMV kernel variants
// interleaved in ONE process (median over rounds), checking each against the
// shipped kernel's output.  Not part of the library; used to pick kernel structure.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o bench/spmv_lab bench/spmv_lab.hip
//   ./bench/spmv_lab [g=4096] [rounds=7] [launches=20]
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>

#include "../sparse_amd/csrc/spmv.hpp"

using namespace samd;
constexpr int kLongFrag = 32;  // threshold used by the historical variants below

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e = (x);                                                              \
    if (e != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

__device__ __forceinline__ int64_t lap_row_start(int64_t i, int64_t g) {
  const int64_t last = i - (g - 1) * g;
  return 5 * i - (i + g - 1) / g - i / g - (i < g ? i : g) - (last > 0 ? last : 0);
}

__global__ void gen_laplacian(int64_t g, int64_t *row_ptr, int32_t *col, double *val, double *x) {
  const int64_t n = g * g;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  const int64_t s = lap_row_start(i, g);
  row_ptr[i] = s;
  if (i == n) return;
  const int64_t gx = i % g, gy = i / g;
  int64_t q = s;
  if (gy > 0) { col[q] = (int32_t)(i - g); val[q++] = -1.0; }
  if (gx > 0) { col[q] = (int32_t)(i - 1); val[q++] = -1.0; }
  col[q] = (int32_t)i; val[q++] = 4.0;
  if (gx < g - 1) { col[q] = (int32_t)(i + 1); val[q++] = -1.0; }
  if (gy < g - 1) { col[q] = (int32_t)(i + g); val[q++] = -1.0; }
  x[i] = 1.0 + (double)(i % 7) / 8.0;
}

// ------------------------------------------------------------------ variants
// V_copy: the same bytes as plain streams (ceiling for this byte mix)
__global__ __launch_bounds__(256) void stream_ceiling(const int4 *__restrict__ col, const double2 *__restrict__ val,
                                                      const double2 *__restrict__ rp, const double2 *__restrict__ x,
                                                      double2 *__restrict__ y, int64_t nnz, int64_t n) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nth = (int64_t)gridDim.x * 256;
  double acc = 0;
  for (int64_t i = tid; i < nnz / 4; i += nth) {
    int4 c = col[i];
    double2 a = val[2 * i], b = val[2 * i + 1];
    acc += a.x + a.y + b.x + b.y + (double)(c.x ^ c.y ^ c.z ^ c.w);
  }
  for (int64_t i = tid; i < n / 2; i += nth) {
    double2 r = rp[i], xx = x[i];
    acc += r.x + r.y + xx.x + xx.y;
    y[i] = make_double2(acc, acc);
  }
}

// V1: row metadata prefetched before the products are staged (no dependent global
// load after the barrier); TILE and rows-per-thread as template parameters.
template <int TILE, int RPT>
__global__ __launch_bounds__(kBlock) void spmv_v1(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                  const double *__restrict__ val, const double *__restrict__ x,
                                                  double *__restrict__ y, const int64_t *__restrict__ tile_lb,
                                                  double *__restrict__ head, int64_t nnz, int64_t ntiles) {
  __shared__ __attribute__((aligned(32))) double prod[TILE];
  __shared__ int longrows[TILE / kLongFrag + 2];
  __shared__ int nlong;
  const int tid = threadIdx.x;
  const int64_t t = xcd_remap(blockIdx.x, ntiles);
  const int64_t base = t * TILE;
  const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);
  const int64_t lb0 = tile_lb[t], lb1 = tile_lb[t + 1];

  constexpr int ITERS = TILE / (kBlock * 4);
  const int32_t *cbase = col + base;
  const double *vbase = val + base;
  Vec4<int32_t> c[ITERS];
  Vec4<double> a[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; it++) {
    const int e = (it * kBlock + tid) * 4;
    if (e + 4 <= cnt) {
      c[it] = *reinterpret_cast<const Vec4<int32_t> *>(cbase + e);
      a[it] = *reinterpret_cast<const Vec4<double> *>(vbase + e);
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const bool ok = e + q < cnt;
        c[it].v[q] = ok ? cbase[e + q] : 0;
        a[it].v[q] = ok ? vbase[e + q] : 0.0;
      }
    }
  }
  // row metadata for the first RPT rows of this lane, issued with the streams
  const int64_t rfirst0 = lb0 > 0 ? lb0 - 1 : 0;  // candidate head row; resolved below
  int64_t rs_[RPT], re_[RPT];
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int64_t r = rfirst0 + tid + k * kBlock;
    const bool ok = r < lb1;
    rs_[k] = ok ? row_ptr[r] : 0;
    re_[k] = ok ? row_ptr[r + 1] : 0;
  }
  if (tid == 0) nlong = 0;
#pragma unroll
  for (int it = 0; it < ITERS; it++) {
    const int e = (it * kBlock + tid) * 4;
    Vec4<double> p;
#pragma unroll
    for (int q = 0; q < 4; q++) p.v[q] = (e + q < cnt) ? a[it].v[q] * x[c[it].v[q]] : 0.0;
    *reinterpret_cast<Vec4<double> *>(&prod[e]) = p;
  }
  __syncthreads();
  const int64_t lim = base + cnt;
  const int nrows = (int)(lb1 - rfirst0);
  auto do_row = [&](int64_t r, int64_t rs, int64_t re) {
    // the candidate head row only counts if it really spills into this tile
    if (r < lb0 && re <= base) return;
    const int s = (int)((rs > base ? rs : base) - base);
    const int e = (int)((re < lim ? re : lim) - base);
    if (e - s > kLongFrag) {
      const int slot = atomicAdd(&nlong, 1);
      longrows[slot] = (int)(r - rfirst0);
      return;
    }
    double acc = 0;
    for (int q = s; q < e; q++) acc += prod[q];
    if (rs >= base) y[r] = acc; else head[t] = acc;
  };
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int64_t r = rfirst0 + tid + k * kBlock;
    if (r < lb1) do_row(r, rs_[k], re_[k]);
  }
  for (int j = tid + RPT * kBlock; j < nrows; j += kBlock) {
    const int64_t r = rfirst0 + j;
    do_row(r, row_ptr[r], row_ptr[r + 1]);
  }
  __syncthreads();
  const int nl = nlong;
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = wave; i < nl; i += kWaves) {
    const int64_t r = rfirst0 + longrows[i];
    const int64_t rs = row_ptr[r], re = row_ptr[r + 1];
    const int s = (int)((rs > base ? rs : base) - base);
    const int e = (int)((re < lim ? re : lim) - base);
    double acc = 0;
    for (int q = s + lane; q < e; q += 64) acc += prod[q];
    acc = wave_reduce(acc, OpAdd());
    if (lane == 0) { if (rs >= base) y[r] = acc; else head[t] = acc; }
  }
}


// V2: V1 + non-temporal streams + ablation switches.

template <int TILE, int RPT, bool NT, bool GATHER, bool REDUCE>
__global__ __launch_bounds__(kBlock) void spmv_v2(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                  const double *__restrict__ val, const double *__restrict__ x,
                                                  double *__restrict__ y, const int64_t *__restrict__ tile_lb,
                                                  double *__restrict__ head, int64_t nnz, int64_t ntiles) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef double v2d __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(32))) double prod[TILE];
  __shared__ int longrows[TILE / kLongFrag + 2];
  __shared__ int nlong;
  const int tid = threadIdx.x;
  const int64_t t = xcd_remap(blockIdx.x, ntiles);
  const int64_t base = t * TILE;
  const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);
  const int64_t lb0 = tile_lb[t], lb1 = tile_lb[t + 1];
  constexpr int ITERS = TILE / (kBlock * 4);
  const int32_t *cbase = col + base;
  const double *vbase = val + base;
  v4i c[ITERS];
  v2d a0[ITERS], a1[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; it++) {
    const int e = (it * kBlock + tid) * 4;
    if (e + 4 <= cnt) {
      if (NT) {
        c[it] = ntload(reinterpret_cast<const v4i *>(cbase + e));
        a0[it] = ntload(reinterpret_cast<const v2d *>(vbase + e));
        a1[it] = ntload(reinterpret_cast<const v2d *>(vbase + e + 2));
      } else {
        c[it] = *reinterpret_cast<const v4i *>(cbase + e);
        a0[it] = *reinterpret_cast<const v2d *>(vbase + e);
        a1[it] = *reinterpret_cast<const v2d *>(vbase + e + 2);
      }
    } else {
      int cc[4]; double aa[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const bool ok = e + q < cnt;
        cc[q] = ok ? cbase[e + q] : 0;
        aa[q] = ok ? vbase[e + q] : 0.0;
      }
      c[it] = v4i{cc[0], cc[1], cc[2], cc[3]};
      a0[it] = v2d{aa[0], aa[1]};
      a1[it] = v2d{aa[2], aa[3]};
    }
  }
  const int64_t rfirst0 = lb0 > 0 ? lb0 - 1 : 0;
  int64_t rs_[RPT], re_[RPT];
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int64_t r = rfirst0 + tid + k * kBlock;
    const bool ok = r < lb1;
    rs_[k] = ok ? (NT ? ntload(&row_ptr[r]) : row_ptr[r]) : 0;
    re_[k] = ok ? (NT ? ntload(&row_ptr[r + 1]) : row_ptr[r + 1]) : 0;
  }
  if (tid == 0) nlong = 0;
#pragma unroll
  for (int it = 0; it < ITERS; it++) {
    const int e = (it * kBlock + tid) * 4;
    double p[4];
    const double av[4] = {a0[it].x, a0[it].y, a1[it].x, a1[it].y};
    const int cv[4] = {c[it].x, c[it].y, c[it].z, c[it].w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (GATHER) p[q] = (e + q < cnt) ? av[q] * x[cv[q]] : 0.0;
      else p[q] = av[q] * (double)cv[q];
    }
    *reinterpret_cast<v2d *>(&prod[e]) = v2d{p[0], p[1]};
    *reinterpret_cast<v2d *>(&prod[e + 2]) = v2d{p[2], p[3]};
  }
  __syncthreads();
  const int64_t lim = base + cnt;
  const int nrows = (int)(lb1 - rfirst0);
  auto do_row = [&](int64_t r, int64_t rs, int64_t re) {
    if (r < lb0 && re <= base) return;
    const int s = (int)((rs > base ? rs : base) - base);
    const int e = (int)((re < lim ? re : lim) - base);
    if (e - s > kLongFrag) {
      const int slot = atomicAdd(&nlong, 1);
      longrows[slot] = (int)(r - rfirst0);
      return;
    }
    double acc = 0;
    if (REDUCE) { for (int q = s; q < e; q++) acc += prod[q]; }
    else acc = prod[s];
    if (rs >= base) { if (NT) __builtin_nontemporal_store(acc, &y[r]); else y[r] = acc; }
    else head[t] = acc;
  };
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int64_t r = rfirst0 + tid + k * kBlock;
    if (r < lb1) do_row(r, rs_[k], re_[k]);
  }
  for (int j = tid + RPT * kBlock; j < nrows; j += kBlock) {
    const int64_t r = rfirst0 + j;
    do_row(r, row_ptr[r], row_ptr[r + 1]);
  }
  __syncthreads();
  const int nl = nlong;
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = wave; i < nl; i += kWaves) {
    const int64_t r = rfirst0 + longrows[i];
    const int64_t rs = row_ptr[r], re = row_ptr[r + 1];
    const int s = (int)((rs > base ? rs : base) - base);
    const int e = (int)((re < lim ? re : lim) - base);
    double acc = 0;
    for (int q = s + lane; q < e; q += 64) acc += prod[q];
    acc = wave_reduce(acc, OpAdd());
    if (lane == 0) { if (rs >= base) y[r] = acc; else head[t] = acc; }
  }
}


// V3: persistent blocks, next tile's col/val prefetched into registers while the
// current tile is gathered/reduced.  Blocks of one XCD (b % 8) walk that XCD's
// contiguous eighth of the tiles side by side.
template <int TILE, int RPT, bool NT>
__global__ __launch_bounds__(kBlock) void spmv_v3(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                  const double *__restrict__ val, const double *__restrict__ x,
                                                  double *__restrict__ y, const int64_t *__restrict__ tile_lb,
                                                  double *__restrict__ head, int64_t nnz, int64_t ntiles) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef double v2d __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(32))) double prod[TILE];
  __shared__ int longrows[TILE / kLongFrag + 2];
  __shared__ int nlong;
  const int tid = threadIdx.x;
  constexpr int ITERS = TILE / (kBlock * 4);
  const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int64_t q8 = ntiles >> 3, r8 = ntiles & 7;
  const int64_t t_begin = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int64_t t_end = t_begin + (xcd < r8 ? q8 + 1 : q8);

  v4i c[ITERS], c2[ITERS];
  v2d a0[ITERS], a1[ITERS], b0[ITERS], b1[ITERS];
  auto load_tile = [&](int64_t t, v4i (&cc)[ITERS], v2d (&x0)[ITERS], v2d (&x1)[ITERS]) {
    const int64_t base = t * TILE;
    const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
      const int e = (it * kBlock + tid) * 4;
      if (e + 4 <= cnt) {
        cc[it] = NT ? ntload(reinterpret_cast<const v4i *>(col + base + e)) : *reinterpret_cast<const v4i *>(col + base + e);
        x0[it] = NT ? ntload(reinterpret_cast<const v2d *>(val + base + e)) : *reinterpret_cast<const v2d *>(val + base + e);
        x1[it] = NT ? ntload(reinterpret_cast<const v2d *>(val + base + e + 2)) : *reinterpret_cast<const v2d *>(val + base + e + 2);
      } else {
        int ci[4]; double ai[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const bool ok = e + q < cnt;
          ci[q] = ok ? col[base + e + q] : 0;
          ai[q] = ok ? val[base + e + q] : 0.0;
        }
        cc[it] = v4i{ci[0], ci[1], ci[2], ci[3]};
        x0[it] = v2d{ai[0], ai[1]};
        x1[it] = v2d{ai[2], ai[3]};
      }
    }
  };
  int64_t t = t_begin + slot;
  if (t < t_end) load_tile(t, c, a0, a1);
  for (; t < t_end; t += nslots) {
    const int64_t base = t * TILE;
    const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);
    const int64_t lb0 = tile_lb[t], lb1 = tile_lb[t + 1];
    const int64_t tn = t + nslots;
    if (tn < t_end) load_tile(tn, c2, b0, b1);
    const int64_t rfirst0 = lb0 > 0 ? lb0 - 1 : 0;
    int64_t rs_[RPT], re_[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int64_t r = rfirst0 + tid + k * kBlock;
      const bool ok = r < lb1;
      rs_[k] = ok ? row_ptr[r] : 0;
      re_[k] = ok ? row_ptr[r + 1] : 0;
    }
    if (tid == 0) nlong = 0;
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
      const int e = (it * kBlock + tid) * 4;
      double p[4];
      const double av[4] = {a0[it].x, a0[it].y, a1[it].x, a1[it].y};
      const int cv[4] = {c[it].x, c[it].y, c[it].z, c[it].w};
#pragma unroll
      for (int q = 0; q < 4; q++) p[q] = (e + q < cnt) ? av[q] * x[cv[q]] : 0.0;
      *reinterpret_cast<v2d *>(&prod[e]) = v2d{p[0], p[1]};
      *reinterpret_cast<v2d *>(&prod[e + 2]) = v2d{p[2], p[3]};
    }
    __syncthreads();
    const int64_t lim = base + cnt;
    const int nrows = (int)(lb1 - rfirst0);
    auto do_row = [&](int64_t r, int64_t rs, int64_t re) {
      if (r < lb0 && re <= base) return;
      const int s = (int)((rs > base ? rs : base) - base);
      const int e = (int)((re < lim ? re : lim) - base);
      if (e - s > kLongFrag) {
        const int sl = atomicAdd(&nlong, 1);
        longrows[sl] = (int)(r - rfirst0);
        return;
      }
      double acc = 0;
      for (int q = s; q < e; q++) acc += prod[q];
      if (rs >= base) y[r] = acc; else head[t] = acc;
    };
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int64_t r = rfirst0 + tid + k * kBlock;
      if (r < lb1) do_row(r, rs_[k], re_[k]);
    }
    for (int j = tid + RPT * kBlock; j < nrows; j += kBlock) {
      const int64_t r = rfirst0 + j;
      do_row(r, row_ptr[r], row_ptr[r + 1]);
    }
    __syncthreads();
    const int nl = nlong;
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = wave; i < nl; i += kWaves) {
      const int64_t r = rfirst0 + longrows[i];
      const int64_t rs = row_ptr[r], re = row_ptr[r + 1];
      const int s = (int)((rs > base ? rs : base) - base);
      const int e = (int)((re < lim ? re : lim) - base);
      double acc = 0;
      for (int q = s + lane; q < e; q += 64) acc += prod[q];
      acc = wave_reduce(acc, OpAdd());
      if (lane == 0) { if (rs >= base) y[r] = acc; else head[t] = acc; }
    }
    if (nl) __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; it++) { c[it] = c2[it]; a0[it] = b0[it]; a1[it] = b1[it]; }
  }
}


// V4: stream col/val into LDS (no x access while streaming), then gather x in ROW-major
// order: at step k every lane touches entry k of its own row, so for banded/stencil
// matrices the lanes of one gather instruction read consecutive x addresses (few L1 tag
// look-ups per instruction instead of one per lane).  L lanes per row, chosen per tile.
template <int TILE, bool NT>
__global__ __launch_bounds__(kBlock) void spmv_v4(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                  const double *__restrict__ val, const double *__restrict__ x,
                                                  double *__restrict__ y, const int64_t *__restrict__ tile_lb,
                                                  double *__restrict__ head, int64_t nnz, int64_t ntiles) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef double v2d __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(32))) double sval[TILE];
  __shared__ __attribute__((aligned(16))) int scol[TILE];
  __shared__ int longrows[TILE / 16 + 2];
  __shared__ int nlong;
  const int tid = threadIdx.x;
  const int64_t t = xcd_remap(blockIdx.x, ntiles);
  const int64_t base = t * TILE;
  const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);
  const int64_t lb0 = tile_lb[t], lb1 = tile_lb[t + 1];
  constexpr int ITERS = TILE / (kBlock * 4);
  const int32_t *cbase = col + base;
  const double *vbase = val + base;
#pragma unroll
  for (int it = 0; it < ITERS; it++) {
    const int e = (it * kBlock + tid) * 4;
    if (e + 4 <= cnt) {
      v4i c = NT ? ntload(reinterpret_cast<const v4i *>(cbase + e)) : *reinterpret_cast<const v4i *>(cbase + e);
      v2d a0 = NT ? ntload(reinterpret_cast<const v2d *>(vbase + e)) : *reinterpret_cast<const v2d *>(vbase + e);
      v2d a1 = NT ? ntload(reinterpret_cast<const v2d *>(vbase + e + 2)) : *reinterpret_cast<const v2d *>(vbase + e + 2);
      *reinterpret_cast<v4i *>(&scol[e]) = c;
      *reinterpret_cast<v2d *>(&sval[e]) = a0;
      *reinterpret_cast<v2d *>(&sval[e + 2]) = a1;
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const bool ok = e + q < cnt;
        scol[e + q] = ok ? cbase[e + q] : 0;
        sval[e + q] = ok ? vbase[e + q] : 0.0;
      }
    }
  }
  const int64_t rfirst0 = lb0 > 0 ? lb0 - 1 : 0;
  const int nrows = (int)(lb1 - rfirst0);
  // lanes per row from the tile's average fragment length (wave-uniform)
  int L = 1;
  {
    const int avg4 = cnt / (nrows > 0 ? nrows : 1) / 4;
    while (L < 64 && L * 2 <= avg4) L *= 2;
  }
  if (tid == 0) nlong = 0;
  // first row of this lane group: metadata prefetched before the barrier
  const int grp = tid / L, sub = tid & (L - 1), ngrp = kBlock / L;
  int64_t rs0 = 0, re0 = 0;
  if (grp < nrows) { rs0 = row_ptr[rfirst0 + grp]; re0 = row_ptr[rfirst0 + grp + 1]; }
  __syncthreads();
  const int64_t lim = base + cnt;
  for (int j = grp; j < nrows; j += ngrp) {
    const int64_t r = rfirst0 + j;
    int64_t rs, re;
    if (j == grp) { rs = rs0; re = re0; } else { rs = row_ptr[r]; re = row_ptr[r + 1]; }
    if (r < lb0 && re <= base) continue;
    const int s = (int)((rs > base ? rs : base) - base);
    const int e = (int)((re < lim ? re : lim) - base);
    if (e - s > 16 * L) {
      if (sub == 0) { const int sl = atomicAdd(&nlong, 1); longrows[sl] = j; }
      continue;
    }
    double acc = 0;
    for (int q = s + sub; q < e; q += 4 * L) {
      double xv[4], av[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int qq = q + u * L;
        const bool ok = qq < e;
        const int cc = scol[ok ? qq : s];
        av[u] = ok ? sval[qq] : 0.0;
        xv[u] = ok ? x[cc] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) acc += av[u] * xv[u];
    }
    for (int d = L >> 1; d >= 1; d >>= 1) acc += wshfl_xor(acc, d);
    if (sub == 0) { if (rs >= base) y[r] = acc; else head[t] = acc; }
  }
  __syncthreads();
  const int nl = nlong;
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = wave; i < nl; i += kWaves) {
    const int64_t r = rfirst0 + longrows[i];
    const int64_t rs = row_ptr[r], re = row_ptr[r + 1];
    const int s = (int)((rs > base ? rs : base) - base);
    const int e = (int)((re < lim ? re : lim) - base);
    double acc = 0;
    for (int q = s + lane; q < e; q += 256) {
      double xv[4], av[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int qq = q + u * 64;
        const bool ok = qq < e;
        const int cc = scol[ok ? qq : s];
        av[u] = ok ? sval[qq] : 0.0;
        xv[u] = ok ? x[cc] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) acc += av[u] * xv[u];
    }
    acc = wave_reduce(acc, OpAdd());
    if (lane == 0) { if (rs >= base) y[r] = acc; else head[t] = acc; }
  }
}


// V5: V4 made persistent over a CONTIGUOUS run of tiles per block, next tile's stream
// prefetched into registers during the row phase.  U = gather unroll.
template <int TILE, int U, bool PERSIST>
__global__ __launch_bounds__(kBlock) void spmv_v5(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                  const double *__restrict__ val, const double *__restrict__ x,
                                                  double *__restrict__ y, const int64_t *__restrict__ tile_lb,
                                                  double *__restrict__ head, int64_t nnz, int64_t ntiles) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef double v2d __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(32))) double sval[TILE];
  __shared__ __attribute__((aligned(16))) int scol[TILE];
  __shared__ int longrows[TILE / 16 + 2];
  __shared__ int nlong;
  const int tid = threadIdx.x;
  constexpr int ITERS = TILE / (kBlock * 4);
  int64_t t0, t1;
  if (PERSIST) {
    const int64_t nb = gridDim.x;
    const int64_t chunk = xcd_remap(blockIdx.x, nb);     // contiguous chunks, XCD-contiguous
    const int64_t per = (ntiles + nb - 1) / nb;
    t0 = chunk * per;
    t1 = t0 + per < ntiles ? t0 + per : ntiles;
  } else {
    t0 = xcd_remap(blockIdx.x, ntiles);
    t1 = t0 + 1;
  }
  v4i c[ITERS];
  v2d a0[ITERS], a1[ITERS];
  auto load_tile = [&](int64_t t) {
    const int64_t base = t * TILE;
    const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
      const int e = (it * kBlock + tid) * 4;
      if (e + 4 <= cnt) {
        c[it] = *reinterpret_cast<const v4i *>(col + base + e);
        a0[it] = *reinterpret_cast<const v2d *>(val + base + e);
        a1[it] = *reinterpret_cast<const v2d *>(val + base + e + 2);
      } else {
        int ci[4]; double ai[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const bool ok = e + q < cnt;
          ci[q] = ok ? col[base + e + q] : 0;
          ai[q] = ok ? val[base + e + q] : 0.0;
        }
        c[it] = v4i{ci[0], ci[1], ci[2], ci[3]};
        a0[it] = v2d{ai[0], ai[1]};
        a1[it] = v2d{ai[2], ai[3]};
      }
    }
  };
  if (t0 < t1) load_tile(t0);
  for (int64_t t = t0; t < t1; t++) {
    const int64_t base = t * TILE;
    const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);
    const int64_t lb0 = tile_lb[t], lb1 = tile_lb[t + 1];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
      const int e = (it * kBlock + tid) * 4;
      *reinterpret_cast<v4i *>(&scol[e]) = c[it];
      *reinterpret_cast<v2d *>(&sval[e]) = a0[it];
      *reinterpret_cast<v2d *>(&sval[e + 2]) = a1[it];
    }
    if (PERSIST && t + 1 < t1) load_tile(t + 1);
    const int64_t rfirst0 = lb0 > 0 ? lb0 - 1 : 0;
    const int nrows = (int)(lb1 - rfirst0);
    int L = 1;
    {
      const int avg4 = cnt / (nrows > 0 ? nrows : 1) / 4;
      while (L < 64 && L * 2 <= avg4) L *= 2;
    }
    if (tid == 0) nlong = 0;
    const int grp = tid / L, sub = tid & (L - 1), ngrp = kBlock / L;
    int64_t rs0 = 0, re0 = 0;
    if (grp < nrows) { rs0 = row_ptr[rfirst0 + grp]; re0 = row_ptr[rfirst0 + grp + 1]; }
    __syncthreads();
    const int64_t lim = base + cnt;
    for (int j = grp; j < nrows; j += ngrp) {
      const int64_t r = rfirst0 + j;
      int64_t rs, re;
      if (j == grp) { rs = rs0; re = re0; } else { rs = row_ptr[r]; re = row_ptr[r + 1]; }
      if (r < lb0 && re <= base) continue;
      const int s = (int)((rs > base ? rs : base) - base);
      const int e = (int)((re < lim ? re : lim) - base);
      if (e - s > 16 * L) {
        if (sub == 0) { const int sl = atomicAdd(&nlong, 1); longrows[sl] = j; }
        continue;
      }
      double acc = 0;
      for (int q = s + sub; q < e; q += U * L) {
        double xv[U], av[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int qq = q + u * L;
          const bool ok = qq < e;
          const int cc = scol[ok ? qq : s];
          av[u] = ok ? sval[qq] : 0.0;
          xv[u] = ok ? x[cc] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += av[u] * xv[u];
      }
      for (int d = L >> 1; d >= 1; d >>= 1) acc += wshfl_xor(acc, d);
      if (sub == 0) { if (rs >= base) y[r] = acc; else head[t] = acc; }
    }
    __syncthreads();
    const int nl = nlong;
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = wave; i < nl; i += kWaves) {
      const int64_t r = rfirst0 + longrows[i];
      const int64_t rs = row_ptr[r], re = row_ptr[r + 1];
      const int s = (int)((rs > base ? rs : base) - base);
      const int e = (int)((re < lim ? re : lim) - base);
      double acc = 0;
      for (int q = s + lane; q < e; q += 256) {
        double xv[4], av[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int qq = q + u * 64;
          const bool ok = qq < e;
          const int cc = scol[ok ? qq : s];
          av[u] = ok ? sval[qq] : 0.0;
          xv[u] = ok ? x[cc] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc += av[u] * xv[u];
      }
      acc = wave_reduce(acc, OpAdd());
      if (lane == 0) { if (rs >= base) y[r] = acc; else head[t] = acc; }
    }
    if (PERSIST) __syncthreads();
  }
}


// V6: V4/V5 (non-persistent) with block size as a parameter and ablation switches.
template <int TILE, int THREADS, bool WRITE_Y, bool READ_ROWPTR, bool NTSTORE>
__global__ __launch_bounds__(THREADS) void spmv_v6(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                   const double *__restrict__ val, const double *__restrict__ x,
                                                   double *__restrict__ y, const int64_t *__restrict__ tile_lb,
                                                   double *__restrict__ head, int64_t nnz, int64_t ntiles, int64_t g) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef double v2d __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(32))) double sval[TILE];
  __shared__ __attribute__((aligned(16))) int scol[TILE];
  __shared__ int longrows[TILE / 16 + 2];
  __shared__ int nlong;
  const int tid = threadIdx.x;
  constexpr int ITERS = TILE / (THREADS * 4);
  constexpr int NW = THREADS / 64;
  const int64_t t = xcd_remap(blockIdx.x, ntiles);
  const int64_t base = t * TILE;
  const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);
  const int64_t lb0 = tile_lb[t], lb1 = tile_lb[t + 1];
#pragma unroll
  for (int it = 0; it < ITERS; it++) {
    const int e = (it * THREADS + tid) * 4;
    if (e + 4 <= cnt) {
      v4i c = *reinterpret_cast<const v4i *>(col + base + e);
      v2d a0 = *reinterpret_cast<const v2d *>(val + base + e);
      v2d a1 = *reinterpret_cast<const v2d *>(val + base + e + 2);
      *reinterpret_cast<v4i *>(&scol[e]) = c;
      *reinterpret_cast<v2d *>(&sval[e]) = a0;
      *reinterpret_cast<v2d *>(&sval[e + 2]) = a1;
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const bool ok = e + q < cnt;
        scol[e + q] = ok ? col[base + e + q] : 0;
        sval[e + q] = ok ? val[base + e + q] : 0.0;
      }
    }
  }
  const int64_t rfirst0 = lb0 > 0 ? lb0 - 1 : 0;
  const int nrows = (int)(lb1 - rfirst0);
  int L = 1;
  {
    const int avg4 = cnt / (nrows > 0 ? nrows : 1) / 4;
    while (L < 64 && L * 2 <= avg4) L *= 2;
  }
  if (tid == 0) nlong = 0;
  const int grp = tid / L, sub = tid & (L - 1), ngrp = THREADS / L;
  int64_t rs0 = 0, re0 = 0;
  if (grp < nrows) {
    if (READ_ROWPTR) { rs0 = row_ptr[rfirst0 + grp]; re0 = row_ptr[rfirst0 + grp + 1]; }
    else { rs0 = lap_row_start(rfirst0 + grp, g); re0 = lap_row_start(rfirst0 + grp + 1, g); }
  }
  __syncthreads();
  const int64_t lim = base + cnt;
  for (int j = grp; j < nrows; j += ngrp) {
    const int64_t r = rfirst0 + j;
    int64_t rs, re;
    if (j == grp) { rs = rs0; re = re0; }
    else if (READ_ROWPTR) { rs = row_ptr[r]; re = row_ptr[r + 1]; }
    else { rs = lap_row_start(r, g); re = lap_row_start(r + 1, g); }
    if (r < lb0 && re <= base) continue;
    const int s = (int)((rs > base ? rs : base) - base);
    const int e = (int)((re < lim ? re : lim) - base);
    if (e - s > 16 * L) {
      if (sub == 0) { const int sl = atomicAdd(&nlong, 1); longrows[sl] = j; }
      continue;
    }
    double acc = 0;
    for (int q = s + sub; q < e; q += 8 * L) {
      double xv[8], av[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int qq = q + u * L;
        const bool ok = qq < e;
        const int cc = scol[ok ? qq : s];
        av[u] = ok ? sval[qq] : 0.0;
        xv[u] = ok ? x[cc] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) acc += av[u] * xv[u];
    }
    for (int d = L >> 1; d >= 1; d >>= 1) acc += wshfl_xor(acc, d);
    if (sub == 0) {
      if (rs >= base) {
        if (WRITE_Y) { if (NTSTORE) __builtin_nontemporal_store(acc, &y[r]); else y[r] = acc; }
        else if (acc == 1.2345e-300) y[r] = acc;
      } else head[t] = acc;
    }
  }
  __syncthreads();
  const int nl = nlong;
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = wave; i < nl; i += NW) {
    const int64_t r = rfirst0 + longrows[i];
    const int64_t rs = row_ptr[r], re = row_ptr[r + 1];
    const int s = (int)((rs > base ? rs : base) - base);
    const int e = (int)((re < lim ? re : lim) - base);
    double acc = 0;
    for (int q = s + lane; q < e; q += 64) acc += sval[q] * x[scol[q]];
    acc = wave_reduce(acc, OpAdd());
    if (lane == 0) { if (rs >= base) y[r] = acc; else head[t] = acc; }
  }
}


// V7: V6 with line-dense stream loads (every wave instruction covers one contiguous KiB)
// so that non-temporal loads never touch a line twice.
template <int TILE, bool NT_STREAM, bool NT_ROWPTR, bool NT_STORE>
__global__ __launch_bounds__(kBlock) void spmv_v7(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                  const double *__restrict__ val, const double *__restrict__ x,
                                                  double *__restrict__ y, const int64_t *__restrict__ tile_lb,
                                                  double *__restrict__ head, int64_t nnz, int64_t ntiles) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef double v2d __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(32))) double sval[TILE];
  __shared__ __attribute__((aligned(16))) int scol[TILE];
  __shared__ int longrows[TILE / 16 + 2];
  __shared__ int nlong;
  const int tid = threadIdx.x;
  const int64_t t = xcd_remap(blockIdx.x, ntiles);
  const int64_t base = t * TILE;
  const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);
  const int64_t lb0 = tile_lb[t], lb1 = tile_lb[t + 1];
  constexpr int CI = TILE / (kBlock * 4), VI = TILE / (kBlock * 2);
  if (cnt == TILE) {
    v4i c[CI];
    v2d a[VI];
#pragma unroll
    for (int it = 0; it < CI; it++) {
      const v4i *p = reinterpret_cast<const v4i *>(col + base) + it * kBlock + tid;
      c[it] = NT_STREAM ? ntload(p) : *p;
    }
#pragma unroll
    for (int it = 0; it < VI; it++) {
      const v2d *p = reinterpret_cast<const v2d *>(val + base) + it * kBlock + tid;
      a[it] = NT_STREAM ? ntload(p) : *p;
    }
#pragma unroll
    for (int it = 0; it < CI; it++) reinterpret_cast<v4i *>(scol)[it * kBlock + tid] = c[it];
#pragma unroll
    for (int it = 0; it < VI; it++) reinterpret_cast<v2d *>(sval)[it * kBlock + tid] = a[it];
  } else {
    for (int e = tid; e < cnt; e += kBlock) { scol[e] = col[base + e]; sval[e] = val[base + e]; }
  }
  const int64_t rfirst0 = lb0 > 0 ? lb0 - 1 : 0;
  const int nrows = (int)(lb1 - rfirst0);
  int L = 1;
  {
    const int avg4 = cnt / (nrows > 0 ? nrows : 1) / 4;
    while (L < 64 && L * 2 <= avg4) L *= 2;
  }
  if (tid == 0) nlong = 0;
  const int grp = tid / L, sub = tid & (L - 1), ngrp = kBlock / L;
  int64_t rs0 = 0, re0 = 0;
  if (grp < nrows) {
    rs0 = NT_ROWPTR ? ntload(&row_ptr[rfirst0 + grp]) : row_ptr[rfirst0 + grp];
    re0 = NT_ROWPTR ? ntload(&row_ptr[rfirst0 + grp + 1]) : row_ptr[rfirst0 + grp + 1];
  }
  __syncthreads();
  const int64_t lim = base + cnt;
  for (int j = grp; j < nrows; j += ngrp) {
    const int64_t r = rfirst0 + j;
    int64_t rs, re;
    if (j == grp) { rs = rs0; re = re0; } else { rs = row_ptr[r]; re = row_ptr[r + 1]; }
    if (r < lb0 && re <= base) continue;
    const int s = (int)((rs > base ? rs : base) - base);
    const int e = (int)((re < lim ? re : lim) - base);
    if (e - s > 16 * L) {
      if (sub == 0) { const int sl = atomicAdd(&nlong, 1); longrows[sl] = j; }
      continue;
    }
    double acc = 0;
    for (int q = s + sub; q < e; q += 8 * L) {
      double xv[8], av[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int qq = q + u * L;
        const bool ok = qq < e;
        const int cc = scol[ok ? qq : s];
        av[u] = ok ? sval[qq] : 0.0;
        xv[u] = ok ? x[cc] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) acc += av[u] * xv[u];
    }
    for (int d = L >> 1; d >= 1; d >>= 1) acc += wshfl_xor(acc, d);
    if (sub == 0) {
      if (rs >= base) { if (NT_STORE) __builtin_nontemporal_store(acc, &y[r]); else y[r] = acc; }
      else head[t] = acc;
    }
  }
  __syncthreads();
  const int nl = nlong;
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = wave; i < nl; i += kWaves) {
    const int64_t r = rfirst0 + longrows[i];
    const int64_t rs = row_ptr[r], re = row_ptr[r + 1];
    const int s = (int)((rs > base ? rs : base) - base);
    const int e = (int)((re < lim ? re : lim) - base);
    double acc = 0;
    for (int q = s + lane; q < e; q += 64) acc += sval[q] * x[scol[q]];
    acc = wave_reduce(acc, OpAdd());
    if (lane == 0) { if (rs >= base) y[r] = acc; else head[t] = acc; }
  }
}


// ---- compact-index variant: 16-bit tile-relative column offsets + 16-bit row starts within the tile
__global__ __launch_bounds__(kBlock) void build_col16(const int32_t *__restrict__ col, int64_t nnz, int64_t ntiles,
                                                      uint16_t *__restrict__ col16, int32_t *__restrict__ c0s,
                                                      int *__restrict__ fail) {
  __shared__ int smin[kWaves], smax[kWaves];
  const int64_t t = blockIdx.x, base = t * 1024;
  int mn = 0x7fffffff, mx = -1;
  int c[4];
  for (int k = 0; k < 4; k++) {
    const int64_t e = base + k * kBlock + threadIdx.x;
    c[k] = e < nnz ? col[e] : -1;
    if (c[k] >= 0) { mn = c[k] < mn ? c[k] : mn; mx = c[k] > mx ? c[k] : mx; }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    const int a = __shfl_xor(mn, d, 64), b = __shfl_xor(mx, d, 64);
    mn = a < mn ? a : mn; mx = b > mx ? b : mx;
  }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
  __syncthreads();
  mn = smin[0]; mx = smax[0];
  for (int w = 1; w < kWaves; w++) { mn = smin[w] < mn ? smin[w] : mn; mx = smax[w] > mx ? smax[w] : mx; }
  if (threadIdx.x == 0) { c0s[t] = mn; if (mx - mn > 65535) *fail = 1; }
  for (int k = 0; k < 4; k++) {
    const int64_t e = base + k * kBlock + threadIdx.x;
    if (e < nnz) col16[e] = (uint16_t)(c[k] - mn);
  }
}

__global__ void build_r16(const int64_t *__restrict__ row_ptr, int64_t n, uint16_t *__restrict__ r16) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r < n) r16[r] = (uint16_t)(row_ptr[r] & 1023);
}

template <bool NTX, int U, int WPE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void spmv_c16(const int64_t *__restrict__ row_ptr, const uint16_t *__restrict__ r16,
                                                   const uint16_t *__restrict__ col16, const int32_t *__restrict__ c0s,
                                                   const double *__restrict__ val, const double *__restrict__ x,
                                                   double *__restrict__ y, const int64_t *__restrict__ tile_lb,
                                                   double *__restrict__ head, int64_t nnz, int64_t ntiles) {
  constexpr int TILE = 1024;
  typedef double T;
  __shared__ __attribute__((aligned(16))) T sval[TILE];
  __shared__ __attribute__((aligned(16))) uint16_t scol[TILE];
  __shared__ int longrows[TILE / 16 + 2];
  __shared__ int nlong;
  const int tid = threadIdx.x;
  const int64_t t = xcd_remap(blockIdx.x, ntiles);
  const int64_t base = t * TILE;
  const int cnt = (int)((nnz - base) < (int64_t)TILE ? (nnz - base) : (int64_t)TILE);
  const int64_t lb0 = tile_lb[t], lb1 = tile_lb[t + 1];
  const int c0 = c0s[t];
  const int64_t S1 = row_ptr[lb1];
  if (cnt == TILE) {
    raw16 a[2], c;
    const raw16 *cp = reinterpret_cast<const raw16 *>(col16 + base);
    const raw16 *vp = reinterpret_cast<const raw16 *>(val + base);
    if (tid < 128) c = ntload(cp + tid);
    a[0] = ntload(vp + tid);
    a[1] = ntload(vp + kBlock + tid);
    if (tid < 128) reinterpret_cast<raw16 *>(scol)[tid] = c;
    reinterpret_cast<raw16 *>(sval)[tid] = a[0];
    reinterpret_cast<raw16 *>(sval)[kBlock + tid] = a[1];
  } else {
    for (int e = tid; e < cnt; e += kBlock) { scol[e] = col16[base + e]; sval[e] = val[base + e]; }
  }
  const int64_t rfirst = lb0 > 0 ? lb0 - 1 : 0;
  const int nrows = (int)(lb1 - rfirst);
  int L = 1;
  {
    const int avg4 = cnt / (nrows > 0 ? nrows : 1) / 4;
    while (L < 64 && L * 2 <= avg4) L *= 2;
  }
  if (tid == 0) nlong = 0;
  const int grp = tid / L, sub = tid & (L - 1), ngrp = kBlock / L;
  auto startf = [&](int64_t r) -> int64_t { return r < lb1 ? base + (int64_t)ntload(&r16[r]) : S1; };
  int64_t rs0 = 0, re0 = 0;
  if (grp < nrows) {
    const int64_t r = rfirst + grp;
    rs0 = r < lb0 ? base - 1 : startf(r);
    re0 = startf(r + 1);
  }
  __syncthreads();
  const int64_t lim = base + cnt;
  for (int j = grp; j < nrows; j += ngrp) {
    const int64_t r = rfirst + j;
    int64_t rs, re;
    if (j == grp) { rs = rs0; re = re0; }
    else { rs = r < lb0 ? base - 1 : startf(r); re = startf(r + 1); }
    if (r < lb0 && re <= base) continue;
    const int s = (int)((rs > base ? rs : base) - base);
    const int e = (int)((re < lim ? re : lim) - base);
    if (e - s > 16 * L) {
      if (sub == 0) { const int slot = atomicAdd(&nlong, 1); longrows[slot] = j; }
      continue;
    }
    T acc = (T)0;
    for (int q = s + sub; q < e; q += U * L) {
      T xv[U], av[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int qq = q + u * L;
        const bool ok = qq < e;
        const int cc = c0 + (int)scol[ok ? qq : s];
        av[u] = ok ? sval[qq] : (T)0;
        xv[u] = ok ? (NTX ? x[cc] : x[cc]) : (T)0;
      }
#pragma unroll
      for (int u = 0; u < U; u++) acc = (T)(acc + (T)(av[u] * xv[u]));
    }
    for (int d = L >> 1; d >= 1; d >>= 1) acc = (T)(acc + wshfl_xor(acc, d));
    if (sub == 0) {
      if (rs >= base) __builtin_nontemporal_store(acc, &y[r]);
      else head[t] = acc;
    }
  }
  __syncthreads();
  const int nl = nlong;
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = wave; i < nl; i += kWaves) {
    const int64_t r = rfirst + longrows[i];
    const int64_t rs = r < lb0 ? base - 1 : startf(r), re = startf(r + 1);
    const int s = (int)((rs > base ? rs : base) - base);
    const int e = (int)((re < lim ? re : lim) - base);
    T acc = (T)0;
    for (int q = s + lane; q < e; q += 64) acc = (T)(acc + (T)(sval[q] * x[c0 + (int)scol[q]]));
    acc = wave_reduce(acc, OpAdd());
    if (lane == 0) {
      if (rs >= base) y[r] = acc;
      else head[t] = acc;
    }
  }
}

struct Variant {
  const char *name;
  std::function<void()> run;
  bool check;
  std::vector<float> ms;
};

int main(int argc, char **argv) {
  const int64_t g = argc > 1 ? atoll(argv[1]) : 4096;
  const int rounds = argc > 2 ? atoi(argv[2]) : 7;
  const int launches = argc > 3 ? atoi(argv[3]) : 20;
  const int64_t n = g * g, nnz = 5 * n - 4 * g;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  int64_t *row_ptr;
  int32_t *col;
  double *val, *x, *y, *yref;
  CK(hipMalloc(&row_ptr, 8 * (n + 1)));
  CK(hipMalloc(&col, 4 * nnz));
  CK(hipMalloc(&val, 8 * nnz));
  CK(hipMalloc(&x, 8 * n));
  CK(hipMalloc(&y, 8 * n));
  CK(hipMalloc(&yref, 8 * n));
  hipLaunchKernelGGL(gen_laplacian, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, st, g, row_ptr, col, val, x);
  CK(hipGetLastError());

  auto make_plan = [&](int64_t tile, int64_t **lb, double **head) {
    const int64_t nt = (nnz + tile - 1) / tile;
    CK(hipMalloc(lb, 8 * (nt + 3)));
    CK(hipMemsetAsync(*lb, 0, 8 * (nt + 3), st));
    CK(hipMalloc(head, 8 * nt));
    hipLaunchKernelGGL(plan_kernel, dim3((unsigned)((nt + 1 + 255) / 256)), dim3(256), 0, st, (const int64_t *)row_ptr,
                       n, tile, nt, *lb);
    CK(hipGetLastError());
    return nt;
  };
  int64_t *lb2k, *lb4k, *lb8k;
  double *hd2k, *hd4k, *hd8k;
  const int64_t nt2k = make_plan(2048, &lb2k, &hd2k);
  const int64_t nt4k = make_plan(4096, &lb4k, &hd4k);
  const int64_t nt8k = make_plan(8192, &lb8k, &hd8k);
  int64_t *lb1k; double *hd1k; int64_t *lb512; double *hd512;
  const int64_t nt512 = make_plan(512, &lb512, &hd512);
  const int64_t nt1k = make_plan(1024, &lb1k, &hd1k);
  (void)nt8k; (void)lb8k; (void)hd8k;

  std::vector<Variant> vs;
  vs.push_back({"shipped stream (spmv.hpp) 1024", [&] {
                  hipLaunchKernelGGL((spmv_stream_kernel<double, int32_t, 1024, false, 8>), dim3((unsigned)nt1k), dim3(256), 0, st,
                                     (const int64_t *)row_ptr, (const int32_t *)col, (const double *)val,
                                     (const double *)x, y, (const int64_t *)lb1k, hd1k, nnz, nt1k, 0);
                  hipLaunchKernelGGL((spmv_fixup_kernel<double, 1024>), dim3((unsigned)((nt1k - 1 + 255) / 256)),
                                     dim3(256), 0, st, (const int64_t *)row_ptr, (const int64_t *)lb1k,
                                     (const double *)hd1k, y, nt1k, false);
                }, true, {}});
  // compact indices
  uint16_t *col16, *r16;
  int32_t *c0s;
  int *cfail;
  CK(hipMalloc(&col16, 2 * nnz));
  CK(hipMalloc(&r16, 2 * (n + 1)));
  CK(hipMalloc(&c0s, 4 * nt1k));
  CK(hipMalloc(&cfail, 4));
  CK(hipMemsetAsync(cfail, 0, 4, st));
  hipLaunchKernelGGL(build_col16, dim3((unsigned)nt1k), dim3(256), 0, st, (const int32_t *)col, nnz, nt1k, col16, c0s, cfail);
  hipLaunchKernelGGL(build_r16, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, st, (const int64_t *)row_ptr, n + 1, r16);
  {
    int hf;
    CK(hipMemcpy(&hf, cfail, 4, hipMemcpyDeviceToHost));
    printf("compact build: %s\n", hf ? "SPAN TOO WIDE" : "ok");
  }
#define C16(U, WPE)                                                                                                 \
  vs.push_back({"c16 compact U=" #U " wpe=" #WPE, [&] {                                                              \
                  hipLaunchKernelGGL((spmv_c16<false, U, WPE>), dim3((unsigned)nt1k), dim3(256), 0, st,              \
                                     (const int64_t *)row_ptr, (const uint16_t *)r16, (const uint16_t *)col16,       \
                                     (const int32_t *)c0s, (const double *)val, (const double *)x, y,                \
                                     (const int64_t *)lb1k, hd1k, nnz, nt1k);                                        \
                  hipLaunchKernelGGL((spmv_fixup_kernel<double, 1024>), dim3((unsigned)((nt1k - 1 + 255) / 256)),    \
                                     dim3(256), 0, st, (const int64_t *)row_ptr, (const int64_t *)lb1k,              \
                                     (const double *)hd1k, y, nt1k, false);                                          \
                }, true, {}})
  C16(8, 1);
  C16(8, 8);
  C16(6, 1);
  C16(6, 8);
  C16(4, 1);
  C16(5, 8);
  vs.push_back({"shipped stream U=6", [&] {
                  hipLaunchKernelGGL((spmv_stream_kernel<double, int32_t, 1024, false, 6>), dim3((unsigned)nt1k), dim3(256), 0, st,
                                     (const int64_t *)row_ptr, (const int32_t *)col, (const double *)val,
                                     (const double *)x, y, (const int64_t *)lb1k, hd1k, nnz, nt1k, 0);
                  hipLaunchKernelGGL((spmv_fixup_kernel<double, 1024>), dim3((unsigned)((nt1k - 1 + 255) / 256)),
                                     dim3(256), 0, st, (const int64_t *)row_ptr, (const int64_t *)lb1k,
                                     (const double *)hd1k, y, nt1k, false);
                }, true, {}});
  vs.push_back({"shipped stream U=5", [&] {
                  hipLaunchKernelGGL((spmv_stream_kernel<double, int32_t, 1024, false, 5>), dim3((unsigned)nt1k), dim3(256), 0, st,
                                     (const int64_t *)row_ptr, (const int32_t *)col, (const double *)val,
                                     (const double *)x, y, (const int64_t *)lb1k, hd1k, nnz, nt1k, 0);
                  hipLaunchKernelGGL((spmv_fixup_kernel<double, 1024>), dim3((unsigned)((nt1k - 1 + 255) / 256)),
                                     dim3(256), 0, st, (const int64_t *)row_ptr, (const int64_t *)lb1k,
                                     (const double *)hd1k, y, nt1k, false);
                }, true, {}});
  vs.push_back({"shipped, no fixup launch", [&] {
                  hipLaunchKernelGGL((spmv_stream_kernel<double, int32_t, 1024, false, 8>), dim3((unsigned)nt1k), dim3(256), 0, st,
                                     (const int64_t *)row_ptr, (const int32_t *)col, (const double *)val,
                                     (const double *)x, y, (const int64_t *)lb1k, hd1k, nnz, nt1k, 0);
                }, false, {}});
  vs.push_back({"stream ceiling (same bytes)", [&] {
                  hipLaunchKernelGGL(stream_ceiling, dim3(2048), dim3(256), 0, st, (const int4 *)col,
                                     (const double2 *)val, (const double2 *)row_ptr, (const double2 *)x, (double2 *)y,
                                     nnz, n);
                }, false, {}});
#define V1(TILE, RPT, LB, HD, NT)                                                                                   \
  vs.push_back({"v1 prefetch rows TILE=" #TILE " RPT=" #RPT, [&] {                                                  \
                  hipLaunchKernelGGL((spmv_v1<TILE, RPT>), dim3((unsigned)NT), dim3(256), 0, st,                     \
                                     (const int64_t *)row_ptr, (const int32_t *)col, (const double *)val,            \
                                     (const double *)x, y, (const int64_t *)LB, HD, nnz, NT);                        \
                  hipLaunchKernelGGL((spmv_fixup_kernel<double, TILE>), dim3((unsigned)((NT - 1 + 255) / 256)),      \
                                     dim3(256), 0, st, (const int64_t *)row_ptr, (const int64_t *)LB,                \
                                     (const double *)HD, y, NT, false);                                                     \
                }, true, {}})
  V1(2048, 2, lb2k, hd2k, nt2k);
  V1(4096, 4, lb4k, hd4k, nt4k);
  V1(4096, 2, lb4k, hd4k, nt4k);


#define V2(TILE, RPT, NT, GA, RE, LB, HD, NT_, CHECK)                                                               \
  vs.push_back({"v2 TILE=" #TILE " RPT=" #RPT " nt=" #NT " gather=" #GA " reduce=" #RE, [&] {                       \
                  hipLaunchKernelGGL((spmv_v2<TILE, RPT, NT, GA, RE>), dim3((unsigned)NT_), dim3(256), 0, st,        \
                                     (const int64_t *)row_ptr, (const int32_t *)col, (const double *)val,            \
                                     (const double *)x, y, (const int64_t *)LB, HD, nnz, NT_);                       \
                  hipLaunchKernelGGL((spmv_fixup_kernel<double, TILE>), dim3((unsigned)((NT_ - 1 + 255) / 256)),     \
                                     dim3(256), 0, st, (const int64_t *)row_ptr, (const int64_t *)LB,                \
                                     (const double *)HD, y, NT_, false);                                                    \
                }, CHECK, {}})
  V2(2048, 2, false, true, true, lb2k, hd2k, nt2k, true);
  V2(2048, 2, true, true, true, lb2k, hd2k, nt2k, true);
  V2(2048, 2, true, false, true, lb2k, hd2k, nt2k, false);
  V2(2048, 2, true, true, false, lb2k, hd2k, nt2k, false);
  V2(2048, 2, true, false, false, lb2k, hd2k, nt2k, false);
  V2(4096, 4, true, true, true, lb4k, hd4k, nt4k, true);
  V2(1024, 1, true, true, true, lb1k, hd1k, nt1k, true);


#define V3(TILE, RPT, NT, GRID, LB, HD, NT_)                                                                        \
  vs.push_back({"v3 persistent TILE=" #TILE " RPT=" #RPT " nt=" #NT " grid=" #GRID, [&] {                           \
                  hipLaunchKernelGGL((spmv_v3<TILE, RPT, NT>), dim3(GRID), dim3(256), 0, st,                         \
                                     (const int64_t *)row_ptr, (const int32_t *)col, (const double *)val,            \
                                     (const double *)x, y, (const int64_t *)LB, HD, nnz, NT_);                       \
                  hipLaunchKernelGGL((spmv_fixup_kernel<double, TILE>), dim3((unsigned)((NT_ - 1 + 255) / 256)),     \
                                     dim3(256), 0, st, (const int64_t *)row_ptr, (const int64_t *)LB,                \
                                     (const double *)HD, y, NT_, false);                                                    \
                }, true, {}})
  V3(2048, 2, false, 2048, lb2k, hd2k, nt2k);
  V3(2048, 2, true, 2048, lb2k, hd2k, nt2k);
  V3(1024, 1, false, 2048, lb1k, hd1k, nt1k);
  V3(1024, 1, true, 2048, lb1k, hd1k, nt1k);
  V3(1024, 1, false, 4096, lb1k, hd1k, nt1k);
  V3(2048, 2, false, 1024, lb2k, hd2k, nt2k);
  V3(4096, 4, false, 1280, lb4k, hd4k, nt4k);


#define V4(TILE, NT, LB, HD, NT_)                                                                                   \
  vs.push_back({"v4 row-gather TILE=" #TILE " nt=" #NT, [&] {                                                       \
                  hipLaunchKernelGGL((spmv_v4<TILE, NT>), dim3((unsigned)NT_), dim3(256), 0, st,                     \
                                     (const int64_t *)row_ptr, (const int32_t *)col, (const double *)val,            \
                                     (const double *)x, y, (const int64_t *)LB, HD, nnz, NT_);                       \
                  hipLaunchKernelGGL((spmv_fixup_kernel<double, TILE>), dim3((unsigned)((NT_ - 1 + 255) / 256)),     \
                                     dim3(256), 0, st, (const int64_t *)row_ptr, (const int64_t *)LB,                \
                                     (const double *)HD, y, NT_, false);                                                    \
                }, true, {}})
  V4(1024, false, lb1k, hd1k, nt1k);
  V4(1024, true, lb1k, hd1k, nt1k);
  V4(2048, false, lb2k, hd2k, nt2k);
  V4(2048, true, lb2k, hd2k, nt2k);


#define V5(TILE, U, PERSIST, GRID, LB, HD, NT_)                                                                     \
  vs.push_back({"v5 TILE=" #TILE " U=" #U " persist=" #PERSIST " grid=" #GRID, [&] {                                 \
                  hipLaunchKernelGGL((spmv_v5<TILE, U, PERSIST>), dim3((unsigned)(PERSIST ? GRID : NT_)), dim3(256), 0, st, \
                                     (const int64_t *)row_ptr, (const int32_t *)col, (const double *)val,            \
                                     (const double *)x, y, (const int64_t *)LB, HD, nnz, NT_);                       \
                  hipLaunchKernelGGL((spmv_fixup_kernel<double, TILE>), dim3((unsigned)((NT_ - 1 + 255) / 256)),     \
                                     dim3(256), 0, st, (const int64_t *)row_ptr, (const int64_t *)LB,                \
                                     (const double *)HD, y, NT_, false);                                                    \
                }, true, {}})
  V5(1024, 4, false, 0, lb1k, hd1k, nt1k);
  V5(1024, 8, false, 0, lb1k, hd1k, nt1k);
  V5(1024, 8, true, 2048, lb1k, hd1k, nt1k);
  V5(1024, 8, true, 4096, lb1k, hd1k, nt1k);
  V5(1024, 8, true, 8192, lb1k, hd1k, nt1k);
  V5(2048, 8, true, 2048, lb2k, hd2k, nt2k);


#define V6(TILE, THREADS, WY, RR, NTS, LB, HD, NT_, CHECK)                                                          \
  vs.push_back({"v6 TILE=" #TILE " thr=" #THREADS " writey=" #WY " rowptr=" #RR " ntstore=" #NTS, [&] {             \
                  hipLaunchKernelGGL((spmv_v6<TILE, THREADS, WY, RR, NTS>), dim3((unsigned)NT_), dim3(THREADS), 0, st, \
                                     (const int64_t *)row_ptr, (const int32_t *)col, (const double *)val,            \
                                     (const double *)x, y, (const int64_t *)LB, HD, nnz, NT_, g);                    \
                  hipLaunchKernelGGL((spmv_fixup_kernel<double, TILE>), dim3((unsigned)((NT_ - 1 + 255) / 256)),     \
                                     dim3(256), 0, st, (const int64_t *)row_ptr, (const int64_t *)LB,                \
                                     (const double *)HD, y, NT_, false);                                                    \
                }, CHECK, {}})
  V6(1024, 256, true, true, false, lb1k, hd1k, nt1k, true);
  V6(1024, 256, false, true, false, lb1k, hd1k, nt1k, false);
  V6(1024, 256, true, false, false, lb1k, hd1k, nt1k, true);
  V6(1024, 256, false, false, false, lb1k, hd1k, nt1k, false);
  V6(1024, 256, true, true, true, lb1k, hd1k, nt1k, true);
  V6(1024, 128, true, true, false, lb1k, hd1k, nt1k, true);
  V6(512, 128, true, true, false, lb512, hd512, nt512, true);
  V6(2048, 512, true, true, false, lb2k, hd2k, nt2k, true);
  V6(4096, 1024, true, true, false, lb4k, hd4k, nt4k, true);


#define V7(TILE, NS, NR, NW, LB, HD, NT_)                                                                           \
  vs.push_back({"v7 dense TILE=" #TILE " ntstream=" #NS " ntrowptr=" #NR " ntstore=" #NW, [&] {                     \
                  hipLaunchKernelGGL((spmv_v7<TILE, NS, NR, NW>), dim3((unsigned)NT_), dim3(256), 0, st,             \
                                     (const int64_t *)row_ptr, (const int32_t *)col, (const double *)val,            \
                                     (const double *)x, y, (const int64_t *)LB, HD, nnz, NT_);                       \
                  hipLaunchKernelGGL((spmv_fixup_kernel<double, TILE>), dim3((unsigned)((NT_ - 1 + 255) / 256)),     \
                                     dim3(256), 0, st, (const int64_t *)row_ptr, (const int64_t *)LB,                \
                                     (const double *)HD, y, NT_, false);                                                    \
                }, true, {}})
  V7(1024, false, false, false, lb1k, hd1k, nt1k);
  V7(1024, true, false, false, lb1k, hd1k, nt1k);
  V7(1024, true, true, false, lb1k, hd1k, nt1k);
  V7(1024, true, true, true, lb1k, hd1k, nt1k);
  V7(2048, true, false, false, lb2k, hd2k, nt2k);
  V7(2048, true, true, false, lb2k, hd2k, nt2k);
  V7(4096, true, true, false, lb4k, hd4k, nt4k);

  // optional filter: keep variants whose name contains argv[4] (variant 0 is always the reference)
  if (argc > 4) {
    std::vector<Variant> keep;
    for (size_t i = 0; i < vs.size(); i++)
      if (i == 0 || strstr(vs[i].name, argv[4])) keep.push_back(vs[i]);
    vs.swap(keep);
  }
  // reference output
  vs[0].run();
  CK(hipMemcpyAsync(yref, y, 8 * n, hipMemcpyDeviceToDevice, st));
  CK(hipStreamSynchronize(st));
  std::vector<double> href(n), hy(n);
  CK(hipMemcpy(href.data(), yref, 8 * n, hipMemcpyDeviceToHost));

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (auto &v : vs) {
    CK(hipMemsetAsync(y, 0xff, 8 * n, st));
    v.run();
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    if (v.check) {
      CK(hipMemcpy(hy.data(), y, 8 * n, hipMemcpyDeviceToHost));
      int64_t bad = 0;
      for (int64_t i = 0; i < n; i++) bad += hy[i] != href[i];
      if (bad) printf("!! %s: %lld mismatches\n", v.name, (long long)bad);
    }
  }
  for (int r = 0; r < rounds; r++) {
    for (auto &v : vs) {
      v.run();
      CK(hipEventRecord(e0, st));
      for (int l = 0; l < launches; l++) v.run();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      v.ms.push_back(ms / launches);
    }
  }
  const double bytes = (double)nnz * 12 + (double)n * 24;
  printf("g=%lld n=%lld nnz=%lld algorithmic bytes=%.0f\n", (long long)g, (long long)n, (long long)nnz, bytes);
  for (auto &v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    const double med = v.ms[v.ms.size() / 2], mn = v.ms.front();
    printf("%-40s median %8.2f us  min %8.2f us  -> %7.1f GB/s (%.1f%% of 8 TB/s)\n", v.name, med * 1e3, mn * 1e3,
           bytes / (med * 1e-3) / 1e9, bytes / (med * 1e-3) / 1e9 / 80.0);
  }
  return 0;
}
