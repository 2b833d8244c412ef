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

#include "assembly.hpp"

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

template <class T>
int spmm(futhark_context *ctx, const CsrBody<T> *a, const T *B, int64_t k, T *C) {
  if (a->n == 0 || k == 0) return 0;
  int64_t want = cdiv(a->n * 64, kBlock);
  const int64_t cap = (int64_t)ctx->num_cu * 64;
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  if (a->col32)
    hipLaunchKernelGGL((spmm_rowpanel_kernel<T, int32_t>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)a->row_ptr, (const int32_t *)a->col32, (const T *)a->vals, B, C, a->n, k);
  else
    hipLaunchKernelGGL((spmm_rowpanel_kernel<T, int64_t>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)a->row_ptr, (const int64_t *)a->col64, (const T *)a->vals, B, C, a->n, k);
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
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
