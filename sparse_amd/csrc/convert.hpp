// convert.hpp -- the remaining members of the matrix_irregular surface on a CSR
// record: coo (compressed.fut:166-177, :208-209), dense (:122-132, :214-215),
// zero/eye/diag (:98-120), scale (:148-152), nnz (:162-164), the record fields in
// the reference's convention, and adoption of an already-normalised record.
#pragma once

#include "assembly.hpp"

namespace samd {

// sr.coo: row ids by replicated_iota over the row lengths (:169-176), restated as
// "row that stores entry e" (find_row) so the work is balanced over entries.
template <class T, class I>
__global__ __launch_bounds__(kBlock) void coo_kernel(const int64_t *__restrict__ row_ptr, const I *__restrict__ col,
                                                     const T *__restrict__ vals, int64_t nnz,
                                                     const int64_t *__restrict__ tile_lb, int64_t tile,
                                                     int64_t *__restrict__ r_out, int64_t *__restrict__ c_out,
                                                     T *__restrict__ v_out) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= nnz) return;
  r_out[e] = find_row(row_ptr, tile_lb, tile, e);
  c_out[e] = (int64_t)col[e];
  v_out[e] = vals[e];
}

// the same, a whole tile per block with the rows from tile_rows (one row_ptr read per row, no search per entry)
template <class T, class I>
__global__ __launch_bounds__(kBlock) void coo_tile_kernel(const int64_t *__restrict__ row_ptr, const I *__restrict__ col,
                                                          const T *__restrict__ vals, int64_t nnz,
                                                          const int64_t *__restrict__ tile_lb,
                                                          int64_t *__restrict__ r_out, int64_t *__restrict__ c_out,
                                                          T *__restrict__ v_out) {
  __shared__ unsigned mark[4 * kBlock];
  __shared__ unsigned sh[kWaves];
  const int64_t t = blockIdx.x;
  int64_t row[4];
  tile_rows(row_ptr, tile_lb, t, mark, sh, row);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int64_t e = t * (4 * kBlock) + wave * 256 + k * 64 + lane;
    if (e < nnz) {
      r_out[e] = row[k];
      c_out[e] = (int64_t)col[e];
      v_out[e] = vals[e];
    }
  }
}

// dense: scatter_2d into a zeroed [n][m] (TRANSPOSED: into [m][n], sc.dense)
template <class T, class I, bool TRANSPOSED>
__global__ __launch_bounds__(kBlock) void dense_kernel(const int64_t *__restrict__ row_ptr, const I *__restrict__ col,
                                                       const T *__restrict__ vals, int64_t nnz,
                                                       const int64_t *__restrict__ tile_lb, int64_t tile, int64_t n,
                                                       int64_t m, T *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= nnz) return;
  const int64_t r = find_row(row_ptr, tile_lb, tile, e);
  const int64_t c = (int64_t)col[e];
  if (TRANSPOSED) out[c * n + r] = vals[e];
  else out[r * m + c] = vals[e];
}

template <class I>
__global__ __launch_bounds__(kBlock) void widen_col_kernel(const I *__restrict__ col, int64_t nnz,
                                                           int64_t *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e < nnz) out[e] = (int64_t)col[e];
}

// from_fields checks: row_off monotone and ending at nnz, 0 <= col < m, columns
// strictly increasing inside a row.
template <class I>
__global__ __launch_bounds__(kBlock) void adopt_col_kernel(const int64_t *__restrict__ col_in, int64_t nnz, int64_t m,
                                                           I *__restrict__ out, int *__restrict__ err) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= nnz) return;
  const int64_t c = col_in[e];
  if (c < 0 || c >= m) { *err = 1; out[e] = 0; }
  else out[e] = (I)c;
}

__global__ __launch_bounds__(kBlock) void adopt_rowoff_kernel(const int64_t *__restrict__ row_off, int64_t n,
                                                              int64_t nnz, int64_t *__restrict__ row_ptr,
                                                              int *__restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i == 0) row_ptr[0] = 0;
  if (i >= n) return;
  const int64_t v = row_off[i];
  const int64_t prev = i == 0 ? 0 : row_off[i - 1];
  if (v < prev || v > nnz || (i == n - 1 && v != nnz)) *err = 1;
  row_ptr[i + 1] = v;
}

template <class I>
__global__ __launch_bounds__(kBlock) void check_sorted_cols_kernel(const int64_t *__restrict__ row_ptr,
                                                                   const I *__restrict__ col, int64_t nnz,
                                                                   const int64_t *__restrict__ tile_lb, int64_t tile,
                                                                   int *__restrict__ err) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= nnz || e == 0) return;
  const int64_t r = find_row(row_ptr, tile_lb, tile, e);
  if (row_ptr[r] < e && !(col[e - 1] < col[e])) *err = 2;
}

template <class T>
__global__ __launch_bounds__(kBlock) void eye_kernel(int64_t n, int64_t e, int64_t *__restrict__ row_ptr,
                                                     int32_t *__restrict__ col32, int64_t *__restrict__ col64,
                                                     T *__restrict__ vals, const T *__restrict__ diag) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i <= n) row_ptr[i] = i < e ? i : e;  // row_off[i] = min(i+1, e) (:107-108)
  if (i < e) {
    if (col32) col32[i] = (int32_t)i; else col64[i] = i;
    vals[i] = diag ? diag[i] : (T)1;
  }
}

template <class T>
__global__ __launch_bounds__(kBlock) void scale_kernel(T v, const T *__restrict__ in, int64_t nnz, T *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e < nnz) out[e] = (T)(v * in[e]);
}

// nnz: stored values that are not zero; eq a b = !(a<b) && !(b<a) (:70,:162-164)
template <class T>
__global__ __launch_bounds__(kBlock) void count_nonzero_kernel(const T *__restrict__ vals, int64_t nnz,
                                                               unsigned long long *__restrict__ out) {
  __shared__ unsigned long long sh[kWaves];
  unsigned long long c = 0;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * kBlock) {
    const T v = vals[e];
    const bool eq0 = !(v < (T)0) && !((T)0 < v);
    c += eq0 ? 0 : 1;
  }
  c = wave_reduce(c, OpAdd());
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kWaves; w++) t += sh[w];
    if (t) atomicAdd(out, t);
  }
}

template <class T>
int alloc_cols(futhark_context *ctx, CsrBody<T> *b) {
  if (use_col32(ctx, b->m)) return dalloc(ctx, &b->col32, b->nnz);
  return dalloc(ctx, &b->col64, b->nnz);
}

// eye n m (diag == nullptr) or diag v (n == m == len v)
template <class T>
int make_eye(futhark_context *ctx, int64_t n, int64_t m, const T *diag, CsrBody<T> **out) {
  if (n < 0 || m < 0) return set_error(ctx, "eye: negative dimension");
  CsrBody<T> *b = new CsrBody<T>();
  CsrGuard<T> guard(ctx, b);
  b->n = n; b->m = m; b->nnz = n < m ? n : m;
  SAMD_TRY(dalloc(ctx, &b->row_ptr, n + 1));
  SAMD_TRY(dalloc(ctx, &b->vals, b->nnz));
  SAMD_TRY(alloc_cols(ctx, b));
  hipLaunchKernelGGL((eye_kernel<T>), dim3((unsigned)cdiv(n + 1, kBlock)), dim3(kBlock), 0, ctx->stream, n, b->nnz,
                     b->row_ptr, b->col32, b->col64, b->vals, diag);
  SAMD_LAUNCH_CHECK(ctx);
  SAMD_TRY(build_plan(ctx, b));
  *out = guard.release();
  return 0;
}

template <class T>
int make_zero(futhark_context *ctx, int64_t n, int64_t m, CsrBody<T> **out) {
  if (n < 0 || m < 0) return set_error(ctx, "zero: negative dimension");
  CsrBody<T> *b = new CsrBody<T>();
  CsrGuard<T> guard(ctx, b);
  b->n = n; b->m = m; b->nnz = 0;
  SAMD_TRY(dalloc(ctx, &b->row_ptr, n + 1));
  SAMD_HIP(ctx, hipMemsetAsync(b->row_ptr, 0, sizeof(int64_t) * (size_t)(n + 1), ctx->stream));
  SAMD_TRY(dalloc(ctx, &b->vals, 0));
  SAMD_TRY(alloc_cols(ctx, b));
  SAMD_TRY(build_plan(ctx, b));
  *out = guard.release();
  return 0;
}

template <class T>
int make_scaled(futhark_context *ctx, T v, const CsrBody<T> *a, CsrBody<T> **out) {
  CsrBody<T> *b = new CsrBody<T>();
  CsrGuard<T> guard(ctx, b);
  b->n = a->n; b->m = a->m; b->nnz = a->nnz;
  SAMD_TRY(dalloc(ctx, &b->row_ptr, a->n + 1));
  SAMD_HIP(ctx, hipMemcpyAsync(b->row_ptr, a->row_ptr, sizeof(int64_t) * (size_t)(a->n + 1), hipMemcpyDeviceToDevice,
                               ctx->stream));
  SAMD_TRY(dalloc(ctx, &b->vals, a->nnz));
  if (a->col32) {
    SAMD_TRY(dalloc(ctx, &b->col32, a->nnz));
    if (a->nnz) SAMD_HIP(ctx, hipMemcpyAsync(b->col32, a->col32, sizeof(int32_t) * (size_t)a->nnz, hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    SAMD_TRY(dalloc(ctx, &b->col64, a->nnz));
    if (a->nnz) SAMD_HIP(ctx, hipMemcpyAsync(b->col64, a->col64, sizeof(int64_t) * (size_t)a->nnz, hipMemcpyDeviceToDevice, ctx->stream));
  }
  if (a->nnz) {
    hipLaunchKernelGGL((scale_kernel<T>), dim3((unsigned)cdiv(a->nnz, kBlock)), dim3(kBlock), 0, ctx->stream, v,
                       (const T *)a->vals, a->nnz, b->vals);
    SAMD_LAUNCH_CHECK(ctx);
  }
  SAMD_TRY(build_plan(ctx, b));
  *out = guard.release();
  return 0;
}

template <class T>
int count_nonzero(futhark_context *ctx, const CsrBody<T> *a, int64_t *out) {
  SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
  if (a->nnz) {
    int64_t want = cdiv(a->nnz, kBlock);
    const unsigned grid = (unsigned)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL((count_nonzero_kernel<T>), dim3(grid), dim3(kBlock), 0, ctx->stream, (const T *)a->vals, a->nnz,
                       (unsigned long long *)ctx->d_small);
    SAMD_LAUNCH_CHECK(ctx);
  }
  return read_small(ctx, ctx->d_small, 1, out);
}

// coo into caller-provided device arrays of length nnz
template <class T>
int to_coo(futhark_context *ctx, const CsrBody<T> *a, int64_t *r, int64_t *c, T *v) {
  if (a->nnz == 0) return 0;
  if (a->tile == 4 * kBlock && a->n < 0x7fffffffLL) {
    if (a->col32)
      hipLaunchKernelGGL((coo_tile_kernel<T, int32_t>), dim3((unsigned)a->ntiles), dim3(kBlock), 0, ctx->stream,
                         (const int64_t *)a->row_ptr, (const int32_t *)a->col32, (const T *)a->vals, a->nnz,
                         (const int64_t *)a->tile_lb, r, c, v);
    else
      hipLaunchKernelGGL((coo_tile_kernel<T, int64_t>), dim3((unsigned)a->ntiles), dim3(kBlock), 0, ctx->stream,
                         (const int64_t *)a->row_ptr, (const int64_t *)a->col64, (const T *)a->vals, a->nnz,
                         (const int64_t *)a->tile_lb, r, c, v);
    SAMD_LAUNCH_CHECK(ctx);
    return 0;
  }
  const unsigned grid = (unsigned)cdiv(a->nnz, kBlock);
  if (a->col32)
    hipLaunchKernelGGL((coo_kernel<T, int32_t>), dim3(grid), dim3(kBlock), 0, ctx->stream, (const int64_t *)a->row_ptr,
                       (const int32_t *)a->col32, (const T *)a->vals, a->nnz, (const int64_t *)a->tile_lb, a->tile, r,
                       c, v);
  else
    hipLaunchKernelGGL((coo_kernel<T, int64_t>), dim3(grid), dim3(kBlock), 0, ctx->stream, (const int64_t *)a->row_ptr,
                       (const int64_t *)a->col64, (const T *)a->vals, a->nnz, (const int64_t *)a->tile_lb, a->tile, r,
                       c, v);
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

// dense into a caller-provided device array: [n][m] row-major, or [m][n] when transposed
template <class T>
int to_dense(futhark_context *ctx, const CsrBody<T> *a, bool transposed, T *out) {
  const int64_t total = a->n * a->m;
  if (total > 0) SAMD_HIP(ctx, hipMemsetAsync(out, 0, sizeof(T) * (size_t)total, ctx->stream));
  if (a->nnz == 0) return 0;
  const unsigned grid = (unsigned)cdiv(a->nnz, kBlock);
#define SAMD_DENSE(I, COL, TR)                                                                                  \
  hipLaunchKernelGGL((dense_kernel<T, I, TR>), dim3(grid), dim3(kBlock), 0, ctx->stream,                        \
                     (const int64_t *)a->row_ptr, (const I *)COL, (const T *)a->vals, a->nnz,                   \
                     (const int64_t *)a->tile_lb, a->tile, a->n, a->m, out)
  if (a->col32) { if (transposed) SAMD_DENSE(int32_t, a->col32, true); else SAMD_DENSE(int32_t, a->col32, false); }
  else { if (transposed) SAMD_DENSE(int64_t, a->col64, true); else SAMD_DENSE(int64_t, a->col64, false); }
#undef SAMD_DENSE
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

// record fields in the reference's convention into caller-provided arrays
template <class T>
int to_fields(futhark_context *ctx, const CsrBody<T> *a, int64_t *row_off, int64_t *col_idx, T *vals) {
  if (a->n > 0)
    SAMD_HIP(ctx, hipMemcpyAsync(row_off, a->row_ptr + 1, sizeof(int64_t) * (size_t)a->n, hipMemcpyDeviceToDevice,
                                 ctx->stream));
  if (a->nnz == 0) return 0;
  SAMD_HIP(ctx, hipMemcpyAsync(vals, a->vals, sizeof(T) * (size_t)a->nnz, hipMemcpyDeviceToDevice, ctx->stream));
  const unsigned grid = (unsigned)cdiv(a->nnz, kBlock);
  if (a->col32)
    hipLaunchKernelGGL((widen_col_kernel<int32_t>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                       (const int32_t *)a->col32, a->nnz, col_idx);
  else
    hipLaunchKernelGGL((widen_col_kernel<int64_t>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)a->col64, a->nnz, col_idx);
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

template <class T>
int from_fields(futhark_context *ctx, int64_t n, int64_t m, const int64_t *row_off, const int64_t *col_idx,
                const T *vals, int64_t nnz, CsrBody<T> **out) {
  if (n < 0 || m < 0) return set_error(ctx, "from_fields: negative dimension");
  if (n == 0 && nnz != 0) return set_error(ctx, "from_fields: entries without rows");
  CsrBody<T> *b = new CsrBody<T>();
  CsrGuard<T> guard(ctx, b);
  b->n = n; b->m = m; b->nnz = nnz;
  Tmp<int> err(ctx);
  SAMD_TRY(err.get(1));
  SAMD_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(int), ctx->stream));
  SAMD_TRY(dalloc(ctx, &b->row_ptr, n + 1));
  hipLaunchKernelGGL(adopt_rowoff_kernel, dim3((unsigned)cdiv(n + 1, kBlock)), dim3(kBlock), 0, ctx->stream, row_off,
                     n, nnz, b->row_ptr, err.p);
  SAMD_LAUNCH_CHECK(ctx);
  SAMD_TRY(dalloc(ctx, &b->vals, nnz));
  SAMD_TRY(alloc_cols(ctx, b));
  if (nnz > 0) {
    SAMD_HIP(ctx, hipMemcpyAsync(b->vals, vals, sizeof(T) * (size_t)nnz, hipMemcpyDeviceToDevice, ctx->stream));
    const unsigned grid = (unsigned)cdiv(nnz, kBlock);
    if (b->col32)
      hipLaunchKernelGGL((adopt_col_kernel<int32_t>), dim3(grid), dim3(kBlock), 0, ctx->stream, col_idx, nnz, m,
                         b->col32, err.p);
    else
      hipLaunchKernelGGL((adopt_col_kernel<int64_t>), dim3(grid), dim3(kBlock), 0, ctx->stream, col_idx, nnz, m,
                         b->col64, err.p);
    SAMD_LAUNCH_CHECK(ctx);
  }
  // row_off must be valid before a plan can be searched
  SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
  SAMD_HIP(ctx, hipMemcpyAsync(ctx->d_small, err.p, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
  int64_t host = 0;
  SAMD_TRY(read_small(ctx, ctx->d_small, 1, &host));
  if (host != 0) return set_error(ctx, "from_fields: row_off not a monotone prefix ending at nnz, or col_idx out of range");
  SAMD_TRY(build_plan(ctx, b));
  if (nnz > 0) {
    const unsigned grid = (unsigned)cdiv(nnz, kBlock);
    if (b->col32)
      hipLaunchKernelGGL((check_sorted_cols_kernel<int32_t>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         (const int64_t *)b->row_ptr, (const int32_t *)b->col32, nnz, (const int64_t *)b->tile_lb,
                         b->tile, err.p);
    else
      hipLaunchKernelGGL((check_sorted_cols_kernel<int64_t>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         (const int64_t *)b->row_ptr, (const int64_t *)b->col64, nnz, (const int64_t *)b->tile_lb,
                         b->tile, err.p);
    SAMD_LAUNCH_CHECK(ctx);
    SAMD_HIP(ctx, hipMemcpyAsync(ctx->d_small, err.p, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    SAMD_TRY(read_small(ctx, ctx->d_small, 1, &host));
    if (host != 0) return set_error(ctx, "from_fields: col_idx not strictly increasing within a row");
  }
  *out = guard.release();
  return 0;
}

}  // namespace samd
