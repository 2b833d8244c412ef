// mono.hpp -- mono sparse row/column matrices: at most one stored element per row
// (module mk_mono, S/mono.fut:61-209; S = lib/github.com/diku-dk/sparse in the reference).
//
// Record (mono.fut:95): col_idx[n] i64, vals[n] T; a row without an element holds (0, zero).
// msc[n][m] = msr[m][n] (mono.fut:203): a column-format handle holds the record of the transpose.
//
// sparse re-uses the compressed assembly (same norm_coo, mono.fut:74-89 = compressed.fut:72-87)
// and then checks the reference's second assert (no two elements in one row, :122-126).
// vsmm (mono.fut:157-159, reduce_by_index) is made order-deterministic: the rows are grouped by
// column once per matrix (a CSR record of the transpose built with the same assembly, cached in
// the body), and every product is then the ordinary SpMV over that record, summing the
// contributions of one column in ascending row order -- the order the oracle uses.
// Everything here is a streaming map/gather over n elements: HBM-bound, 8+vb bytes per row.
#pragma once

#include "convert.hpp"
#include "spmm.hpp"

namespace samd {

template <class T>
struct MonoBody {
  int64_t n = 0, m = 0;
  int64_t *col = nullptr;
  T *vals = nullptr;
  // some col_idx >= m: only `eye n m` with n > m produces it (mono.fut:102-105 stores iota n);
  // dense and vsmm ignore such elements (scatter / reduce_by_index semantics), smvm is an error.
  bool col_oob = false;
  CsrBody<T> *bins = nullptr;  // lazily built: (m+1) x n record grouping rows by column; bin m = out of range
  int refs = 1;
};

template <class T>
inline void mono_release(futhark_context *ctx, MonoBody<T> *b) {
  if (!b) return;
  if (--b->refs > 0) return;
  dfree(ctx, b->col);
  dfree(ctx, b->vals);
  csr_release(ctx, b->bins);
  delete b;
}

template <class T>
struct MonoGuard {
  futhark_context *ctx;
  MonoBody<T> *b;
  MonoGuard(futhark_context *c, MonoBody<T> *p) : ctx(c), b(p) {}
  ~MonoGuard() { if (b) mono_release(ctx, b); }
  MonoBody<T> *release() { MonoBody<T> *q = b; b = nullptr; return q; }
};

template <class T>
inline int mono_alloc(futhark_context *ctx, int64_t n, int64_t m, MonoBody<T> **out) {
  if (n < 0 || m < 0) return set_error(ctx, "mono: negative dimension");
  MonoBody<T> *b = new MonoBody<T>();
  MonoGuard<T> g(ctx, b);
  b->n = n;
  b->m = m;
  SAMD_TRY(dalloc(ctx, &b->col, n));
  SAMD_TRY(dalloc(ctx, &b->vals, n));
  *out = g.release();
  return 0;
}

// mode 0: zero (:97-100)  1: eye (:102-105)  2: diag v (:107-110)
template <class T>
__global__ __launch_bounds__(kBlock) void mono_fill_kernel(int64_t *__restrict__ col, T *__restrict__ vals, int64_t n,
                                                           int mode, const T *__restrict__ diag) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  col[i] = mode == 0 ? 0 : i;
  vals[i] = mode == 0 ? (T)0 : (mode == 1 ? (T)1 : diag[i]);
}

template <class T>
int mono_make(futhark_context *ctx, int64_t n, int64_t m, int mode, const T *diag, MonoBody<T> **out) {
  MonoBody<T> *b = nullptr;
  SAMD_TRY(mono_alloc<T>(ctx, n, m, &b));
  MonoGuard<T> g(ctx, b);
  if (n > 0) {
    hipLaunchKernelGGL((mono_fill_kernel<T>), dim3((unsigned)cdiv(n, kBlock)), dim3(kBlock), 0, ctx->stream, b->col,
                       b->vals, n, mode, diag);
    SAMD_LAUNCH_CHECK(ctx);
  }
  b->col_oob = mode == 1 && n > m;
  *out = g.release();
  return 0;
}

// scatter of the normalised list into rows (:127-128); flag = some row holds two elements (:122-126)
template <class T, class I>
__global__ __launch_bounds__(kBlock) void mono_from_csr_kernel(const int64_t *__restrict__ row_ptr,
                                                               const I *__restrict__ cin, const T *__restrict__ vin,
                                                               int64_t n, int64_t *__restrict__ col,
                                                               T *__restrict__ vals, int *__restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t s = row_ptr[i], e = row_ptr[i + 1];
  if (e - s > 1) *flag = 1;  // every writer stores the same value
  col[i] = e > s ? (int64_t)cin[s] : 0;
  vals[i] = e > s ? vin[s] : (T)0;
}

template <class T>
int mono_sparse(futhark_context *ctx, int64_t n, int64_t m, const int64_t *rs, const int64_t *cs, const T *vs,
                int64_t count, MonoBody<T> **out) {
  CsrBody<T> *c = nullptr;
  SAMD_TRY(assemble_coo<T>(ctx, n, m, rs, cs, vs, count, &c));  // norm_coo + the bounds assert (:120-121)
  CsrGuard<T> cg(ctx, c);
  MonoBody<T> *b = nullptr;
  SAMD_TRY(mono_alloc<T>(ctx, n, m, &b));
  MonoGuard<T> g(ctx, b);
  if (n > 0) {
    int *flag = (int *)ctx->d_small;
    SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
    const unsigned grid = (unsigned)cdiv(n, kBlock);
    if (c->col32)
      hipLaunchKernelGGL((mono_from_csr_kernel<T, int32_t>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         (const int64_t *)c->row_ptr, (const int32_t *)c->col32, (const T *)c->vals, n, b->col,
                         b->vals, flag);
    else
      hipLaunchKernelGGL((mono_from_csr_kernel<T, int64_t>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         (const int64_t *)c->row_ptr, (const int64_t *)c->col64, (const T *)c->vals, n, b->col,
                         b->vals, flag);
    SAMD_LAUNCH_CHECK(ctx);
    int64_t host = 0;
    SAMD_TRY(read_small(ctx, ctx->d_small, 1, &host));
    if ((int)(host & 0xffffffff) != 0)
      return set_error(ctx, "Error: two elements in one row of a mono sparse matrix (assertion r1 != r2 failed)");
  }
  *out = g.release();
  return 0;
}

// dense (:112-114): scatter_2d ignores out-of-range columns.  out is [n][m], or [m][n] when transposed
// (sc.dense = sr.dense |> transpose, :191-192).
template <class T>
__global__ __launch_bounds__(kBlock) void mono_dense_kernel(const int64_t *__restrict__ col, const T *__restrict__ vals,
                                                            int64_t n, int64_t m, bool transposed,
                                                            T *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t c = col[i];
  if (c < 0 || c >= m) return;
  out[transposed ? c * n + i : i * m + c] = vals[i];
}

template <class T>
int mono_dense(futhark_context *ctx, const MonoBody<T> *a, bool transposed, T *out) {
  const int64_t total = a->n * a->m;
  if (total > 0) SAMD_HIP(ctx, hipMemsetAsync(out, 0, sizeof(T) * (size_t)total, ctx->stream));
  if (a->n > 0 && a->m > 0) {
    hipLaunchKernelGGL((mono_dense_kernel<T>), dim3((unsigned)cdiv(a->n, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)a->col, (const T *)a->vals, a->n, a->m, transposed, out);
    SAMD_LAUNCH_CHECK(ctx);
  }
  return 0;
}

// scale (:116-117), + and - (:139-149; the col_idx of both operands must agree)
template <class T>
__global__ __launch_bounds__(kBlock) void mono_map_kernel(const int64_t *__restrict__ colA, const T *__restrict__ va,
                                                          const int64_t *__restrict__ colB, const T *__restrict__ vb,
                                                          T s, int op, int64_t n, int64_t *__restrict__ col,
                                                          T *__restrict__ vals, int *__restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t c = colA[i];
  col[i] = c;
  if (op == 0) { vals[i] = (T)(s * va[i]); return; }
  if (colB[i] != c) *flag = 1;
  vals[i] = op == 1 ? (T)(va[i] + vb[i]) : (T)(va[i] - vb[i]);
}

template <class T>
int mono_map(futhark_context *ctx, const MonoBody<T> *a, const MonoBody<T> *b2, T s, int op, MonoBody<T> **out) {
  if (op != 0 && (a->n != b2->n || a->m != b2->m)) return set_error(ctx, "mono +/-: operands differ in shape");
  MonoBody<T> *b = nullptr;
  SAMD_TRY(mono_alloc<T>(ctx, a->n, a->m, &b));
  MonoGuard<T> g(ctx, b);
  b->col_oob = a->col_oob;
  if (a->n > 0) {
    int *flag = (int *)ctx->d_small;
    if (op != 0) SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
    hipLaunchKernelGGL((mono_map_kernel<T>), dim3((unsigned)cdiv(a->n, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)a->col, (const T *)a->vals, (const int64_t *)(b2 ? b2->col : nullptr),
                       (const T *)(b2 ? b2->vals : nullptr), s, op, a->n, b->col, b->vals, flag);
    SAMD_LAUNCH_CHECK(ctx);
    if (op != 0) {
      int64_t host = 0;
      SAMD_TRY(read_small(ctx, ctx->d_small, 1, &host));
      if ((int)(host & 0xffffffff) != 0)
        return set_error(ctx, "Error: mono +/-: column indices differ (assertion c == c' failed)");
    }
  }
  *out = g.release();
  return 0;
}

// nnz (:131-133): values that are not zero
template <class T>
int mono_nnz(futhark_context *ctx, const MonoBody<T> *a, int64_t *out) {
  SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
  if (a->n) {
    const int64_t want = cdiv(a->n, kBlock);
    hipLaunchKernelGGL((count_nonzero_kernel<T>), dim3((unsigned)(want < 4096 ? want : 4096)), dim3(kBlock), 0,
                       ctx->stream, (const T *)a->vals, a->n, (unsigned long long *)ctx->d_small);
    SAMD_LAUNCH_CHECK(ctx);
  }
  return read_small(ctx, ctx->d_small, 1, out);
}

// coo (:135-137): filter (v < 0 || 0 < v) over zip3 (iota n) col_idx vals -- order preserved
template <class T>
struct NonZeroIn {
  const T *vals;
  __device__ __forceinline__ uint32_t operator()(int64_t i) const {
    const T v = vals[i];
    return (v < (T)0 || (T)0 < v) ? 1u : 0u;
  }
};

template <class T>
__global__ __launch_bounds__(kBlock) void mono_coo_emit_kernel(const int64_t *__restrict__ col,
                                                               const T *__restrict__ vals,
                                                               const uint32_t *__restrict__ slots, int64_t n,
                                                               bool swap, int64_t *__restrict__ r,
                                                               int64_t *__restrict__ c, T *__restrict__ v) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const T x = vals[i];
  if (!(x < (T)0 || (T)0 < x)) return;
  const uint32_t s = slots[i] - 1;
  r[s] = swap ? col[i] : i;  // sc.coo = sr.coo |> map swap (:185-186)
  c[s] = swap ? i : col[i];
  v[s] = x;
}

// Two steps so that the C layer can size the output arrays: count, then emit.
template <class T>
int mono_coo_count(futhark_context *ctx, const MonoBody<T> *a, Tmp<uint32_t> &slots, int64_t *count) {
  if (a->n >= 0xffff0000LL) return set_error(ctx, "mono coo: more than 2^32 rows");
  SAMD_TRY(slots.get(a->n));
  SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
  SAMD_TRY((device_scan<uint32_t, OpAdd, NonZeroIn<T>, false>(ctx, NonZeroIn<T>{a->vals}, a->n, 0u, OpAdd(), slots.p,
                                                               (uint32_t *)ctx->d_small)));
  int64_t host = 0;
  SAMD_TRY(read_small(ctx, ctx->d_small, 1, &host));
  *count = (int64_t)(uint32_t)(host & 0xffffffffLL);
  return 0;
}

template <class T>
int mono_coo_emit(futhark_context *ctx, const MonoBody<T> *a, const uint32_t *slots, bool swap, int64_t *r, int64_t *c,
                  T *v) {
  if (a->n > 0) {
    hipLaunchKernelGGL((mono_coo_emit_kernel<T>), dim3((unsigned)cdiv(a->n, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)a->col, (const T *)a->vals, slots, a->n, swap, r, c, v);
    SAMD_LAUNCH_CHECK(ctx);
  }
  return 0;
}

// smvm (:154-155): y[i] = vals[i] * v[col_idx[i]] -- one gather per row
template <class T>
__global__ __launch_bounds__(kBlock) void mono_smvm_kernel(const int64_t *__restrict__ col, const T *__restrict__ vals,
                                                           const T *__restrict__ x, T *__restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) y[i] = (T)(vals[i] * x[col[i]]);
}

template <class T>
int mono_smvm(futhark_context *ctx, const MonoBody<T> *a, const T *x, T *y) {
  if (a->col_oob) return set_error(ctx, "Error: index out of bounds in mono smvm (col_idx >= m; eye n m with n > m)");
  if (a->n > 0) {
    hipLaunchKernelGGL((mono_smvm_kernel<T>), dim3((unsigned)cdiv(a->n, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)a->col, (const T *)a->vals, x, y, a->n);
    SAMD_LAUNCH_CHECK(ctx);
  }
  return 0;
}

// ---- rows grouped by column, built once per matrix
__global__ __launch_bounds__(kBlock) void mono_bin_keys_kernel(const int64_t *__restrict__ col, int64_t n, int64_t m,
                                                               int64_t *__restrict__ major,
                                                               int64_t *__restrict__ minor) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t c = col[i];
  major[i] = (c < 0 || c >= m) ? m : c;  // bin m collects what reduce_by_index would ignore
  minor[i] = i;
}

template <class T>
int mono_bins(futhark_context *ctx, const MonoBody<T> *a, const CsrBody<T> **out) {
  if (!a->bins) {
    Tmp<int64_t> major(ctx), minor(ctx);
    SAMD_TRY(major.get(a->n));
    SAMD_TRY(minor.get(a->n));
    if (a->n > 0) {
      hipLaunchKernelGGL(mono_bin_keys_kernel, dim3((unsigned)cdiv(a->n, kBlock)), dim3(kBlock), 0, ctx->stream,
                         (const int64_t *)a->col, a->n, a->m, major.p, minor.p);
      SAMD_LAUNCH_CHECK(ctx);
    }
    CsrBody<T> *c = nullptr;
    SAMD_TRY(assemble_coo<T>(ctx, a->m + 1, a->n > 0 ? a->n : 1, major.p, minor.p, (const T *)a->vals, a->n, &c));
    const_cast<MonoBody<T> *>(a)->bins = c;
  }
  *out = a->bins;
  return 0;
}

// vsmm (:157-159): out[c] = sum over rows i with col_idx[i] == c of v[i] * vals[i]  == SpMV over the bins.
// y must have room for m + 1 elements (the last one is the out-of-range bin and is not part of the result).
template <class T>
int mono_vsmm(futhark_context *ctx, const MonoBody<T> *a, const T *v, T *y_m_plus_1) {
  const CsrBody<T> *bins = nullptr;
  SAMD_TRY(mono_bins(ctx, a, &bins));
  return spmv<T>(ctx, bins, v, y_m_plus_1);
}

// dmsmm (:161-162): out[r][:] = vsmm D[r][:] S, for D [n][k] and S msr[k][m].  One thread per output
// element; the contributions of a column are added in ascending row order of S.
template <class T, class I>
__global__ __launch_bounds__(kBlock) void mono_dmsmm_kernel(const int64_t *__restrict__ bin_ptr,
                                                            const I *__restrict__ rows, const T *__restrict__ vals,
                                                            const T *__restrict__ D, int64_t n, int64_t k, int64_t m,
                                                            T *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n * m) return;
  const int64_t r = e / m, c = e - r * m;
  T acc = (T)0;
  for (int64_t q = bin_ptr[c]; q < bin_ptr[c + 1]; q++) acc = (T)(acc + (T)(D[r * k + (int64_t)rows[q]] * vals[q]));
  out[e] = acc;
}

template <class T>
int mono_dmsmm(futhark_context *ctx, const T *D, int64_t n, const MonoBody<T> *s, T *out) {
  const int64_t k = s->n, m = s->m;
  if (n * m == 0) return 0;
  const CsrBody<T> *bins = nullptr;
  SAMD_TRY(mono_bins(ctx, s, &bins));
  const unsigned grid = (unsigned)cdiv(n * m, kBlock);
  if (bins->col32)
    hipLaunchKernelGGL((mono_dmsmm_kernel<T, int32_t>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)bins->row_ptr, (const int32_t *)bins->col32, (const T *)bins->vals, D, n, k, m,
                       out);
  else
    hipLaunchKernelGGL((mono_dmsmm_kernel<T, int64_t>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)bins->row_ptr, (const int64_t *)bins->col64, (const T *)bins->vals, D, n, k, m,
                       out);
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

}  // namespace samd
