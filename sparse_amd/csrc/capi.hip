// capi.hip -- the extern "C" boundary declared in include/sparse_amd.h.
//
// Every function: lock the context, make its device current, run stream-ordered
// work, return 0 / non-zero with the message left for futhark_context_get_error.
#include "common.hpp"
#include "scan.hpp"
#include "radix_sort.hpp"
#include "assembly.hpp"
#include "spmv.hpp"
#include "spmv_pb.hpp"
#include "spmm.hpp"
#include "convert.hpp"
#include "mono.hpp"
#include "mtx.hpp"
#include "store_format.hpp"
#include "bwprobe.hpp"

using namespace samd;

// ---------------------------------------------------------------- static scratch
// Device memory that exists without any allocation call, for the one entry point that must allocate nothing
// (sparse_amd_csr_*_smvm_into_noalloc).  Every GPU the library runs on holds its own copy of this array, at its
// own address, until the process ends.  In host code the array's name is only a stand-in: the address a kernel
// may use comes from hipGetSymbolAddress with the device current (futhark_context_new keeps it in the context).
constexpr size_t kStaticScratchBytes = (size_t)4 << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

// ---------------------------------------------------------------- records
struct futhark_i64_1d : DevArray<int64_t, 1> {};
struct futhark_i32_1d : DevArray<int32_t, 1> {};
struct futhark_f32_1d : DevArray<float, 1> {};
struct futhark_f64_1d : DevArray<double, 1> {};
struct futhark_i32_2d : DevArray<int32_t, 2> {};
struct futhark_f32_2d : DevArray<float, 2> {};
struct futhark_f64_2d : DevArray<double, 2> {};

struct futhark_opaque_csr_f64 { CsrBody<double> *b; };
struct futhark_opaque_csc_f64 { CsrBody<double> *b; };
struct futhark_opaque_csr_f32 { CsrBody<float> *b; };
struct futhark_opaque_csc_f32 { CsrBody<float> *b; };
struct futhark_opaque_csr_i32 { CsrBody<int32_t> *b; };
struct futhark_opaque_csc_i32 { CsrBody<int32_t> *b; };
struct futhark_opaque_msr_f64 { MonoBody<double> *b; };
struct futhark_opaque_msc_f64 { MonoBody<double> *b; };
struct futhark_opaque_msr_f32 { MonoBody<float> *b; };
struct futhark_opaque_msc_f32 { MonoBody<float> *b; };
struct futhark_opaque_msr_i32 { MonoBody<int32_t> *b; };
struct futhark_opaque_msc_i32 { MonoBody<int32_t> *b; };

namespace {

struct Enter {
  std::lock_guard<std::mutex> g;
  bool ok;
  explicit Enter(futhark_context *ctx) : g(ctx->mu), ok(ctx->own_stream != nullptr) {
    if (ok) (void)hipSetDevice(ctx->device);
    ctx->n_entries++;
  }
};

// A context whose device could not be opened fails every call loudly: there is
// no CPU path behind this boundary.
struct NoDevice {
  futhark_context *ctx;
  operator int() const {
    if (!ctx->has_err) set_error(ctx, "context has no usable HIP device; libsparse_amd has no CPU path");
    return 1;
  }
  template <class P>
  operator P *() const {
    if (!ctx->has_err) set_error(ctx, "context has no usable HIP device; libsparse_amd has no CPU path");
    return nullptr;
  }
};
#define ENTER(ctx) \
  Enter g(ctx);    \
  if (!g.ok) return NoDevice{ctx}

template <class A>
int arr_alloc(futhark_context *ctx, A **out, const int64_t *shape, int rank) {
  A *a = new A();
  int64_t count = 1;
  for (int i = 0; i < rank; i++) {
    if (shape[i] < 0) { delete a; return set_error(ctx, "negative array dimension"); }
    a->shape[i] = shape[i];
    count *= shape[i];
  }
  a->owned = true;
  if (dalloc(ctx, &a->data, count) != 0) { delete a; return 1; }
  *out = a;
  return 0;
}

template <class A>
int arr_free(futhark_context *ctx, A *a) {
  if (!a) return 0;
  if (a->owned) dfree(ctx, a->data);
  delete a;
  return 0;
}

template <class A, class T>
A *arr_new(futhark_context *ctx, const T *host, const int64_t *shape, int rank, hipMemcpyKind kind) {
  A *a = nullptr;
  if (arr_alloc(ctx, &a, shape, rank) != 0) return nullptr;
  const int64_t count = a->count();
  if (count > 0) {
    hipError_t e = hipMemcpyAsync(a->data, host, sizeof(T) * (size_t)count, kind, ctx->stream);
    if (e == hipSuccess && kind == hipMemcpyHostToDevice) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error(ctx, "copy into new array failed: %s", hipGetErrorString(e));
      arr_free(ctx, a);
      return nullptr;
    }
  }
  return a;
}

template <class A>
A *arr_wrap(futhark_context *ctx, void *dev, const int64_t *shape, int rank) {
  (void)ctx;
  A *a = new A();
  for (int i = 0; i < rank; i++) a->shape[i] = shape[i];
  a->data = (decltype(a->data))dev;
  a->owned = false;
  return a;
}

template <class A, class T>
int arr_values(futhark_context *ctx, A *a, T *host) {
  const int64_t count = a->count();
  if (count > 0) SAMD_HIP(ctx, hipMemcpyAsync(host, a->data, sizeof(T) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

template <class H, class T>
int wrap_body(futhark_context *ctx, H **out, CsrBody<T> *b) {
  (void)ctx;
  H *h = new H();
  h->b = b;
  *out = h;
  return 0;
}

template <class H>
int free_handle(futhark_context *ctx, H *h) {
  if (!h) return 0;
  csr_release(ctx, h->b);
  delete h;
  return 0;
}

// ---- typed entry bodies (TT-independent logic; the macros below only name them)
template <class T, class H, class VA>
int e_sparse(futhark_context *ctx, H **out, int64_t n_major, int64_t n_minor, const futhark_i64_1d *major,
             const futhark_i64_1d *minor, const VA *vs) {
  if (major->shape[0] != minor->shape[0] || major->shape[0] != vs->shape[0])
    return set_error(ctx, "sparse: coordinate arrays differ in length");
  CsrBody<T> *b = nullptr;
  SAMD_TRY(assemble_coo<T>(ctx, n_major, n_minor, major->data, minor->data, vs->data, major->shape[0], &b));
  return wrap_body(ctx, out, b);
}

template <class T, class H>
int e_eye(futhark_context *ctx, H **out, int64_t n, int64_t m) {
  CsrBody<T> *b = nullptr;
  SAMD_TRY(make_eye<T>(ctx, n, m, nullptr, &b));
  return wrap_body(ctx, out, b);
}

template <class T, class H>
int e_zero(futhark_context *ctx, H **out, int64_t n, int64_t m) {
  CsrBody<T> *b = nullptr;
  SAMD_TRY(make_zero<T>(ctx, n, m, &b));
  return wrap_body(ctx, out, b);
}

template <class T, class H, class VA>
int e_diag(futhark_context *ctx, H **out, const VA *v) {
  CsrBody<T> *b = nullptr;
  SAMD_TRY(make_eye<T>(ctx, v->shape[0], v->shape[0], v->data, &b));
  return wrap_body(ctx, out, b);
}

template <class T, class H, class A2>
int e_dense(futhark_context *ctx, A2 **out, const H *a, bool is_csc) {
  // a csc handle stores the transposed record; its dense is [stored m][stored n]
  const int64_t shape[2] = {is_csc ? a->b->m : a->b->n, is_csc ? a->b->n : a->b->m};
  A2 *o = nullptr;
  SAMD_TRY(arr_alloc(ctx, &o, shape, 2));
  if (to_dense<T>(ctx, a->b, is_csc, o->data) != 0) { arr_free(ctx, o); return 1; }
  *out = o;
  return 0;
}

template <class T, class H>
int e_scale(futhark_context *ctx, H **out, T v, const H *a) {
  CsrBody<T> *b = nullptr;
  SAMD_TRY(make_scaled<T>(ctx, v, a->b, &b));
  return wrap_body(ctx, out, b);
}

template <class T, class H, class VA>
int e_coo(futhark_context *ctx, futhark_i64_1d **o0, futhark_i64_1d **o1, VA **o2, const H *a, bool is_csc) {
  const int64_t shape[1] = {a->b->nnz};
  futhark_i64_1d *r = nullptr, *c = nullptr;
  VA *v = nullptr;
  SAMD_TRY(arr_alloc(ctx, &r, shape, 1));
  if (arr_alloc(ctx, &c, shape, 1) != 0) { arr_free(ctx, r); return 1; }
  if (arr_alloc(ctx, &v, shape, 1) != 0) { arr_free(ctx, r); arr_free(ctx, c); return 1; }
  if (to_coo<T>(ctx, a->b, r->data, c->data, v->data) != 0) {
    arr_free(ctx, r); arr_free(ctx, c); arr_free(ctx, v);
    return 1;
  }
  // sc.coo = sr.coo |> map swap (compressed.fut:208-209): free in SoA, swap the handles
  *o0 = is_csc ? c : r;
  *o1 = is_csc ? r : c;
  *o2 = v;
  return 0;
}

template <class T, class H>
int e_addsub(futhark_context *ctx, H **out, const H *a, const H *b, bool sub) {
  CsrBody<T> *r = nullptr;
  SAMD_TRY(add_sub<T>(ctx, a->b, b->b, sub, &r));
  return wrap_body(ctx, out, r);
}

template <class HO, class HI>
int e_relabel(futhark_context *ctx, HO **out, const HI *a) {
  a->b->refs++;
  return wrap_body(ctx, out, a->b);
}

template <class T, class H, class VA>
int e_smvm(futhark_context *ctx, VA **out, const H *a, const VA *x) {
  if (x->shape[0] != a->b->m) return set_error(ctx, "smvm: vector length %lld does not match matrix (%lld)",
                                               (long long)x->shape[0], (long long)a->b->m);
  const int64_t shape[1] = {a->b->n};
  VA *y = nullptr;
  SAMD_TRY(arr_alloc(ctx, &y, shape, 1));
  if (spmv<T>(ctx, a->b, x->data, y->data) != 0) { arr_free(ctx, y); return 1; }
  *out = y;
  return 0;
}

// y[rows[i]] += part[i]; rows are distinct (the caller's contract), so no atomics
template <class T>
__global__ __launch_bounds__(kBlock) void rows_add_kernel(const T *__restrict__ part, const int64_t *__restrict__ rows,
                                                          int64_t count, int64_t ylen, T *__restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= count) return;
  const int64_t r = rows[i];
  if (r >= 0 && r < ylen) y[r] = (T)(y[r] + part[i]);
}

template <class T, class H, class VA>
int e_smvm_into(futhark_context *ctx, VA *y, const H *a, const VA *x, const futhark_i64_1d *rows) {
  if (x->shape[0] != a->b->m) return set_error(ctx, "smvm_into: vector length %lld does not match matrix (%lld)",
                                               (long long)x->shape[0], (long long)a->b->m);
  if (!rows) {
    if (y->shape[0] != a->b->n) return set_error(ctx, "smvm_into: result has %lld elements, matrix has %lld rows",
                                                 (long long)y->shape[0], (long long)a->b->n);
    return spmv<T>(ctx, a->b, x->data, y->data);
  }
  if (rows->shape[0] != a->b->n) return set_error(ctx, "smvm_into: %lld row indices for %lld rows",
                                                  (long long)rows->shape[0], (long long)a->b->n);
  if (a->b->n == 0) return 0;
  bool done = false;  // small records: one launch, no temporary
  SAMD_TRY(spmv_rows_add_small<T>(ctx, a->b, x->data, y->data, rows->data, y->shape[0], &done));
  if (done) return 0;
  Tmp<T> part(ctx);
  SAMD_TRY(part.get(a->b->n));
  SAMD_TRY(spmv<T>(ctx, a->b, x->data, part.p));
  hipLaunchKernelGGL((rows_add_kernel<T>), dim3((unsigned)cdiv(a->b->n, kBlock)), dim3(kBlock), 0, ctx->stream,
                     (const T *)part.p, (const int64_t *)rows->data, a->b->n, y->shape[0], y->data);
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

template <class T, class H, class VA>
int e_smvm_into_noalloc(futhark_context *ctx, VA *y, const H *a, const VA *x) {
  if (x->shape[0] != a->b->m) return set_error(ctx, "smvm_into_noalloc: vector length %lld does not match matrix (%lld)",
                                               (long long)x->shape[0], (long long)a->b->m);
  if (y->shape[0] != a->b->n) return set_error(ctx, "smvm_into_noalloc: result has %lld elements, matrix has %lld rows",
                                               (long long)y->shape[0], (long long)a->b->n);
  size_t room = kStaticScratchBytes;  // test override: pretend the array is smaller (never larger)
  if (ctx->opt_noalloc_scratch_bytes > 0 && (size_t)ctx->opt_noalloc_scratch_bytes < room)
    room = (size_t)ctx->opt_noalloc_scratch_bytes;
  return spmv_noalloc<T>(ctx, a->b, x->data, y->data, ctx->static_scratch, room);
}

template <class T, class H, class A2>
int e_smm(futhark_context *ctx, A2 **out, const H *a, const A2 *B) {
  if (B->shape[0] != a->b->m) return set_error(ctx, "smm: panel has %lld rows, matrix has %lld columns",
                                               (long long)B->shape[0], (long long)a->b->m);
  const int64_t shape[2] = {a->b->n, B->shape[1]};
  A2 *C = nullptr;
  SAMD_TRY(arr_alloc(ctx, &C, shape, 2));
  if (spmm<T>(ctx, a->b, B->data, B->shape[1], C->data) != 0) { arr_free(ctx, C); return 1; }
  *out = C;
  return 0;
}

template <class T, class HO, class HI>
int e_convert(futhark_context *ctx, HO **out, const HI *a) {
  CsrBody<T> *b = nullptr;
  SAMD_TRY(transpose_physical<T>(ctx, a->b, &b));
  return wrap_body(ctx, out, b);
}

template <class T, class HR, class HC>
int e_smsmm(futhark_context *ctx, HR **out, const HR *a, const HC *b) {
  CsrBody<T> *r = nullptr;
  SAMD_TRY(smsmm<T>(ctx, a->b, b->b, &r));
  return wrap_body(ctx, out, r);
}

template <class T, class H, class VA>
int e_fields(futhark_context *ctx, futhark_i64_1d **ro, futhark_i64_1d **ci, VA **va, const H *a) {
  const int64_t sn[1] = {a->b->n}, sz[1] = {a->b->nnz};
  futhark_i64_1d *r = nullptr, *c = nullptr;
  VA *v = nullptr;
  SAMD_TRY(arr_alloc(ctx, &r, sn, 1));
  if (arr_alloc(ctx, &c, sz, 1) != 0) { arr_free(ctx, r); return 1; }
  if (arr_alloc(ctx, &v, sz, 1) != 0) { arr_free(ctx, r); arr_free(ctx, c); return 1; }
  if (to_fields<T>(ctx, a->b, r->data, c->data, v->data) != 0) {
    arr_free(ctx, r); arr_free(ctx, c); arr_free(ctx, v);
    return 1;
  }
  *ro = r; *ci = c; *va = v;
  return 0;
}

template <class T, class H, class VA>
int e_from_fields(futhark_context *ctx, H **out, int64_t n, int64_t m, const futhark_i64_1d *ro,
                  const futhark_i64_1d *ci, const VA *va) {
  if (ro->shape[0] != n) return set_error(ctx, "from_fields: row_off must have length n");
  if (ci->shape[0] != va->shape[0]) return set_error(ctx, "from_fields: col_idx and vals differ in length");
  CsrBody<T> *b = nullptr;
  SAMD_TRY(from_fields<T>(ctx, n, m, ro->data, ci->data, va->data, ci->shape[0], &b));
  return wrap_body(ctx, out, b);
}

// ---- store / restore of the opaque sr / sc values (the serialisers `futhark c --library` generates for
// every opaque type; the byte format is the library's own: store_format.hpp).
template <class T> struct TagOf;
template <> struct TagOf<double> { static const char *name() { return "f64"; } };
template <> struct TagOf<float> { static const char *name() { return "f32"; } };
template <> struct TagOf<int32_t> { static const char *name() { return "i32"; } };

template <class T, class H>
int e_store(futhark_context *ctx, const H *a, void **p, size_t *n) {
  const CsrBody<T> *b = a->b;
  const size_t bytes = sizeof(StoreHeader) + sizeof(int64_t) * (size_t)(b->n + b->nnz) + sizeof(T) * (size_t)b->nnz;
  if (n) *n = bytes;
  if (!p) return 0;  // size query
  Tmp<int64_t> ro(ctx), ci(ctx);
  Tmp<T> va(ctx);
  SAMD_TRY(ro.get(b->n));
  SAMD_TRY(ci.get(b->nnz));
  SAMD_TRY(va.get(b->nnz));
  SAMD_TRY(to_fields<T>(ctx, b, ro.p, ci.p, va.p));
  char *out = (char *)*p;
  const bool mine = out == nullptr;
  if (mine) {
    out = (char *)malloc(bytes);
    if (!out) return set_error(ctx, "store: host allocation of %zu bytes failed", bytes);
  }
  StoreHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "SAMDCSR1", 8);
  strncpy(h.tag, TagOf<T>::name(), sizeof h.tag - 1);
  h.n = b->n; h.m = b->m; h.nnz = b->nnz;
  memcpy(out, &h, sizeof h);
  char *q = out + sizeof h;
  hipError_t e = hipSuccess;
  if (b->n > 0) e = hipMemcpyAsync(q, ro.p, sizeof(int64_t) * (size_t)b->n, hipMemcpyDeviceToHost, ctx->stream);
  q += sizeof(int64_t) * (size_t)b->n;
  if (e == hipSuccess && b->nnz > 0)
    e = hipMemcpyAsync(q, ci.p, sizeof(int64_t) * (size_t)b->nnz, hipMemcpyDeviceToHost, ctx->stream);
  q += sizeof(int64_t) * (size_t)b->nnz;
  if (e == hipSuccess && b->nnz > 0)
    e = hipMemcpyAsync(q, va.p, sizeof(T) * (size_t)b->nnz, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (mine) free(out);
    return set_error(ctx, "store: copy to host failed: %s", hipGetErrorString(e));
  }
  *p = out;
  return 0;
}

template <class T, class H>
H *e_restore(futhark_context *ctx, const void *p) {
  StoreHeader h;
  char msg[160];
  if (store_parse_header(p, (size_t)-1, TagOf<T>::name(), sizeof(T), &h, msg, sizeof msg) != 0) {
    set_error(ctx, "%s", msg);
    return nullptr;
  }
  Tmp<int64_t> ro(ctx), ci(ctx);
  Tmp<T> va(ctx);
  if (ro.get(h.n) != 0 || ci.get(h.nnz) != 0 || va.get(h.nnz) != 0) return nullptr;
  const char *q = (const char *)p + sizeof h;
  hipError_t e = hipSuccess;
  if (h.n > 0) e = hipMemcpyAsync(ro.p, q, sizeof(int64_t) * (size_t)h.n, hipMemcpyHostToDevice, ctx->stream);
  q += sizeof(int64_t) * (size_t)h.n;
  if (e == hipSuccess && h.nnz > 0)
    e = hipMemcpyAsync(ci.p, q, sizeof(int64_t) * (size_t)h.nnz, hipMemcpyHostToDevice, ctx->stream);
  q += sizeof(int64_t) * (size_t)h.nnz;
  if (e == hipSuccess && h.nnz > 0)
    e = hipMemcpyAsync(va.p, q, sizeof(T) * (size_t)h.nnz, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the caller may release p on return
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error(ctx, "restore: copy to device failed: %s", hipGetErrorString(e));
    return nullptr;
  }
  CsrBody<T> *b = nullptr;
  if (from_fields<T>(ctx, h.n, h.m, ro.p, ci.p, va.p, h.nnz, &b) != 0) return nullptr;  // validates the record
  H *out = nullptr;
  wrap_body(ctx, &out, b);
  return out;
}

// ---- mono (S/mono.fut): typed entry bodies
template <class H, class T>
int m_wrap(H **out, MonoBody<T> *b) {
  H *h = new H();
  h->b = b;
  *out = h;
  return 0;
}

template <class H>
int m_free(futhark_context *ctx, H *h) {
  if (!h) return 0;
  mono_release(ctx, h->b);
  delete h;
  return 0;
}

template <class T, class H, class VA>
int m_sparse(futhark_context *ctx, H **out, int64_t n, int64_t m, const futhark_i64_1d *rows,
             const futhark_i64_1d *cols, const VA *vs) {
  if (rows->shape[0] != cols->shape[0] || rows->shape[0] != vs->shape[0])
    return set_error(ctx, "sparse: coordinate arrays differ in length");
  MonoBody<T> *b = nullptr;
  SAMD_TRY(mono_sparse<T>(ctx, n, m, rows->data, cols->data, vs->data, rows->shape[0], &b));
  return m_wrap(out, b);
}

template <class T, class H>
int m_make(futhark_context *ctx, H **out, int64_t n, int64_t m, int mode, const T *diag) {
  MonoBody<T> *b = nullptr;
  SAMD_TRY(mono_make<T>(ctx, n, m, mode, diag, &b));
  return m_wrap(out, b);
}

template <class T, class H, class A2>
int m_dense(futhark_context *ctx, A2 **out, const H *a, bool is_msc) {
  const int64_t shape[2] = {is_msc ? a->b->m : a->b->n, is_msc ? a->b->n : a->b->m};
  A2 *o = nullptr;
  SAMD_TRY(arr_alloc(ctx, &o, shape, 2));
  if (mono_dense<T>(ctx, a->b, is_msc, o->data) != 0) { arr_free(ctx, o); return 1; }
  *out = o;
  return 0;
}

template <class T, class H>
int m_map(futhark_context *ctx, H **out, const H *a, const H *b2, T s, int op) {
  MonoBody<T> *b = nullptr;
  SAMD_TRY(mono_map<T>(ctx, a->b, b2 ? b2->b : nullptr, s, op, &b));
  return m_wrap(out, b);
}

template <class T, class H, class VA>
int m_coo(futhark_context *ctx, futhark_i64_1d **o0, futhark_i64_1d **o1, VA **o2, const H *a, bool is_msc) {
  Tmp<uint32_t> slots(ctx);
  int64_t count = 0;
  SAMD_TRY(mono_coo_count<T>(ctx, a->b, slots, &count));
  const int64_t shape[1] = {count};
  futhark_i64_1d *r = nullptr, *c = nullptr;
  VA *v = nullptr;
  SAMD_TRY(arr_alloc(ctx, &r, shape, 1));
  if (arr_alloc(ctx, &c, shape, 1) != 0) { arr_free(ctx, r); return 1; }
  if (arr_alloc(ctx, &v, shape, 1) != 0) { arr_free(ctx, r); arr_free(ctx, c); return 1; }
  if (mono_coo_emit<T>(ctx, a->b, slots.p, is_msc, r->data, c->data, v->data) != 0) {
    arr_free(ctx, r); arr_free(ctx, c); arr_free(ctx, v);
    return 1;
  }
  *o0 = r; *o1 = c; *o2 = v;
  return 0;
}

template <class HO, class HI>
int m_relabel(HO **out, const HI *a) {
  a->b->refs++;
  return m_wrap(out, a->b);
}

template <class T, class H, class VA>
int m_smvm(futhark_context *ctx, VA **out, const H *a, const VA *x) {
  if (x->shape[0] != a->b->m) return set_error(ctx, "smvm: vector length %lld does not match matrix (%lld)",
                                               (long long)x->shape[0], (long long)a->b->m);
  const int64_t shape[1] = {a->b->n};
  VA *y = nullptr;
  SAMD_TRY(arr_alloc(ctx, &y, shape, 1));
  if (mono_smvm<T>(ctx, a->b, x->data, y->data) != 0) { arr_free(ctx, y); return 1; }
  *out = y;
  return 0;
}

template <class T, class H, class VA>
int m_vsmm(futhark_context *ctx, VA **out, const VA *v, const H *a) {
  if (v->shape[0] != a->b->n) return set_error(ctx, "vsmm: vector length %lld does not match matrix (%lld)",
                                               (long long)v->shape[0], (long long)a->b->n);
  const int64_t room[1] = {a->b->m + 1};  // + the bin of out-of-range columns, not part of the result
  VA *y = nullptr;
  SAMD_TRY(arr_alloc(ctx, &y, room, 1));
  if (mono_vsmm<T>(ctx, a->b, v->data, y->data) != 0) { arr_free(ctx, y); return 1; }
  y->shape[0] = a->b->m;
  *out = y;
  return 0;
}

template <class T, class H, class A2>
int m_dmsmm(futhark_context *ctx, A2 **out, const A2 *d, const H *s) {
  if (d->shape[1] != s->b->n) return set_error(ctx, "dmsmm: dense operand has %lld columns, matrix has %lld rows",
                                               (long long)d->shape[1], (long long)s->b->n);
  const int64_t shape[2] = {d->shape[0], s->b->m};
  A2 *o = nullptr;
  SAMD_TRY(arr_alloc(ctx, &o, shape, 2));
  if (mono_dmsmm<T>(ctx, d->data, d->shape[0], s->b, o->data) != 0) { arr_free(ctx, o); return 1; }
  *out = o;
  return 0;
}

template <class T, class H, class VA>
int m_fields(futhark_context *ctx, futhark_i64_1d **ci, VA **va, const H *a) {
  const int64_t sn[1] = {a->b->n};
  futhark_i64_1d *c = nullptr;
  VA *v = nullptr;
  SAMD_TRY(arr_alloc(ctx, &c, sn, 1));
  if (arr_alloc(ctx, &v, sn, 1) != 0) { arr_free(ctx, c); return 1; }
  if (a->b->n > 0) {
    SAMD_HIP(ctx, hipMemcpyAsync(c->data, a->b->col, sizeof(int64_t) * (size_t)a->b->n, hipMemcpyDeviceToDevice, ctx->stream));
    SAMD_HIP(ctx, hipMemcpyAsync(v->data, a->b->vals, sizeof(T) * (size_t)a->b->n, hipMemcpyDeviceToDevice, ctx->stream));
  }
  *ci = c; *va = v;
  return 0;
}

}  // namespace

#include "group.hpp"

extern "C" {

// ------------------------------------------------------------------ context
struct futhark_context_config *futhark_context_config_new(void) { return new futhark_context_config(); }
void futhark_context_config_free(struct futhark_context_config *cfg) { delete cfg; }
void futhark_context_config_set_debugging(struct futhark_context_config *cfg, int flag) { cfg->debugging = flag; }
void futhark_context_config_set_profiling(struct futhark_context_config *cfg, int flag) { cfg->profiling = flag; }
void futhark_context_config_set_logging(struct futhark_context_config *cfg, int flag) { cfg->logging = flag; }
// Generated by the multicore back end (worker pool size).  Accepted and recorded; a GPU context has no
// worker pool, so it changes nothing -- a harness written for `futhark multicore --library` links and runs.
void futhark_context_config_set_num_threads(struct futhark_context_config *cfg, int n) { cfg->num_threads = n; }
void futhark_context_config_set_device(struct futhark_context_config *cfg, const char *s) {
  if (!s) return;
  if (s[0] == '#') s++;
  cfg->device = atoi(s);
}

struct futhark_context *futhark_context_new(struct futhark_context_config *cfg) {
  futhark_context *ctx = new futhark_context();
  ctx->device = cfg ? cfg->device : 0;
  ctx->logging = cfg ? cfg->logging : 0;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    set_error(ctx, "no HIP device available (%s); libsparse_amd has no CPU path",
              e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return ctx;  // futhark convention: a context is always returned; the error is pending
  }
  if (ctx->device < 0 || ctx->device >= ndev) {
    set_error(ctx, "device #%d out of range (%d present)", ctx->device, ndev);
    return ctx;
  }
  hipDeviceProp_t prop;
  if (hipSetDevice(ctx->device) != hipSuccess || hipGetDeviceProperties(&prop, ctx->device) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipHostMalloc((void **)&ctx->h_small, sizeof(int64_t) * 16, hipHostMallocDefault) != hipSuccess ||
      hipMalloc((void **)&ctx->d_small, sizeof(int64_t) * 16) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_pre, hipEventDisableTiming) != hipSuccess) {
    set_error(ctx, "context initialisation failed: %s", hipGetErrorString(hipGetLastError()));
    return ctx;
  }
  {
    void *sp = nullptr;
    const hipError_t es = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));  // this device's copy (the device is current)
    if (es != hipSuccess) {
      (void)hipGetLastError();
      set_error(ctx, "context initialisation failed: hipGetSymbolAddress(scratch): %s", hipGetErrorString(es));
      return ctx;
    }
    ctx->static_scratch = sp;
  }
  ctx->stream = ctx->own_stream;
  ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (ctx->logging) fprintf(stderr, "[sparse_amd] device #%d %s (%s), %d CUs\n", ctx->device, prop.name,
                            prop.gcnArchName, ctx->num_cu);
  return ctx;
}

void futhark_context_free(struct futhark_context *ctx) {
  if (!ctx) return;
  if (ctx->own_stream) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->pool.clear();
    (void)hipFree(ctx->d_small);
    (void)hipHostFree(ctx->h_small);
    if (ctx->ev_pre) (void)hipEventDestroy(ctx->ev_pre);
    (void)hipStreamDestroy(ctx->own_stream);
  }
  delete ctx;
}

int futhark_context_sync(struct futhark_context *ctx) {
  ENTER(ctx);
  if (ctx->has_err) return 1;
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

char *futhark_context_get_error(struct futhark_context *ctx) {
  std::lock_guard<std::mutex> g(ctx->mu);
  if (!ctx->has_err) return nullptr;
  char *s = strdup(ctx->err.c_str());
  ctx->has_err = false;
  ctx->err.clear();
  return s;
}

int futhark_context_clear_caches(struct futhark_context *ctx) {
  ENTER(ctx);
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->pool.clear();
  return 0;
}

char *futhark_context_report(struct futhark_context *ctx) {
  std::lock_guard<std::mutex> g(ctx->mu);
  char buf[512];
  snprintf(buf, sizeof buf,
           "device: #%d (%d CUs)\nAPI calls: %lld\ndevice memory held: %zu bytes (peak %zu, cached %zu, live blocks %zu)\n",
           ctx->device, ctx->num_cu, (long long)ctx->n_entries, ctx->pool.total(), ctx->pool.peak(),
           ctx->pool.cached(), ctx->pool.live_blocks());
  return strdup(buf);
}

int sparse_amd_context_set_stream(struct futhark_context *ctx, void *hip_stream) {
  ENTER(ctx);
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = (hipStream_t)hip_stream;  // NULL is the HIP null stream (torch's default stream)
  return 0;
}
void *sparse_amd_context_get_stream(struct futhark_context *ctx) { return (void *)ctx->stream; }

int sparse_amd_context_set_option(struct futhark_context *ctx, const char *key, int64_t value) {
  std::lock_guard<std::mutex> g(ctx->mu);
  if (!strcmp(key, "spmv_algo")) ctx->opt_spmv_algo = value;
  else if (!strcmp(key, "spmv_lanes")) ctx->opt_spmv_lanes = value;
  else if (!strcmp(key, "wide_col")) ctx->opt_wide_col = value;
  else if (!strcmp(key, "spmm_algo")) ctx->opt_spmm_algo = value;
  else if (!strcmp(key, "spmv_cblk")) ctx->opt_spmv_cblk = value;
  else if (!strcmp(key, "sort_algo")) ctx->opt_sort_algo = value;
  else if (!strcmp(key, "sort_indirect")) ctx->opt_sort_indirect = value;
  else if (!strcmp(key, "sort_msd")) ctx->opt_sort_msd = value;
  else if (!strcmp(key, "sort_inject")) ctx->opt_sort_inject = value;
  else if (!strcmp(key, "add_sort")) ctx->opt_add_sort = value;
  else if (!strcmp(key, "fold_emit")) ctx->opt_fold_emit = value;
  else if (!strcmp(key, "sort_msd_chunk")) ctx->opt_sort_msd_chunk = value;
  else if (!strcmp(key, "sort_pairs")) ctx->opt_sort_pairs = value;
  else if (!strcmp(key, "spmv_reorder")) ctx->opt_spmv_reorder = value;
  else if (!strcmp(key, "spmv_col16")) ctx->opt_spmv_col16 = value;
  else if (!strcmp(key, "spmv_eager")) ctx->opt_spmv_eager = value;
  else if (!strcmp(key, "spmv_pb")) ctx->opt_spmv_pb = value;
  else if (!strcmp(key, "spmv_pb_rows")) ctx->opt_spmv_pb_rows = value;
  else if (!strcmp(key, "spmv_pb_lines")) ctx->opt_spmv_pb_lines = value;
  else if (!strcmp(key, "spmv_pb_entries")) ctx->opt_spmv_pb_entries = value;
  else if (!strcmp(key, "spmv_pb_fold")) ctx->opt_spmv_pb_fold = value;
  else if (!strcmp(key, "spmv_pb_oround")) ctx->opt_spmv_pb_oround = value;
  else if (!strcmp(key, "spmv_pb_xcopy")) ctx->opt_spmv_pb_xcopy = value;
  else if (!strcmp(key, "spmv_pb_shares")) ctx->opt_spmv_pb_shares = value;
  else if (!strcmp(key, "spmv_pb_deal")) ctx->opt_spmv_pb_deal = value;
  else if (!strcmp(key, "ablate")) ctx->opt_ablate = value;
  else if (!strcmp(key, "noalloc_scratch_bytes")) ctx->opt_noalloc_scratch_bytes = value;
  else return set_error(ctx, "unknown option '%s'", key);
  return 0;
}
int64_t sparse_amd_context_get_option(struct futhark_context *ctx, const char *key) {
  std::lock_guard<std::mutex> g(ctx->mu);
  if (!strcmp(key, "spmv_algo")) return ctx->opt_spmv_algo;
  if (!strcmp(key, "spmv_lanes")) return ctx->opt_spmv_lanes;
  if (!strcmp(key, "wide_col")) return ctx->opt_wide_col;
  if (!strcmp(key, "spmm_algo")) return ctx->opt_spmm_algo;
  if (!strcmp(key, "spmv_cblk")) return ctx->opt_spmv_cblk;
  if (!strcmp(key, "sort_algo")) return ctx->opt_sort_algo;
  if (!strcmp(key, "sort_msd")) return ctx->opt_sort_msd;
  if (!strcmp(key, "sort_last_msd")) return ctx->last_sort_msd;
  if (!strcmp(key, "sort_last_long_runs")) return ctx->last_long_runs;
  if (!strcmp(key, "spmv_reorder")) return ctx->opt_spmv_reorder;
  if (!strcmp(key, "spmv_eager")) return ctx->opt_spmv_eager;
  if (!strcmp(key, "spmv_pb")) return ctx->opt_spmv_pb;
  if (!strcmp(key, "spmv_pb_fold")) return ctx->opt_spmv_pb_fold;
  if (!strcmp(key, "spmv_last_pb_fold")) return ctx->last_pb_fold;
  if (!strcmp(key, "spmv_last_pb_outputs")) return ctx->last_pb_outputs;
  if (!strcmp(key, "num_cu")) return ctx->num_cu;
  if (!strcmp(key, "spmv_last_path")) return ctx->last_spmv_path;
  if (!strcmp(key, "spmv_last_col_bytes")) return ctx->last_spmv_col_bytes;
  if (!strcmp(key, "spmv_col16")) return ctx->opt_spmv_col16;
  return -1;
}
const char *sparse_amd_version(void) { return "sparse_amd 0.1 (gfx950)"; }

int sparse_amd_context_measure_bandwidth(struct futhark_context *ctx, int64_t read_bytes_per_group,
                                         int64_t write_bytes_per_group, int64_t groups, double *gb_per_s, double *us) {
  ENTER(ctx);
  return measure_bandwidth(ctx, read_bytes_per_group, write_bytes_per_group, groups, gb_per_s, us);
}

// ------------------------------------------------------------------- arrays
#define SAMD_ARRAY_1D(NAME, CT)                                                                              \
  struct futhark_##NAME##_1d *futhark_new_##NAME##_1d(struct futhark_context *ctx, const CT *data, int64_t d0) { \
    ENTER(ctx);                                                                                            \
    const int64_t shape[1] = {d0};                                                                           \
    return arr_new<futhark_##NAME##_1d, CT>(ctx, data, shape, 1, hipMemcpyHostToDevice);                     \
  }                                                                                                          \
  struct futhark_##NAME##_1d *futhark_new_raw_##NAME##_1d(struct futhark_context *ctx, const void *data,     \
                                                          int64_t d0) {                                      \
    ENTER(ctx);                                                                                            \
    const int64_t shape[1] = {d0};                                                                           \
    return arr_new<futhark_##NAME##_1d, CT>(ctx, (const CT *)data, shape, 1, hipMemcpyDeviceToDevice);       \
  }                                                                                                          \
  struct futhark_##NAME##_1d *sparse_amd_wrap_##NAME##_1d(struct futhark_context *ctx, void *data, int64_t d0) { \
    ENTER(ctx);                                                                                            \
    const int64_t shape[1] = {d0};                                                                           \
    return arr_wrap<futhark_##NAME##_1d>(ctx, data, shape, 1);                                               \
  }                                                                                                          \
  int futhark_free_##NAME##_1d(struct futhark_context *ctx, struct futhark_##NAME##_1d *arr) {               \
    ENTER(ctx);                                                                                            \
    return arr_free(ctx, arr);                                                                               \
  }                                                                                                          \
  int futhark_values_##NAME##_1d(struct futhark_context *ctx, struct futhark_##NAME##_1d *arr, CT *data) {   \
    ENTER(ctx);                                                                                            \
    return arr_values(ctx, arr, data);                                                                       \
  }                                                                                                          \
  void *futhark_values_raw_##NAME##_1d(struct futhark_context *ctx, struct futhark_##NAME##_1d *arr) {       \
    (void)ctx;                                                                                               \
    return (void *)arr->data;                                                                                \
  }                                                                                                          \
  const int64_t *futhark_shape_##NAME##_1d(struct futhark_context *ctx, struct futhark_##NAME##_1d *arr) {   \
    (void)ctx;                                                                                               \
    return arr->shape;                                                                                       \
  }

#define SAMD_ARRAY_2D(NAME, CT)                                                                              \
  struct futhark_##NAME##_2d *futhark_new_##NAME##_2d(struct futhark_context *ctx, const CT *data, int64_t d0, \
                                                      int64_t d1) {                                          \
    ENTER(ctx);                                                                                            \
    const int64_t shape[2] = {d0, d1};                                                                       \
    return arr_new<futhark_##NAME##_2d, CT>(ctx, data, shape, 2, hipMemcpyHostToDevice);                     \
  }                                                                                                          \
  struct futhark_##NAME##_2d *futhark_new_raw_##NAME##_2d(struct futhark_context *ctx, const void *data,     \
                                                          int64_t d0, int64_t d1) {                          \
    ENTER(ctx);                                                                                            \
    const int64_t shape[2] = {d0, d1};                                                                       \
    return arr_new<futhark_##NAME##_2d, CT>(ctx, (const CT *)data, shape, 2, hipMemcpyDeviceToDevice);       \
  }                                                                                                          \
  struct futhark_##NAME##_2d *sparse_amd_wrap_##NAME##_2d(struct futhark_context *ctx, void *data, int64_t d0, \
                                                          int64_t d1) {                                      \
    ENTER(ctx);                                                                                            \
    const int64_t shape[2] = {d0, d1};                                                                       \
    return arr_wrap<futhark_##NAME##_2d>(ctx, data, shape, 2);                                               \
  }                                                                                                          \
  int futhark_free_##NAME##_2d(struct futhark_context *ctx, struct futhark_##NAME##_2d *arr) {               \
    ENTER(ctx);                                                                                            \
    return arr_free(ctx, arr);                                                                               \
  }                                                                                                          \
  int futhark_values_##NAME##_2d(struct futhark_context *ctx, struct futhark_##NAME##_2d *arr, CT *data) {   \
    ENTER(ctx);                                                                                            \
    return arr_values(ctx, arr, data);                                                                       \
  }                                                                                                          \
  void *futhark_values_raw_##NAME##_2d(struct futhark_context *ctx, struct futhark_##NAME##_2d *arr) {       \
    (void)ctx;                                                                                               \
    return (void *)arr->data;                                                                                \
  }                                                                                                          \
  const int64_t *futhark_shape_##NAME##_2d(struct futhark_context *ctx, struct futhark_##NAME##_2d *arr) {   \
    (void)ctx;                                                                                               \
    return arr->shape;                                                                                       \
  }

SAMD_ARRAY_1D(i64, int64_t)
SAMD_ARRAY_1D(i32, int32_t)
SAMD_ARRAY_1D(f32, float)
SAMD_ARRAY_1D(f64, double)
SAMD_ARRAY_2D(i32, int32_t)
SAMD_ARRAY_2D(f32, float)
SAMD_ARRAY_2D(f64, double)

// -------------------------------------------------------- typed matrix entries
#define CSR(TT) struct futhark_opaque_csr_##TT
#define CSC(TT) struct futhark_opaque_csc_##TT
#define A1(TT) struct futhark_##TT##_1d
#define A2(TT) struct futhark_##TT##_2d
#define CTX struct futhark_context *ctx
#define I64A const struct futhark_i64_1d

#define SAMD_MATRIX(TT, CT)                                                                                   \
  int futhark_free_opaque_csr_##TT(CTX, CSR(TT) * o) { ENTER(ctx); return free_handle(ctx, o); }            \
  int futhark_free_opaque_csc_##TT(CTX, CSC(TT) * o) { ENTER(ctx); return free_handle(ctx, o); }            \
  int futhark_store_opaque_csr_##TT(CTX, const CSR(TT) * obj, void **p, size_t *n) {                          \
    ENTER(ctx);                                                                                             \
    return e_store<CT>(ctx, obj, p, n);                                                                       \
  }                                                                                                           \
  int futhark_store_opaque_csc_##TT(CTX, const CSC(TT) * obj, void **p, size_t *n) {                          \
    ENTER(ctx);                                                                                             \
    return e_store<CT>(ctx, obj, p, n);                                                                       \
  }                                                                                                           \
  CSR(TT) * futhark_restore_opaque_csr_##TT(CTX, const void *p) {                                             \
    ENTER(ctx);                                                                                             \
    return e_restore<CT, CSR(TT)>(ctx, p);                                                                    \
  }                                                                                                           \
  CSC(TT) * futhark_restore_opaque_csc_##TT(CTX, const void *p) {                                             \
    ENTER(ctx);                                                                                             \
    return e_restore<CT, CSC(TT)>(ctx, p);                                                                    \
  }                                                                                                           \
  int futhark_entry_csr_##TT##_sparse(CTX, CSR(TT) * *out0, const int64_t n, const int64_t m, I64A *rs,       \
                                      I64A *cs, const A1(TT) * vs) {                                          \
    ENTER(ctx);                                                                                             \
    return e_sparse<CT>(ctx, out0, n, m, rs, cs, vs);                                                         \
  }                                                                                                           \
  int futhark_entry_csc_##TT##_sparse(CTX, CSC(TT) * *out0, const int64_t n, const int64_t m, I64A *rs,       \
                                      I64A *cs, const A1(TT) * vs) {                                          \
    ENTER(ctx); /* sr.sparse m n (swap coo) */                                                              \
    return e_sparse<CT>(ctx, out0, m, n, cs, rs, vs);                                                         \
  }                                                                                                           \
  int futhark_entry_csr_##TT##_zero(CTX, CSR(TT) * *out0, const int64_t n, const int64_t m) {                 \
    ENTER(ctx);                                                                                             \
    return e_zero<CT>(ctx, out0, n, m);                                                                       \
  }                                                                                                           \
  int futhark_entry_csc_##TT##_zero(CTX, CSC(TT) * *out0, const int64_t n, const int64_t m) {                 \
    ENTER(ctx);                                                                                             \
    return e_zero<CT>(ctx, out0, m, n);                                                                       \
  }                                                                                                           \
  int futhark_entry_csr_##TT##_eye(CTX, CSR(TT) * *out0, const int64_t n, const int64_t m) {                  \
    ENTER(ctx);                                                                                             \
    return e_eye<CT>(ctx, out0, n, m);                                                                        \
  }                                                                                                           \
  int futhark_entry_csc_##TT##_eye(CTX, CSC(TT) * *out0, const int64_t n, const int64_t m) {                  \
    ENTER(ctx);                                                                                             \
    return e_eye<CT>(ctx, out0, m, n);                                                                        \
  }                                                                                                           \
  int futhark_entry_csr_##TT##_diag(CTX, CSR(TT) * *out0, const A1(TT) * v) {                                 \
    ENTER(ctx);                                                                                             \
    return e_diag<CT>(ctx, out0, v);                                                                          \
  }                                                                                                           \
  int futhark_entry_csc_##TT##_diag(CTX, CSC(TT) * *out0, const A1(TT) * v) {                                 \
    ENTER(ctx);                                                                                             \
    return e_diag<CT>(ctx, out0, v);                                                                          \
  }                                                                                                           \
  int futhark_entry_csr_##TT##_dense(CTX, A2(TT) * *out0, const CSR(TT) * a) {                                \
    ENTER(ctx);                                                                                             \
    return e_dense<CT>(ctx, out0, a, false);                                                                  \
  }                                                                                                           \
  int futhark_entry_csc_##TT##_dense(CTX, A2(TT) * *out0, const CSC(TT) * a) {                                \
    ENTER(ctx);                                                                                             \
    return e_dense<CT>(ctx, out0, a, true);                                                                   \
  }                                                                                                           \
  int futhark_entry_csr_##TT##_scale(CTX, CSR(TT) * *out0, const CT v, const CSR(TT) * a) {                   \
    ENTER(ctx);                                                                                             \
    return e_scale<CT>(ctx, out0, v, a);                                                                      \
  }                                                                                                           \
  int futhark_entry_csc_##TT##_scale(CTX, CSC(TT) * *out0, const CT v, const CSC(TT) * a) {                   \
    ENTER(ctx);                                                                                             \
    return e_scale<CT>(ctx, out0, v, a);                                                                      \
  }                                                                                                           \
  int futhark_entry_csr_##TT##_nnz(CTX, int64_t *out0, const CSR(TT) * a) {                                   \
    ENTER(ctx);                                                                                             \
    return count_nonzero<CT>(ctx, a->b, out0);                                                                \
  }                                                                                                           \
  int futhark_entry_csc_##TT##_nnz(CTX, int64_t *out0, const CSC(TT) * a) {                                   \
    ENTER(ctx);                                                                                             \
    return count_nonzero<CT>(ctx, a->b, out0);                                                                \
  }                                                                                                           \
  int futhark_entry_csr_##TT##_coo(CTX, struct futhark_i64_1d **out0, struct futhark_i64_1d **out1,           \
                                   A1(TT) * *out2, const CSR(TT) * a) {                                       \
    ENTER(ctx);                                                                                             \
    return e_coo<CT>(ctx, out0, out1, out2, a, false);                                                        \
  }                                                                                                           \
  int futhark_entry_csc_##TT##_coo(CTX, struct futhark_i64_1d **out0, struct futhark_i64_1d **out1,           \
                                   A1(TT) * *out2, const CSC(TT) * a) {                                       \
    ENTER(ctx);                                                                                             \
    return e_coo<CT>(ctx, out0, out1, out2, a, true);                                                         \
  }                                                                                                           \
  int futhark_entry_csr_##TT##_add(CTX, CSR(TT) * *out0, const CSR(TT) * a, const CSR(TT) * b) {              \
    ENTER(ctx);                                                                                             \
    return e_addsub<CT>(ctx, out0, a, b, false);                                                              \
  }                                                                                                           \
  int futhark_entry_csr_##TT##_sub(CTX, CSR(TT) * *out0, const CSR(TT) * a, const CSR(TT) * b) {              \
    ENTER(ctx);                                                                                             \
    return e_addsub<CT>(ctx, out0, a, b, true);                                                               \
  }                                                                                                           \
  int futhark_entry_csc_##TT##_add(CTX, CSC(TT) * *out0, const CSC(TT) * a, const CSC(TT) * b) {              \
    ENTER(ctx);                                                                                             \
    return e_addsub<CT>(ctx, out0, a, b, false);                                                              \
  }                                                                                                           \
  int futhark_entry_csc_##TT##_sub(CTX, CSC(TT) * *out0, const CSC(TT) * a, const CSC(TT) * b) {              \
    ENTER(ctx);                                                                                             \
    return e_addsub<CT>(ctx, out0, a, b, true);                                                               \
  }                                                                                                           \
  int futhark_entry_csr_##TT##_transpose(CTX, CSC(TT) * *out0, const CSR(TT) * a) {                           \
    ENTER(ctx);                                                                                             \
    return e_relabel(ctx, out0, a);                                                                           \
  }                                                                                                           \
  int futhark_entry_csc_##TT##_transpose(CTX, CSR(TT) * *out0, const CSC(TT) * a) {                           \
    ENTER(ctx);                                                                                             \
    return e_relabel(ctx, out0, a);                                                                           \
  }                                                                                                           \
  int futhark_entry_csr_##TT##_smvm(CTX, A1(TT) * *out0, const CSR(TT) * a, const A1(TT) * x) {               \
    ENTER(ctx);                                                                                             \
    return e_smvm<CT>(ctx, out0, a, x);                                                                       \
  }                                                                                                           \
  int futhark_entry_csc_##TT##_vsmm(CTX, A1(TT) * *out0, const A1(TT) * a, const CSC(TT) * b) {               \
    ENTER(ctx); /* sr.smvm (transpose b) a */                                                               \
    return e_smvm<CT>(ctx, out0, b, a);                                                                       \
  }                                                                                                           \
  int futhark_entry_csr_##TT##_smm(CTX, A2(TT) * *out0, const CSR(TT) * a, const A2(TT) * b) {                \
    ENTER(ctx);                                                                                             \
    return e_smm<CT>(ctx, out0, a, b);                                                                        \
  }                                                                                                           \
  int futhark_entry_csr_##TT##_to_csc(CTX, CSC(TT) * *out0, const CSR(TT) * a) {                              \
    ENTER(ctx);                                                                                             \
    return e_convert<CT>(ctx, out0, a);                                                                       \
  }                                                                                                           \
  int futhark_entry_csc_##TT##_to_csr(CTX, CSR(TT) * *out0, const CSC(TT) * a) {                              \
    ENTER(ctx);                                                                                             \
    return e_convert<CT>(ctx, out0, a);                                                                       \
  }                                                                                                           \
  int futhark_entry_smsmm_##TT(CTX, CSR(TT) * *out0, const CSR(TT) * a, const CSC(TT) * b) {                  \
    ENTER(ctx);                                                                                             \
    return e_smsmm<CT>(ctx, out0, a, b);                                                                      \
  }                                                                                                           \
  int sparse_amd_csr_##TT##_dims(CTX, const CSR(TT) * a, int64_t *n, int64_t *m, int64_t *stored) {           \
    (void)ctx;                                                                                                \
    *n = a->b->n; *m = a->b->m; *stored = a->b->nnz;                                                          \
    return 0;                                                                                                 \
  }                                                                                                           \
  int sparse_amd_csc_##TT##_dims(CTX, const CSC(TT) * a, int64_t *n, int64_t *m, int64_t *stored) {           \
    (void)ctx;                                                                                                \
    *n = a->b->n; *m = a->b->m; *stored = a->b->nnz;                                                          \
    return 0;                                                                                                 \
  }                                                                                                           \
  int sparse_amd_csr_##TT##_fields(CTX, struct futhark_i64_1d **row_off, struct futhark_i64_1d **col_idx,     \
                                   A1(TT) * *vals, const CSR(TT) * a) {                                       \
    ENTER(ctx);                                                                                             \
    return e_fields<CT>(ctx, row_off, col_idx, vals, a);                                                      \
  }                                                                                                           \
  int sparse_amd_csc_##TT##_fields(CTX, struct futhark_i64_1d **row_off, struct futhark_i64_1d **col_idx,     \
                                   A1(TT) * *vals, const CSC(TT) * a) {                                       \
    ENTER(ctx);                                                                                             \
    return e_fields<CT>(ctx, row_off, col_idx, vals, a);                                                      \
  }                                                                                                           \
  int sparse_amd_csr_##TT##_from_fields(CTX, CSR(TT) * *out0, const int64_t n, const int64_t m,               \
                                        I64A *row_off, I64A *col_idx, const A1(TT) * vals) {                  \
    ENTER(ctx);                                                                                             \
    return e_from_fields<CT>(ctx, out0, n, m, row_off, col_idx, vals);                                        \
  }                                                                                                           \
  int sparse_amd_csr_##TT##_smvm_into(CTX, A1(TT) * y, const CSR(TT) * a, const A1(TT) * x, I64A *rows) {     \
    ENTER(ctx);                                                                                             \
    return e_smvm_into<CT>(ctx, y, a, x, rows);                                                               \
  }                                                                                                           \
  int sparse_amd_csr_##TT##_smvm_into_noalloc(CTX, A1(TT) * y, const CSR(TT) * a, const A1(TT) * x) {         \
    ENTER(ctx);                                                                                             \
    return e_smvm_into_noalloc<CT>(ctx, y, a, x);                                                             \
  }                                                                                                           \
  int sparse_amd_csr_##TT##_prepare(CTX, const CSR(TT) * a, const int64_t expected_products) {                \
    ENTER(ctx);                                                                                             \
    return spmv_prepare<CT>(ctx, a->b, expected_products);                                                    \
  }                                                                                                           \
  int sparse_amd_csc_##TT##_prepare(CTX, const CSC(TT) * a, const int64_t expected_products) {                \
    ENTER(ctx);                                                                                             \
    return spmv_prepare<CT>(ctx, a->b, expected_products);                                                    \
  }

SAMD_MATRIX(f64, double)
SAMD_MATRIX(f32, float)
SAMD_MATRIX(i32, int32_t)

// ------------------------------------ the reference's own entries (module i32)
// Each body is the composition written in compressed_test.fut at the cited lines.
typedef struct futhark_opaque_csr_i32 csr_i32;
typedef struct futhark_opaque_csc_i32 csc_i32;
typedef struct futhark_i32_1d arr_i32;
typedef struct futhark_i32_2d mat_i32;
typedef const struct futhark_i64_1d carr_i64;

// compressed_test.fut:30-31  csr.eye n m |> csr.dense
int futhark_entry_test_csr_eye(CTX, mat_i32 **out0, const int64_t in0, const int64_t in1) {
  ENTER(ctx);
  csr_i32 *a = nullptr;
  SAMD_TRY(e_eye<int32_t>(ctx, &a, in0, in1));
  int rc = e_dense<int32_t>(ctx, out0, a, false);
  free_handle(ctx, a);
  return rc;
}
// :44-45  csr.sparse n m (zip3 xs ys vs) |> csr.dense
int futhark_entry_test_csr_sparse(CTX, mat_i32 **out0, const int64_t in0, const int64_t in1, carr_i64 *in2,
                                  carr_i64 *in3, const arr_i32 *in4) {
  ENTER(ctx);
  csr_i32 *a = nullptr;
  SAMD_TRY(e_sparse<int32_t>(ctx, &a, in0, in1, in2, in3, in4));
  int rc = e_dense<int32_t>(ctx, out0, a, false);
  free_handle(ctx, a);
  return rc;
}
// :57-59  csr.smvm (csr.sparse n m ...) v
int futhark_entry_test_csr_smvm(CTX, arr_i32 **out0, const int64_t in0, const int64_t in1, carr_i64 *in2,
                                carr_i64 *in3, const arr_i32 *in4, const arr_i32 *in5) {
  ENTER(ctx);
  csr_i32 *a = nullptr;
  SAMD_TRY(e_sparse<int32_t>(ctx, &a, in0, in1, in2, in3, in4));
  int rc = e_smvm<int32_t>(ctx, out0, a, in5);
  free_handle(ctx, a);
  return rc;
}
// :68-69
int futhark_entry_test_csr_nnz(CTX, int64_t *out0, const int64_t in0, const int64_t in1, carr_i64 *in2,
                               carr_i64 *in3, const arr_i32 *in4) {
  ENTER(ctx);
  csr_i32 *a = nullptr;
  SAMD_TRY(e_sparse<int32_t>(ctx, &a, in0, in1, in2, in3, in4));
  int rc = count_nonzero<int32_t>(ctx, a->b, out0);
  free_handle(ctx, a);
  return rc;
}
// :76-78
int futhark_entry_test_csr_coo(CTX, struct futhark_i64_1d **out0, struct futhark_i64_1d **out1, arr_i32 **out2,
                               const int64_t in0, const int64_t in1, carr_i64 *in2, carr_i64 *in3,
                               const arr_i32 *in4) {
  ENTER(ctx);
  csr_i32 *a = nullptr;
  SAMD_TRY(e_sparse<int32_t>(ctx, &a, in0, in1, in2, in3, in4));
  int rc = e_coo<int32_t>(ctx, out0, out1, out2, a, false);
  free_handle(ctx, a);
  return rc;
}
// :104-105  csc.eye n m |> csc.dense
int futhark_entry_test_csc_eye(CTX, mat_i32 **out0, const int64_t in0, const int64_t in1) {
  ENTER(ctx);
  csc_i32 *a = nullptr;
  SAMD_TRY(e_eye<int32_t>(ctx, &a, in1, in0));
  int rc = e_dense<int32_t>(ctx, out0, a, true);
  free_handle(ctx, a);
  return rc;
}
// :118-119
int futhark_entry_test_csc_sparse(CTX, mat_i32 **out0, const int64_t in0, const int64_t in1, carr_i64 *in2,
                                  carr_i64 *in3, const arr_i32 *in4) {
  ENTER(ctx);
  csc_i32 *a = nullptr;
  SAMD_TRY(e_sparse<int32_t>(ctx, &a, in1, in0, in3, in2, in4));
  int rc = e_dense<int32_t>(ctx, out0, a, true);
  free_handle(ctx, a);
  return rc;
}
// :128-129
int futhark_entry_test_csc_nnz(CTX, int64_t *out0, const int64_t in0, const int64_t in1, carr_i64 *in2,
                               carr_i64 *in3, const arr_i32 *in4) {
  ENTER(ctx);
  csc_i32 *a = nullptr;
  SAMD_TRY(e_sparse<int32_t>(ctx, &a, in1, in0, in3, in2, in4));
  int rc = count_nonzero<int32_t>(ctx, a->b, out0);
  free_handle(ctx, a);
  return rc;
}
// :136-138
int futhark_entry_test_csc_coo(CTX, struct futhark_i64_1d **out0, struct futhark_i64_1d **out1, arr_i32 **out2,
                               const int64_t in0, const int64_t in1, carr_i64 *in2, carr_i64 *in3,
                               const arr_i32 *in4) {
  ENTER(ctx);
  csc_i32 *a = nullptr;
  SAMD_TRY(e_sparse<int32_t>(ctx, &a, in1, in0, in3, in2, in4));
  int rc = e_coo<int32_t>(ctx, out0, out1, out2, a, true);
  free_handle(ctx, a);
  return rc;
}
// :145-147  csr.sparse n m ... |> csr.transpose |> csc.dense
int futhark_entry_test_csr_transpose(CTX, mat_i32 **out0, const int64_t in0, const int64_t in1, carr_i64 *in2,
                                     carr_i64 *in3, const arr_i32 *in4) {
  ENTER(ctx);
  csr_i32 *a = nullptr;
  SAMD_TRY(e_sparse<int32_t>(ctx, &a, in0, in1, in2, in3, in4));
  csc_i32 *t = nullptr;
  e_relabel(ctx, &t, a);
  int rc = e_dense<int32_t>(ctx, out0, t, true);
  free_handle(ctx, t);
  free_handle(ctx, a);
  return rc;
}
// :154-156  csc.sparse n m ... |> csc.transpose |> csr.dense
int futhark_entry_test_csc_transpose(CTX, mat_i32 **out0, const int64_t in0, const int64_t in1, carr_i64 *in2,
                                     carr_i64 *in3, const arr_i32 *in4) {
  ENTER(ctx);
  csc_i32 *a = nullptr;
  SAMD_TRY(e_sparse<int32_t>(ctx, &a, in1, in0, in3, in2, in4));
  csr_i32 *t = nullptr;
  e_relabel(ctx, &t, a);
  int rc = e_dense<int32_t>(ctx, out0, t, false);
  free_handle(ctx, t);
  free_handle(ctx, a);
  return rc;
}
// :172-178
int futhark_entry_test_smsmm(CTX, mat_i32 **out0, const int64_t in0, const int64_t in1, const int64_t in2,
                             carr_i64 *in3, carr_i64 *in4, const arr_i32 *in5, carr_i64 *in6, carr_i64 *in7,
                             const arr_i32 *in8) {
  ENTER(ctx);
  csr_i32 *A = nullptr;
  csc_i32 *B = nullptr;
  csr_i32 *C = nullptr;
  SAMD_TRY(e_sparse<int32_t>(ctx, &A, in0, in1, in3, in4, in5));
  int rc = e_sparse<int32_t>(ctx, &B, in2, in1, in7, in6, in8);  // csc.sparse m k = sr.sparse k m (swap)
  if (rc == 0) rc = e_smsmm<int32_t>(ctx, &C, A, B);
  if (rc == 0) rc = e_dense<int32_t>(ctx, out0, C, false);
  free_handle(ctx, C);
  free_handle(ctx, B);
  free_handle(ctx, A);
  return rc;
}

// --------------------------------------------------------- mono (S/mono.fut)
#define MSR(TT) struct futhark_opaque_msr_##TT
#define MSC(TT) struct futhark_opaque_msc_##TT

#define SAMD_MONO(TT, CT)                                                                                      \
  int futhark_free_opaque_msr_##TT(CTX, MSR(TT) * o) { ENTER(ctx); return m_free(ctx, o); }                   \
  int futhark_free_opaque_msc_##TT(CTX, MSC(TT) * o) { ENTER(ctx); return m_free(ctx, o); }                   \
  int futhark_entry_msr_##TT##_sparse(CTX, MSR(TT) * *out0, const int64_t n, const int64_t m, I64A *rs,        \
                                      I64A *cs, const A1(TT) * vs) {                                           \
    ENTER(ctx);                                                                                                \
    return m_sparse<CT>(ctx, out0, n, m, rs, cs, vs);                                                          \
  }                                                                                                            \
  int futhark_entry_msc_##TT##_sparse(CTX, MSC(TT) * *out0, const int64_t n, const int64_t m, I64A *rs,        \
                                      I64A *cs, const A1(TT) * vs) {                                           \
    ENTER(ctx); /* map swap coo |> sr.sparse m n */                                                            \
    return m_sparse<CT>(ctx, out0, m, n, cs, rs, vs);                                                          \
  }                                                                                                            \
  int futhark_entry_msr_##TT##_zero(CTX, MSR(TT) * *out0, const int64_t n, const int64_t m) {                  \
    ENTER(ctx);                                                                                                \
    return m_make<CT>(ctx, out0, n, m, 0, nullptr);                                                            \
  }                                                                                                            \
  int futhark_entry_msc_##TT##_zero(CTX, MSC(TT) * *out0, const int64_t n, const int64_t m) {                  \
    ENTER(ctx);                                                                                                \
    return m_make<CT>(ctx, out0, m, n, 0, nullptr);                                                            \
  }                                                                                                            \
  int futhark_entry_msr_##TT##_eye(CTX, MSR(TT) * *out0, const int64_t n, const int64_t m) {                   \
    ENTER(ctx);                                                                                                \
    return m_make<CT>(ctx, out0, n, m, 1, nullptr);                                                            \
  }                                                                                                            \
  int futhark_entry_msc_##TT##_eye(CTX, MSC(TT) * *out0, const int64_t n, const int64_t m) {                   \
    ENTER(ctx);                                                                                                \
    return m_make<CT>(ctx, out0, m, n, 1, nullptr);                                                            \
  }                                                                                                            \
  int futhark_entry_msr_##TT##_diag(CTX, MSR(TT) * *out0, const A1(TT) * v) {                                  \
    ENTER(ctx);                                                                                                \
    return m_make<CT>(ctx, out0, v->shape[0], v->shape[0], 2, v->data);                                        \
  }                                                                                                            \
  int futhark_entry_msc_##TT##_diag(CTX, MSC(TT) * *out0, const A1(TT) * v) {                                  \
    ENTER(ctx);                                                                                                \
    return m_make<CT>(ctx, out0, v->shape[0], v->shape[0], 2, v->data);                                        \
  }                                                                                                            \
  int futhark_entry_msr_##TT##_dense(CTX, A2(TT) * *out0, const MSR(TT) * a) {                                 \
    ENTER(ctx);                                                                                                \
    return m_dense<CT>(ctx, out0, a, false);                                                                   \
  }                                                                                                            \
  int futhark_entry_msc_##TT##_dense(CTX, A2(TT) * *out0, const MSC(TT) * a) {                                 \
    ENTER(ctx);                                                                                                \
    return m_dense<CT>(ctx, out0, a, true);                                                                    \
  }                                                                                                            \
  int futhark_entry_msr_##TT##_scale(CTX, MSR(TT) * *out0, const CT v, const MSR(TT) * a) {                    \
    ENTER(ctx);                                                                                                \
    return m_map<CT>(ctx, out0, a, (const MSR(TT) *)nullptr, v, 0);                                            \
  }                                                                                                            \
  int futhark_entry_msc_##TT##_scale(CTX, MSC(TT) * *out0, const CT v, const MSC(TT) * a) {                    \
    ENTER(ctx);                                                                                                \
    return m_map<CT>(ctx, out0, a, (const MSC(TT) *)nullptr, v, 0);                                            \
  }                                                                                                            \
  int futhark_entry_msr_##TT##_nnz(CTX, int64_t *out0, const MSR(TT) * a) {                                    \
    ENTER(ctx);                                                                                                \
    return mono_nnz<CT>(ctx, a->b, out0);                                                                      \
  }                                                                                                            \
  int futhark_entry_msc_##TT##_nnz(CTX, int64_t *out0, const MSC(TT) * a) {                                    \
    ENTER(ctx);                                                                                                \
    return mono_nnz<CT>(ctx, a->b, out0);                                                                      \
  }                                                                                                            \
  int futhark_entry_msr_##TT##_coo(CTX, struct futhark_i64_1d **out0, struct futhark_i64_1d **out1,            \
                                   A1(TT) * *out2, const MSR(TT) * a) {                                        \
    ENTER(ctx);                                                                                                \
    return m_coo<CT>(ctx, out0, out1, out2, a, false);                                                         \
  }                                                                                                            \
  int futhark_entry_msc_##TT##_coo(CTX, struct futhark_i64_1d **out0, struct futhark_i64_1d **out1,            \
                                   A1(TT) * *out2, const MSC(TT) * a) {                                        \
    ENTER(ctx);                                                                                                \
    return m_coo<CT>(ctx, out0, out1, out2, a, true);                                                          \
  }                                                                                                            \
  int futhark_entry_msr_##TT##_add(CTX, MSR(TT) * *out0, const MSR(TT) * a, const MSR(TT) * b) {               \
    ENTER(ctx);                                                                                                \
    return m_map<CT>(ctx, out0, a, b, (CT)0, 1);                                                               \
  }                                                                                                            \
  int futhark_entry_msr_##TT##_sub(CTX, MSR(TT) * *out0, const MSR(TT) * a, const MSR(TT) * b) {               \
    ENTER(ctx);                                                                                                \
    return m_map<CT>(ctx, out0, a, b, (CT)0, 2);                                                               \
  }                                                                                                            \
  int futhark_entry_msc_##TT##_add(CTX, MSC(TT) * *out0, const MSC(TT) * a, const MSC(TT) * b) {               \
    ENTER(ctx);                                                                                                \
    return m_map<CT>(ctx, out0, a, b, (CT)0, 1);                                                               \
  }                                                                                                            \
  int futhark_entry_msc_##TT##_sub(CTX, MSC(TT) * *out0, const MSC(TT) * a, const MSC(TT) * b) {               \
    ENTER(ctx);                                                                                                \
    return m_map<CT>(ctx, out0, a, b, (CT)0, 2);                                                               \
  }                                                                                                            \
  int futhark_entry_msr_##TT##_transpose(CTX, MSC(TT) * *out0, const MSR(TT) * a) {                            \
    ENTER(ctx);                                                                                                \
    return m_relabel(out0, a);                                                                                 \
  }                                                                                                            \
  int futhark_entry_msc_##TT##_transpose(CTX, MSR(TT) * *out0, const MSC(TT) * a) {                            \
    ENTER(ctx);                                                                                                \
    return m_relabel(out0, a);                                                                                 \
  }                                                                                                            \
  int futhark_entry_msr_##TT##_smvm(CTX, A1(TT) * *out0, const MSR(TT) * a, const A1(TT) * v) {                \
    ENTER(ctx);                                                                                                \
    return m_smvm<CT>(ctx, out0, a, v);                                                                        \
  }                                                                                                            \
  int futhark_entry_msr_##TT##_vsmm(CTX, A1(TT) * *out0, const A1(TT) * v, const MSR(TT) * a) {                \
    ENTER(ctx);                                                                                                \
    return m_vsmm<CT>(ctx, out0, v, a);                                                                        \
  }                                                                                                            \
  int futhark_entry_msc_##TT##_vsmm(CTX, A1(TT) * *out0, const A1(TT) * a, const MSC(TT) * b) {                \
    ENTER(ctx); /* sr.smvm (transpose b) a */                                                                  \
    return m_smvm<CT>(ctx, out0, b, a);                                                                        \
  }                                                                                                            \
  int futhark_entry_msr_##TT##_dmsmm(CTX, A2(TT) * *out0, const A2(TT) * d, const MSR(TT) * s) {               \
    ENTER(ctx);                                                                                                \
    return m_dmsmm<CT>(ctx, out0, d, s);                                                                       \
  }                                                                                                            \
  int sparse_amd_msr_##TT##_fields(CTX, struct futhark_i64_1d **col_idx, A1(TT) * *vals, const MSR(TT) * a) {  \
    ENTER(ctx);                                                                                                \
    return m_fields<CT>(ctx, col_idx, vals, a);                                                                \
  }                                                                                                            \
  int sparse_amd_msc_##TT##_fields(CTX, struct futhark_i64_1d **col_idx, A1(TT) * *vals, const MSC(TT) * a) {  \
    ENTER(ctx);                                                                                                \
    return m_fields<CT>(ctx, col_idx, vals, a);                                                                \
  }

SAMD_MONO(f64, double)
SAMD_MONO(f32, float)
SAMD_MONO(i32, int32_t)

// The reference's own mono entries (module i32, mono_test.fut): each body is the composition
// written at the cited lines.
typedef struct futhark_opaque_msr_i32 msr_i32;
typedef struct futhark_opaque_msc_i32 msc_i32;
#define COO_ARGS const int64_t in0, const int64_t in1, carr_i64 *in2, carr_i64 *in3, const arr_i32 *in4

// mono_test.fut:30-31  msr.eye n m |> msr.dense
int futhark_entry_test_msr_eye(CTX, mat_i32 **out0, const int64_t in0, const int64_t in1) {
  ENTER(ctx);
  msr_i32 *a = nullptr;
  SAMD_TRY(m_make<int32_t>(ctx, &a, in0, in1, 1, nullptr));
  int rc = m_dense<int32_t>(ctx, out0, a, false);
  m_free(ctx, a);
  return rc;
}
// :42-43
int futhark_entry_test_msr_sparse(CTX, mat_i32 **out0, COO_ARGS) {
  ENTER(ctx);
  msr_i32 *a = nullptr;
  SAMD_TRY(m_sparse<int32_t>(ctx, &a, in0, in1, in2, in3, in4));
  int rc = m_dense<int32_t>(ctx, out0, a, false);
  m_free(ctx, a);
  return rc;
}
// :55-57
int futhark_entry_test_msr_smvm(CTX, arr_i32 **out0, COO_ARGS, const arr_i32 *in5) {
  ENTER(ctx);
  msr_i32 *a = nullptr;
  SAMD_TRY(m_sparse<int32_t>(ctx, &a, in0, in1, in2, in3, in4));
  int rc = m_smvm<int32_t>(ctx, out0, a, in5);
  m_free(ctx, a);
  return rc;
}
// :66-67
int futhark_entry_test_msr_nnz(CTX, int64_t *out0, COO_ARGS) {
  ENTER(ctx);
  msr_i32 *a = nullptr;
  SAMD_TRY(m_sparse<int32_t>(ctx, &a, in0, in1, in2, in3, in4));
  int rc = mono_nnz<int32_t>(ctx, a->b, out0);
  m_free(ctx, a);
  return rc;
}
// :74-76
int futhark_entry_test_msr_coo(CTX, struct futhark_i64_1d **out0, struct futhark_i64_1d **out1, arr_i32 **out2,
                               COO_ARGS) {
  ENTER(ctx);
  msr_i32 *a = nullptr;
  SAMD_TRY(m_sparse<int32_t>(ctx, &a, in0, in1, in2, in3, in4));
  int rc = m_coo<int32_t>(ctx, out0, out1, out2, a, false);
  m_free(ctx, a);
  return rc;
}
// :87-89  (n, m, v, xs, ys, vs)
int futhark_entry_test_msr_vsmm(CTX, arr_i32 **out0, const int64_t in0, const int64_t in1, const arr_i32 *in2,
                                carr_i64 *in3, carr_i64 *in4, const arr_i32 *in5) {
  ENTER(ctx);
  msr_i32 *a = nullptr;
  SAMD_TRY(m_sparse<int32_t>(ctx, &a, in0, in1, in3, in4, in5));
  int rc = m_vsmm<int32_t>(ctx, out0, in2, a);
  m_free(ctx, a);
  return rc;
}
// :114-115  msc.eye n m |> msc.dense
int futhark_entry_test_msc_eye(CTX, mat_i32 **out0, const int64_t in0, const int64_t in1) {
  ENTER(ctx);
  msc_i32 *a = nullptr;
  SAMD_TRY(m_make<int32_t>(ctx, &a, in1, in0, 1, nullptr));
  int rc = m_dense<int32_t>(ctx, out0, a, true);
  m_free(ctx, a);
  return rc;
}
// :126-127
int futhark_entry_test_msc_sparse(CTX, mat_i32 **out0, COO_ARGS) {
  ENTER(ctx);
  msc_i32 *a = nullptr;
  SAMD_TRY(m_sparse<int32_t>(ctx, &a, in1, in0, in3, in2, in4));
  int rc = m_dense<int32_t>(ctx, out0, a, true);
  m_free(ctx, a);
  return rc;
}
// :136-137
int futhark_entry_test_msc_nnz(CTX, int64_t *out0, COO_ARGS) {
  ENTER(ctx);
  msc_i32 *a = nullptr;
  SAMD_TRY(m_sparse<int32_t>(ctx, &a, in1, in0, in3, in2, in4));
  int rc = mono_nnz<int32_t>(ctx, a->b, out0);
  m_free(ctx, a);
  return rc;
}
// :144-146
int futhark_entry_test_msc_coo(CTX, struct futhark_i64_1d **out0, struct futhark_i64_1d **out1, arr_i32 **out2,
                               COO_ARGS) {
  ENTER(ctx);
  msc_i32 *a = nullptr;
  SAMD_TRY(m_sparse<int32_t>(ctx, &a, in1, in0, in3, in2, in4));
  int rc = m_coo<int32_t>(ctx, out0, out1, out2, a, true);
  m_free(ctx, a);
  return rc;
}
// :153-155  msr.sparse n m ... |> msr.transpose |> msc.dense
int futhark_entry_test_msr_transpose(CTX, mat_i32 **out0, COO_ARGS) {
  ENTER(ctx);
  msr_i32 *a = nullptr;
  SAMD_TRY(m_sparse<int32_t>(ctx, &a, in0, in1, in2, in3, in4));
  msc_i32 *t = nullptr;
  m_relabel(&t, a);
  int rc = m_dense<int32_t>(ctx, out0, t, true);
  m_free(ctx, t);
  m_free(ctx, a);
  return rc;
}
// :162-164  msc.sparse n m ... |> msc.transpose |> msr.dense
int futhark_entry_test_msc_transpose(CTX, mat_i32 **out0, COO_ARGS) {
  ENTER(ctx);
  msc_i32 *a = nullptr;
  SAMD_TRY(m_sparse<int32_t>(ctx, &a, in1, in0, in3, in2, in4));
  msr_i32 *t = nullptr;
  m_relabel(&t, a);
  int rc = m_dense<int32_t>(ctx, out0, t, false);
  m_free(ctx, t);
  m_free(ctx, a);
  return rc;
}

// ------------------------------------------------------ Matrix Market (host side)
int sparse_amd_mtx_read(const char *path, int threads, int64_t *n, int64_t *m, int64_t *count, int64_t **rows,
                        int64_t **cols, double **vals, char **error) {
  MtxResult r;
  std::string err;
  if (error) *error = nullptr;
  if (mtx_read(path, threads, &r, &err) != 0) {
    if (error) *error = strdup(err.c_str());
    return 1;
  }
  *n = r.n; *m = r.m; *count = r.count;
  *rows = r.rows; *cols = r.cols; *vals = r.vals;
  return 0;
}

int sparse_amd_mtx_write(const char *path, int64_t n, int64_t m, int64_t count, const int64_t *rows,
                         const int64_t *cols, const double *vals, char **error) {
  std::string err;
  if (error) *error = nullptr;
  if (mtx_write(path, n, m, count, rows, cols, vals, &err) != 0) {
    if (error) *error = strdup(err.c_str());
    return 1;
  }
  return 0;
}

void sparse_amd_mtx_free(void *p) { free(p); }

// --------------------------------------------- one process, several devices (group.hpp)
int sparse_amd_group_new(struct sparse_amd_group **out, int ndev, const int *devices, const char *transport) {
  sparse_amd_group *g = new sparse_amd_group();
  *out = g;  // always returned: the error text lives in it
  if (ndev < 1 || ndev > 64) return group_error(g, "group: %d members asked for (1..64)", ndev);
  g->G = ndev;
  bool distinct = true;
  for (int k = 0; k < ndev; k++) {
    g->dev.push_back(devices ? devices[k] : k);
    for (int j = 0; j < k; j++) distinct = distinct && g->dev[(size_t)j] != g->dev[(size_t)k];
  }
  const bool want_rccl = transport && !strcmp(transport, "rccl");
  const bool want_peer = transport && !strcmp(transport, "peer");
  if (transport && !want_rccl && !want_peer) return group_error(g, "group: unknown transport '%s' (rccl, peer)", transport);
  if (want_rccl && !distinct) return group_error(g, "group: RCCL needs distinct devices (the peer transport rehearses several members on one)");
  for (int k = 0; k < ndev; k++) {
    futhark_context_config *cfg = futhark_context_config_new();
    cfg->device = g->dev[(size_t)k];
    futhark_context *c = futhark_context_new(cfg);
    futhark_context_config_free(cfg);
    g->ctx.push_back(c);
    if (c->has_err) return group_member_error(g, k, "context");
  }
  g->ev_x.assign((size_t)ndev, nullptr);
  g->ev_done.assign((size_t)ndev, nullptr);
  for (int k = 0; k < ndev; k++) {
    SAMD_GHIP(g, hipSetDevice(g->dev[(size_t)k]));
    SAMD_GHIP(g, hipEventCreateWithFlags(&g->ev_x[(size_t)k], hipEventDisableTiming));
    SAMD_GHIP(g, hipEventCreateWithFlags(&g->ev_done[(size_t)k], hipEventDisableTiming));
    for (int j = 0; j < ndev; j++) {  // direct peer access where the devices differ (ignored if already on / unsupported)
      if (g->dev[(size_t)j] == g->dev[(size_t)k]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, g->dev[(size_t)k], g->dev[(size_t)j]) == hipSuccess && can)
        (void)hipDeviceEnablePeerAccess(g->dev[(size_t)j], 0);
      (void)hipGetLastError();
    }
  }
  // RCCL: by default for several distinct devices; a group of ONE member opens it only when asked to by name (its
  // slice of x then travels through a grouped send-to-self, which is how the transport is exercised on a one-GPU box)
  if (((ndev > 1 && !want_peer) || (ndev == 1 && want_rccl)) && distinct) {
    std::string why;
    if (!g->rccl.load(&why)) {
      if (want_rccl) return group_error(g, "group: %s", why.c_str());
    } else {
      g->comm.assign((size_t)ndev, nullptr);
      SAMD_GNCCL(g, g->rccl.CommInitAll(g->comm.data(), ndev, g->dev.data()));
      g->use_rccl = true;
    }
  }
  return 0;
}

void sparse_amd_group_free(struct sparse_amd_group *g) { group_destroy(g); }

char *sparse_amd_group_get_error(struct sparse_amd_group *g) {
  if (!g || !g->has_err) return nullptr;
  g->has_err = false;
  return strdup(g->err.c_str());
}

int sparse_amd_group_size(struct sparse_amd_group *g) { return g->G; }
const char *sparse_amd_group_transport(struct sparse_amd_group *g) { return g->use_rccl ? "rccl" : (g->G <= 1 ? "none" : "peer"); }
struct futhark_context *sparse_amd_group_context(struct sparse_amd_group *g, int k) {
  return (k >= 0 && k < g->G) ? g->ctx[(size_t)k] : nullptr;
}

int sparse_amd_group_sync(struct sparse_amd_group *g) {
  for (int k = 0; k < g->G; k++)
    if (futhark_context_sync(g->ctx[(size_t)k]) != 0) return group_member_error(g, k, "sync");
  return 0;
}

int sparse_amd_dcsr_f64_free(struct sparse_amd_group *g, struct sparse_amd_dcsr_f64 *A) {
  if (!A) return 0;
  for (int k = 0; k < (int)A->A.size(); k++) {
    futhark_context *c = g->ctx[(size_t)k];
    if (A->A[(size_t)k]) futhark_free_opaque_csr_f64(c, A->A[(size_t)k]);
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(g->dev[(size_t)k]);
    if (k < (int)A->x_full.size()) dfree(c, A->x_full[(size_t)k]);
  }
  delete A;
  return 0;
}

// sr.sparse n m coo (compressed.fut:154-160) with the rows dealt to the members: host triples in, one csr handle per
// member out.  Splitters: nearly equal triple counts, cut at the bins of a coarse row histogram.
int sparse_amd_dcsr_f64_sparse(struct sparse_amd_group *g, struct sparse_amd_dcsr_f64 **out, int64_t n, int64_t m,
                               const int64_t *rs, const int64_t *cs, const double *vs, int64_t count) {
  *out = nullptr;
  if (n < 0 || m < 0 || count < 0) return group_error(g, "sparse: negative size");
  const int G = g->G;
  const int shift = bits_for(n) > 16 ? bits_for(n) - 16 : 0;
  const int64_t nbins = n > 0 ? ((n - 1) >> shift) + 1 : 1;
  std::vector<int64_t> hist((size_t)nbins, 0);
  for (int64_t i = 0; i < count; i++) {
    if (rs[i] < 0 || rs[i] >= n || cs[i] < 0 || cs[i] >= m)
      return group_error(g, "Error: index out of bounds in sparse (assertion 0 <= r < n && 0 <= c < m failed)");
    hist[(size_t)(rs[i] >> shift)]++;
  }
  sparse_amd_dcsr_f64 *A = new sparse_amd_dcsr_f64();
  A->n = n; A->m = m;
  A->rb.assign((size_t)G + 1, n);
  A->rb[0] = 0;
  {
    int64_t cum = 0;
    int next = 1;
    for (int64_t b = 0; b < nbins && next < G; b++) {
      cum += hist[(size_t)b];
      while (next < G && cum >= count * next / G) {  // bins [0, b] hold at least member next's share
        const int64_t cut = (b + 1) << shift;
        A->rb[(size_t)next++] = cut < n ? cut : n;
      }
    }
  }
  if (n == m) A->cb = A->rb;
  else {
    const int64_t per = cdiv(m, G);
    A->cb.assign((size_t)G + 1, m);
    for (int k = 0; k < G; k++) A->cb[(size_t)k] = k * per < m ? k * per : m;
  }
  // triples to their owners, input order kept (duplicates are then summed as on one device)
  std::vector<int64_t> cnt((size_t)G, 0);
  std::vector<int> owner((size_t)count);
  for (int64_t i = 0; i < count; i++) {
    int k = 0;
    while (rs[i] >= A->rb[(size_t)k + 1]) k++;
    owner[(size_t)i] = k;
    cnt[(size_t)k]++;
  }
  A->A.assign((size_t)G, nullptr);
  A->x_full.assign((size_t)G, nullptr);
  int rc = 0;
  for (int k = 0; k < G && rc == 0; k++) {
    std::vector<int64_t> r((size_t)cnt[(size_t)k]), c((size_t)cnt[(size_t)k]);
    std::vector<double> v((size_t)cnt[(size_t)k]);
    int64_t q = 0;
    for (int64_t i = 0; i < count; i++)
      if (owner[(size_t)i] == k) { r[(size_t)q] = rs[i] - A->rb[(size_t)k]; c[(size_t)q] = cs[i]; v[(size_t)q] = vs[i]; q++; }
    futhark_context *cx = g->ctx[(size_t)k];
    futhark_i64_1d *ar = futhark_new_i64_1d(cx, r.data(), q), *ac = futhark_new_i64_1d(cx, c.data(), q);
    futhark_f64_1d *av = futhark_new_f64_1d(cx, v.data(), q);
    if (!ar || !ac || !av ||
        futhark_entry_csr_f64_sparse(cx, &A->A[(size_t)k], A->rb[(size_t)k + 1] - A->rb[(size_t)k], m, ar, ac, av) != 0)
      rc = group_member_error(g, k, "sparse");
    if (ar) futhark_free_i64_1d(cx, ar);
    if (ac) futhark_free_i64_1d(cx, ac);
    if (av) futhark_free_f64_1d(cx, av);
    if (rc == 0) {
      std::lock_guard<std::mutex> lk(cx->mu);
      (void)hipSetDevice(g->dev[(size_t)k]);
      if (dalloc(cx, &A->x_full[(size_t)k], m) != 0) rc = 2;
    }
    if (rc == 2) rc = group_member_error(g, k, "sparse (x buffer)");
  }
  if (rc != 0) { sparse_amd_dcsr_f64_free(g, A); return rc; }
  *out = A;
  return 0;
}

int sparse_amd_dcsr_f64_bounds(struct sparse_amd_group *g, const struct sparse_amd_dcsr_f64 *A, int64_t *row_bounds,
                               int64_t *col_bounds) {
  for (int k = 0; k <= g->G; k++) {
    if (row_bounds) row_bounds[k] = A->rb[(size_t)k];
    if (col_bounds) col_bounds[k] = A->cb[(size_t)k];
  }
  return 0;
}

struct futhark_opaque_csr_f64 *sparse_amd_dcsr_f64_member(struct sparse_amd_group *g, const struct sparse_amd_dcsr_f64 *A, int k) {
  return (k >= 0 && k < g->G) ? A->A[(size_t)k] : nullptr;
}

int sparse_amd_dvec_f64_free(struct sparse_amd_group *g, struct sparse_amd_dvec_f64 *v) {
  if (!v) return 0;
  for (int k = 0; k < (int)v->d.size(); k++) {
    futhark_context *c = g->ctx[(size_t)k];
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(g->dev[(size_t)k]);
    dfree(c, v->d[(size_t)k]);
  }
  delete v;
  return 0;
}

// A vector cut like A's columns (like_rows == 0: an operand of smvm) or like its rows (a result); host == NULL
// leaves it uninitialised.
int sparse_amd_dvec_f64_new(struct sparse_amd_group *g, struct sparse_amd_dvec_f64 **out, const struct sparse_amd_dcsr_f64 *A,
                            int like_rows, const double *host) {
  *out = nullptr;
  sparse_amd_dvec_f64 *v = new sparse_amd_dvec_f64();
  v->b = like_rows ? A->rb : A->cb;
  v->d.assign((size_t)g->G, nullptr);
  int bad = -1;
  hipError_t he = hipSuccess;
  for (int k = 0; k < g->G && bad < 0; k++) {
    futhark_context *c = g->ctx[(size_t)k];
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(g->dev[(size_t)k]);
    const int64_t cnt = v->b[(size_t)k + 1] - v->b[(size_t)k];
    if (dalloc(c, &v->d[(size_t)k], cnt) != 0) { bad = k; break; }
    if (host && cnt > 0) {
      he = hipMemcpyAsync(v->d[(size_t)k], host + v->b[(size_t)k], sizeof(double) * (size_t)cnt, hipMemcpyHostToDevice, c->stream);
      if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
      if (he != hipSuccess) { (void)hipGetLastError(); bad = k; }
    }
  }
  if (bad >= 0) {
    if (he != hipSuccess) group_error(g, "dvec: copy to member %d failed: %s", bad, hipGetErrorString(he));
    else group_member_error(g, bad, "dvec");
    sparse_amd_dvec_f64_free(g, v);
    return 1;
  }
  *out = v;
  return 0;
}

int sparse_amd_dvec_f64_values(struct sparse_amd_group *g, const struct sparse_amd_dvec_f64 *v, double *host) {
  for (int k = 0; k < g->G; k++) {
    futhark_context *c = g->ctx[(size_t)k];
    std::lock_guard<std::mutex> lk(c->mu);
    SAMD_GHIP(g, hipSetDevice(g->dev[(size_t)k]));
    const int64_t cnt = v->b[(size_t)k + 1] - v->b[(size_t)k];
    if (cnt > 0)
      SAMD_GHIP(g, hipMemcpyAsync(host + v->b[(size_t)k], v->d[(size_t)k], sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost,
                                  c->stream));
  }
  for (int k = 0; k < g->G; k++) {
    SAMD_GHIP(g, hipSetDevice(g->dev[(size_t)k]));
    SAMD_GHIP(g, hipStreamSynchronize(g->ctx[(size_t)k]->stream));
  }
  return 0;
}

void *sparse_amd_dvec_f64_member_ptr(struct sparse_amd_group *g, const struct sparse_amd_dvec_f64 *v, int k, int64_t *len) {
  if (k < 0 || k >= g->G) return nullptr;
  if (len) *len = v->b[(size_t)k + 1] - v->b[(size_t)k];
  return v->d[(size_t)k];
}

// sr.smvm (compressed.fut:134-146) over the members: every member's slice of x reaches every member, then every
// member multiplies its rows.  Asynchronous; sparse_amd_group_sync or sparse_amd_dvec_f64_values wait.
int sparse_amd_dcsr_f64_smvm(struct sparse_amd_group *g, struct sparse_amd_dvec_f64 *y, const struct sparse_amd_dcsr_f64 *A,
                             const struct sparse_amd_dvec_f64 *x) {
  return group_smvm(g, y, A, x);
}

}  // extern "C"
