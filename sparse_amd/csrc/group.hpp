// group.hpp -- one process that owns G devices: row-sharded CSR, distributed vectors, y = A x with the x slices
// exchanged by RCCL (grouped ncclSend / ncclRecv, all pairs at once: xGMI is a point-to-point mesh) or by direct
// peer copies.  The native counterpart of sparse_amd/distributed.py for callers without Python (SURVEY 8b
// "Threading": one context may own 1..8 GPUs); the reference has no distributed layer at all.
//
// Partition: contiguous row blocks with nearly equal triple counts (splitters from a coarse row
// histogram), member k holds rows [rb[k], rb[k+1]) as an ordinary csr handle of its own context with GLOBAL column
// numbers, and a full-length x buffer on its device.  x is cut like the rows when the matrix is square (so a
// product's result can be the next product's operand), else into equal slices.  Triples are dealt to the members
// in input order, so duplicates are summed in the same order as on one device: the members' records are, bit for
// bit, the row blocks of the single-device record.
//
// librccl is opened with dlopen when a group asks for it: single-GPU users of the library do not need it.
// Included by capi.hip after its entry templates.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <string>
#include <vector>

namespace {

struct RcclApi {
  void *lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load(std::string *why) {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) { *why = std::string("librccl.so not found: ") + dlerror(); return false; }
#define SAMD_RCCL_SYM(field, sym)                                                      \
  field = reinterpret_cast<decltype(field)>(dlsym(lib, sym));                          \
  if (!field) { *why = std::string("librccl.so lacks ") + sym; return false; }
    SAMD_RCCL_SYM(CommInitAll, "ncclCommInitAll")
    SAMD_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    SAMD_RCCL_SYM(GroupStart, "ncclGroupStart")
    SAMD_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    SAMD_RCCL_SYM(Send, "ncclSend")
    SAMD_RCCL_SYM(Recv, "ncclRecv")
    SAMD_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef SAMD_RCCL_SYM
    return true;
  }
};

}  // namespace

struct sparse_amd_group {
  int G = 0;
  std::vector<int> dev;
  std::vector<futhark_context *> ctx;
  bool use_rccl = false;
  RcclApi rccl;
  std::vector<ncclComm_t> comm;
  std::vector<hipEvent_t> ev_x, ev_done;  // member k's slice has left / member k's product has finished
  std::string err;
  bool has_err = false;
};

struct sparse_amd_dvec_f64 {
  std::vector<int64_t> b;   // G + 1 bounds
  std::vector<double *> d;  // member k's slice on its device (from its context's pool)
};

struct sparse_amd_dcsr_f64 {
  int64_t n = 0, m = 0;
  std::vector<int64_t> rb, cb;  // row / column bounds, G + 1 each
  std::vector<futhark_opaque_csr_f64 *> A;
  std::vector<double *> x_full;  // [m] on every member's device
};

namespace {

int group_error(sparse_amd_group *g, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g->err = buf;
  g->has_err = true;
  return 1;
}

// a member's own failure: take the text from its context
int group_member_error(sparse_amd_group *g, int k, const char *what) {
  char *e = futhark_context_get_error(g->ctx[(size_t)k]);
  group_error(g, "%s on member %d (device #%d): %s", what, k, g->dev[(size_t)k], e ? e : "(no message)");
  free(e);
  return 1;
}

#define SAMD_GHIP(g, call)                                                                          \
  do {                                                                                              \
    hipError_t e__ = (call);                                                                        \
    if (e__ != hipSuccess) {                                                                        \
      (void)hipGetLastError();                                                                      \
      return group_error(g, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    }                                                                                               \
  } while (0)

#define SAMD_GNCCL(g, call)                                                                         \
  do {                                                                                              \
    ncclResult_t r__ = (call);                                                                      \
    if (r__ != ncclSuccess)                                                                         \
      return group_error(g, "%s failed: %s (%s:%d)", #call, (g)->rccl.GetErrorString(r__), __FILE__, __LINE__); \
  } while (0)

void group_destroy(sparse_amd_group *g) {
  if (!g) return;
  for (int k = 0; k < (int)g->ctx.size(); k++) {
    (void)hipSetDevice(g->dev[(size_t)k]);
    if (k < (int)g->ev_x.size() && g->ev_x[(size_t)k]) (void)hipEventDestroy(g->ev_x[(size_t)k]);
    if (k < (int)g->ev_done.size() && g->ev_done[(size_t)k]) (void)hipEventDestroy(g->ev_done[(size_t)k]);
  }
  for (ncclComm_t c : g->comm)
    if (c) (void)g->rccl.CommDestroy(c);
  for (futhark_context *c : g->ctx) futhark_context_free(c);
  if (g->rccl.lib) dlclose(g->rccl.lib);
  delete g;
}

// x slices to every member's x_full, then every member's product.  Single host thread; all ordering between
// devices is by events (peer copies) or inside RCCL.
int group_smvm(sparse_amd_group *g, sparse_amd_dvec_f64 *y, const sparse_amd_dcsr_f64 *A, const sparse_amd_dvec_f64 *x) {
  const int G = g->G;
  if ((int)x->d.size() != G || (int)y->d.size() != G) return group_error(g, "smvm: vector belongs to another group");
  for (int k = 0; k <= G; k++) {
    if (x->b[(size_t)k] != A->cb[(size_t)k]) return group_error(g, "smvm: x is not partitioned like the matrix's columns");
    if (y->b[(size_t)k] != A->rb[(size_t)k]) return group_error(g, "smvm: y is not partitioned like the matrix's rows");
  }
  // own slices into place (a one-member RCCL group sends its slice to itself below instead)
  const bool self_rccl = G == 1 && g->use_rccl;
  for (int k = 0; k < G && !self_rccl; k++) {
    futhark_context *c = g->ctx[(size_t)k];
    std::lock_guard<std::mutex> lk(c->mu);
    SAMD_GHIP(g, hipSetDevice(g->dev[(size_t)k]));
    const int64_t cnt = A->cb[(size_t)k + 1] - A->cb[(size_t)k];
    if (cnt > 0)
      SAMD_GHIP(g, hipMemcpyAsync(A->x_full[(size_t)k] + A->cb[(size_t)k], x->d[(size_t)k], sizeof(double) * (size_t)cnt,
                                  hipMemcpyDeviceToDevice, c->stream));
  }
  if (g->use_rccl) {
    SAMD_GNCCL(g, g->rccl.GroupStart());
    for (int k = 0; k < G; k++) {
      const int64_t mine = A->cb[(size_t)k + 1] - A->cb[(size_t)k];
      for (int j = 0; j < G; j++) {
        if (j == k && !self_rccl) continue;
        const int64_t theirs = A->cb[(size_t)j + 1] - A->cb[(size_t)j];
        if (mine > 0) SAMD_GNCCL(g, g->rccl.Send(x->d[(size_t)k], (size_t)mine, ncclDouble, j, g->comm[(size_t)k], g->ctx[(size_t)k]->stream));
        if (theirs > 0)
          SAMD_GNCCL(g, g->rccl.Recv(A->x_full[(size_t)k] + A->cb[(size_t)j], (size_t)theirs, ncclDouble, j, g->comm[(size_t)k],
                                     g->ctx[(size_t)k]->stream));
      }
    }
    SAMD_GNCCL(g, g->rccl.GroupEnd());
  } else if (G > 1) {
    // direct peer copies, source-driven: member k writes its slice into every other member's buffer on its own
    // stream, after that member's previous product has finished reading the buffer
    for (int k = 0; k < G; k++) {
      SAMD_GHIP(g, hipSetDevice(g->dev[(size_t)k]));
      const int64_t cnt = A->cb[(size_t)k + 1] - A->cb[(size_t)k];
      hipStream_t sk = g->ctx[(size_t)k]->stream;
      for (int j = 0; j < G; j++) {
        if (j == k || cnt == 0) continue;
        SAMD_GHIP(g, hipStreamWaitEvent(sk, g->ev_done[(size_t)j], 0));
        SAMD_GHIP(g, hipMemcpyPeerAsync(A->x_full[(size_t)j] + A->cb[(size_t)k], g->dev[(size_t)j], x->d[(size_t)k],
                                        g->dev[(size_t)k], sizeof(double) * (size_t)cnt, sk));
      }
      SAMD_GHIP(g, hipEventRecord(g->ev_x[(size_t)k], sk));
    }
    for (int j = 0; j < G; j++) {
      SAMD_GHIP(g, hipSetDevice(g->dev[(size_t)j]));
      for (int k = 0; k < G; k++)
        if (k != j) SAMD_GHIP(g, hipStreamWaitEvent(g->ctx[(size_t)j]->stream, g->ev_x[(size_t)k], 0));
    }
  }
  for (int k = 0; k < G; k++) {
    futhark_context *c = g->ctx[(size_t)k];
    int rc;
    {
      std::lock_guard<std::mutex> lk(c->mu);
      SAMD_GHIP(g, hipSetDevice(g->dev[(size_t)k]));
      rc = spmv<double>(c, A->A[(size_t)k]->b, A->x_full[(size_t)k], y->d[(size_t)k]);
      if (rc == 0) SAMD_GHIP(g, hipEventRecord(g->ev_done[(size_t)k], c->stream));
    }
    if (rc != 0) return group_member_error(g, k, "smvm");  // reads the member's message under its own lock
  }
  return 0;
}

}  // namespace
