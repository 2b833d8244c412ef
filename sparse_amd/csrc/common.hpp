// common.hpp -- context, device-memory pool, array/opaque records, error plumbing.
//
// One context = one HIP device + one stream.  All library work is stream-ordered
// on that stream, so a block released to the pool can be handed to a later
// operation without further synchronisation.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>

#include "../../include/sparse_amd.h"

namespace samd {

// ------------------------------------------------------------------ pool
// Size-bucketed free list in front of hipMalloc/hipFree (the generated Futhark
// runtimes keep the same kind of free list): steady-state entry calls allocate
// nothing from the driver.
class Pool {
 public:
  void *alloc(size_t bytes, hipError_t *err) {
    size_t need = round_up(bytes);
    auto it = free_.lower_bound(need);
    if (it != free_.end() && it->first <= need + need / 2 + 4096) {
      void *p = it->second;
      size_t sz = it->first;
      free_.erase(it);
      live_[p] = sz;
      cached_ -= sz;
      return p;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, need);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      clear();
      e = hipMalloc(&p, need);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      *err = e;
      return nullptr;
    }
    live_[p] = need;
    total_ += need;
    if (total_ > peak_) peak_ = total_;
    return p;
  }
  void release(void *p) {
    if (!p) return;
    auto it = live_.find(p);
    if (it == live_.end()) return;  // not ours (wrapped memory)
    free_.emplace(it->second, p);
    cached_ += it->second;
    live_.erase(it);
  }
  void clear() {
    for (auto &kv : free_) {
      (void)hipFree(kv.second);
      total_ -= kv.first;
    }
    free_.clear();
    cached_ = 0;
  }
  size_t total() const { return total_; }
  size_t peak() const { return peak_; }
  size_t cached() const { return cached_; }
  size_t live_blocks() const { return live_.size(); }

 private:
  static size_t round_up(size_t b) {
    if (b < 256) return 256;
    return (b + 255) & ~size_t(255);
  }
  std::multimap<size_t, void *> free_;
  std::unordered_map<void *, size_t> live_;
  size_t total_ = 0, peak_ = 0, cached_ = 0;
};

}  // namespace samd

// ------------------------------------------------------------- C-visible records
struct futhark_context_config {
  int device = 0;
  int debugging = 0, profiling = 0, logging = 0;
  int num_threads = 0;  // multicore back end's knob: recorded, unused (no CPU path)
};

struct futhark_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  int num_cu = 256;
  int logging = 0;
  std::mutex mu;
  std::string err;
  bool has_err = false;
  samd::Pool pool;
  // pinned staging for small device->host reads (counts, flags)
  int64_t *h_small = nullptr;  // 16 int64
  int64_t *d_small = nullptr;  // 16 int64
  // this device's copy of the library's static scratch array (capi.hip), looked up at context creation
  void *static_scratch = nullptr;
  // options (sparse_amd_context_set_option)
  int64_t opt_spmv_algo = 0;
  int64_t opt_spmv_lanes = 0;
  int64_t opt_wide_col = 0;
  int64_t opt_spmv_tile = 0;
  int64_t opt_spmm_algo = 0;
  int64_t opt_spmv_cblk = 0;  // 0 auto, > 0 always with this many columns per block, -1 never
  int64_t opt_spmv_reorder = 0;  // 0 auto, 1 always (when the record qualifies by size), -1 never
  int64_t opt_spmv_eager = 0;  // 1: a record's first product decides and builds everything (the behaviour before round 3; tests)
  int64_t opt_spmv_pb = 0;  // two-pass SpMV (spmv_pb.hpp), 0 large skewed gather-bound records of 4-byte values, 1 also flat popularity, 2 always (tests), -1 never
  int64_t opt_spmv_pb_rows = 0, opt_spmv_pb_lines = 0, opt_spmv_pb_entries = 0;  // test overrides: rows per panel,
                                                                                 // 32-column lines per block, entries per panel
  int64_t opt_spmv_pb_xcopy = 0;  // two-pass SpMV, folded layout: 1 = x copied into block order first (lab / tests), 0 = pass 1 gathers its slices from x
  int64_t opt_spmv_pb_fold = 0;  // two-pass SpMV, products of one row inside one piece summed in pass 1: 0 and 1 always (round 3: it wins whatever folds), -1 never
  int64_t opt_spmv_pb_deal = 0;    // lines into column blocks: 0 sorted by popularity for skewed records (with the folded layout), dealt round otherwise; 1 always sorted; -1 always dealt
  int64_t opt_spmv_pb_shares = 0;  // lab: folded layout, pass 1's workgroups per CU (0 = 4)
  int64_t opt_spmv_pb_oround = 0;  // lab: folded layout, alignment of a piece's outputs in products (0 = kPbOutRound)
  int64_t last_pb_fold = 0;      // 1 if the last two-pass product ran on a folded layout; last_pb_outputs: products it wrote (pads included)
  int64_t last_pb_outputs = 0;
  int64_t opt_sort_indirect = 0;  // lab: 1 = assembly sorts 8-byte values by 4-byte source index and gathers them at the end
  int64_t opt_sort_algo = 0;  // 0 one-sweep (look-back), 1 histogram pass + scan per digit
  int64_t opt_fold_emit = 0;       // assembly after an LSD sort: 0 = head flags, scan, fold and emission in one kernel (look-back), -1 = scan + merge_emit
  int64_t opt_add_sort = 0;        // lab/tests: 1 = sr.+ / sr.- sort the concatenated entries instead of merging the two runs
  int64_t opt_sort_inject = 0;     // tests: 1 = the next assembly's first attempt is treated as if a look-back wait had timed out
  bool sort_force_scan = false;    // set while an assembly is repeated after a look-back wait timed out: histogram + scan passes only
  int64_t opt_sort_msd = 0;        // assembly: 0 = top key bits by global passes + the rest in LDS when the sampled buckets are even, -1 never, T > 0 force T top bits
  int64_t opt_sort_msd_chunk = 0;  // lab/tests: nominal chunk length of the in-LDS sort
  int64_t opt_sort_pairs = 0;      // assembly, 8-byte values carried through the top-bit passes: 0 = key and value side by side in one array, -1 = separate arrays
  int64_t last_long_runs = 0;      // runs of one coordinate the last assembly handed to fold_long_runs_kernel (tests read it back)
  int64_t last_sort_msd = 0;       // top bits the last assembly's sort gave to the global passes (0 = plain LSD, -1 = tried, did not fit, sorted again)
  int64_t opt_noalloc_scratch_bytes = 0;  // tests: treat the static scratch array as this small (0 = its real size)
  int64_t opt_ablate = 0;  // kernel-lab switch: results are wrong when non-zero
  // digit histograms left by the pack kernel for the sort that follows (radix_sort.hpp): pre_hist[d][256] = counts of
  // bits [pre_shift[d], pre_shift[d] + pre_bits[d]) of pre_hist_keys[0 .. pre_hist_n); null = none
  // samples of the keys taken BEFORE the keys were packed (assembly.hpp: the host then waits for them on ev_pre while
  // the pack kernel runs, instead of idling the GPU while it decides): valid for pre_sample_n keys ending at bit
  // pre_sample_end; the five words land in h_small[8 .. 13)
  hipEvent_t ev_pre = nullptr;
  bool pre_sample = false;
  int64_t pre_sample_n = 0;
  int pre_sample_end = 0;
  uint32_t *pre_hist = nullptr;
  const void *pre_hist_keys = nullptr;
  int64_t pre_hist_n = 0;
  int pre_digits = 0, pre_shift[12] = {0}, pre_bits[12] = {0};
  // statistics for futhark_context_report
  int64_t n_entries = 0;
  int64_t last_spmv_col_bytes = 0;  // bytes per column index the last SpMV kernel read (2, 4 or 8)
  int64_t opt_spmv_col16 = 0;  // 16-bit tile-relative columns: 0 decided per record, -1 never
  int64_t last_spmv_path = 0;  // 0 one-pass, 1 hot-column renumbering, 2 column blocks, 3 two passes (tests read it back)
};

namespace samd {

inline int set_error(futhark_context *ctx, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  ctx->err = buf;
  ctx->has_err = true;
  return 1;
}

#define SAMD_HIP(ctx, call)                                                                  \
  do {                                                                                       \
    hipError_t e__ = (call);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      (void)hipGetLastError();                                                               \
      return samd::set_error(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),    \
                             __FILE__, __LINE__);                                            \
    }                                                                                        \
  } while (0)

#define SAMD_TRY(expr)        \
  do {                        \
    int rc__ = (expr);        \
    if (rc__ != 0) return rc__; \
  } while (0)

// Every kernel launch is followed by this: catches bad launch configurations at
// the call site instead of at the next synchronisation.
#define SAMD_LAUNCH_CHECK(ctx) SAMD_HIP(ctx, hipGetLastError())

template <class T>
inline int dalloc(futhark_context *ctx, T **out, int64_t count) {
  hipError_t e = hipSuccess;
  size_t bytes = sizeof(T) * (size_t)(count > 0 ? count : 1);
  void *p = ctx->pool.alloc(bytes, &e);
  if (!p) return set_error(ctx, "device allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
  *out = (T *)p;
  return 0;
}

inline void dfree(futhark_context *ctx, void *p) { ctx->pool.release(p); }

// RAII temporary from the pool.
template <class T>
struct Tmp {
  futhark_context *ctx;
  T *p = nullptr;
  explicit Tmp(futhark_context *c) : ctx(c) {}
  ~Tmp() { dfree(ctx, p); }
  int get(int64_t count) { return dalloc(ctx, &p, count); }
  T *release() { T *q = p; p = nullptr; return q; }
  Tmp(const Tmp &) = delete;
  Tmp &operator=(const Tmp &) = delete;
};

// Copy a few int64 from device to host through the pinned slot and wait.
inline int read_small(futhark_context *ctx, const int64_t *d, int count, int64_t *out) {
  SAMD_HIP(ctx, hipMemcpyAsync(ctx->h_small, d, sizeof(int64_t) * count, hipMemcpyDeviceToHost, ctx->stream));
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < count; i++) out[i] = ctx->h_small[i];
  return 0;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline int bits_for(int64_t count) {  // bits needed to hold values 0..count-1
  if (count <= 1) return 0;
  return 64 - __builtin_clzll((unsigned long long)(count - 1));
}

// -------------------------------------------------------------- device arrays
template <class T, int R>
struct DevArray {
  T *data = nullptr;
  int64_t shape[R];
  bool owned = true;
  int64_t count() const {
    int64_t c = 1;
    for (int i = 0; i < R; i++) c *= shape[i];
    return c;
  }
};

// ---------------------------------------------------------------- CSR record
// The reference record {row_off[n] inclusive, col_idx[nnz] i64, vals[nnz]}
// (compressed.fut:93-96) is held as
//   row_ptr[n+1]  int64, row_ptr[0] = 0  -> row_off = row_ptr + 1 (same bytes)
//   col32[nnz]    int32 when m <= 2^31-1 (declared in the roofline arithmetic:
//                 ib = 4), else col64[nnz] int64
//   vals[nnz]     T
// plus the nnz-balanced tiling used by the streaming kernels:
//   tile_lb[ntiles+1]: first row whose start is >= t*tile (tile_lb[ntiles] = n)
//   head[ntiles]     : partial sum of the row fragment that enters tile t from
//                      an earlier tile (SpMV scratch, rewritten every call)
template <class T>
struct CsrBody {
  int64_t n = 0, m = 0, nnz = 0;
  int64_t *row_ptr = nullptr;
  int32_t *col32 = nullptr;
  int64_t *col64 = nullptr;
  T *vals = nullptr;
  // plan
  int64_t tile = 0, ntiles = 0;
  int64_t *tile_lb = nullptr;
  T *head = nullptr;
  int64_t max_row_len = -1;  // filled lazily by the SpMV dispatcher
  // amortisation (spmv.hpp: spmv_may_build): products multiplied so far, and how many the caller has said it will
  // run in all (sparse_amd_csr_*_prepare; 0 = not said)
  mutable int64_t products = 0;
  mutable int64_t expected_products = 0;
  // 16-bit columns relative to each tile's smallest column (spmv.hpp: col16_prepare), for records whose tiles all
  // span fewer than 65536 columns (banded, stencil); built lazily at the first product of a large record
  mutable int col16_state = 0;  // 0 undecided, 1 active, -1 rejected
  mutable uint16_t *col16 = nullptr;   // [nnz]
  mutable int32_t *tile_c0 = nullptr;  // [ntiles]
  mutable int32_t *fix_list = nullptr;  // [4 ntiles] rows the fix-up launch folds: row (low 31 bits, high bits), first spill tile, tiles spanned
  mutable int64_t fix_count = 0;
  mutable uint16_t *row_rel = nullptr;  // [n + 1] row starts relative to their tile (spmv.hpp: row_rel_kernel); built lazily
  // plan statistics (tile_lb[ntiles+1], [ntiles+2]), read back once by the first product
  mutable int64_t spill_max = -1;      // longest first spill
  mutable int64_t max_tile_rows = -1;  // most rows one tile's block visits (max over tiles of the kernels' nrows64)
  // hot-column renumbering for gather-bound SpMV (spmv.hpp, "reorder"); built lazily, owned by the record
  mutable int reorder_state = 0;   // 0 undecided, 1 active, -1 rejected
  mutable int32_t *col_hot = nullptr;    // [nnz] rank of each entry's column by popularity
  mutable uint32_t *hot_perm = nullptr;  // [m]   hot_perm[rank] = column
  mutable T *x_hot = nullptr;            // [m]   x gathered into rank order, rewritten by every product
  // column-blocked copy for gather-bound SpMV on flat, wide matrices (spmv.hpp, "column blocks"); built lazily
  mutable bool gather_flat = false;  // the renumbering was rejected because column popularity is flat
  mutable int cblk_state = 0;        // 0 undecided, 1 active, -1 rejected
  mutable int64_t cblk_nb = 0;       // column blocks
  mutable int64_t *cblk_piece_ptr = nullptr;  // [panels * nb + 1] start of every piece
  mutable uint16_t *cblk_end = nullptr;       // [panels * nb * rows per panel] end of each row inside its piece
  mutable int32_t *cblk_col = nullptr;        // [nnz] entries in piece order
  mutable T *cblk_val = nullptr;
  // two-pass copy for gather-bound SpMV on power-law records (spmv_pb.hpp); built lazily
  mutable int pb_state = 0;  // 0 undecided, 1 active, -1 rejected
  mutable int pb_fold = 0;   // 1: folded layout (pb_col16 bit 15 = first entry of a run, pb_dgroup = first output of every group of 32 entries;
                             // both and pb_val in chunks of 512 entries, a lane's eight entries side by side: pb_pass1f_kernel)
  mutable int64_t pb_outputs = 0;  // products pass 1 writes (pads included)
  mutable int64_t *pb_wtab = nullptr;  // folded layout: [pb_nwg][3] pass 1's shares (first chunk, end, column block)
  mutable int64_t pb_nwg = 0;
  mutable int64_t pb_np = 0, pb_nb = 0, pb_w = 0, pb_nl = 0;  // panels, column blocks, columns per block, 32-column lines
  mutable int64_t *pb_panel_row = nullptr;   // [np + 1] first row of every panel
  mutable int64_t *pb_piece_ptr = nullptr;   // [np * nb + 1] start of every piece, panel-major
  mutable uint32_t *pb_line_of_rank = nullptr;  // [nl] lines by descending entry count
  mutable uint32_t *pb_line_blk = nullptr;      // folded layout: the same list block by block (pass 1 takes its slices of x through it)
  mutable uint16_t *pb_col16 = nullptr, *pb_row16 = nullptr;  // [nnz] column inside the block, row inside the panel
  mutable T *pb_val = nullptr, *pb_prod = nullptr;            // [nnz] values in piece order; products (rewritten per call)
  mutable uint32_t *pb_dgroup = nullptr;                      // [padded entries / 32] pass 1's destination groups
  mutable T *pb_x = nullptr;                                  // [nb * w] x in block order (rewritten per call)
  // SpMM: tiles that go to the matrix cores (spmm.hpp), decided at the first product
  mutable int32_t *tile_split = nullptr;  // [ntiles] ids of the dense-block tiles, then of the others
  mutable int64_t n_mfma_tiles = -1;
  int refs = 1;
};

template <class T>
inline void csr_release(futhark_context *ctx, CsrBody<T> *b) {
  if (!b) return;
  if (--b->refs > 0) return;
  dfree(ctx, b->row_ptr);
  dfree(ctx, b->col32);
  dfree(ctx, b->col64);
  dfree(ctx, b->vals);
  dfree(ctx, b->tile_lb);
  dfree(ctx, b->head);
  dfree(ctx, b->row_rel);
  dfree(ctx, b->fix_list);
  dfree(ctx, b->col16);
  dfree(ctx, b->tile_c0);
  dfree(ctx, b->col_hot);
  dfree(ctx, b->hot_perm);
  dfree(ctx, b->x_hot);
  dfree(ctx, b->tile_split);
  dfree(ctx, b->cblk_piece_ptr);
  dfree(ctx, b->cblk_end);
  dfree(ctx, b->cblk_col);
  dfree(ctx, b->cblk_val);
  dfree(ctx, b->pb_panel_row);
  dfree(ctx, b->pb_piece_ptr);
  dfree(ctx, b->pb_line_of_rank);
  dfree(ctx, b->pb_line_blk);
  dfree(ctx, b->pb_col16);
  dfree(ctx, b->pb_row16);
  dfree(ctx, b->pb_val);
  dfree(ctx, b->pb_prod);
  dfree(ctx, b->pb_x);
  dfree(ctx, b->pb_dgroup);
  dfree(ctx, b->pb_wtab);
  delete b;
}

template <class T>
struct CsrGuard {  // releases a half-built body on an error path
  futhark_context *ctx;
  CsrBody<T> *b;
  CsrGuard(futhark_context *c, CsrBody<T> *p) : ctx(c), b(p) {}
  ~CsrGuard() { if (b) csr_release(ctx, b); }
  CsrBody<T> *release() { CsrBody<T> *q = b; b = nullptr; return q; }
};

}  // namespace samd
