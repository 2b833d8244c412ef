// spmv_pb.hpp -- two-pass SpMV for gather-bound records (power-law graphs: BASELINE configs[2]).
//
// On an RMAT matrix the one-pass kernel (spmv.hpp) is bound by the rate at which a CU turns over x
// gathers that miss its L1 (~140-150 G gathers/s chip-wide, measured with every gather hitting L2), not by
// HBM: 3e8 entries cost 2 ms where their bytes would stream in 0.35 ms.  Here no gather leaves the CU.
// At its first product such a record gets a second copy of its entries cut into PIECES = (row panel) x
// (column block): a panel is <= R consecutive rows holding <= ~E entries, a block is W columns, R and W sized so
// that a slice of y (f64 accumulators), or of x, is 128 KB of LDS.  A product is then
//   pass 0  x copied into block order (whole 32-column lines);
//   pass 1  (workgroup = share of one column block, its slice of x in LDS): every entry's product val * x[col],
//           the gather served by LDS; the block's entries are ONE contiguous range, read and written as a stream;
//   pass 2  (workgroup = row panel, its slice of y in LDS): each wavefront walks pieces (panel, block) of the
//           panel, adds their products into the slice by LDS atomics (runs of equal rows folded inside the
//           wavefront first), then the slice is written to y.
// Entries are laid out [block][panel][row][column].  Columns are dealt to blocks by popularity (32-column lines
// ranked by sampled entry count, line of rank q -> block q mod NB): every block, hence every piece, holds about
// the same number of entries whatever the skew.
// Bytes per entry: 4+2 read, 4 written (pass 1), 4+2 read (pass 2) = 16 against 8 algorithmic (f32).
// The slice of y is accumulated in f64 (i32: in i32), so the sum of a row does not follow the oracle's
// left-to-right f32 order: within the 1e-5 bar (in fact closer to the exact sum), exact for i32.
//
// What the hardware taught (C3 = RMAT scale 24, f32, 2.96e8 entries; rocprofv3 kernel stats via bench/pb_lab.sh,
// LDS rates via bench/lds_lab.hip):
//   * ds_add_f32 retires one lane every 3.0 cycles per CU, ds_add_f64 and ds_add_u32 one every 0.4 (same
//     address: 3-4 cycles each for all).  With an f32 slice pass 2 took 1.99 ms, with f64 accumulators 0.85-0.95.
//   * a load behind a branch (`ok ? load : 0`) makes the compiler wait for it at once: the predicated form of
//     pass 1 streamed at 2 TB/s (0.87 ms without its store), the clamped-address form at 4.7 TB/s (0.38 ms).
//   * stores that straddle 128-byte lines: pass 1 0.91 ms, the same loop started on a 32-entry boundary 0.67 ms
//     (4.5 TB/s for 1.8 GB read + 1.2 GB written); non-temporal or plain stores make no difference.
//   * walking pieces costs: with the layout [panel][block] (pass 1 walks ~300-entry pieces and writes them, pass 2
//     streams) pass 1 took 1.1-1.4 ms; entries per panel E = 1 M / 512 K / 256 K / 128 K / 64 K give pass 2
//     1.22 / 0.96 / 0.94 / 0.99 / 1.27 ms (heavy panels are slow per entry, small pieces are slow per piece).
//   * pass 2 is not waiting for memory: a ring of four chunks (three in flight per wavefront instead of one) ran
//     0.99 ms against 0.94; without the LDS accumulation the same walk takes 0.51 ms -- the ~56 cycles of LDS work
//     per 64 entries (one ds_add_f64 = 27 of them) are what it pays.
//   * pass 2 was issuing 63 vector instructions per 64 entries, most of them the in-wavefront fold: folding only
//     when a wavefront holds <= 16 runs (instead of <= 32) 0.93 -> 0.89 ms.
//   * shipped: 0.06 + 0.66 + 0.89 = 1.60 ms per product against 2.08 ms for the one-pass kernel with hot-column
//     renumbering (20.5 % of HBM peak on the algorithmic bytes against 15.9 %).  One-off set-up 40 ms.
// Default for large, wide, skewed records of 4-byte values (reorder_prepare hands them over); 8-byte values stay
// on the one-pass paths (28 B per entry here against 12).
#pragma once

#include <vector>

#include "spmv.hpp"

namespace samd {

constexpr int kPbThreads = 1024;  // one workgroup per CU: the slice takes most of the CU's 160 KB of LDS
constexpr int kPbWaves = kPbThreads / 64;
constexpr int kPbLine = 32;                         // columns per line (the unit dealt to blocks)
constexpr int kPbSliceBytes = 128 << 10;            // slice of x (pass 1) or y (pass 2) in LDS
constexpr int64_t kPbPanelEntries = 256 << 10;      // a panel is closed once it holds this many entries (measured on C3:
                                                    // 1 M 1.22 ms, 512 K 0.96, 256 K 0.94, 128 K 0.99, 64 K 1.27 in pass 2)
constexpr int kPbFoldMax = 16;                      // pass 2 folds a wavefront's runs when it holds at most this many (C3: 2: 0.95 ms, 8: 0.91, 16: 0.89, 32: 0.93, 48: 1.01)
constexpr int64_t kPbMinBytes = 16 << 20;            // x at least this large (RMAT f32, bench/pb_scale_sweep.py, two-pass / one-pass ms:
                                                    // scale 20 0.21 / 0.15, 21 0.264 / 0.259, 22 0.47 / 0.51, 23 0.81 / 1.01, 24 1.62 / 2.08)
constexpr int kPbFoldRows = 64;                     // a panel of fewer rows than this keeps its entries sorted by row: pass 2 folds runs
constexpr int kPbFoldFlag = 0x8000;                 // row16 bit that says so
constexpr int kPbRunCap = 32;                       // folded layout: a row's entries inside a piece are cut into runs of at most this many
constexpr int64_t kPbMinPiece = 128;               // auto mode: mean entries per piece below this -> not worth it

// Popularity of the 32-column lines from every stride-th entry: the ranking only has to deal the lines to the
// blocks evenly, and counting all 3e8 entries of C3 took 42 ms of atomics on the hottest counters.
__global__ __launch_bounds__(kBlock) void pb_line_count_kernel(const int32_t *__restrict__ col, int64_t nnz,
                                                               int64_t stride, uint32_t *__restrict__ counts) {
  const int64_t samples = (nnz + stride - 1) / stride;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < samples; i += (int64_t)gridDim.x * kBlock)
    atomicAdd(&counts[col[i * stride] / kPbLine], 1u);
}

// Place q' of the line of popularity rank q: block q' mod nb, slot q' / nb.  Dealt (q' = q): the lines go round the
// blocks, every block holds about the same number of entries.  Sorted (q' = (q mod lpb) * nb + q / lpb): block 0 holds
// the lpb most popular lines, block 1 the next, ... -- a row then meets its entries in few blocks and pass 1 can add
// most of them up (folded layout below).  place_line[q'] = the line at place q' (0xffffffff: none).
__global__ __launch_bounds__(kBlock) void pb_rank_kernel(const uint32_t *__restrict__ line_of_rank, int64_t nl, int64_t nb,
                                                         int64_t lpb, int sorted, uint32_t *__restrict__ rank_of_line,
                                                         uint32_t *__restrict__ place_line) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= nl) return;
  const int64_t q2 = sorted ? (q % lpb) * nb + q / lpb : q;
  const uint32_t line = line_of_rank[q];
  rank_of_line[line] = (uint32_t)q2;
  place_line[q2] = line;
}

// key of entry e = (block * np + panel) * 32 + (row inside the panel) mod 32: block-major (pass 1 streams a block's
// entries as one range), and inside a piece the entries come grouped by the LDS bank pair their row's accumulator
// lives in (pb_build_kernel deals them out so that a group of 32 entries holds each bank pair about once)
__global__ __launch_bounds__(kBlock) void pb_key_kernel(const int64_t *__restrict__ row_ptr,
                                                        const int32_t *__restrict__ col, int64_t nnz,
                                                        const int64_t *__restrict__ tile_lb, int64_t tile,
                                                        const int64_t *__restrict__ panel_row, int64_t np,
                                                        const uint32_t *__restrict__ rank_of_line, int64_t nb,
                                                        uint32_t *__restrict__ keys) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= nnz) return;
  const int64_t r = find_row(row_ptr, tile_lb, tile, e);
  int64_t lo = 0, hi = np;  // largest p with panel_row[p] <= r
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (panel_row[mid] <= r) lo = mid; else hi = mid;
  }
  const int64_t q = rank_of_line[col[e] / kPbLine];
  keys[e] = (uint32_t)((((q % nb) * np + lo) << 5) | ((r - panel_row[lo]) & 31));
}

// piece_ptr[k] = first position whose key (shifted right by `shift` bits) is >= k, k = 0 .. pieces
__global__ __launch_bounds__(kBlock) void pb_piece_ptr_kernel(const uint32_t *__restrict__ sorted_keys, int64_t nnz,
                                                              int64_t pieces, int64_t *__restrict__ piece_ptr,
                                                              int shift) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k > pieces) return;
  int64_t lo = 0, hi = nnz;
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    if ((int64_t)(sorted_keys[mid] >> shift) < k) lo = mid + 1; else hi = mid;
  }
  piece_ptr[k] = lo;
}

// Padded piece sizes (multiples of 32 entries: every aligned group of 32 entries then belongs to one piece, in the
// block-major layout pass 1 reads and in the panel-major layout it writes) in both orders, for the two scans.
__global__ __launch_bounds__(kBlock) void pb_psz_kernel(const int64_t *__restrict__ piece_ptr, int64_t pieces, int64_t np,
                                                        int64_t nb, int64_t *__restrict__ boff, int64_t *__restrict__ poff) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k > pieces) return;
  if (k == pieces) { boff[k] = 0; poff[k] = 0; return; }
  const int64_t sz = (piece_ptr[k + 1] - piece_ptr[k] + 31) & ~(int64_t)31;
  const int64_t b = k / np, p = k - b * np;
  boff[k] = sz;
  poff[p * nb + b] = sz;
}

// Entry i of the sorted (block-major) list goes to its place in both padded layouts: column and value where pass 1
// streams them (block-major), row where pass 2 streams it (panel-major).  Pads keep the fill: value 0, row 0xFFFF.
template <class T>
__global__ __launch_bounds__(kBlock) void pb_build_kernel(const int64_t *__restrict__ row_ptr,
                                                          const int32_t *__restrict__ col, const T *__restrict__ val,
                                                          int64_t nnz, const int64_t *__restrict__ tile_lb,
                                                          int64_t tile, const int64_t *__restrict__ panel_row,
                                                          const uint32_t *__restrict__ rank_of_line, int64_t nb,
                                                          int64_t np, const uint32_t *__restrict__ sorted_keys,
                                                          const uint32_t *__restrict__ perm,
                                                          const int64_t *__restrict__ piece_ptr,
                                                          const int64_t *__restrict__ boff, const int64_t *__restrict__ poff,
                                                          uint16_t *__restrict__ col16, uint16_t *__restrict__ row16,
                                                          T *__restrict__ pval) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nnz) return;
  const int64_t src = perm ? (int64_t)perm[i] : i;  // no sort pass ran (one piece): identity
  const int64_t k = (int64_t)(sorted_keys[i] >> 5);
  const int64_t b = k / np, p = k - b * np;
  int64_t j = i - piece_ptr[k];
  const int64_t r = find_row(row_ptr, tile_lb, tile, src);
  const int32_t c = col[src];
  const int64_t q = rank_of_line[c / kPbLine];
  // The sorted piece lists its entries by (row mod 32, row, column).  A panel of few rows (heavy rows) keeps that
  // order -- equal rows adjacent -- and is flagged: pass 2 folds its runs inside the wavefront.  Every other piece
  // is dealt out over its G groups of 32, entry j to group j mod G: neighbours in the sorted order (same bank pair)
  // land in different groups, so one ds_add_f64 instruction sees each bank pair about once instead of ~4 times.
  int flag = 0;
  if (panel_row[p + 1] - panel_row[p] < kPbFoldRows) flag = kPbFoldFlag;
  else {
    const int64_t cnt = piece_ptr[k + 1] - piece_ptr[k], G = (cnt + 31) >> 5;
    const int64_t sl = j / G;
    j = (j % G) * 32 + ((sl & 1) * 16 + (sl >> 1));  // neighbours (possibly one bank pair twice) into different 16-lane halves
  }
  const int64_t db = boff[k] + j, dp = poff[p * nb + b] + j;
  col16[db] = (uint16_t)((q / nb) * kPbLine + (c % kPbLine));
  pval[db] = val[src];
  row16[dp] = (uint16_t)((r - panel_row[p]) | flag);
}

__global__ __launch_bounds__(kBlock) void pb_bounds_kernel(const int64_t *__restrict__ boff, const int64_t *__restrict__ poff,
                                                           int64_t np, int64_t nb, int64_t *__restrict__ bounds) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i <= nb) bounds[i] = boff[i * np];                       // boff has pieces + 1 = nb * np + 1 elements
  else if (i <= nb + 1 + np) bounds[i] = poff[(i - nb - 1) * nb];
}

// dgroup[g] = the group of 32 entries in the panel-major layout that receives group g of the block-major one
__global__ __launch_bounds__(kBlock) void pb_dgroup_kernel(const int64_t *__restrict__ boff, const int64_t *__restrict__ poff,
                                                           int64_t pieces, int64_t np, int64_t nb,
                                                           uint32_t *__restrict__ dgroup) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= pieces) return;
  const int64_t b = k / np, p = k - b * np;
  const int64_t g0 = boff[k] >> 5, g1 = boff[k + 1] >> 5, d0 = poff[p * nb + b] >> 5;
  for (int64_t g = g0; g < g1; g++) dgroup[g] = (uint32_t)(d0 + (g - g0));
}

// x in block order: the line at place q -> block q mod nb, slot q / nb; w = columns per block; nl = places
template <class T>
__global__ __launch_bounds__(kBlock) void pb_xperm_kernel(const T *__restrict__ x, int64_t m,
                                                          const uint32_t *__restrict__ line_of_rank, int64_t nl,
                                                          int64_t nb, int64_t w, T *__restrict__ xperm) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t q = t / kPbLine, j = t % kPbLine;
  if (q >= nl) return;
  const uint32_t line = line_of_rank[q];  // the line at place q
  const int64_t c = (int64_t)line * kPbLine + j;
  xperm[(q % nb) * w + (q / nb) * kPbLine + j] = (line != 0xffffffffu && c < m) ? x[c] : (T)0;
}

// the same in 16-byte pieces (x 16-byte aligned): a quarter (f32) or half (f64) of the threads and accesses
template <class T>
__global__ __launch_bounds__(kBlock) void pb_xperm_vec_kernel(const T *__restrict__ x, int64_t m,
                                                              const uint32_t *__restrict__ line_of_rank, int64_t nl,
                                                              int64_t nb, int64_t w, T *__restrict__ xperm) {
  constexpr int PER = 16 / (int)sizeof(T), PIECES = kPbLine / PER;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t q = t / PIECES, j = (t % PIECES) * PER;
  if (q >= nl) return;
  const uint32_t line = line_of_rank[q];  // the line at place q
  const int64_t c = (int64_t)line * kPbLine + j;
  T *dst = xperm + (q % nb) * w + (q / nb) * kPbLine + j;  // w and the places are multiples of 32 columns: 16-byte aligned
  if (line != 0xffffffffu && c + PER <= m) {
    *reinterpret_cast<raw16 *>(dst) = ntload(reinterpret_cast<const raw16 *>(x + c));
  } else {
#pragma unroll
    for (int u = 0; u < PER; u++) dst[u] = (line != 0xffffffffu && c + u < m) ? x[c + u] : (T)0;
  }
}

__device__ __forceinline__ int64_t cdiv_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

// value of lane src, the same in every lane and known to be so (scalar registers)
__device__ __forceinline__ int64_t pb_bcast64(int64_t v, int src) {
  const int lo = __builtin_amdgcn_readfirstlane(__shfl((int)(uint32_t)v, src, 64));
  const int hi = __builtin_amdgcn_readfirstlane(__shfl((int)(uint32_t)((uint64_t)v >> 32), src, 64));
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

template <class T, int U>
__global__ __launch_bounds__(kPbThreads) void pb_pass1_kernel(const int64_t *__restrict__ block_ptr,
                                                              const uint16_t *__restrict__ col16,
                                                              const T *__restrict__ pval, const T *__restrict__ xperm,
                                                              const uint32_t *__restrict__ dgroup,
                                                              T *__restrict__ prod, int w, int lab) {
  __shared__ __attribute__((aligned(16))) T xs[kPbSliceBytes / sizeof(T)];
  const int64_t b = blockIdx.x;
  // (tried: the slice straight from x through line_of_rank, no renumbered copy -- the scattered 128-byte lines in
  // front of the stream cost pass 1 0.11 ms where the separate copy kernel costs 0.06)
  {
    constexpr int PER = 16 / (int)sizeof(T);  // w is a multiple of 32 columns: whole 128-bit pieces
    const raw16 *src = reinterpret_cast<const raw16 *>(xperm + b * (int64_t)w);
    for (int j = threadIdx.x; j < w / PER; j += kPbThreads) reinterpret_cast<raw16 *>(xs)[j] = src[j];
  }
  // the block's entries (all its pieces, each padded to a multiple of 32) are one contiguous range; the gridDim.y
  // workgroups of a block take equal shares of it
  constexpr int64_t kStep = (int64_t)kPbThreads * U;
  const int64_t s0 = block_ptr[b], e0 = block_ptr[b + 1];
  const int64_t per = cdiv_dev(cdiv_dev(e0 - s0, kStep), (int64_t)gridDim.y) * kStep;
  const int64_t s = s0 + (int64_t)blockIdx.y * per;
  const int64_t e = s + per < e0 ? s + per : e0;
  __syncthreads();
  // Loads are never predicated: a load behind a branch makes the compiler wait for it at once (measured: the
  // predicated form of this loop ran at 2 TB/s).  Partial steps clamp their addresses instead.  Every range
  // starts at a multiple of 32 entries and a group of 32 entries has ONE destination group (dgroup), so every
  // half-wavefront store covers one whole 128-byte line of the panel-major product stream.
  for (int64_t i0 = s; i0 < e; i0 += kStep) {
    const bool full = i0 + kStep <= e;  // uniform
    uint16_t c[U];
    T v[U];
    uint32_t g[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t ii = i0 + (int64_t)u * kPbThreads + threadIdx.x;
      const int64_t ic = full ? ii : (ii < e ? ii : e - 1);
      c[u] = ntload(&col16[ic]);
      v[u] = ntload(&pval[ic]);
      g[u] = dgroup[ic >> 5];
    }
    T pr[U];
#pragma unroll
    for (int u = 0; u < U; u++) pr[u] = lab == 2 ? v[u] : (T)(v[u] * xs[c[u]]);  // lab 2: no LDS gather
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t ii = i0 + (int64_t)u * kPbThreads + threadIdx.x;
      if (full || ii < e) {
        const int64_t d = ((int64_t)g[u] << 5) | (ii & 31);
        if (lab == 3) prod[d] = pr[u];                                                      // lab 3: plain store
        else if (lab != 1 || pr[u] == (T)12345) __builtin_nontemporal_store(pr[u], &prod[d]);  // lab 1: no store
      }
    }
  }
}

// Adds v into ys[row] for the lanes of one wavefront whose rows come in runs (entries of a piece are sorted by
// row): a run is folded inside the wavefront and only its last lane touches LDS, so a heavy row costs one
// atomic per 64 entries instead of 64 colliding ones.  row < 0: lane holds nothing.
// The slice of y is accumulated in Acc<T>: double for float as well -- measured (bench/lds_lab.hip) ds_add_f32
// retires one lane every 3.0 cycles per CU, ds_add_f64 and ds_add_u32 one every 0.4 -- which also makes the sum
// of a float row insensitive to the order the atomics land in.
template <class T> struct PbAcc { typedef double type; };
template <> struct PbAcc<int32_t> { typedef int32_t type; };

template <class T>
__device__ __forceinline__ void pb_seg_add(typename PbAcc<T>::type *ys, int row, T v, int lane, int fold_max) {
  typedef typename PbAcc<T>::type acc_t;
  const int rprev = wshfl_up(row, 1);
  const bool head = lane == 0 || rprev != row;
  const unsigned long long heads = __ballot(head);
  // fold when at least half of the lanes continue their neighbour's row (heavy rows: atomics on one address are
  // serial, 3 cycles each); otherwise the few repeats are cheaper than the fold
  if (__popcll(heads) <= fold_max) {
    const unsigned long long below = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    const int head_pos = 63 - __builtin_clzll(below);  // lane 0 is always a head, so below != 0
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const T up = wshfl_up(v, d);
      if (lane - d >= head_pos) v = (T)(v + up);
    }
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    if (row >= 0 && tail) atomicAdd(&ys[row], (acc_t)v);
  } else if (row >= 0) {
    atomicAdd(&ys[row], (acc_t)v);
  }
}

// Workgroup = panel: its products (all blocks, panel-major layout) are ONE contiguous range, streamed by the 16
// wavefronts in chunks of 64 U entries; chunk i + 1 is requested before chunk i is added.  Pads carry the row
// number of a trash slot behind the slice, so nothing in the loop looks at them.
// The loop is bound by the LDS atomics: a ds_add_f64 whose 64 lanes hold random rows runs ~109 cycles (27
// conflict-free: 32 lanes into 32 bank pairs collide ~4 deep), 0.87 ms for C3's 3e8 entries.  So the entries of a
// piece are stored dealt out by bank pair (pb_build_kernel): an instruction then sees each bank pair about once.
// Panels of a few heavy rows are the exception: same-address atomics serialise whatever the banks, so those keep
// their rows adjacent and fold the runs inside the wavefront first (flag bit in row16).
template <class T> struct PbSlice {  // rows of y a panel holds; at most 2^14 so that the trash slot's number leaves bit 15 free
  static constexpr int kFit = kPbSliceBytes / (int)sizeof(typename PbAcc<T>::type);
  static constexpr int kRows = kFit < 16384 ? kFit : 16384;
};

template <class V>
__device__ __forceinline__ V pb_wave_sum(V v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = (V)(v + wshfl_xor(v, d));
  return v;
}

// pads of the panel-major stream: rows kRows .. kRows + 31 (the trash slots behind the slice), by position
__global__ __launch_bounds__(kBlock) void pb_fill_trash_kernel(uint16_t *__restrict__ row16, int64_t n, int base) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) row16[i] = (uint16_t)(base + (int)(i & 31));
}

template <class T, int U>
__global__ __launch_bounds__(kPbThreads) void pb_pass2_kernel(const int64_t *__restrict__ panel_ptr,
                                                              const int64_t *__restrict__ panel_row,
                                                              const uint16_t *__restrict__ row16,
                                                              const T *__restrict__ prod, T *__restrict__ y,
                                                              int lab, int fold_max) {
  typedef typename PbAcc<T>::type acc_t;
  constexpr int kTrash = PbSlice<T>::kRows;
  __shared__ acc_t ys[kTrash + 32];  // 32 trash slots: the pads of one instruction land in different bank pairs
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t p = blockIdx.x;
  const int64_t r0 = panel_row[p];
  const int rows = (int)(panel_row[p + 1] - r0);
  for (int j = tid; j < rows; j += kPbThreads) ys[j] = (acc_t)0;
  __syncthreads();
  const int64_t ps = panel_ptr[p], pe = panel_ptr[p + 1];
  constexpr int64_t kChunk = 64 * U, kStride = (int64_t)kPbWaves * kChunk;
  auto load = [&](int64_t cs, int (&r)[U], T (&v)[U]) {
    const uint16_t *rp = row16 + cs;
    const T *vp = prod + cs;
    if (cs + kChunk <= pe) {  // uniform
#pragma unroll
      for (int u = 0; u < U; u++) {
        r[u] = (int)ntload(rp + lane + 64 * u);
        v[u] = ntload(vp + lane + 64 * u);
      }
    } else {  // the panel's last chunk: clamped offsets (never predicated loads), lanes beyond the end go to the trash slot
      const int n = (int)(pe - cs);
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int o = lane + 64 * u;
        const int oc = o < n ? o : n - 1;
        const int rr = (int)ntload(rp + oc);
        r[u] = o < n ? rr : kTrash;
        v[u] = ntload(vp + oc);
      }
    }
  };
  auto add = [&](const int (&r)[U], const T (&v)[U]) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (lab == 1) { if (r[u] == 12345 && v[u] == (T)77) ys[0] = (acc_t)v[u]; continue; }  // lab: stream only
      if (fold_max > 0 && __ballot(r[u] & kPbFoldFlag) != 0ull)  // a panel of few, heavy rows: fold the runs first
        pb_seg_add<T>(ys, r[u] & (kPbFoldFlag - 1), v[u], lane, fold_max < 64 ? 64 : fold_max);
      else
        atomicAdd(&ys[r[u] & (kPbFoldFlag - 1)], (acc_t)v[u]);
    }
  };
  int64_t cs = ps + (int64_t)wave * kChunk;
  if (cs < pe) {
    int ra[U], rb[U];
    T va[U], vb[U];
    load(cs, ra, va);
    for (;;) {
      const int64_t c1 = cs + kStride;
      if (c1 < pe) load(c1, rb, vb);
      add(ra, va);
      if (c1 >= pe) break;
      const int64_t c2 = c1 + kStride;
      if (c2 < pe) load(c2, ra, va);
      add(rb, vb);
      if (c2 >= pe) break;
      cs = c2;
    }
  }
  __syncthreads();
  for (int j2 = tid; j2 < rows; j2 += kPbThreads) y[r0 + j2] = (T)ys[j2];
}

// ------------------------------------------------------------------------------------------------ folded layout
// On a power-law matrix a heavy row has many entries in one column block: their products belong to one element of
// y, and pass 1 can add them up before anything is written.  C3: a row of 1300 entries meets each of the 512 blocks
// 2.5 times, the 13 000 heaviest rows (28 % of the entries) 8 times and more; with the lines sorted into the blocks
// by popularity 72 % of the products disappear.
// Layout (round 3, third form).  A piece's entries are grouped in RUNS: a row's n entries inside the piece are cut
// into runs of 32 and then along the bits of what is left (13 = 8 + 4 + 1), so a run is 1, 2, 4, 8, 16 or 32 entries
// long.  The runs of a piece are sorted by length: a CLASS = the runs of one piece with one length L, its entries
// stored run after run, its sums run after run in the panel-major output stream.  Pass 1 walks a list of STEPS of
// up to 256 entries of one class: lane l loads its four entries as one 8-byte and one 16-byte piece, multiplies,
// adds up what belongs to one run -- 4 / L runs inside the lane for L <= 4, L / 4 neighbouring lanes by DPP moves
// above that -- and the step's sums leave as one contiguous store.  Regular by construction: no run boundaries to
// find, no step that depends on another, any step range can be any wavefront's.
// What led here (C3, pass 1 alone): runs as neighbouring lanes of one instruction, summed by segmented scans, a
// run's last lane storing: 0.53 ms, bound by its 36 vector instructions per 64 entries and by stores of a third of a
// wavefront; lane = run with the entries as jagged diagonals (64 entries per step, two 2- and 4-byte loads): 0.58 ms,
// bound by the number of vector-memory instructions (a CU retires one per ~16 cycles whatever its width).
// Among the runs of a class the order is the bank-pair deal of the plain layout: pass 2's LDS atomics of one
// instruction then meet each bank pair about once.
// desc[step] = {first entry, first output, lanes | class << 8, -}; class c: L = 32 >> c.
constexpr int kPbClasses = 6;
constexpr int kPbStepEntries = 256;  // 64 lanes x 4 entries

struct PbRunHeadIn {  // 1 where sorted entry i starts a row's entries inside a piece
  const uint32_t *keys;
  const uint16_t *rr;
  __device__ __forceinline__ uint32_t operator()(int64_t i) const {
    return (i == 0 || (keys[i] >> 5) != (keys[i - 1] >> 5) || rr[i] != rr[i - 1]) ? 1u : 0u;
  }
};

struct PbSubHeadIn {  // 1 where a run starts: n entries of a row = runs of 32, then the bits of n mod 32 from the top
  const uint32_t *id0;         // inclusive sums of PbRunHeadIn
  const uint32_t *row_start;   // first entry of every row-in-piece, then nnz
  __device__ __forceinline__ uint32_t operator()(int64_t i) const {
    const uint32_t id = id0[i] - 1, s = row_start[id], n = row_start[id + 1] - s, t = (uint32_t)i - s;
    const uint32_t full = n & ~(uint32_t)(kPbRunCap - 1);
    if (t < full) return (t & (kPbRunCap - 1)) == 0 ? 1u : 0u;
    uint32_t pos = full;
#pragma unroll
    for (uint32_t bit = kPbRunCap / 2; bit >= 1; bit >>= 1)
      if (n & bit) {
        if (t < pos + bit) return t == pos ? 1u : 0u;
        pos += bit;
      }
    return 1u;  // not reached
  }
};

__global__ __launch_bounds__(kBlock) void pbf_rows_kernel(const int64_t *__restrict__ row_ptr, int64_t nnz,
                                                          const int64_t *__restrict__ tile_lb, int64_t tile,
                                                          const int64_t *__restrict__ panel_row, int64_t np,
                                                          const uint32_t *__restrict__ sorted_keys,
                                                          const uint32_t *__restrict__ perm, uint16_t *__restrict__ rr) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nnz) return;
  const int64_t src = perm ? (int64_t)perm[i] : i;
  const int64_t p = (int64_t)(sorted_keys[i] >> 5) % np;
  rr[i] = (uint16_t)(find_row(row_ptr, tile_lb, tile, src) - panel_row[p]);
}

// start[j] = position of the j-th head, start[heads] = n (hscan: inclusive sums of the head flags)
template <class HeadIn>
__global__ __launch_bounds__(kBlock) void pbf_starts_kernel(HeadIn head, const uint32_t *__restrict__ hscan, int64_t n,
                                                            uint32_t *__restrict__ start) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (head(i)) start[hscan[i] - 1] = (uint32_t)i;
  if (i == n - 1) start[hscan[i]] = (uint32_t)n;
}

// sort key of run j: its piece, then its class (longest first)
__global__ __launch_bounds__(kBlock) void pbf_runkey_kernel(const uint32_t *__restrict__ sorted_keys,
                                                            const uint32_t *__restrict__ run_start, int64_t runs,
                                                            uint32_t *__restrict__ rk) {
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= runs) return;
  const uint32_t i0 = run_start[j], len = run_start[j + 1] - i0;  // 1, 2, 4, .. 32
  rk[j] = (sorted_keys[i0] & ~31u) | (uint32_t)(__clz(len) - 26);
}

// cstart[e] = first sorted run of class e = piece * 6 + c (e = 0 .. 6 pieces; the last = runs)
__global__ __launch_bounds__(kBlock) void pbf_class_start_kernel(const uint32_t *__restrict__ rks, int64_t runs,
                                                                 int64_t classes, uint32_t *__restrict__ cstart) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e > classes) return;
  const int64_t key = ((e / kPbClasses) << 5) | (e % kPbClasses);
  int64_t lo = 0, hi = runs;
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    if ((int64_t)rks[mid] < key) lo = mid + 1; else hi = mid;
  }
  cstart[e] = (uint32_t)lo;
}

// per class: its runs rounded up so that every lane's four entries are whole runs (phantom runs: value 0, trash
// row), its entries, its steps; element `classes` of each = 0 for the exclusive scans
__global__ __launch_bounds__(kBlock) void pbf_class_kernel(const uint32_t *__restrict__ cstart, int64_t classes,
                                                           uint32_t *__restrict__ cls_runs, uint32_t *__restrict__ cls_ent,
                                                           uint32_t *__restrict__ cls_steps) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e > classes) return;
  uint32_t padded = 0, ent = 0;
  if (e < classes) {
    const uint32_t L = (uint32_t)kPbRunCap >> (e % kPbClasses), per = L >= 4 ? 1u : 4u / L;
    padded = (cstart[e + 1] - cstart[e] + per - 1) / per * per;
    ent = padded * L;
  }
  cls_runs[e] = padded;
  cls_ent[e] = ent;
  cls_steps[e] = (ent + kPbStepEntries - 1) / kPbStepEntries;
}

// per piece: where each of its classes' outputs begin inside it; poff[panel-major k] = its outputs, rounded up to
// whole groups of 32
__global__ __launch_bounds__(kBlock) void pbf_piece_kernel(const uint32_t *__restrict__ cls_runs, int64_t pieces, int64_t np,
                                                           int64_t nb, uint32_t *__restrict__ cls_out,
                                                           int64_t *__restrict__ poff) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k > pieces) return;
  if (k == pieces) { poff[k] = 0; return; }
  uint32_t o = 0;
#pragma unroll
  for (int c = 0; c < kPbClasses; c++) {
    cls_out[k * kPbClasses + c] = o;
    o += cls_runs[k * kPbClasses + c];
  }
  const int64_t b = k / np, p = k - b * np;
  poff[p * nb + b] = ((int64_t)o + 31) & ~(int64_t)31;
}

// The run at sorted place q gets its place in its class: inside a class of N runs (which the stable sort left
// ordered by row mod 32, then row) the deal of pb_build_kernel -- G = ceil(N / 32) columns, run jl to column
// jl mod G, the columns one behind the other -- so that 32 neighbouring outputs belong to rows of distinct row mod 32.
__global__ __launch_bounds__(kBlock) void pbf_deal_kernel(const uint32_t *__restrict__ rks, const uint32_t *__restrict__ rperm,
                                                          const uint32_t *__restrict__ cstart, int64_t runs,
                                                          const uint32_t *__restrict__ run_start,
                                                          const uint16_t *__restrict__ rr, const uint32_t *__restrict__ ebase,
                                                          const uint32_t *__restrict__ cls_out,
                                                          const int64_t *__restrict__ poff, int64_t np, int64_t nb,
                                                          uint32_t *__restrict__ run_ent, uint16_t *__restrict__ row16) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= runs) return;
  const int64_t k = (int64_t)(rks[q] >> 5), c = (int64_t)(rks[q] & 31), e = k * kPbClasses + c;
  const int64_t s = cstart[e], N = (int64_t)cstart[e + 1] - s, jl = q - s;
  const int64_t G = (N + 31) >> 5, col = jl % G, per = N / G, rem = N % G;
  int64_t sl = jl / G;
  if (per + (col < rem ? 1 : 0) == 32) sl = (sl & 1) * 16 + (sl >> 1);  // a full column: neighbours (possibly one bank pair twice) into different 16-lane halves
  const int64_t place = col * per + (col < rem ? col : rem) + sl;
  const uint32_t j = rperm[q];
  const int64_t b = k / np, p = k - b * np;
  run_ent[j] = ebase[e] + (uint32_t)(place * (kPbRunCap >> c));
  row16[poff[p * nb + b] + (int64_t)cls_out[e] + place] = rr[run_start[j]];
}

template <class T>
__global__ __launch_bounds__(kBlock) void pbf_place_kernel(const int32_t *__restrict__ col, const T *__restrict__ val,
                                                           int64_t nnz, const uint32_t *__restrict__ perm,
                                                           const uint32_t *__restrict__ rank_of_line, int64_t nb,
                                                           const uint32_t *__restrict__ hscan,
                                                           const uint32_t *__restrict__ run_start,
                                                           const uint32_t *__restrict__ run_ent, uint16_t *__restrict__ colj,
                                                           T *__restrict__ valj) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nnz) return;
  const int64_t src = perm ? (int64_t)perm[i] : i;
  const uint32_t j = hscan[i] - 1;
  const int64_t dst = (int64_t)run_ent[j] + (i - (int64_t)run_start[j]);
  const int32_t c = col[src];
  const int64_t q = rank_of_line[c / kPbLine];
  colj[dst] = (uint16_t)((q / nb) * kPbLine + (c % kPbLine));
  valj[dst] = val[src];
}

// the steps of every class
__global__ __launch_bounds__(kBlock) void pbf_desc_kernel(const uint32_t *__restrict__ cls_ent, const uint32_t *__restrict__ ebase,
                                                          const uint32_t *__restrict__ stbase,
                                                          const uint32_t *__restrict__ cls_out, const int64_t *__restrict__ poff,
                                                          int64_t classes, int64_t np, int64_t nb, uint32_t *__restrict__ desc) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= classes) return;
  const uint32_t ent = cls_ent[e];
  if (ent == 0) return;
  const int64_t k = e / kPbClasses, c = e % kPbClasses, b = k / np, p = k - b * np;
  const uint32_t ob = (uint32_t)(poff[p * nb + b] + (int64_t)cls_out[e]), L = (uint32_t)kPbRunCap >> c;
  uint32_t *d = desc + 4 * (int64_t)stbase[e];
  for (uint32_t done = 0; done < ent; done += kPbStepEntries, d += 4) {
    const uint32_t left = ent - done;
    d[0] = ebase[e] + done;
    d[1] = ob + done / L;
    d[2] = (left >= (uint32_t)kPbStepEntries ? 64u : left / 4) | ((uint32_t)c << 8);
    d[3] = 0;
  }
}

// bounds[0 .. nb]: block b's first step; then [0 .. np]: panel p's first output
__global__ __launch_bounds__(kBlock) void pbf_bounds_kernel(const uint32_t *__restrict__ stbase, const int64_t *__restrict__ poff,
                                                            int64_t np, int64_t nb, int64_t *__restrict__ bounds) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i <= nb) bounds[i] = (int64_t)stbase[i * np * kPbClasses];
  else if (i <= nb + 1 + np) bounds[i] = poff[(i - nb - 1) * nb];
}

// ---- pass 1 of the folded layout: workgroup = kPbWaves consecutive wavefront ranges of one column block (its slice
// of x in LDS), wavefront = a range of steps, walked a batch at a time; the descriptors of batch g + 2 and the
// entries of batch g + 1 are requested before batch g is summed.  Every step loads 256 entries whatever its count
// (behind the array: slack), so no load is predicated; the stores go through a buffer whose range is the step's
// outputs, so the lanes behind them are dropped by the address check, not by a branch.
typedef unsigned int pb_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int pb_u32x4 __attribute__((ext_vector_type(4)));
constexpr int kPbRsrcFlags = 0x00020000;  // gfx9 raw buffer, 32-bit data format

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pb_rsrc(const void *base, int bytes = 0x7ffffffe) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, kPbRsrcFlags);
}

template <int CTRL, class T>
__device__ __forceinline__ T pb_dpp0(T v) {  // lanes without a source read 0
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
  } else {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(T, ((uint64_t)hi << 32) | lo);
  }
}

template <class T>
struct PbStepRegs {  // a lane's four entries of one step
  pb_u32x2 c;                       // four 16-bit columns
  pb_u32x4 v[sizeof(T) / 4];        // four values
  __device__ __forceinline__ uint32_t col(int k) const { return (k & 1) ? c[k >> 1] >> 16 : c[k >> 1] & 0xffffu; }
  __device__ __forceinline__ T val(int k) const {
    if constexpr (sizeof(T) == 4) {
      const uint32_t w = v[0][k];  // (a bit cast of the element expression itself reads element 0: clang 20)
      return __builtin_bit_cast(T, w);
    } else {
      const uint64_t w = ((uint64_t)v[k >> 1][(k & 1) * 2 + 1] << 32) | v[k >> 1][(k & 1) * 2];
      return __builtin_bit_cast(T, w);
    }
  }
};

template <int AUX, int N, class T>
__device__ __forceinline__ void pb_bstore_n(const T (&x)[N], __amdgpu_buffer_rsrc_t r, int off) {  // N consecutive elements
  constexpr int W = N * (int)sizeof(T) / 4;  // dwords: 1, 2, 4 or 8
  uint32_t w[W];
#pragma unroll
  for (int k = 0; k < N; k++) {
    if constexpr (sizeof(T) == 4) {
      w[k] = __builtin_bit_cast(uint32_t, x[k]);
    } else {
      const uint64_t u = __builtin_bit_cast(uint64_t, x[k]);
      w[2 * k] = (uint32_t)u;
      w[2 * k + 1] = (uint32_t)(u >> 32);
    }
  }
  if constexpr (W == 1) {
    __builtin_amdgcn_raw_buffer_store_b32(w[0], r, off, 0, AUX);
  } else if constexpr (W == 2) {
    const pb_u32x2 t = {w[0], w[1]};
    __builtin_amdgcn_raw_buffer_store_b64(t, r, off, 0, AUX);
  } else {
#pragma unroll
    for (int h = 0; h < W / 4; h++) {
      const pb_u32x4 t = {w[4 * h], w[4 * h + 1], w[4 * h + 2], w[4 * h + 3]};
      __builtin_amdgcn_raw_buffer_store_b128(t, r, off + 16 * h, 0, AUX);
    }
  }
}

// LAB (kernel lab; results wrong for 1 and 2): 1 no store, 2 no LDS gather, 3 non-temporal stores
template <class T, int LAB>
__global__ __launch_bounds__(kPbThreads) void pb_pass1f_kernel(const int64_t *__restrict__ wtab,
                                                               const uint32_t *__restrict__ desc,
                                                               const uint16_t *__restrict__ colj,
                                                               const T *__restrict__ valj, const T *__restrict__ xperm,
                                                               T *__restrict__ prod, int w) {
  __shared__ __attribute__((aligned(16))) T xs[kPbSliceBytes / sizeof(T)];
  constexpr int B = sizeof(T) == 4 ? 4 : 2;  // steps per batch
  constexpr int AUXS = LAB == 3 ? 2 : 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t r = (int64_t)blockIdx.x * kPbWaves + wave;
  const int64_t b = wtab[2 * (int64_t)blockIdx.x * kPbWaves + 1];
  const int64_t step0 = wtab[2 * r], step1 = wtab[2 * r + 2];
  const int nsteps = (int)(step1 - step0), nbat = (nsteps + B - 1) / B;
  // descriptors through a range-checked buffer: the steps behind the range read as zeros, i.e. no lanes
  const __amdgpu_buffer_rsrc_t rd = pb_rsrc(desc + 4 * step0, nsteps > 0 ? nsteps * 16 : 0);
  // batch g = step g of each of the range's B quarters: steps that are neighbours in memory share a line of each
  // array, and a load that meets a line still on its way stalls the L1 (TCP_PENDING_STALL_CYCLES); this way they are
  // a whole batch apart
  auto dload = [&](int g) -> pb_u32x4 {  // lane u (mod B) gets step u nbat + g
    const int idx = g < nbat ? (lane & (B - 1)) * nbat + g : nsteps;  // behind the range: zeros
    return __builtin_amdgcn_raw_buffer_load_b128(rd, idx * 16, 0, 0);
  };
  auto issue = [&](const pb_u32x4 d, PbStepRegs<T> (&e)[B]) {
#pragma unroll
    for (int u = 0; u < B; u++) {
      // a buffer per step whose range ends with the step's last lane: the lanes behind it read zeros and ask for
      // nothing (asking anyway -- the next step's lines -- made that step's own loads wait on lines already
      // pending: TCP_PENDING_STALL_CYCLES as large as the kernel itself)
      const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)d.x, u);
      const int lanes = __builtin_amdgcn_readlane((int)d.z, u) & 0xff;
      const __amdgpu_buffer_rsrc_t rc = pb_rsrc(colj + off, lanes * 8), rv = pb_rsrc(valj + off, lanes * 4 * (int)sizeof(T));
      e[u].c = __builtin_amdgcn_raw_buffer_load_b64(rc, lane * 8, 0, 2);  // aux 2: non-temporal
#pragma unroll
      for (int h = 0; h < (int)sizeof(T) / 4; h++)
        e[u].v[h] = __builtin_amdgcn_raw_buffer_load_b128(rv, lane * 4 * (int)sizeof(T) + 16 * h, 0, 2);
    }
  };
  auto consume = [&](const pb_u32x4 d, const PbStepRegs<T> (&e)[B]) {
    T p[B][4];
#pragma unroll
    for (int u = 0; u < B; u++)
#pragma unroll
      for (int k = 0; k < 4; k++) p[u][k] = LAB == 2 ? e[u].val(k) : (T)(e[u].val(k) * xs[e[u].col(k)]);
#pragma unroll
    for (int u = 0; u < B; u++) {
      const uint32_t ob = (uint32_t)__builtin_amdgcn_readlane((int)d.y, u);
      const int meta = __builtin_amdgcn_readlane((int)d.z, u);
      const int lanes = meta & 0xff, c = meta >> 8;  // class c: runs of 32 >> c entries
      if (LAB == 1 && p[u][0] != (T)12345) continue;
      // the step's outputs: 4 lanes / L of them; the lanes behind them are out of the buffer's range and dropped
      const __amdgpu_buffer_rsrc_t rp = pb_rsrc(prod + ob, ((lanes * 4) >> (5 - c)) * (int)sizeof(T));
      if (c == 5) {
        pb_bstore_n<AUXS, 4>(p[u], rp, lane * 4 * (int)sizeof(T));
      } else if (c == 4) {
        const T s2[2] = {(T)(p[u][0] + p[u][1]), (T)(p[u][2] + p[u][3])};
        pb_bstore_n<AUXS, 2>(s2, rp, lane * 2 * (int)sizeof(T));
      } else {
        T s1[1] = {(T)((T)(p[u][0] + p[u][1]) + (T)(p[u][2] + p[u][3]))};
        if (c == 3) {
          pb_bstore_n<AUXS, 1>(s1, rp, lane * (int)sizeof(T));
        } else {  // a run is 2, 4 or 8 neighbouring lanes: its last lane collects
          s1[0] = (T)(s1[0] + pb_dpp0<0x111>(s1[0]));                // row_shr:1
          if (c <= 1) s1[0] = (T)(s1[0] + pb_dpp0<0x112>(s1[0]));    // row_shr:2
          if (c == 0) s1[0] = (T)(s1[0] + pb_dpp0<0x114>(s1[0]));    // row_shr:4
          const int per = 8 >> c;  // lanes per run
          if ((lane & (per - 1)) == per - 1) pb_bstore_n<AUXS, 1>(s1, rp, (lane >> (3 - c)) * (int)sizeof(T));
        }
      }
    }
  };
  pb_u32x4 d0 = dload(0), d1 = dload(1), d2 = dload(2);
  {
    constexpr int PER = 16 / (int)sizeof(T);
    const raw16 *src = reinterpret_cast<const raw16 *>(xperm + b * (int64_t)w);
    for (int j = threadIdx.x; j < w / PER; j += kPbThreads) reinterpret_cast<raw16 *>(xs)[j] = src[j];
  }
  PbStepRegs<T> ea[B], eb[B];
  issue(d0, ea);
  __syncthreads();
  for (int g = 0; g < nbat; g += 2) {
    const pb_u32x4 d3 = dload(g + 3);
    issue(d1, eb);
    consume(d0, ea);
    const pb_u32x4 d4 = dload(g + 4);
    issue(d2, ea);
    consume(d1, eb);
    d0 = d2;
    d1 = d3;
    d2 = d4;
  }
}

// Builds the folded layout from the entries sorted by (piece, row mod 32, row, column).  *done = false (and nothing
// kept) when folding would not shorten the product stream enough, or the sizes leave the 32-bit indices.
template <class T>
int pbf_build(futhark_context *ctx, const CsrBody<T> *a, const uint32_t *eks, const uint32_t *eperm,
              const int64_t *piece_ptr, const int64_t *panel_row, const uint32_t *rank_of_line, int64_t nb, int64_t np,
              int64_t pieces, bool sorted, bool *done) {
  *done = false;
  const int64_t nnz = a->nnz;
  const unsigned egrid = (unsigned)cdiv(nnz, kBlock);
  int *sort_err = reinterpret_cast<int *>(ctx->d_small + 2);
  Tmp<uint16_t> rr(ctx);
  Tmp<uint32_t> hscan(ctx), run_start(ctx);
  SAMD_TRY(rr.get(nnz));
  SAMD_TRY(hscan.get(nnz));
  hipLaunchKernelGGL(pbf_rows_kernel, dim3(egrid), dim3(kBlock), 0, ctx->stream, (const int64_t *)a->row_ptr, nnz,
                     (const int64_t *)a->tile_lb, a->tile, panel_row, np, eks, eperm, rr.p);
  SAMD_LAUNCH_CHECK(ctx);
  uint32_t *d_runs = reinterpret_cast<uint32_t *>(ctx->d_small + 4);
  int64_t small[8];
  int64_t runs = 0;
  {
    // ---- a row's entries inside a piece, then the runs they are cut into
    Tmp<uint32_t> id0(ctx), row_start(ctx);
    SAMD_TRY(id0.get(nnz));
    const PbRunHeadIn head0{eks, rr.p};
    SAMD_TRY((device_scan<uint32_t, OpAdd, PbRunHeadIn, false>(ctx, head0, nnz, 0u, OpAdd(), id0.p, d_runs)));
    SAMD_TRY(read_small(ctx, ctx->d_small, 5, small));
    const int64_t rows0 = (int64_t)(uint32_t)(small[4] & 0xffffffff);
    if (ctx->logging) fprintf(stderr, "[sparse_amd] two-pass SpMV: %lld rows-in-pieces for %lld entries\n", (long long)rows0, (long long)nnz);
    // not worth the fold?  (With sorted blocks there is no other layout: the plain pass 1 splits every block into the
    // same number of shares.)
    if (!sorted && ctx->opt_spmv_pb_fold <= 0 && (double)rows0 > 0.85 * (double)nnz) return 0;
    SAMD_TRY(row_start.get(rows0 + 1));
    hipLaunchKernelGGL((pbf_starts_kernel<PbRunHeadIn>), dim3(egrid), dim3(kBlock), 0, ctx->stream, head0, (const uint32_t *)id0.p,
                       nnz, row_start.p);
    SAMD_LAUNCH_CHECK(ctx);
    const PbSubHeadIn head{id0.p, row_start.p};
    SAMD_TRY((device_scan<uint32_t, OpAdd, PbSubHeadIn, false>(ctx, head, nnz, 0u, OpAdd(), hscan.p, d_runs)));
    SAMD_TRY(read_small(ctx, ctx->d_small, 5, small));
    runs = (int64_t)(uint32_t)(small[4] & 0xffffffff);
    if (runs + 64 * pieces >= (int64_t)0xffffff00LL) return 0;  // 32-bit places and outputs
    SAMD_TRY(run_start.get(runs + 1));
    hipLaunchKernelGGL((pbf_starts_kernel<PbSubHeadIn>), dim3(egrid), dim3(kBlock), 0, ctx->stream, head,
                       (const uint32_t *)hscan.p, nnz, run_start.p);
    SAMD_LAUNCH_CHECK(ctx);
    SAMD_TRY(read_small(ctx, ctx->d_small, 1, small));  // id0 and row_start are released after their use
  }
  // ---- the runs sorted by (piece, class); classes, their entries, steps and outputs
  const int64_t classes = pieces * kPbClasses;
  const unsigned cgrid = (unsigned)cdiv(classes + 1, kBlock), rgrid = (unsigned)cdiv(runs, kBlock);
  Tmp<uint32_t> rk_a(ctx), rk_b(ctx), rp_a(ctx), rp_b(ctx), cstart(ctx), cls_runs(ctx), cls_ent(ctx), ebase(ctx), stbase(ctx),
      cls_out(ctx), run_ent(ctx);
  Tmp<int64_t> poff(ctx), bounds(ctx);
  SAMD_TRY(rk_a.get(runs));
  SAMD_TRY(rk_b.get(runs));
  SAMD_TRY(rp_a.get(runs));
  SAMD_TRY(rp_b.get(runs));
  SAMD_TRY(cstart.get(classes + 1));
  SAMD_TRY(cls_runs.get(classes + 1));
  SAMD_TRY(cls_ent.get(classes + 1));
  SAMD_TRY(ebase.get(classes + 1));
  SAMD_TRY(stbase.get(classes + 1));
  SAMD_TRY(cls_out.get(classes + 1));
  SAMD_TRY(run_ent.get(runs));
  SAMD_TRY(poff.get(pieces + 1));
  SAMD_TRY(bounds.get(nb + 1 + np + 1));
  hipLaunchKernelGGL(pbf_runkey_kernel, dim3(rgrid), dim3(kBlock), 0, ctx->stream, eks, (const uint32_t *)run_start.p, runs, rk_a.p);
  SAMD_LAUNCH_CHECK(ctx);
  uint32_t *rks = nullptr;
  const uint32_t *rperm = nullptr;
  SAMD_TRY((radix_sort_pairs<uint32_t, uint32_t>(ctx, rk_a.p, rk_b.p, (const uint32_t *)nullptr, rp_a.p, rp_b.p, runs, 0,
                                                  bits_for(pieces) + 5, &rks, &rperm, sort_err)));
  hipLaunchKernelGGL(pbf_class_start_kernel, dim3(cgrid), dim3(kBlock), 0, ctx->stream, (const uint32_t *)rks, runs, classes,
                     cstart.p);
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(pbf_class_kernel, dim3(cgrid), dim3(kBlock), 0, ctx->stream, (const uint32_t *)cstart.p, classes, cls_runs.p,
                     cls_ent.p, stbase.p);
  SAMD_LAUNCH_CHECK(ctx);
  uint32_t *d_tot = reinterpret_cast<uint32_t *>(ctx->d_small);  // [0]: entries (64-bit slot 0), [2]: steps (64-bit slot 1)
  SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, 2 * sizeof(int64_t), ctx->stream));
  SAMD_TRY((device_scan<uint32_t, OpAdd, PtrIn<uint32_t>, true>(ctx, PtrIn<uint32_t>{cls_ent.p}, classes + 1, 0u, OpAdd(), ebase.p,
                                                                 d_tot)));
  SAMD_TRY((device_scan<uint32_t, OpAdd, PtrIn<uint32_t>, true>(ctx, PtrIn<uint32_t>{stbase.p}, classes + 1, 0u, OpAdd(),
                                                                 stbase.p, d_tot + 2)));
  hipLaunchKernelGGL(pbf_piece_kernel, dim3((unsigned)cdiv(pieces + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                     (const uint32_t *)cls_runs.p, pieces, np, nb, cls_out.p, poff.p);
  SAMD_LAUNCH_CHECK(ctx);
  SAMD_TRY((device_scan<int64_t, OpAdd, PtrIn<int64_t>, true>(ctx, PtrIn<int64_t>{poff.p}, pieces + 1, (int64_t)0, OpAdd(),
                                                               poff.p, (int64_t *)nullptr)));
  SAMD_HIP(ctx, hipMemcpyAsync(ctx->d_small + 3, poff.p + pieces, sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
  SAMD_TRY(read_small(ctx, ctx->d_small, 4, small));
  if ((int)(small[2] & 0xffffffff) != 0) return set_error(ctx, "internal: radix sort look-back timed out");
  const int64_t etotal = (int64_t)(uint32_t)(small[0] & 0xffffffff), nsteps = (int64_t)(uint32_t)(small[1] & 0xffffffff);
  const int64_t ototal = small[3];  // entries (phantom runs included), steps, outputs (pads included)
  if (ctx->logging)
    fprintf(stderr, "[sparse_amd] two-pass SpMV, folded: %lld runs, %lld entries laid out, %lld steps (%.0f entries per step), %lld products\n",
            (long long)runs, (long long)etotal, (long long)nsteps, (double)etotal / (double)(nsteps > 0 ? nsteps : 1), (long long)ototal);
  // 32-bit byte offsets in pass 1 (a wrapped 32-bit entry total shows as fewer entries than the record has)
  if (ototal >= (int64_t)0x7fffff00LL || etotal < nnz || (etotal + 1024) * (int64_t)sizeof(T) >= (int64_t)0xffffff00LL) return 0;
  // ---- the layout itself
  constexpr int64_t kSlack = 512;  // a step loads 256 entries whatever its count
  Tmp<uint32_t> desc(ctx);
  Tmp<uint16_t> colj(ctx), row16(ctx);
  Tmp<T> valj(ctx), prod(ctx);
  SAMD_TRY(desc.get(4 * nsteps));
  SAMD_TRY(colj.get(etotal + kSlack));
  SAMD_TRY(valj.get(etotal + kSlack));
  SAMD_TRY(row16.get(ototal));
  SAMD_TRY(prod.get(ototal));
  SAMD_HIP(ctx, hipMemsetAsync(colj.p, 0, sizeof(uint16_t) * (size_t)(etotal + kSlack), ctx->stream));  // phantom runs: column 0, value 0
  SAMD_HIP(ctx, hipMemsetAsync(valj.p, 0, sizeof(T) * (size_t)(etotal + kSlack), ctx->stream));
  SAMD_HIP(ctx, hipMemsetAsync(prod.p, 0, sizeof(T) * (size_t)ototal, ctx->stream));  // a piece's pads (up to the next group of 32) are read by pass 2, into the trash slots
  if (ototal > 0) {
    hipLaunchKernelGGL(pb_fill_trash_kernel, dim3((unsigned)cdiv(ototal, kBlock)), dim3(kBlock), 0, ctx->stream, row16.p, ototal,
                       (int)PbSlice<T>::kRows);  // pads and phantom runs: the trash slots behind the slice
    SAMD_LAUNCH_CHECK(ctx);
  }
  hipLaunchKernelGGL(pbf_deal_kernel, dim3(rgrid), dim3(kBlock), 0, ctx->stream, (const uint32_t *)rks, rperm,
                     (const uint32_t *)cstart.p, runs, (const uint32_t *)run_start.p, (const uint16_t *)rr.p,
                     (const uint32_t *)ebase.p, (const uint32_t *)cls_out.p, (const int64_t *)poff.p, np, nb, run_ent.p, row16.p);
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL((pbf_place_kernel<T>), dim3(egrid), dim3(kBlock), 0, ctx->stream, (const int32_t *)a->col32,
                     (const T *)a->vals, nnz, eperm, rank_of_line, nb, (const uint32_t *)hscan.p, (const uint32_t *)run_start.p,
                     (const uint32_t *)run_ent.p, colj.p, valj.p);
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(pbf_desc_kernel, dim3((unsigned)cdiv(classes, kBlock)), dim3(kBlock), 0, ctx->stream,
                     (const uint32_t *)cls_ent.p, (const uint32_t *)ebase.p, (const uint32_t *)stbase.p,
                     (const uint32_t *)cls_out.p, (const int64_t *)poff.p, classes, np, nb, desc.p);
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(pbf_bounds_kernel, dim3((unsigned)cdiv(nb + 1 + np + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                     (const uint32_t *)stbase.p, (const int64_t *)poff.p, np, nb, bounds.p);
  SAMD_LAUNCH_CHECK(ctx);
  std::vector<int64_t> hb((size_t)nb + 1);
  SAMD_HIP(ctx, hipMemcpyAsync(hb.data(), bounds.p, sizeof(int64_t) * (size_t)(nb + 1), hipMemcpyDeviceToHost, ctx->stream));
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // ---- pass 1's wavefront ranges: about four workgroups per CU, every block cut into workgroups in proportion to
  // its steps (at least one), every workgroup's steps into kPbWaves ranges.  wtab[r] = {first step, block}
  std::vector<int64_t> wt;
  {
    const int64_t target = (int64_t)ctx->num_cu * (ctx->opt_spmv_pb_shares > 0 ? ctx->opt_spmv_pb_shares : 4);
    const int64_t total = nsteps - hb[0] > 0 ? nsteps - hb[0] : 1;
    for (int64_t b = 0; b < nb; b++) {
      const int64_t s0 = hb[(size_t)b], s1 = hb[(size_t)b + 1];
      if (s1 <= s0) continue;
      int64_t shares = ((s1 - s0) * target + total / 2) / total;
      const int64_t most = cdiv(s1 - s0, (int64_t)kPbWaves * 4);  // a wavefront wants at least a batch
      if (shares > most) shares = most;
      if (shares < 1) shares = 1;
      const int64_t nr = shares * kPbWaves;
      for (int64_t q = 0; q < nr; q++) {
        wt.push_back(s0 + (int64_t)((__int128)(s1 - s0) * q / nr));
        wt.push_back(b);
      }
    }
    wt.push_back(nsteps);
    wt.push_back(-1);
  }
  const int64_t nr = (int64_t)wt.size() / 2 - 1;
  Tmp<int64_t> wtab(ctx);
  SAMD_TRY(wtab.get((int64_t)wt.size()));
  SAMD_HIP(ctx, hipMemcpyAsync(wtab.p, wt.data(), sizeof(int64_t) * wt.size(), hipMemcpyHostToDevice, ctx->stream));
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));  // wt is host memory that goes away; the temporaries are released after their use
  a->pb_wtab = wtab.release();
  a->pb_nwg = nr / kPbWaves;
  a->pb_piece_ptr = bounds.release();
  a->pb_dgroup = desc.release();
  a->pb_col16 = colj.release();
  a->pb_row16 = row16.release();
  a->pb_val = valj.release();
  a->pb_prod = prod.release();
  a->pb_fold = 1;
  a->pb_outputs = ototal;
  *done = true;
  return 0;
}

// Decides and, if worthwhile, builds the piece copy.  One-time host synchronisations.
// vetted: called by reorder_prepare, which has already established that the record is large, that most tiles
// span a wide column range and that column popularity is skewed (the default route, option 0).  Option 1 also
// admits records of flat popularity (own checks below), 2 admits anything (tests), -1 nothing.
template <class T>
int pb_prepare(futhark_context *ctx, const CsrBody<T> *a, bool vetted, bool flat) {
  a->pb_state = -1;
  if (ctx->opt_spmv_pb < 0 || (ctx->opt_spmv_pb == 0 && !vetted)) return 0;
  if (!a->col32 || a->nnz == 0 || a->n == 0 || a->tile <= 0) return 0;
  const bool force = ctx->opt_spmv_pb > 1;
  // slice sizes (overridable for tests so that small records have many panels and blocks)
  const int64_t slice = kPbSliceBytes / (int64_t)sizeof(T);
  const int64_t yslice = PbSlice<T>::kRows;  // rows of y a panel can hold
  int64_t R = ctx->opt_spmv_pb_rows > 0 ? ctx->opt_spmv_pb_rows : yslice;
  int64_t lpb = ctx->opt_spmv_pb_lines > 0 ? ctx->opt_spmv_pb_lines : slice / kPbLine;  // lines per block
  int64_t E = ctx->opt_spmv_pb_entries > 0 ? ctx->opt_spmv_pb_entries : kPbPanelEntries;
  if (R > yslice) R = yslice;
  if (R > 65535) R = 65535;  // 16-bit row offsets
  if (lpb * kPbLine > slice) lpb = slice / kPbLine;
  if (lpb * kPbLine > 65536) lpb = 65536 / kPbLine;
  if (E < a->tile) E = a->tile;
  const int64_t nl = cdiv(a->m, kPbLine);
  const int64_t nb = cdiv(nl, lpb);
  const int64_t w = cdiv(nl, nb) * kPbLine;  // columns per block actually used (<= lpb * 32)
  const int64_t nq = (w / kPbLine) * nb;     // places for lines (>= nl)
  // lines sorted into the blocks by popularity (and pass 1 adding up what a row holds in a block) when the popularity
  // is skewed: that is where rows meet the same few blocks again and again
  const bool sorted = ctx->opt_spmv_pb_fold >= 0 && (ctx->opt_spmv_pb_deal > 0 || (ctx->opt_spmv_pb_deal == 0 && vetted && !flat));
  if (!force) {
    // the gather has to be the problem: 4-byte values (16 B moved per entry against 8; for 8-byte values the
    // one-pass paths are on par), x beyond the L2s, most tiles spanning a wide column range
    // skewed popularity: 4-byte values only (16 B moved per entry against 8; with 8-byte values the hot-column
    // renumbering is on par).  Flat popularity (uniform random: no renumbering can help, every gather of the
    // one-pass kernels misses L2): any value size -- measured f64, m = 1e7: 100 entries per row 0.78 ms against
    // 1.04 with column blocks, 12.5 per row 0.89 against 2.42, the 1e9-entry record 5.7 against 6.3.
    // (8-byte values with skewed popularity: only with the lines sorted into the blocks and the folded layout, which
    // writes a fraction of the products)
    if ((sizeof(T) != 4 && !flat && !sorted) || a->m * (int64_t)sizeof(T) < kPbMinBytes) return 0;
    const int64_t np_est = cdiv(a->n, R) + cdiv(a->nnz, E);
    if (a->nnz / (np_est * nb) < kPbMinPiece) return 0;
  }
  if (!force && !vetted) {
    unsigned long long *stat = reinterpret_cast<unsigned long long *>(ctx->d_small);
    SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t), ctx->stream));
    hipLaunchKernelGGL(reorder_wide_tiles_kernel, dim3((unsigned)cdiv(a->ntiles, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const int32_t *)a->col32, a->nnz, a->ntiles, a->tile, stat);
    SAMD_LAUNCH_CHECK(ctx);
    int64_t wide = 0;
    SAMD_TRY(read_small(ctx, ctx->d_small, 1, &wide));
    if (wide * 2 < a->ntiles) return 0;  // banded / stencil-like: the gather is already local
  }
  // ---- row panels: cut where E more entries have started and at every R-th row (host: a few thousand values)
  std::vector<int64_t> lb((size_t)a->ntiles + 1);
  SAMD_HIP(ctx, hipMemcpyAsync(lb.data(), a->tile_lb, sizeof(int64_t) * lb.size(), hipMemcpyDeviceToHost, ctx->stream));
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<int64_t> cuts;
  cuts.push_back(0);
  {
    const int64_t et = E / a->tile > 0 ? E / a->tile : 1;
    int64_t next_entry_cut = et;  // in tiles
    int64_t cur = 0;
    while (cur < a->n) {
      int64_t nxt = cur + R < a->n ? cur + R : a->n;  // at most R rows
      while (next_entry_cut <= a->ntiles && lb[(size_t)next_entry_cut] <= cur) next_entry_cut += et;
      if (next_entry_cut <= a->ntiles && lb[(size_t)next_entry_cut] < nxt) nxt = lb[(size_t)next_entry_cut];
      cuts.push_back(nxt);
      cur = nxt;
    }
  }
  const int64_t np = (int64_t)cuts.size() - 1;
  const int64_t pieces = np * nb;
  if (pieces >= (1ll << 26) || a->nnz >= (int64_t)0xffff0000LL) return 0;  // 32-bit keys: piece, then 5 bits of row
  if (!force && a->nnz / pieces < kPbMinPiece) return 0;

  Tmp<int64_t> panel_row(ctx), piece_ptr(ctx), bounds(ctx);
  Tmp<uint32_t> line_of_rank(ctx), dgroup(ctx);
  int64_t ptotal = 0;  // entries of the padded layouts
  Tmp<uint16_t> col16(ctx), row16(ctx);
  Tmp<T> pval(ctx), prod(ctx), xperm(ctx);
  SAMD_TRY(panel_row.get(np + 1));
  SAMD_TRY(piece_ptr.get(pieces + 1));
  SAMD_TRY(line_of_rank.get(nq));  // the line at every place
  SAMD_HIP(ctx, hipMemsetAsync(line_of_rank.p, 0xff, sizeof(uint32_t) * (size_t)nq, ctx->stream));
  SAMD_HIP(ctx, hipMemcpyAsync(panel_row.p, cuts.data(), sizeof(int64_t) * (size_t)(np + 1), hipMemcpyHostToDevice,
                               ctx->stream));
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));  // cuts is host memory that goes away
  int *sort_err = reinterpret_cast<int *>(ctx->d_small + 2);
  SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small + 2, 0, sizeof(int64_t), ctx->stream));
  {
    // ---- lines ranked by popularity
    Tmp<uint32_t> counts(ctx), keys_b(ctx), pay_a(ctx), pay_b(ctx), rank_of_line(ctx);
    SAMD_TRY(counts.get(nl));
    SAMD_TRY(keys_b.get(nl));
    SAMD_TRY(pay_a.get(nl));
    SAMD_TRY(pay_b.get(nl));
    SAMD_TRY(rank_of_line.get(nl));
    SAMD_HIP(ctx, hipMemsetAsync(counts.p, 0, sizeof(uint32_t) * (size_t)nl, ctx->stream));
    {
      int64_t stride = a->nnz >> 24;  // about 2^24 samples
      if (stride < 1) stride = 1;
      if (stride % 2 == 0) stride++;  // odd: no resonance with even row lengths
      const int64_t want = cdiv(cdiv(a->nnz, stride), kBlock), cap = (int64_t)ctx->num_cu * 32;
      hipLaunchKernelGGL(pb_line_count_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(kBlock), 0, ctx->stream,
                         (const int32_t *)a->col32, a->nnz, stride, counts.p);
      SAMD_LAUNCH_CHECK(ctx);
    }
    const unsigned lgrid = (unsigned)cdiv(nl, kBlock);
    hipLaunchKernelGGL(reorder_keys_kernel, dim3(lgrid), dim3(kBlock), 0, ctx->stream, (const uint32_t *)counts.p, nl,
                       counts.p);  // in place: ~count, ascending = most popular first, ties by line number
    SAMD_LAUNCH_CHECK(ctx);
    uint32_t *ks = nullptr;
    const uint32_t *perm = nullptr;
    SAMD_TRY((radix_sort_pairs<uint32_t, uint32_t>(ctx, counts.p, keys_b.p, (const uint32_t *)nullptr, pay_a.p, pay_b.p,
                                                    nl, 0, 32, &ks, &perm, sort_err)));
    hipLaunchKernelGGL(pb_rank_kernel, dim3(lgrid), dim3(kBlock), 0, ctx->stream, perm, nl, nb, w / kPbLine, sorted ? 1 : 0,
                       rank_of_line.p, line_of_rank.p);
    SAMD_LAUNCH_CHECK(ctx);
    // ---- entries sorted by piece (stable: a piece keeps the record's (row, column) order)
    Tmp<uint32_t> ek_a(ctx), ek_b(ctx), ep_a(ctx), ep_b(ctx);
    SAMD_TRY(ek_a.get(a->nnz));
    SAMD_TRY(ek_b.get(a->nnz));
    SAMD_TRY(ep_a.get(a->nnz));
    SAMD_TRY(ep_b.get(a->nnz));
    const unsigned egrid = (unsigned)cdiv(a->nnz, kBlock);
    hipLaunchKernelGGL(pb_key_kernel, dim3(egrid), dim3(kBlock), 0, ctx->stream, (const int64_t *)a->row_ptr,
                       (const int32_t *)a->col32, a->nnz, (const int64_t *)a->tile_lb, a->tile,
                       (const int64_t *)panel_row.p, np, (const uint32_t *)rank_of_line.p, nb, ek_a.p);
    SAMD_LAUNCH_CHECK(ctx);
    uint32_t *eks = nullptr;
    const uint32_t *eperm = nullptr;
    SAMD_TRY((radix_sort_pairs<uint32_t, uint32_t>(ctx, ek_a.p, ek_b.p, (const uint32_t *)nullptr, ep_a.p, ep_b.p, a->nnz,
                                                    0, bits_for(pieces) + 5, &eks, &eperm, sort_err)));
    hipLaunchKernelGGL(pb_piece_ptr_kernel, dim3((unsigned)cdiv(pieces + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const uint32_t *)eks, a->nnz, pieces, piece_ptr.p, 5);
    SAMD_LAUNCH_CHECK(ctx);
    bool folded = false;
    {
      int64_t sflag[3];
      SAMD_TRY(read_small(ctx, ctx->d_small, 3, sflag));
      if ((int)(sflag[2] & 0xffffffff) != 0) return set_error(ctx, "internal: radix sort look-back timed out");
    }
    if (ctx->opt_spmv_pb_fold >= 0)
      SAMD_TRY(pbf_build<T>(ctx, a, eks, eperm, (const int64_t *)piece_ptr.p, (const int64_t *)panel_row.p,
                            (const uint32_t *)rank_of_line.p, nb, np, pieces, sorted, &folded));
    if (!folded) {
    // ---- both padded layouts: every piece rounded up to a multiple of 32 entries
    Tmp<int64_t> boff(ctx), poff(ctx);
    SAMD_TRY(boff.get(pieces + 1));
    SAMD_TRY(poff.get(pieces + 1));
    const unsigned pgrid = (unsigned)cdiv(pieces + 1, kBlock);
    hipLaunchKernelGGL(pb_psz_kernel, dim3(pgrid), dim3(kBlock), 0, ctx->stream, (const int64_t *)piece_ptr.p, pieces, np, nb,
                       boff.p, poff.p);
    SAMD_LAUNCH_CHECK(ctx);
    SAMD_TRY((device_scan<int64_t, OpAdd, PtrIn<int64_t>, true>(ctx, PtrIn<int64_t>{boff.p}, pieces + 1, (int64_t)0, OpAdd(),
                                                                 boff.p, (int64_t *)nullptr)));
    SAMD_TRY((device_scan<int64_t, OpAdd, PtrIn<int64_t>, true>(ctx, PtrIn<int64_t>{poff.p}, pieces + 1, (int64_t)0, OpAdd(),
                                                                 poff.p, (int64_t *)nullptr)));
    SAMD_HIP(ctx, hipMemcpyAsync(ctx->d_small, boff.p + pieces, sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
    int64_t flag[3];
    SAMD_TRY(read_small(ctx, ctx->d_small, 3, flag));
    if ((int)(flag[2] & 0xffffffff) != 0) return set_error(ctx, "internal: radix sort look-back timed out");
    ptotal = flag[0];
    if (ptotal >= (int64_t)0xffffffe0LL * 32) return 0;  // 32-bit group numbers
    SAMD_TRY(col16.get(ptotal));
    SAMD_TRY(row16.get(ptotal));
    SAMD_TRY(pval.get(ptotal));
    SAMD_TRY(dgroup.get(ptotal / 32));
    SAMD_TRY(bounds.get(nb + 1 + np + 1));
    SAMD_HIP(ctx, hipMemsetAsync(col16.p, 0, sizeof(uint16_t) * (size_t)ptotal, ctx->stream));
    SAMD_HIP(ctx, hipMemsetAsync(pval.p, 0, sizeof(T) * (size_t)ptotal, ctx->stream));
    hipLaunchKernelGGL(pb_fill_trash_kernel, dim3((unsigned)cdiv(ptotal, kBlock)), dim3(kBlock), 0, ctx->stream, row16.p, ptotal,
                       (int)PbSlice<T>::kRows);  // pads: the trash slots behind the slice
    SAMD_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL((pb_build_kernel<T>), dim3(egrid), dim3(kBlock), 0, ctx->stream, (const int64_t *)a->row_ptr,
                       (const int32_t *)a->col32, (const T *)a->vals, a->nnz, (const int64_t *)a->tile_lb, a->tile,
                       (const int64_t *)panel_row.p, (const uint32_t *)rank_of_line.p, nb, np, (const uint32_t *)eks, eperm,
                       (const int64_t *)piece_ptr.p, (const int64_t *)boff.p, (const int64_t *)poff.p, col16.p, row16.p,
                       pval.p);
    SAMD_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(pb_dgroup_kernel, dim3((unsigned)cdiv(pieces, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)boff.p, (const int64_t *)poff.p, pieces, np, nb, dgroup.p);
    SAMD_LAUNCH_CHECK(ctx);
    // block b's range in the block-major layout, then panel p's range in the panel-major one
    hipLaunchKernelGGL(pb_bounds_kernel, dim3((unsigned)cdiv(nb + 1 + np + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)boff.p, (const int64_t *)poff.p, np, nb, bounds.p);
    SAMD_LAUNCH_CHECK(ctx);
    SAMD_TRY(read_small(ctx, ctx->d_small, 1, flag));  // orders the temporaries' release after their use
    }
  }
  if (a->pb_fold == 0) SAMD_TRY(prod.get(ptotal));
  SAMD_TRY(xperm.get(nb * w));
  a->pb_np = np;
  a->pb_nb = nb;
  a->pb_w = w;
  a->pb_nl = nq;  // places of lines in block order
  a->pb_panel_row = panel_row.release();
  if (a->pb_fold == 0) {
    a->pb_piece_ptr = bounds.release();  // [nb + 1] block ranges (block-major layout), then [np + 1] panel ranges (panel-major)
    a->pb_dgroup = dgroup.release();
    a->pb_col16 = col16.release();
    a->pb_row16 = row16.release();
    a->pb_val = pval.release();
    a->pb_prod = prod.release();
    a->pb_outputs = ptotal;
  }
  a->pb_line_of_rank = line_of_rank.release();
  a->pb_x = xperm.release();
  a->pb_state = 1;
  if (ctx->logging) fprintf(stderr, "[sparse_amd] two-pass SpMV: %lld panels x %lld blocks of %lld columns, %.0f entries per piece\n",
                            (long long)np, (long long)nb, (long long)w, (double)a->nnz / (double)pieces);
  return 0;
}

template <class T>
int spmv_pb(futhark_context *ctx, const CsrBody<T> *a, const T *x, T *y) {
  if (((uintptr_t)x & 15) == 0)
    hipLaunchKernelGGL((pb_xperm_vec_kernel<T>), dim3((unsigned)cdiv(a->pb_nl * (kPbLine * (int64_t)sizeof(T) / 16), kBlock)),
                       dim3(kBlock), 0, ctx->stream, x, a->m, (const uint32_t *)a->pb_line_of_rank, a->pb_nl, a->pb_nb, a->pb_w,
                       a->pb_x);
  else
    hipLaunchKernelGGL((pb_xperm_kernel<T>), dim3((unsigned)cdiv(a->pb_nl * kPbLine, kBlock)), dim3(kBlock), 0, ctx->stream,
                       x, a->m, (const uint32_t *)a->pb_line_of_rank, a->pb_nl, a->pb_nb, a->pb_w, a->pb_x);
  SAMD_LAUNCH_CHECK(ctx);
  // pass 1: about 4 workgroups per CU so that the tail is short
  int64_t split = cdiv((int64_t)ctx->num_cu * 4, a->pb_nb);
  if (split < 1) split = 1;
  if (split > 65535) split = 65535;
  ctx->last_pb_fold = a->pb_fold ? 1 : 0;
  ctx->last_pb_outputs = a->pb_outputs;
  if (a->pb_fold) {
    const int lab1 = (int)((ctx->opt_ablate >> 2) & 7);
    const dim3 grid1((unsigned)a->pb_nwg);
#define SAMD_PB1F(LAB)                                                                                                 \
  hipLaunchKernelGGL((pb_pass1f_kernel<T, LAB>), grid1, dim3(kPbThreads), 0, ctx->stream, (const int64_t *)a->pb_wtab,  \
                     (const uint32_t *)a->pb_dgroup, (const uint16_t *)a->pb_col16, (const T *)a->pb_val,               \
                     (const T *)a->pb_x, a->pb_prod, (int)a->pb_w)
    if (a->pb_nwg <= 0) {}  // nothing stored
    else if (lab1 == 1) SAMD_PB1F(1);
    else if (lab1 == 2) SAMD_PB1F(2);
    else if (lab1 == 3) SAMD_PB1F(3);
    else SAMD_PB1F(0);
#undef SAMD_PB1F
  }
  else
    hipLaunchKernelGGL((pb_pass1_kernel<T, 8>), dim3((unsigned)a->pb_nb, (unsigned)split), dim3(kPbThreads), 0, ctx->stream,
                       (const int64_t *)a->pb_piece_ptr, (const uint16_t *)a->pb_col16, (const T *)a->pb_val,
                       (const T *)a->pb_x, (const uint32_t *)a->pb_dgroup, a->pb_prod, (int)a->pb_w,
                       (int)((ctx->opt_ablate >> 2) & 3));
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL((pb_pass2_kernel<T, 8>), dim3((unsigned)a->pb_np), dim3(kPbThreads), 0, ctx->stream,
                     (const int64_t *)(a->pb_piece_ptr + a->pb_nb + 1), (const int64_t *)a->pb_panel_row,
                     (const uint16_t *)a->pb_row16, (const T *)a->pb_prod, y, (int)(ctx->opt_ablate & 3),
                     ((ctx->opt_ablate & 3) == 2 || a->pb_fold) ? 0 : (ctx->opt_ablate >> 5) ? (int)(ctx->opt_ablate >> 5) : kPbFoldMax);
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

}  // namespace samd
