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
constexpr int kPbSliceBytes = 128 << 10;            // slice of y (pass 2) in LDS (f64 accumulators: 16 K rows)
constexpr int kPbXSliceBytes = 160 << 10;           // slice of x (pass 1) in LDS: all of it (f32: 40 K columns per block)
constexpr int64_t kPbPanelEntries = 256 << 10;      // a panel is closed once it holds this many entries (measured on C3:
                                                    // 1 M 1.22 ms, 512 K 0.96, 256 K 0.94, 128 K 0.99, 64 K 1.27 in pass 2)
constexpr int kPbFoldMax = 16;                      // pass 2 folds a wavefront's runs when it holds at most this many (C3: 2: 0.95 ms, 8: 0.91, 16: 0.89, 32: 0.93, 48: 1.01)
constexpr int64_t kPbMinBytes = 16 << 20;            // x at least this large (RMAT f32, bench/pb_scale_sweep.py, two-pass / one-pass ms:
                                                    // scale 20 0.21 / 0.15, 21 0.264 / 0.259, 22 0.47 / 0.51, 23 0.81 / 1.01, 24 1.62 / 2.08)
constexpr int kPbFoldRows = 64;                     // a panel of fewer rows than this keeps its entries sorted by row: pass 2 folds runs
constexpr int kPbFoldFlag = 0x8000;                 // row16 bit that says so
constexpr int kPbHeadFlag = 0x8000;                 // folded layout, while it is built: bit of the entry's row that says it starts a run
constexpr int kPbRunCap = 32;                       // folded layout: a run is cut at every multiple of this many positions (C3, 16 -- runs inside one DPP row, one scan step fewer: pass 1 the same 0.557 ms, 7 % more products, pass 2 0.160 -> 0.184)
constexpr int kPbOutRound = 8;                      // folded layout: a piece's outputs start at a multiple of this many products (C3, round 3, chunked layout, 32 / 16 / 8: pass 1 0.504 / 0.501 / 0.498 ms, pass 2 0.158 / 0.155 / 0.1545, 95.7 / 89.0 / 86.1 M products; round 2's kernel liked 32: 1 / 8 / 16 / 32 gave pass 1 0.714 / 0.707 / 0.694 / 0.687)
constexpr int64_t kPbPanelEntriesFolded = 512 << 10; // the same for the folded layout, whose pass 2 sees a third of the entries (C3, 256 K / 512 K / 1 M: pass 1 0.521 / 0.504 / 0.501 ms, pass 2 0.160 / 0.162 / 0.165, 100.6 / 95.7 / 93.7 M products)
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

// place_line (place q = slot * nb + block) turned block-major: line_blk[block * slots + slot]
__global__ __launch_bounds__(kBlock) void pb_line_blk_kernel(const uint32_t *__restrict__ place_line, int64_t nq, int64_t nb,
                                                             uint32_t *__restrict__ line_blk) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= nq) return;
  line_blk[(q % nb) * (nq / nb) + q / nb] = place_line[q];
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
  __shared__ __attribute__((aligned(16))) T xs[kPbXSliceBytes / sizeof(T)];
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
        atomicAdd(&ys[lab == 3 ? (((r[u] & (kPbFoldFlag - 1)) & ~31) | (lane & 31)) : (r[u] & (kPbFoldFlag - 1))], (acc_t)v[u]);  // lab 3: every half-wavefront a permutation of the 32 bank pairs (wrong sums)
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

typedef unsigned int pb_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int pb_u32x4 __attribute__((ext_vector_type(4)));
constexpr int kPbRsrcFlags = 0x00020000;  // gfx9 raw buffer, 32-bit data format

// a raw buffer (stride 0) over `bytes` bytes from base: offsets behind them read as zeros and are not stored
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pb_rsrc(const void *base, int bytes = 0x7ffffffe) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, kPbRsrcFlags);
}

// ---- pass 2 of the folded layout, two products per lane.  A group of 16 lanes then adds products 2 l and, in a
// second instruction, 2 l + 1 of 32 NEIGHBOURING products: half of one dealt group of 32, whose rows all differ in
// row mod 32, so the bank deal holds as it stands -- with one 4-byte load of rows and one load of sums per 128
// products instead of two of each.  (Four per lane, the stream stored transposed inside every 64 so that the deal
// still holds: pass 2 0.145 -> 0.138 ms on C3, but pass 1's stores, strided inside their 256 bytes, 0.521 -> 0.547.)
// The panel's stream is a multiple of 64 products long (pbf_panel_align_kernel); the places behind it are out of
// the buffers' range: row 0, sum 0.
template <class T>
__global__ __launch_bounds__(kPbThreads) void pb_pass2p_kernel(const int64_t *__restrict__ panel_ptr,
                                                               const int64_t *__restrict__ panel_row,
                                                               const uint16_t *__restrict__ row16,
                                                               const T *__restrict__ prod, T *__restrict__ y, int lab) {
  typedef typename PbAcc<T>::type acc_t;
  constexpr int kTrash = PbSlice<T>::kRows;
  constexpr int S = 4, kStep = 128;  // loads per batch, products per load
  __shared__ acc_t ys[kTrash + 32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int64_t p = blockIdx.x;
  const int64_t r0 = panel_row[p];
  const int rows = (int)(panel_row[p + 1] - r0);
  for (int j = tid; j < rows; j += kPbThreads) ys[j] = (acc_t)0;
  __syncthreads();
  const int64_t ps = panel_ptr[p], pe = panel_ptr[p + 1];
  struct Regs {
    uint32_t r[S];
    T v[S][2];
  };
  auto load = [&](int64_t cs, Regs &g) {  // cs may lie behind the end: nothing is asked for then
    const int64_t left = pe - cs;
    const int n = left <= 0 ? 0 : left < S * kStep ? (int)left : S * kStep;
    const int64_t at = left > 0 ? cs : ps;
    const __amdgpu_buffer_rsrc_t rr = pb_rsrc(row16 + at, n * 2), rv = pb_rsrc(prod + at, n * (int)sizeof(T));
#pragma unroll
    for (int s = 0; s < S; s++) {
      g.r[s] = __builtin_amdgcn_raw_buffer_load_b32(rr, lane * 4 + s * (kStep * 2), 0, 2);
      if constexpr (sizeof(T) == 4) {
        const pb_u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rv, lane * 8 + s * (kStep * 4), 0, 2);
        const uint32_t w0 = w[0], w1 = w[1];
        g.v[s][0] = __builtin_bit_cast(T, w0);
        g.v[s][1] = __builtin_bit_cast(T, w1);
      } else {
        const pb_u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rv, lane * 16 + s * (kStep * 8), 0, 2);
        const uint64_t w0 = ((uint64_t)w[1] << 32) | w[0], w1 = ((uint64_t)w[3] << 32) | w[2];
        g.v[s][0] = __builtin_bit_cast(T, w0);
        g.v[s][1] = __builtin_bit_cast(T, w1);
      }
    }
  };
  auto add = [&](const Regs &g) {
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint32_t row = (k ? g.r[s] >> 16 : g.r[s]) & 0x7fffu;
        if (lab == 1) { if (row == 12345 && g.v[s][k] == (T)77) ys[0] = (acc_t)g.v[s][k]; continue; }  // lab: stream only
        atomicAdd(&ys[lab == 3 ? ((row & ~31u) | (uint32_t)(lane & 31)) : row], (acc_t)g.v[s][k]);  // lab 3: conflict-free by force (wrong sums)
      }
  };
  constexpr int64_t kBatch = (int64_t)S * kStep, kStride = (int64_t)kPbWaves * kBatch;
  int64_t cs = ps + (int64_t)wave * kBatch;
  if (cs < pe) {
    Regs ga, gb;
    load(cs, ga);
    for (;;) {
      const int64_t c1 = cs + kStride;
      load(c1, gb);
      add(ga);
      if (c1 >= pe) break;
      const int64_t c2 = c1 + kStride;
      load(c2, ga);
      add(gb);
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
// 2.5 times, the 13 000 heaviest rows (28 % of the entries) 8 times and more; the product stream shrinks to about
// half, which takes 10 bytes per dropped product off the 16 the two passes move per entry.
// Layout: a piece's entries are grouped in RUNS (one row each), a run's entries are neighbours in the block-major
// stream, bit 15 of col16 marks a run's first entry and every position that is a multiple of 32 starts a new run
// (so a run never leaves the 32 lanes that hold its group).  Pass 1 forms the products, adds each run up inside the
// wavefront (segmented scan over DPP moves, skipped when the instruction holds no run longer than one entry) and the
// last lane of a run stores the sum at pb_dgroup[group] + (runs before it in the group): the outputs of a piece
// are contiguous in the panel-major stream, in the order of its runs.  That order is the bank-pair deal of the
// plain layout applied to runs, so pass 2 (the same kernel, nothing to fold any more) still sees each bank pair
// about once per instruction.  Pads: the first pad of a piece starts a run whose row is the trash slot.
template <int CTRL, int ROWMASK, class T>
__device__ __forceinline__ T pb_dpp0(T v) {  // lanes without a source (or outside ROWMASK) read 0
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));
  } else {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, ROWMASK, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, ROWMASK, 0xf, false);
    return __builtin_bit_cast(T, ((uint64_t)hi << 32) | lo);
  }
}

// Per-lane constants of the run arithmetic.  A run never leaves its 32-lane half (positions that are multiples of 32
// start a run), so every lane works on its half's 32 bits of the wavefront's masks.
struct PbLane {
  int lane, l5;
  bool hi;
  uint32_t upto, bit, under;  // bits 0..l5, bit l5, bits 0..l5-1
  __device__ __forceinline__ explicit PbLane(int ln) : lane(ln), l5(ln & 31), hi(ln >= 32) {
    bit = 1u << l5;
    under = bit - 1u;
    upto = under | bit;
  }
  __device__ __forceinline__ uint32_t half(unsigned long long m) const { return hi ? (uint32_t)(m >> 32) : (uint32_t)m; }
};

// heads: lanes that start a run (lanes 0 and 32 among them).  Returns in a run's last lane the sum of the run.
template <class T>
__device__ __forceinline__ T pb_run_sum(T p, unsigned long long heads, const PbLane &L) {
  if (heads == ~0ull) return p;  // every run is one entry (the thin part of the matrix)
  const int dist = L.l5 - 31 + __builtin_clz(L.half(heads) & L.upto);  // distance to the run's first lane
  {
    const T t = pb_dpp0<0x111, 0xf>(p);  // row_shr:1
    if (dist >= 1) p = (T)(p + t);
  }
  if (__ballot(dist >= 2) != 0ull) {
    {
      const T t = pb_dpp0<0x112, 0xf>(p);
      if (dist >= 2) p = (T)(p + t);
    }
    if (__ballot(dist >= 4) != 0ull) {
      {
        const T t = pb_dpp0<0x114, 0xf>(p);
        if (dist >= 4) p = (T)(p + t);
      }
      {
        const T t = pb_dpp0<0x118, 0xf>(p);
        if (dist >= 8) p = (T)(p + t);
      }
    }
  }
  // a run that began in the 16 lanes before this row: lane 15 of that row holds its part there
  if (kPbRunCap > 16) {
    const T t = pb_dpp0<0x142, 0xa>(p);  // row_bcast:15 into rows 1 and 3
    if (dist > (L.lane & 15)) p = (T)(p + t);
  }
  return p;
}

__device__ __forceinline__ int pb_rank_in_half(unsigned long long mask, int lane) {  // set bits below the lane inside its 32-lane half
  const PbLane L(lane);
  return __popc(L.half(mask) & L.under);
}

template <class P>
__device__ __forceinline__ P *pb_uniform_ptr(P *p) {  // p is the same in every lane: keep it in scalar registers
  const uint64_t u = (uint64_t)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
  typedef __attribute__((address_space(1))) P G;  // global memory: an integer cast alone would leave a generic (flat) pointer
  return (P *)(G *)(((uint64_t)hi << 32) | lo);
}

// ---- buffer addressing for pass 1's streams.  A global load or store through a 64-bit vector address costs a
// 64-bit vector add per access (v_lshl_add_u64, not a full-rate instruction), and pass 1 is bound by vector issue:
// 36 vector instructions per 64 entries, 10 of them address arithmetic.  Buffer instructions take a scalar
// descriptor (the base moves with the step: scalar adds, which issue beside the vector ones), one 32-bit lane offset
// that never changes, and an immediate: no vector instruction per load at all, three (and, bcnt + base, shift) per
// store.  The descriptors are raw (stride 0) with the range check out of the way: every offset is in range by
// construction of the layout.
template <int AUX>
__device__ __forceinline__ float pb_bload(__amdgpu_buffer_rsrc_t r, int off, const float *) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX));
}
template <int AUX>
__device__ __forceinline__ int32_t pb_bload(__amdgpu_buffer_rsrc_t r, int off, const int32_t *) {
  return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ double pb_bload(__amdgpu_buffer_rsrc_t r, int off, const double *) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, AUX));
}
template <int AUX>
__device__ __forceinline__ void pb_bstore(float v, __amdgpu_buffer_rsrc_t r, int off) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void pb_bstore(int32_t v, __amdgpu_buffer_rsrc_t r, int off) {
  __builtin_amdgcn_raw_buffer_store_b32((unsigned)v, r, off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void pb_bstore(double v, __amdgpu_buffer_rsrc_t r, int off) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(pb_u32x2, v), r, off, 0, AUX);
}

// the run sums of one instruction's 64 products; h: this lane's half of the heads mask.  As pb_run_sum, on 32-bit
// masks only (a 64-bit vector shift per use is what the compiler makes of `lane < 32 ? lo : hi` otherwise).
template <class T>
__device__ __forceinline__ T pb_run_sum32(T p, bool all_heads, uint32_t h, const PbLane &L) {
  if (all_heads) return p;  // every run is one entry (the thin part of the matrix)
  const int dist = L.l5 - 31 + __builtin_clz(h & L.upto);  // distance to the run's first lane
  {
    const T t = pb_dpp0<0x111, 0xf>(p);  // row_shr:1
    if (dist >= 1) p = (T)(p + t);
  }
  if (__ballot(dist >= 2) != 0ull) {
    {
      const T t = pb_dpp0<0x112, 0xf>(p);
      if (dist >= 2) p = (T)(p + t);
    }
    if (__ballot(dist >= 4) != 0ull) {
      {
        const T t = pb_dpp0<0x114, 0xf>(p);
        if (dist >= 4) p = (T)(p + t);
      }
      {
        const T t = pb_dpp0<0x118, 0xf>(p);
        if (dist >= 8) p = (T)(p + t);
      }
    }
  }
  if (kPbRunCap > 16) {  // a run that began in the 16 lanes before this row: lane 15 of that row holds its part there
    const T t = pb_dpp0<0x142, 0xa>(p);  // row_bcast:15 into rows 1 and 3
    if (dist > (L.lane & 15)) p = (T)(p + t);
  }
  return p;
}

// ---- pass 1 of the folded layout.
// Layout read here (pbf_chunk_kernel makes it from the block-major order at the end of the build): every column block
// is cut into CHUNKS of 512 entries, the block's last chunk filled up with pads; chunk ci is one wavefront's step, its
// entry u * 64 + l (u = 0 .. 7: the u-th instruction's lane l) stored so that a lane's eight entries are neighbours:
//   colc[ci * 512 + l * 8 + u]                                    one 16-byte load per lane and chunk
//   valc[ci * 512 + (u / P) * 64 P + l * P + u % P], P = 16 / sizeof(T)   8 / P 16-byte loads, each one contiguous KiB
//   outc[ci * 32 + h * 8 + u]                                     first output of half h (lanes 32 h ..) of instruction u
//   outc[ci * 32 + 16 + h * 8 + u]                                the run starts among that half's 32 lanes, as a mask
// (the masks ready-made: no ballot in the loop, and all 16 bits of a column are the column, so a block can be as
// wide as the whole LDS holds: 40 K columns of f32)
// Why: with one workgroup per CU (the slice of x takes the LDS) a CU has 16 wavefronts to keep its share of the HBM
// stream in flight, and a wavefront that asks for a step as 8 + 8 + 8 two- and four-byte loads cannot have more than
// two steps outstanding (the 6-bit counter of outstanding vector-memory instructions also counts the 8 stores of a
// step).  Five 16-byte loads per step leave room for a ring of three steps: two are in flight while one is summed.
constexpr int kPbChunk = 512;  // entries per chunk: 8 per lane

template <class T>
struct PbChunkRegs {
  static constexpr int kPer = 16 / (int)sizeof(T), kVecs = 8 / kPer;
  pb_u32x4 c, v[kVecs], g[2], h[2];
  __device__ __forceinline__ uint32_t col(int u) const { return (u & 1) ? c[u >> 1] >> 16 : c[u >> 1] & 0xffffu; }
  __device__ __forceinline__ uint32_t heads(int u) const { return h[u >> 2][u & 3]; }  // run starts among the 32 lanes of this lane's half
  __device__ __forceinline__ uint32_t out(int u) const { return g[u >> 2][u & 3]; }
  __device__ __forceinline__ T val(int u) const {
    if constexpr (sizeof(T) == 4) {
      const uint32_t w = v[u >> 2][u & 3];  // (a bit cast of the element expression itself reads element 0: clang 20)
      return __builtin_bit_cast(T, w);
    } else {
      const uint64_t w = ((uint64_t)v[u >> 1][(u & 1) * 2 + 1] << 32) | v[u >> 1][(u & 1) * 2];
      return __builtin_bit_cast(T, w);
    }
  }
};

// LAB (kernel lab; results wrong for 1, 2, 4, 5, 6): 1 no store, 2 no LDS gather, 3 non-temporal stores, 4 no run sums,
// 5 neither stores nor run sums, 6 three whole-wavefront stores per chunk
// WIDE: the products do not fit a 32-bit byte offset (2^32 / sizeof(T) outputs and more): stores through 64-bit addresses
// wtab: per workgroup its first chunk, the chunk behind its last, its column block
template <class T, int LAB, bool WIDE>
__global__ __launch_bounds__(kPbThreads) void pb_pass1f_kernel(const int64_t *__restrict__ wtab,
                                                               const uint16_t *__restrict__ colc,
                                                               const T *__restrict__ valc, const T *__restrict__ xperm,
                                                               const uint32_t *__restrict__ outc,
                                                               T *__restrict__ prod, int w,
                                                               const uint32_t *__restrict__ line_blk,
                                                               const T *__restrict__ x, int64_t m) {
  __shared__ __attribute__((aligned(16))) T xs[kPbXSliceBytes / sizeof(T)];
  typedef PbChunkRegs<T> Regs;
  const int64_t s = wtab[3 * (int64_t)blockIdx.x], e = wtab[3 * (int64_t)blockIdx.x + 1], b = wtab[3 * (int64_t)blockIdx.x + 2];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const PbLane L(lane);
  const int off_c = lane * 16, off_g = (lane >> 5) * 32;
  const __amdgpu_buffer_rsrc_t rprod = pb_rsrc(prod);
  auto load = [&](int64_t ci, Regs &r) {  // a whole chunk; ci is the same in every lane
    const __amdgpu_buffer_rsrc_t rc = pb_rsrc(colc + ci * kPbChunk), rv = pb_rsrc(valc + ci * kPbChunk),
                                 rg = pb_rsrc(outc + ci * (kPbChunk / 16));
    r.c = __builtin_amdgcn_raw_buffer_load_b128(rc, off_c, 0, 2);  // aux 2: non-temporal
#pragma unroll
    for (int q = 0; q < Regs::kVecs; q++) r.v[q] = __builtin_amdgcn_raw_buffer_load_b128(rv, off_c + q * 1024, 0, 2);
    r.g[0] = __builtin_amdgcn_raw_buffer_load_b128(rg, off_g, 0, 0);
    r.g[1] = __builtin_amdgcn_raw_buffer_load_b128(rg, off_g + 16, 0, 0);
    r.h[0] = __builtin_amdgcn_raw_buffer_load_b128(rg, off_g + 64, 0, 0);
    r.h[1] = __builtin_amdgcn_raw_buffer_load_b128(rg, off_g + 80, 0, 0);
  };
  // The sums are stored where they fall, a group's few at a time: short, unaligned pieces of lines, which the memory
  // system charges per piece rather than per byte (C3, lines sorted into the blocks: 84 M sums = 0.34 GB cost 0.16 ms
  // where the 2 GB of entries stream in 0.40).  Plain stores, so that L2 can put the pieces of a line together
  // (non-temporal: pass 1 0.615 instead of 0.558 ms; the next kernel then pays 0.02 ms for the write-backs).
  // Tried twice and dropped: collecting the sums in a per-wavefront row of LDS and writing whole aligned 256-byte
  // rows -- (a) across piece boundaries with scalar bookkeeping per half group: 0.687 -> 0.91 ms; (b) only for steps
  // whose 16 groups continue each other (a flag bit per group from the build, one test per step): 0.558 -> 0.584 ms,
  // the LDS round trip costs more than the fuller lines save.
  auto work = [&](const Regs &r) {
    T pr[8];
#pragma unroll
    for (int u = 0; u < 8; u++) pr[u] = LAB == 2 ? r.val(u) : (T)(r.val(u) * xs[r.col(u)]);
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t h = r.heads(u);  // this lane's half of the mask (bit 0 is always set: lanes 0 and 32 start a run)
      const T sum = (LAB == 4 || LAB == 5) ? pr[u] : pb_run_sum32<T>(pr[u], __ballot(h != 0xffffffffu) == 0ull, h, L);
      const uint32_t tails = (h >> 1) | 0x80000000u;  // a run ends where the next lane starts one, and at the end of its half
      if (LAB == 6) {  // lab: three whole-wavefront stores per chunk (about the same bytes as the eight partial ones)
        if (u < 3) pb_bstore<0>(sum, rprod, (int)((r.out(u) + (uint32_t)L.l5) * (uint32_t)sizeof(T)));
        else if (sum == (T)12345) pb_bstore<0>(sum, rprod, 0);
        continue;
      }
      if ((tails & L.bit) != 0) {
        const uint32_t slot = r.out(u) + (uint32_t)__popc(tails & L.under);
        if (WIDE) {
          T *d = prod + (int64_t)slot;
          if (LAB == 3) __builtin_nontemporal_store(sum, d);
          else if ((LAB != 1 && LAB != 5) || sum == (T)12345) *d = sum;
        } else {
          if (LAB == 3) pb_bstore<2>(sum, rprod, (int)(slot * (uint32_t)sizeof(T)));
          else if ((LAB != 1 && LAB != 5) || sum == (T)12345) pb_bstore<0>(sum, rprod, (int)(slot * (uint32_t)sizeof(T)));
        }
      }
    }
  };
  // the wavefront's chunks: s + wave, s + wave + 16, ...  Requests run two chunks ahead; behind the last chunk the
  // last one is asked for again (it is in cache): behind a branch the compiler cannot tell which requests are
  // outstanding where the paths join, and makes the sums of this chunk wait for the loads of the next ones.
  // The first two chunks are asked for before the slice of x: both wait for memory, neither for the other.
  const int64_t first = s + wave;
  const bool any = first < e;
  const int nf = any ? (int)((e - first + kPbWaves - 1) / kPbWaves) : 0;
  const int64_t last = first + (int64_t)(nf - 1) * kPbWaves;
  auto at = [&](int j) {
    const int64_t ci = first + (int64_t)j * kPbWaves;
    return ci < last ? ci : last;
  };
  Regs ra, rb, rc;
  if (any) {
    load(at(0), ra);
    load(at(1), rb);
  }
  if (line_blk == nullptr) {  // the slice from the copy of x in block order (pb_xperm_kernel: x is not 16-byte aligned)
    constexpr int PER = 16 / (int)sizeof(T);
    const raw16 *src = reinterpret_cast<const raw16 *>(xperm + b * (int64_t)w);
    for (int j = threadIdx.x; j < w / PER; j += kPbThreads) reinterpret_cast<raw16 *>(xs)[j] = src[j];
  } else {
    // The slice straight from x, line by line through the block's list of lines (line_blk[b][slot]).  With the lines
    // sorted into the blocks most blocks have one workgroup, so every line is fetched once, as the copy kernel would
    // have fetched it, and the copy's 64 MB written and read again drop out (C3: 25 us).  (Round 2 measured this
    // slower: the blocks were dealt then, four workgroups each fetching every scattered line again.)
    constexpr int PER = 16 / (int)sizeof(T), PIECES = kPbLine / PER;
    const uint32_t *tab = line_blk + b * (int64_t)(w / kPbLine);
    for (int j = threadIdx.x; j < w / PER; j += kPbThreads) {
      const uint32_t line = tab[j / PIECES];
      const int64_t c = (int64_t)line * kPbLine + (j % PIECES) * PER;
      raw16 v = raw16{0u, 0u, 0u, 0u};
      if (line != 0xffffffffu) {
        if (c + PER <= m) {
          v = *reinterpret_cast<const raw16 *>(x + c);
        } else {
          T t[PER];
#pragma unroll
          for (int u = 0; u < PER; u++) t[u] = c + u < m ? x[c + u] : (T)0;
          v = *reinterpret_cast<const raw16 *>(t);
        }
      }
      reinterpret_cast<raw16 *>(xs)[j] = v;
    }
  }
  __syncthreads();
  if (any) {
    for (int j = 0;;) {
      load(at(j + 2), rc);
      work(ra);
      if (++j >= nf) break;
      load(at(j + 2), ra);
      work(rb);
      if (++j >= nf) break;
      load(at(j + 2), rb);
      work(rc);
      if (++j >= nf) break;
    }
  }
}

// The chunked layout pass 1 reads, from the block-major one the build works in (positions = places of the dealt
// order; bounds[b] = first position of block b).  One wavefront per chunk.  Pads behind a block's end: a run of its
// own each (bit 15), value 0, outputs in the trash behind the products.
template <class T>
__global__ __launch_bounds__(kBlock) void pbf_chunk_kernel(const int64_t *__restrict__ bounds, const int64_t *__restrict__ cbase,
                                                           int64_t nb, int64_t nchunks, const uint16_t *__restrict__ col16,
                                                           const uint16_t *__restrict__ rowt, const T *__restrict__ pval,
                                                           const uint32_t *__restrict__ uout, uint32_t trash,
                                                           uint16_t *__restrict__ colc, T *__restrict__ valc,
                                                           uint32_t *__restrict__ outc) {
  constexpr int kPer = 16 / (int)sizeof(T);
  const int64_t ci = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (ci >= nchunks) return;
  int64_t lo = 0, hi = nb;  // largest b with cbase[b] <= ci (behind empty blocks: the one that holds the chunk)
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (cbase[mid] <= ci) lo = mid; else hi = mid;
  }
  const int64_t ls = bounds[lo] + (ci - cbase[lo]) * kPbChunk;
  const int64_t left = bounds[lo + 1] - ls;
  const int n = left < kPbChunk ? (int)left : kPbChunk;
  uint16_t *cd = colc + ci * kPbChunk + lane * 8;
  T *vd = valc + ci * kPbChunk;
  uint32_t *od = outc + ci * (kPbChunk / 16);
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int o = u * 64 + lane;
    const bool ok = o < n;
    cd[u] = ok ? col16[ls + o] : (uint16_t)0;
    vd[(u / kPer) * (64 * kPer) + lane * kPer + u % kPer] = ok ? pval[ls + o] : (T)0;
    const unsigned long long heads = __ballot(!ok || (rowt[ls + o] & kPbHeadFlag) != 0);  // pads behind the block: a run each
    if (lane == 0) {
      od[16 + u] = (uint32_t)heads;
      od[16 + 8 + u] = (uint32_t)(heads >> 32);
    }
  }
  if (lane < kPbChunk / 32) {
    const int k = 2 * (lane & 7) + (lane >> 3);  // group of instruction u = lane & 7, half lane >> 3
    od[lane] = k * 32 < n ? uout[(ls >> 5) + k] : trash;
  }
}

// ---- building the folded layout (once per record) from the entries sorted by (piece, row mod 32, row, column)
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

struct PbRunHeadIn {  // 1 where sorted entry i starts a run: first of its piece or another row than its predecessor
  const uint32_t *keys;
  const uint16_t *rr;
  __device__ __forceinline__ uint32_t operator()(int64_t i) const {
    return (i == 0 || (keys[i] >> 5) != (keys[i - 1] >> 5) || rr[i] != rr[i - 1]) ? 1u : 0u;
  }
};

struct PbU32AsI64In {
  const uint32_t *p;
  __device__ __forceinline__ int64_t operator()(int64_t i) const { return (int64_t)p[i]; }
};

// run_start[j] = sorted position of run j's first entry, run_start[runs] = nnz
__global__ __launch_bounds__(kBlock) void pbf_runstart_kernel(PbRunHeadIn head, const uint32_t *__restrict__ hscan,
                                                              int64_t nnz, uint32_t *__restrict__ run_start) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nnz) return;
  if (head(i)) run_start[hscan[i] - 1] = (uint32_t)i;
  if (i == nnz - 1) run_start[hscan[i]] = (uint32_t)nnz;
}

struct PbSingleIn {  // 1 where run j is a single entry
  const uint32_t *run_start;
  __device__ __forceinline__ uint32_t operator()(int64_t j) const { return run_start[j + 1] - run_start[j] == 1u ? 1u : 0u; }
};

// Order of a piece's runs: the runs of two entries and more first, the single entries behind them, each of the two
// classes dealt by itself (the deal of pb_build_kernel, applied to runs).  On C3 two runs in three are single entries
// (a fifth of the entries): side by side they fill whole instructions of pass 1, which then skips its segmented sums
// (every lane a run of its own) and stores 64 neighbouring sums at once.
// piece_run[k] = runs in front of piece k (k = 0 .. pieces); piece_long[k] = how many of piece k's runs are longer than
// one entry; slots[k] = places of the piece: both classes rounded up to whole groups of 32, plus one place that
// carries the piece's padding
__global__ __launch_bounds__(kBlock) void pbf_piece_runs_kernel(const int64_t *__restrict__ piece_ptr,
                                                                const uint32_t *__restrict__ hscan,
                                                                const uint32_t *__restrict__ sscan, int64_t pieces,
                                                                uint32_t *__restrict__ piece_run,
                                                                uint32_t *__restrict__ piece_long,
                                                                int64_t *__restrict__ slots) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k > pieces) return;
  const int64_t a = piece_ptr[k];
  const uint32_t ra = a > 0 ? hscan[a - 1] : 0u;
  piece_run[k] = ra;
  if (k == pieces) { slots[k] = 0; piece_long[k] = 0; return; }
  const int64_t b = piece_ptr[k + 1];
  const uint32_t rb = b > 0 ? hscan[b - 1] : 0u;
  const uint32_t singles = (rb > 0 ? sscan[rb - 1] : 0u) - (ra > 0 ? sscan[ra - 1] : 0u), longs = (rb - ra) - singles;
  piece_long[k] = longs;
  slots[k] = (int64_t)((longs + 31u) & ~31u) + (int64_t)((singles + 31u) & ~31u) + 1;
}

// run j -> its place in its piece; len[place] = entries of the run; the piece's extra place gets its padding
__global__ __launch_bounds__(kBlock) void pbf_slot_kernel(const uint32_t *__restrict__ sorted_keys,
                                                          const uint32_t *__restrict__ run_start, int64_t runs,
                                                          const uint32_t *__restrict__ piece_run,
                                                          const uint32_t *__restrict__ piece_long,
                                                          const uint32_t *__restrict__ sscan,
                                                          const int64_t *__restrict__ sbase,
                                                          const int64_t *__restrict__ piece_ptr,
                                                          uint32_t *__restrict__ len, uint32_t *__restrict__ run_slot) {
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= runs) return;
  const uint32_t i0 = run_start[j], n = run_start[j + 1] - i0;
  const int64_t k = (int64_t)(sorted_keys[i0] >> 5);
  const int64_t first = (int64_t)piece_run[k], all = (int64_t)piece_run[k + 1] - first, longs = (int64_t)piece_long[k];
  const int64_t sb = (int64_t)sscan[j] - (first > 0 ? (int64_t)sscan[first - 1] : 0);  // single entries of the piece up to and including run j
  const int64_t lbase = (longs + 31) & ~(int64_t)31;
  int64_t jl, N, base;
  if (n == 1) { jl = sb - 1; N = all - longs; base = sbase[k] + lbase; }
  else { jl = (j - first) - sb; N = longs; base = sbase[k]; }
  const int64_t G = (N + 31) >> 5, sl = jl / G;
  const int64_t slot = base + (jl % G) * 32 + ((sl & 1) * 16 + (sl >> 1));
  len[slot] = n;
  run_slot[j] = (uint32_t)slot;
  if (j == first) len[sbase[k] + lbase + (((all - longs) + 31) & ~(int64_t)31)] = (uint32_t)((-(piece_ptr[k + 1] - piece_ptr[k])) & 31);
}

template <class T>
__global__ __launch_bounds__(kBlock) void pbf_place_kernel(const int32_t *__restrict__ col, const T *__restrict__ val,
                                                           int64_t nnz, const uint32_t *__restrict__ rank_of_line,
                                                           int64_t nb, const uint32_t *__restrict__ perm,
                                                           const uint32_t *__restrict__ hscan,
                                                           const uint32_t *__restrict__ run_start,
                                                           const uint32_t *__restrict__ run_slot,
                                                           const int64_t *__restrict__ spos, const uint16_t *__restrict__ rr,
                                                           uint16_t *__restrict__ col16, T *__restrict__ pval,
                                                           uint16_t *__restrict__ rowt) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nnz) return;
  const int64_t src = perm ? (int64_t)perm[i] : i;
  const uint32_t j = hscan[i] - 1;
  const uint32_t i0 = run_start[j];
  const int64_t pos = spos[run_slot[j]] + (i - (int64_t)i0);
  const int32_t c = col[src];
  const int64_t q = rank_of_line[c / kPbLine];
  const bool head = i == (int64_t)i0 || (pos & (kPbRunCap - 1)) == 0;
  col16[pos] = (uint16_t)((q / nb) * kPbLine + (c % kPbLine));
  pval[pos] = val[src];
  rowt[pos] = (uint16_t)(rr[i] | (head ? kPbHeadFlag : 0));
}

// per piece: its first pad starts the trash run; heads per group counted afterwards
__global__ __launch_bounds__(kBlock) void pbf_padhead_kernel(const int64_t *__restrict__ piece_ptr,
                                                             const uint32_t *__restrict__ piece_run,
                                                             const uint32_t *__restrict__ piece_long,
                                                             const int64_t *__restrict__ sbase,
                                                             const int64_t *__restrict__ spos, int64_t pieces, int trash_row,
                                                             uint16_t *__restrict__ rowt) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= pieces) return;
  if (((piece_ptr[k + 1] - piece_ptr[k]) & 31) == 0) return;
  const int64_t all = (int64_t)piece_run[k + 1] - (int64_t)piece_run[k], longs = (int64_t)piece_long[k];
  rowt[spos[sbase[k] + ((longs + 31) & ~(int64_t)31) + (((all - longs) + 31) & ~(int64_t)31)]] = (uint16_t)(trash_row | kPbHeadFlag);
}

__global__ __launch_bounds__(kBlock) void pbf_group_heads_kernel(const uint16_t *__restrict__ rowt, int64_t groups,
                                                                 uint32_t *__restrict__ cnt) {
  const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (g > groups) return;
  uint32_t n = 0;
  if (g < groups) {
    const uint4 *p = reinterpret_cast<const uint4 *>(rowt + g * 32);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint4 w = p[q];
      n += __popc(w.x & 0x80008000u) + __popc(w.y & 0x80008000u) + __popc(w.z & 0x80008000u) + __popc(w.w & 0x80008000u);
    }
  }
  cnt[g] = n;  // element `groups` = 0: the exclusive scan leaves the total there
}

// outputs of every piece, rounded up (pass 1's stores like a piece to start on a line; the pads cost pass 2 nothing
// but their bytes since each of an instruction's pads has a trash slot of its own: with one shared slot they took
// 16 turns on it and pass 2 ran 0.325 ms instead of 0.260), written where the panel-major scan wants them
__global__ __launch_bounds__(kBlock) void pbf_osz_kernel(const int64_t *__restrict__ sbase, const int64_t *__restrict__ spos,
                                                         const uint32_t *__restrict__ hx, int64_t pieces, int64_t np,
                                                         int64_t nb, uint32_t round, int64_t *__restrict__ poff) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k > pieces) return;
  if (k == pieces) { poff[k] = 0; return; }
  const int64_t g0 = spos[sbase[k]] >> 5, g1 = spos[sbase[k + 1]] >> 5;
  const int64_t b = k / np, p = k - b * np;
  poff[p * nb + b] = (int64_t)((hx[g1] - hx[g0] + round - 1u) / round * round);
}

// A panel's products start on a multiple of 64: pass 2 reads them 64 at a time (128 bytes of rows, 256 of sums per
// instruction), and instructions that share a line make the second wait in the L1 for the line the first asked for.
__global__ __launch_bounds__(kBlock) void pbf_panel_align_kernel(int64_t np, int64_t nb, int64_t *__restrict__ poff) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= np) return;
  int64_t sum = 0;
  for (int64_t b = 0; b < nb; b++) sum += poff[p * nb + b];
  poff[p * nb + nb - 1] += (-sum) & 63;
}

__global__ __launch_bounds__(kBlock) void pbf_uout_kernel(const int64_t *__restrict__ sbase, const int64_t *__restrict__ spos,
                                                          const uint32_t *__restrict__ hx, const int64_t *__restrict__ poff,
                                                          int64_t pieces, int64_t np, int64_t nb,
                                                          uint32_t *__restrict__ uout) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= pieces) return;
  const int64_t g0 = spos[sbase[k]] >> 5, g1 = spos[sbase[k + 1]] >> 5;
  const int64_t b = k / np, p = k - b * np;
  const int64_t base = poff[p * nb + b];
  const uint32_t h0 = hx[g0];
  for (int64_t g = g0; g < g1; g++) uout[g] = (uint32_t)(base + (int64_t)(hx[g] - h0));
}

// the row of every output, where pass 2 streams it
__global__ __launch_bounds__(kBlock) void pbf_rows_out_kernel(const uint16_t *__restrict__ rowt, int64_t ptotal,
                                                              const uint32_t *__restrict__ uout,
                                                              uint16_t *__restrict__ row16) {
  const int64_t pos = (int64_t)blockIdx.x * kBlock + threadIdx.x;  // ptotal is a multiple of 32, blocks of 256
  const bool ok = pos < ptotal;
  const bool head = ok && (rowt[pos] & kPbHeadFlag) != 0;
  const unsigned long long heads = __ballot(head);
  if (head) row16[(int64_t)uout[pos >> 5] + pb_rank_in_half(heads, (int)(threadIdx.x & 63))] = (uint16_t)(rowt[pos] & (kPbHeadFlag - 1));
}

__global__ __launch_bounds__(kBlock) void pbf_bounds_kernel(const int64_t *__restrict__ sbase, const int64_t *__restrict__ spos,
                                                            const int64_t *__restrict__ poff, int64_t np, int64_t nb,
                                                            int64_t *__restrict__ bounds) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i <= nb) bounds[i] = spos[sbase[i * np]];
  else if (i <= nb + 1 + np) bounds[i] = poff[(i - nb - 1) * nb];
}

// Builds the folded layout from the sorted entries.  *done = false (and nothing kept) when folding would not shorten
// the product stream enough, or the sizes leave the 32-bit indices.
template <class T>
int pbf_build(futhark_context *ctx, const CsrBody<T> *a, const uint32_t *eks, const uint32_t *eperm,
              const int64_t *piece_ptr, const int64_t *panel_row, const uint32_t *rank_of_line, int64_t nb, int64_t np,
              int64_t pieces, bool sorted, bool *done) {
  *done = false;
  const int64_t nnz = a->nnz;
  const unsigned egrid = (unsigned)cdiv(nnz, kBlock), pgrid = (unsigned)cdiv(pieces + 1, kBlock);
  Tmp<uint16_t> rr(ctx);
  Tmp<uint32_t> hscan(ctx);
  SAMD_TRY(rr.get(nnz));
  SAMD_TRY(hscan.get(nnz));
  hipLaunchKernelGGL(pbf_rows_kernel, dim3(egrid), dim3(kBlock), 0, ctx->stream, (const int64_t *)a->row_ptr, nnz,
                     (const int64_t *)a->tile_lb, a->tile, panel_row, np, eks, eperm, rr.p);
  SAMD_LAUNCH_CHECK(ctx);
  const PbRunHeadIn head{eks, rr.p};
  uint32_t *d_runs = reinterpret_cast<uint32_t *>(ctx->d_small + 4);
  SAMD_TRY((device_scan<uint32_t, OpAdd, PbRunHeadIn, false>(ctx, head, nnz, 0u, OpAdd(), hscan.p, d_runs)));
  int64_t small[8];
  SAMD_TRY(read_small(ctx, ctx->d_small, 5, small));
  const int64_t runs = (int64_t)(uint32_t)(small[4] & 0xffffffff);
  if (ctx->logging) fprintf(stderr, "[sparse_amd] two-pass SpMV: %lld runs for %lld entries\n", (long long)runs, (long long)nnz);
  // not worth the fold arithmetic?  (With sorted blocks there is no other layout: the plain pass 1 splits every block
  // into the same number of shares.)
  // (Round 2 declined below 15 % fewer products.  Round 3's pass 1 reads wide and gathers x itself: measured
  // against the plain layout it wins whatever folds -- C5 whole, 8.5 % fewer products: 5.84 -> 5.10 ms; its shard
  // 0.760 -> 0.725; 12.5 entries per row, 1 % fewer: 0.884 -> 0.867 -- so the plain layout is what option
  // spmv_pb_fold = -1 and the 32-bit limits below fall back to.)
  if (runs + 65 * pieces >= (int64_t)0xffffff00LL) return 0;                      // 32-bit places and outputs
  Tmp<uint32_t> run_start(ctx), piece_run(ctx), run_slot(ctx), len(ctx);
  Tmp<int64_t> sbase(ctx), spos(ctx);
  SAMD_TRY(run_start.get(runs + 1));
  SAMD_TRY(piece_run.get(pieces + 1));
  SAMD_TRY(run_slot.get(runs));
  SAMD_TRY(sbase.get(pieces + 1));
  hipLaunchKernelGGL(pbf_runstart_kernel, dim3(egrid), dim3(kBlock), 0, ctx->stream, head, (const uint32_t *)hscan.p, nnz,
                     run_start.p);
  SAMD_LAUNCH_CHECK(ctx);
  Tmp<uint32_t> sscan(ctx), piece_long(ctx);
  SAMD_TRY(sscan.get(runs));
  SAMD_TRY(piece_long.get(pieces + 1));
  SAMD_TRY((device_scan<uint32_t, OpAdd, PbSingleIn, false>(ctx, PbSingleIn{run_start.p}, runs, 0u, OpAdd(), sscan.p,
                                                             (uint32_t *)nullptr)));
  hipLaunchKernelGGL(pbf_piece_runs_kernel, dim3(pgrid), dim3(kBlock), 0, ctx->stream, piece_ptr, (const uint32_t *)hscan.p,
                     (const uint32_t *)sscan.p, pieces, piece_run.p, piece_long.p, sbase.p);
  SAMD_LAUNCH_CHECK(ctx);
  SAMD_TRY((device_scan<int64_t, OpAdd, PtrIn<int64_t>, true>(ctx, PtrIn<int64_t>{sbase.p}, pieces + 1, (int64_t)0, OpAdd(),
                                                               sbase.p, (int64_t *)nullptr)));
  SAMD_HIP(ctx, hipMemcpyAsync(ctx->d_small, sbase.p + pieces, sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
  SAMD_TRY(read_small(ctx, ctx->d_small, 1, small));
  const int64_t places = small[0];  // all pieces' places in the dealt order
  SAMD_TRY(len.get(places + 1));
  SAMD_TRY(spos.get(places + 1));
  SAMD_HIP(ctx, hipMemsetAsync(len.p, 0, sizeof(uint32_t) * (size_t)(places + 1), ctx->stream));
  hipLaunchKernelGGL(pbf_slot_kernel, dim3((unsigned)cdiv(runs, kBlock)), dim3(kBlock), 0, ctx->stream, eks,
                     (const uint32_t *)run_start.p, runs, (const uint32_t *)piece_run.p, (const uint32_t *)piece_long.p,
                     (const uint32_t *)sscan.p, (const int64_t *)sbase.p, piece_ptr, len.p, run_slot.p);
  SAMD_LAUNCH_CHECK(ctx);
  SAMD_TRY((device_scan<int64_t, OpAdd, PbU32AsI64In, true>(ctx, PbU32AsI64In{len.p}, places + 1, (int64_t)0, OpAdd(), spos.p,
                                                             (int64_t *)nullptr)));
  SAMD_HIP(ctx, hipMemcpyAsync(ctx->d_small, spos.p + places, sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
  SAMD_TRY(read_small(ctx, ctx->d_small, 1, small));
  const int64_t ptotal = small[0];  // entries of the block-major layout, pads included: a multiple of 32
  const int64_t groups = ptotal >> 5;
  if ((ptotal & 31) != 0) return set_error(ctx, "internal: folded two-pass layout is not made of whole groups");
  Tmp<uint16_t> col16(ctx), rowt(ctx), row16(ctx);
  Tmp<T> pval(ctx), prod(ctx), xperm(ctx);
  Tmp<uint32_t> hx(ctx), uout(ctx);
  Tmp<int64_t> poff(ctx), bounds(ctx);
  SAMD_TRY(col16.get(ptotal));
  SAMD_TRY(rowt.get(ptotal));
  SAMD_TRY(pval.get(ptotal));
  SAMD_TRY(hx.get(groups + 1));
  SAMD_TRY(uout.get(groups));
  SAMD_TRY(poff.get(pieces + 1));
  SAMD_TRY(bounds.get(nb + 1 + np + 1));
  SAMD_HIP(ctx, hipMemsetAsync(col16.p, 0, sizeof(uint16_t) * (size_t)ptotal, ctx->stream));
  SAMD_HIP(ctx, hipMemsetAsync(pval.p, 0, sizeof(T) * (size_t)ptotal, ctx->stream));
  SAMD_HIP(ctx, hipMemsetD16Async(reinterpret_cast<hipDeviceptr_t>(rowt.p), (unsigned short)PbSlice<T>::kRows, (size_t)ptotal,
                                  ctx->stream));
  hipLaunchKernelGGL((pbf_place_kernel<T>), dim3(egrid), dim3(kBlock), 0, ctx->stream, (const int32_t *)a->col32,
                     (const T *)a->vals, nnz, rank_of_line, nb, eperm, (const uint32_t *)hscan.p, (const uint32_t *)run_start.p,
                     (const uint32_t *)run_slot.p, (const int64_t *)spos.p, (const uint16_t *)rr.p, col16.p, pval.p, rowt.p);
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(pbf_padhead_kernel, dim3(pgrid), dim3(kBlock), 0, ctx->stream, piece_ptr, (const uint32_t *)piece_run.p,
                     (const uint32_t *)piece_long.p,
                     (const int64_t *)sbase.p, (const int64_t *)spos.p, pieces, (int)PbSlice<T>::kRows, rowt.p);
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(pbf_group_heads_kernel, dim3((unsigned)cdiv(groups + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                     (const uint16_t *)rowt.p, groups, hx.p);
  SAMD_LAUNCH_CHECK(ctx);
  SAMD_TRY((device_scan<uint32_t, OpAdd, PtrIn<uint32_t>, true>(ctx, PtrIn<uint32_t>{hx.p}, groups + 1, 0u, OpAdd(), hx.p,
                                                                 (uint32_t *)nullptr)));
  hipLaunchKernelGGL(pbf_osz_kernel, dim3(pgrid), dim3(kBlock), 0, ctx->stream, (const int64_t *)sbase.p, (const int64_t *)spos.p,
                     (const uint32_t *)hx.p, pieces, np, nb,
                     (uint32_t)(ctx->opt_spmv_pb_oround > 0 ? ctx->opt_spmv_pb_oround : kPbOutRound), poff.p);
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(pbf_panel_align_kernel, dim3((unsigned)cdiv(np, kBlock)), dim3(kBlock), 0, ctx->stream, np, nb, poff.p);
  SAMD_LAUNCH_CHECK(ctx);
  SAMD_TRY((device_scan<int64_t, OpAdd, PtrIn<int64_t>, true>(ctx, PtrIn<int64_t>{poff.p}, pieces + 1, (int64_t)0, OpAdd(),
                                                               poff.p, (int64_t *)nullptr)));
  SAMD_HIP(ctx, hipMemcpyAsync(ctx->d_small, poff.p + pieces, sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
  SAMD_TRY(read_small(ctx, ctx->d_small, 1, small));
  const int64_t ototal = small[0];  // outputs
  if (ototal >= (int64_t)0xffffff00LL) return 0;
  SAMD_TRY(row16.get(ototal));
  hipLaunchKernelGGL(pb_fill_trash_kernel, dim3((unsigned)cdiv(ototal, kBlock)), dim3(kBlock), 0, ctx->stream, row16.p, ototal,
                     (int)PbSlice<T>::kRows);
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(pbf_uout_kernel, dim3((unsigned)cdiv(pieces, kBlock)), dim3(kBlock), 0, ctx->stream, (const int64_t *)sbase.p,
                     (const int64_t *)spos.p, (const uint32_t *)hx.p, (const int64_t *)poff.p, pieces, np, nb, uout.p);
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(pbf_rows_out_kernel, dim3((unsigned)cdiv(ptotal, kBlock)), dim3(kBlock), 0, ctx->stream,
                     (const uint16_t *)rowt.p, ptotal, (const uint32_t *)uout.p, row16.p);
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(pbf_bounds_kernel, dim3((unsigned)cdiv(nb + 1 + np + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                     (const int64_t *)sbase.p, (const int64_t *)spos.p, (const int64_t *)poff.p, np, nb, bounds.p);
  SAMD_LAUNCH_CHECK(ctx);
  SAMD_TRY(prod.get(ototal + 64));  // behind the outputs: the trash the pads of the chunked layout write to
  SAMD_HIP(ctx, hipMemsetAsync(prod.p, 0, sizeof(T) * (size_t)(ototal + 64), ctx->stream));  // output pads are read by pass 2 (into the trash slots)
  // ---- the chunked layout pass 1 reads, and pass 1's shares: about four workgroups per CU, every block cut into
  // shares of whole chunks in proportion to its entries (at least one)
  std::vector<int64_t> hb((size_t)nb + 1);
  SAMD_HIP(ctx, hipMemcpyAsync(hb.data(), bounds.p, sizeof(int64_t) * (size_t)(nb + 1), hipMemcpyDeviceToHost, ctx->stream));
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<int64_t> wt, cb((size_t)nb + 1);
  cb[0] = 0;
  for (int64_t b = 0; b < nb; b++) cb[(size_t)b + 1] = cb[(size_t)b] + cdiv(hb[(size_t)b + 1] - hb[(size_t)b], kPbChunk);
  const int64_t nchunks = cb[(size_t)nb];
  {
    const int64_t target = (int64_t)ctx->num_cu * (ctx->opt_spmv_pb_shares > 0 ? ctx->opt_spmv_pb_shares : 4);
    const int64_t total = nchunks > 0 ? nchunks : 1;
    for (int64_t b = 0; b < nb; b++) {
      const int64_t c0 = cb[(size_t)b], c1 = cb[(size_t)b + 1];
      if (c1 <= c0) continue;
      int64_t shares = ((c1 - c0) * target + total / 2) / total;
      if (shares < 1) shares = 1;
      if (shares > c1 - c0) shares = c1 - c0;
      const int64_t per = cdiv(c1 - c0, shares);
      for (int64_t c = c0; c < c1; c += per) {
        wt.push_back(c);
        wt.push_back(c + per < c1 ? c + per : c1);
        wt.push_back(b);
      }
    }
  }
  Tmp<int64_t> wtab(ctx), cbase(ctx);
  Tmp<uint16_t> colc(ctx);
  Tmp<T> valc(ctx);
  Tmp<uint32_t> outc(ctx);
  SAMD_TRY(wtab.get((int64_t)wt.size()));
  SAMD_TRY(cbase.get(nb + 1));
  SAMD_TRY(colc.get(nchunks * kPbChunk));
  SAMD_TRY(valc.get(nchunks * kPbChunk));
  SAMD_TRY(outc.get(nchunks * (kPbChunk / 16)));
  SAMD_HIP(ctx, hipMemcpyAsync(cbase.p, cb.data(), sizeof(int64_t) * cb.size(), hipMemcpyHostToDevice, ctx->stream));
  if (!wt.empty())
    SAMD_HIP(ctx, hipMemcpyAsync(wtab.p, wt.data(), sizeof(int64_t) * wt.size(), hipMemcpyHostToDevice, ctx->stream));
  if (nchunks > 0) {
    hipLaunchKernelGGL((pbf_chunk_kernel<T>), dim3((unsigned)cdiv(nchunks, kBlock / 64)), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)bounds.p, (const int64_t *)cbase.p, nb, nchunks, (const uint16_t *)col16.p,
                       (const uint16_t *)rowt.p, (const T *)pval.p, (const uint32_t *)uout.p, (uint32_t)ototal, colc.p, valc.p,
                       outc.p);
    SAMD_LAUNCH_CHECK(ctx);
  }
  SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));  // wt and cb are host memory that goes away; the temporaries are released after their use
  a->pb_wtab = wtab.release();
  a->pb_nwg = (int64_t)wt.size() / 3;
  a->pb_piece_ptr = bounds.release();
  a->pb_dgroup = outc.release();
  a->pb_col16 = colc.release();
  a->pb_row16 = row16.release();
  a->pb_val = valc.release();
  a->pb_prod = prod.release();
  a->pb_fold = 1;
  a->pb_outputs = ototal;
  if (ctx->logging) fprintf(stderr, "[sparse_amd] two-pass SpMV, folded: %lld entries (padded %lld) -> %lld products\n",
                            (long long)nnz, (long long)ptotal, (long long)ototal);
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
  const int64_t slice = kPbXSliceBytes / (int64_t)sizeof(T);
  const int64_t yslice = PbSlice<T>::kRows;  // rows of y a panel can hold
  int64_t R = ctx->opt_spmv_pb_rows > 0 ? ctx->opt_spmv_pb_rows : yslice;
  int64_t lpb = ctx->opt_spmv_pb_lines > 0 ? ctx->opt_spmv_pb_lines : slice / kPbLine;  // lines per block
  int64_t E = ctx->opt_spmv_pb_entries > 0 ? ctx->opt_spmv_pb_entries : kPbPanelEntries;  // (the folded layout's default: below, once `sorted` is known)
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
  if (sorted && ctx->opt_spmv_pb_entries <= 0) E = kPbPanelEntriesFolded > a->tile ? kPbPanelEntriesFolded : a->tile;
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
  if (a->pb_fold) {  // the same list block by block, for pass 1's slice loads
    Tmp<uint32_t> line_blk(ctx);
    SAMD_TRY(line_blk.get(nq));
    hipLaunchKernelGGL(pb_line_blk_kernel, dim3((unsigned)cdiv(nq, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const uint32_t *)line_of_rank.p, nq, nb, line_blk.p);
    SAMD_LAUNCH_CHECK(ctx);
    a->pb_line_blk = line_blk.release();
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
  // folded layout, x 16-byte aligned: pass 1 takes its slices from x itself
  const bool gather = a->pb_fold && a->pb_line_blk != nullptr && ((uintptr_t)x & 15) == 0 && ctx->opt_spmv_pb_xcopy <= 0;
  if (gather) {}
  else if (((uintptr_t)x & 15) == 0)
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
#define SAMD_PB1F(LAB, WIDE)                                                                                           \
  hipLaunchKernelGGL((pb_pass1f_kernel<T, LAB, WIDE>), grid1, dim3(kPbThreads), 0, ctx->stream,                        \
                     (const int64_t *)a->pb_wtab, (const uint16_t *)a->pb_col16, (const T *)a->pb_val,                 \
                     (const T *)a->pb_x, (const uint32_t *)a->pb_dgroup, a->pb_prod, (int)a->pb_w,                     \
                     gather ? (const uint32_t *)a->pb_line_blk : (const uint32_t *)nullptr, x, a->m)
    // products beyond a 32-bit byte offset (or option ablate bit 30, for the tests): stores through 64-bit addresses
    const bool wide = (a->pb_outputs + 64) * (int64_t)sizeof(T) >= (int64_t)0x7fffffff || ((ctx->opt_ablate >> 30) & 1);
    if (a->pb_nwg <= 0) {}  // nothing stored
    else if (lab1 == 1) SAMD_PB1F(1, false);
    else if (lab1 == 2) SAMD_PB1F(2, false);
    else if (lab1 == 3) SAMD_PB1F(3, false);
    else if (lab1 == 4) SAMD_PB1F(4, false);
    else if (lab1 == 5) SAMD_PB1F(5, false);
    else if (lab1 == 6) SAMD_PB1F(6, false);
    else if (wide) SAMD_PB1F(0, true);
    else SAMD_PB1F(0, false);
#undef SAMD_PB1F
  }
  else
    hipLaunchKernelGGL((pb_pass1_kernel<T, 8>), dim3((unsigned)a->pb_nb, (unsigned)split), dim3(kPbThreads), 0, ctx->stream,
                       (const int64_t *)a->pb_piece_ptr, (const uint16_t *)a->pb_col16, (const T *)a->pb_val,
                       (const T *)a->pb_x, (const uint32_t *)a->pb_dgroup, a->pb_prod, (int)a->pb_w,
                       (int)((ctx->opt_ablate >> 2) & 3));
  SAMD_LAUNCH_CHECK(ctx);
  if (a->pb_fold && !((ctx->opt_ablate >> 29) & 1))  // (bit 29: the one-product-per-lane kernel, for the tests)
    hipLaunchKernelGGL((pb_pass2p_kernel<T>), dim3((unsigned)a->pb_np), dim3(kPbThreads), 0, ctx->stream,
                       (const int64_t *)(a->pb_piece_ptr + a->pb_nb + 1), (const int64_t *)a->pb_panel_row,
                       (const uint16_t *)a->pb_row16, (const T *)a->pb_prod, y, (int)(ctx->opt_ablate & 3));
  else
    hipLaunchKernelGGL((pb_pass2_kernel<T, 8>), dim3((unsigned)a->pb_np), dim3(kPbThreads), 0, ctx->stream,
                     (const int64_t *)(a->pb_piece_ptr + a->pb_nb + 1), (const int64_t *)a->pb_panel_row,
                     (const uint16_t *)a->pb_row16, (const T *)a->pb_prod, y, (int)(ctx->opt_ablate & 3),
                     ((ctx->opt_ablate & 3) == 2 || a->pb_fold) ? 0 : (ctx->opt_ablate >> 5) ? (int)(ctx->opt_ablate >> 5) : kPbFoldMax);
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

}  // namespace samd
