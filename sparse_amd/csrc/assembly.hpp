// assembly.hpp -- COO -> CSR record (sr.sparse / sc.sparse, compressed.fut:154-160,
// :211-212, norm_coo :72-87) and everything that re-enters it (+, -, physical
// CSR<->CSC conversion), plus the nnz-balanced tiling ("plan") of a record.
//
// Pipeline (all on the context stream):
//   pack      (row, col) i64 pair -> one unsigned key (row << cbits | col), bounds
//             check (the assert at :156) folded in                         [1 pass]
//   sort      stable LSD radix sort on the significant key bits only, payload =
//             the entry's value bits (radix_sort.hpp)               [<= 8 passes]
//   heads     inclusive scan of the head flags key[i] != key[i-1] (:77-81) gives
//             every run its output slot; the scan total is the merged count
//   emit      one thread per element: run heads fold their run left to right in
//             sorted (= input) order (segmented_reduce :83-84) and write
//             col_idx/vals; the last element of each row writes
//             row_ptr[row+1] = slots so far
//   rowptr    inclusive max-scan over row_ptr fills empty rows.  Result equals
//             reduce_by_index + scan (:158-159): row_ptr[r+1] = #entries in
//             rows 0..r = row_off[r].
#pragma once

#include "radix_sort_msd.hpp"

namespace samd {

template <class K>
__device__ __forceinline__ K shl_safe(K v, int s) { return s >= (int)(sizeof(K) * 8) ? (K)0 : (K)(v << s); }
template <class K>
__device__ __forceinline__ K shr_safe(K v, int s) { return s >= (int)(sizeof(K) * 8) ? (K)0 : (K)(v >> s); }

// ---------------------------------------------------------------- pack kernels
template <class K>
__global__ __launch_bounds__(kBlock) void pack_coo_kernel(const int64_t *__restrict__ major,
                                                          const int64_t *__restrict__ minor, int64_t count,
                                                          int64_t n_major, int64_t n_minor, int minor_bits,
                                                          K *__restrict__ keys, int *__restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= count) return;
  const int64_t r = major[i], c = minor[i];
  K k = 0;
  if (r < 0 || r >= n_major || c < 0 || c >= n_minor) *err = 1;
  else k = shl_safe<K>((K)r, minor_bits) | (K)c;
  keys[i] = k;
}

// The same for large inputs, with digit histograms of the keys taken along the way: the digits of an LSD sort from bit
// 0 and the (up to three) top-aligned digits the top-bits-first sort starts with (radix_sort_msd.hpp; the same digits
// when the key width is a multiple of 8).  Whichever sort the assembly then picks finds its histograms made, and the
// separate read of all keys for them (rs_hist_all: C2 0.18 ms, C5 2 ms, C3 1.3 ms) drops out.  A fixed grid walks
// the input; one add per run of equal neighbouring digits as in rs_hist_all.
constexpr int kPackDigitsMax = 12;
struct PackDigits {
  int n;
  int shift[kPackDigitsMax], bits[kPackDigitsMax];
};

__global__ __launch_bounds__(kBlock) void pack_coo_hist_kernel(const int64_t *__restrict__ major,
                                                               const int64_t *__restrict__ minor, int64_t count,
                                                               int64_t n_major, int64_t n_minor, int minor_bits,
                                                               uint64_t *__restrict__ keys, int *__restrict__ err,
                                                               PackDigits dg, uint32_t *__restrict__ ghist) {
  __shared__ uint32_t h[kPackDigitsMax][256];
  for (int d = threadIdx.x; d < dg.n * 256; d += kBlock) (&h[0][0])[d] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  bool bad = false;
  for (int64_t base = (int64_t)blockIdx.x * kBlock; base < count; base += (int64_t)gridDim.x * kBlock) {
    const int64_t i = base + threadIdx.x;
    const bool ok = i < count;
    uint64_t k = 0;
    if (ok) {
      const int64_t r = major[i], c = minor[i];
      if (r < 0 || r >= n_major || c < 0 || c >= n_minor) bad = true;
      else k = shl_safe<uint64_t>((uint64_t)r, minor_bits) | (uint64_t)c;
      keys[i] = k;
    }
    for (int p = 0; p < dg.n; p++) {  // uniform
      const uint32_t d = ok ? ((uint32_t)(k >> dg.shift[p]) & ((1u << dg.bits[p]) - 1u)) : 0xffffffffu;
      const uint32_t before = wave_prev_lane_dpp(d);
      const bool head = lane == 0 || d != before;
      const uint64_t hm = __ballot(head);
      if (head && ok) {
        const uint64_t rest = lane == 63 ? 0ull : (hm >> (lane + 1));
        const uint32_t len = rest ? (uint32_t)__builtin_ctzll(rest) + 1u : (uint32_t)(64 - lane);
        atomicAdd(&h[p][d], len);
      }
    }
  }
  if (bad) *err = 1;
  __syncthreads();
  for (int d = threadIdx.x; d < dg.n * 256; d += kBlock) {
    const uint32_t v = (&h[0][0])[d];
    if (v) atomicAdd(&ghist[d], v);
  }
}

// Row that stores entry e of a record, searched only inside e's tile
// (tile_lb brackets the candidates, so the search is a handful of steps in
// cache-resident data instead of log2(n) over all of row_ptr).
__device__ __forceinline__ int64_t find_row(const int64_t *__restrict__ row_ptr, const int64_t *__restrict__ tile_lb,
                                            int64_t tile, int64_t e) {
  const int64_t t = e / tile;
  int64_t lo = tile_lb[t] > 0 ? tile_lb[t] - 1 : 0;
  int64_t hi = tile_lb[t + 1];  // rows >= hi start beyond this tile
  // largest r in [lo, hi) with row_ptr[r] <= e
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (row_ptr[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// Rows of all entries of tile t at once, for a block of kBlock threads (tile = kPlanTile = 4 * kBlock entries):
// every row that starts in the tile leaves its number at its start (empty rows share a start: the largest wins, and
// that is the row the entry belongs to), an inclusive max-scan in entry order spreads it.  Thread holds the entries
// at tile positions wave * 256 + k * 64 + lane, k < 4 (the layout rule of scan.hpp).  Reads row_ptr once per row
// instead of a binary search per entry (find_row: ~10 dependent L2 loads): pack_csr 0.97 -> 0.3 ms on C2.
__device__ __forceinline__ void tile_rows(const int64_t *__restrict__ row_ptr, const int64_t *__restrict__ tile_lb,
                                          int64_t t, unsigned *mark /* LDS, 4 * kBlock */, unsigned *sh /* LDS, kWaves */,
                                          int64_t (&row)[4]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = t * (4 * kBlock);
  const int64_t lb0 = tile_lb[t], lb1 = tile_lb[t + 1];
  const int64_t rfirst = lb0 > 0 ? lb0 - 1 : 0;
#pragma unroll
  for (int k = 0; k < 4; k++) mark[k * kBlock + tid] = 0u;
  __syncthreads();
  // The last tile lists the record's trailing empty rows too (tile_lb[ntiles] = n): their start is nnz, which is one
  // past the tile when nnz is a multiple of the tile size.  No entry reads that mark, so such rows leave none.
  for (int64_t r = lb0 + tid; r < lb1; r += kBlock) {
    const int64_t off = row_ptr[r] - base;
    if (off < 4 * kBlock) atomicMax(&mark[(int)off], (unsigned)(r - rfirst));
  }
  __syncthreads();
  unsigned v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = mark[wave * 256 + k * 64 + lane];
  unsigned total;
  block_scan_striped<unsigned, OpMax, 4>(v, 0u, OpMax(), sh, total);
#pragma unroll
  for (int k = 0; k < 4; k++) row[k] = rfirst + (int64_t)v[k];
}

// keys of a whole tile per block (grid = ntiles); rows through tile_rows
template <class K, class I, bool SWAP>
__global__ __launch_bounds__(kBlock) void pack_csr_tile_kernel(const int64_t *__restrict__ row_ptr,
                                                               const I *__restrict__ col, int64_t nnz,
                                                               const int64_t *__restrict__ tile_lb, int minor_bits,
                                                               K *__restrict__ keys) {
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
      const K r = (K)row[k], c = (K)col[e];
      keys[e] = SWAP ? (shl_safe<K>(c, minor_bits) | r) : (shl_safe<K>(r, minor_bits) | c);
    }
  }
}

// The transposing form for large records, with the digit histograms the sort will start from taken along the way
// (as pack_coo_hist_kernel does): a fixed grid walks the tiles.
template <class K, class I>
__global__ __launch_bounds__(kBlock) void pack_csr_tile_hist_kernel(const int64_t *__restrict__ row_ptr,
                                                                    const I *__restrict__ col, int64_t nnz, int64_t ntiles,
                                                                    const int64_t *__restrict__ tile_lb, int minor_bits,
                                                                    K *__restrict__ keys, PackDigits dg,
                                                                    uint32_t *__restrict__ ghist) {
  __shared__ unsigned mark[4 * kBlock];
  __shared__ unsigned sh[kWaves];
  __shared__ uint32_t h[kPackDigitsMax][256];
  for (int d = threadIdx.x; d < dg.n * 256; d += kBlock) (&h[0][0])[d] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    int64_t row[4];
    tile_rows(row_ptr, tile_lb, t, mark, sh, row);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t e = t * (4 * kBlock) + wave * 256 + k * 64 + lane;
      const bool ok = e < nnz;
      K key = 0;
      if (ok) {
        key = shl_safe<K>((K)col[e], minor_bits) | (K)row[k];
        keys[e] = key;
      }
      for (int p = 0; p < dg.n; p++) {  // uniform
        const uint32_t d = ok ? ((uint32_t)(key >> dg.shift[p]) & ((1u << dg.bits[p]) - 1u)) : 0xffffffffu;
        const uint32_t before = wave_prev_lane_dpp(d);
        const bool head = lane == 0 || d != before;
        const uint64_t hm = __ballot(head);
        if (head && ok) {
          const uint64_t rest = lane == 63 ? 0ull : (hm >> (lane + 1));
          atomicAdd(&h[p][d], rest ? (uint32_t)__builtin_ctzll(rest) + 1u : (uint32_t)(64 - lane));
        }
      }
    }
    __syncthreads();  // tile_rows' LDS is reused by the next tile
  }
  __syncthreads();
  for (int d = threadIdx.x; d < dg.n * 256; d += kBlock) {
    const uint32_t v = (&h[0][0])[d];
    if (v) atomicAdd(&ghist[d], v);
  }
}

// key of stored entry e: (row << minor_bits | col), or swapped (col << minor_bits | row)
template <class K, class I, bool SWAP>
__global__ __launch_bounds__(kBlock) void pack_csr_kernel(const int64_t *__restrict__ row_ptr,
                                                          const I *__restrict__ col, int64_t nnz,
                                                          const int64_t *__restrict__ tile_lb, int64_t tile,
                                                          int minor_bits, K *__restrict__ keys) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= nnz) return;
  const K r = (K)find_row(row_ptr, tile_lb, tile, e);
  const K c = (K)col[e];
  keys[e] = SWAP ? (shl_safe<K>(c, minor_bits) | r) : (shl_safe<K>(r, minor_bits) | c);
}

template <class T>
__global__ __launch_bounds__(kBlock) void concat_vals_kernel(const T *__restrict__ a, int64_t na,
                                                             const T *__restrict__ b, int64_t nb, bool negate_b,
                                                             T *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < na) out[i] = a[i];
  else if (i < na + nb) out[i] = negate_b ? (T)((T)(-1) * b[i - na]) : b[i - na];  // scale (T.i64 (-1)), :183
}

// How far from its place in the sorted order does an input pair sit?  Sum over 16 * kBlock evenly spread
// pairs of |i - key[i] / keyspace * count| (the rank a key would have if keys were spread evenly), each
// capped at `cap`.  Small for inputs that are sorted, nearly sorted, or a banded record being transposed
// (key = column, position ~ row): the final gather vals[perm[i]] then stays inside a cache-sized window.
template <class KF>
__global__ __launch_bounds__(kBlock) void displacement_sample_kernel(KF keys, int64_t count,
                                                                     double inv_keyspace, unsigned long long cap,
                                                                     unsigned long long *__restrict__ out) {
  const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x, total = (int64_t)gridDim.x * kBlock;
  const int64_t i = (count - 1) / total * s;
  const double pred = (double)keys(i) * inv_keyspace * (double)count;
  const double d = fabs(pred - (double)i);
  unsigned long long v = d >= (double)cap ? cap : (unsigned long long)d;
  v = wave_reduce(v, OpAdd());
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

// ---------------------------------------------------------------- two sorted runs -> one
// sr.+ / sr.- concatenate the entries of two records (compressed.fut:179-183): each half is already in key order, so
// the stable sort of the concatenation is a stable MERGE (first run before second on equal keys).  Merge path: a
// workgroup owns kMergeTile consecutive outputs, finds where its diagonal cuts the two runs (binary search in
// global memory, once per workgroup), stages exactly those kMergeTile keys in LDS, and every thread merges its 8
// outputs after a second search in LDS.  Also writes each output's position in the concatenation (the value's source).
constexpr int kMergeItems = 8;
constexpr int kMergeTile = kBlock * kMergeItems;

template <class K, class A, class B>
__device__ __forceinline__ int64_t merge_path_split(const A &a, int64_t na, const B &b, int64_t nb, int64_t d) {
  int64_t lo = d > nb ? d - nb : 0, hi = d < na ? d : na;  // number of first-run elements among the first d outputs
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (a(mid) <= b(d - 1 - mid)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

template <class K>
__global__ __launch_bounds__(kBlock) void merge_runs_kernel(const K *__restrict__ keys, int64_t na, int64_t nb,
                                                            K *__restrict__ keys_out, uint32_t *__restrict__ src_out) {
  __shared__ K sk[kMergeTile];
  __shared__ int64_t s_split[2];
  const int tid = threadIdx.x;
  const int64_t total = na + nb;
  const int64_t d0 = (int64_t)blockIdx.x * kMergeTile;
  const int64_t d1 = d0 + kMergeTile < total ? d0 + kMergeTile : total;
  const K *ka = keys, *kb = keys + na;
  if (tid < 2) {
    auto ga = [&](int64_t i) { return ka[i]; };
    auto gb = [&](int64_t j) { return kb[j]; };
    s_split[tid] = merge_path_split<K>(ga, na, gb, nb, tid == 0 ? d0 : d1);
  }
  __syncthreads();
  const int64_t i0 = s_split[0], i1 = s_split[1];
  const int64_t j0 = d0 - i0, j1 = d1 - i1;
  const int la = (int)(i1 - i0), lb = (int)(j1 - j0);  // la + lb = d1 - d0 <= kMergeTile
  for (int q = tid; q < la; q += kBlock) sk[q] = ka[i0 + q];
  for (int q = tid; q < lb; q += kBlock) sk[la + q] = kb[j0 + q];
  __syncthreads();
  const int ld = tid * kMergeItems;
  if (ld >= la + lb) return;
  auto sa = [&](int64_t i) { return sk[i]; };
  auto sb = [&](int64_t j) { return sk[la + j]; };
  int ia = (int)merge_path_split<K>(sa, la, sb, lb, ld);
  int ib = ld - ia;
#pragma unroll
  for (int q = 0; q < kMergeItems; q++) {
    if (ld + q >= la + lb) break;
    const bool take_a = ib >= lb || (ia < la && sk[ia] <= sk[la + ib]);
    const K k = take_a ? sk[ia] : sk[la + ib];
    const int64_t src = take_a ? i0 + ia : na + j0 + ib;
    keys_out[d0 + ld + q] = k;
    src_out[d0 + ld + q] = (uint32_t)src;
    if (take_a) ia++; else ib++;
  }
}

// ---------------------------------------------------------------- merge / emit
template <class K>
struct HeadFlagIn {
  const K *keys;
  __device__ __forceinline__ uint32_t operator()(int64_t i) const {
    return (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
  }
};

template <class T> struct BitsOf;
template <> struct BitsOf<double> { typedef uint64_t type; };
template <> struct BitsOf<float> { typedef uint32_t type; };
template <> struct BitsOf<int32_t> { typedef uint32_t type; };

// ---------------------------------------------------------------- long runs of one coordinate
// A run of equal keys is folded by the thread that holds its first pair, one add after the other, in input order:
// the reference's order after a stable sort (compressed.fut:76-87), and what lets the parity tests hold merged
// floating-point values bit for bit.  That is a dependent loop as long as the run, so it is bounded: a run of more
// than kLongRun pairs (found with ONE load -- the keys are sorted, so keys[i + kLongRun] == keys[i] says it all) is
// put on a list instead and summed afterwards by a whole workgroup (fold_long_runs_kernel): every thread adds the
// pairs i + t, i + t + 256, ... in that order into four interleaved partial sums, the partial sums are combined in
// a fixed tree.  Deterministic (run to run, bit for bit); exact for i32 (wrapping adds commute); for floats it is a
// different order than left to right -- compressed.fut:83-84 is a segmented_reduce, whose order the language leaves
// open -- and at least as accurate (partial sums of n / 1024 terms instead of one of n).
constexpr int64_t kLongRun = 4096;

template <class K, class T>
__global__ __launch_bounds__(kBlock) void fold_long_runs_kernel(const K *__restrict__ keys, const uint32_t *__restrict__ perm,
                                                                const T *__restrict__ vals_in, int64_t count,
                                                                const unsigned long long *__restrict__ long_list,
                                                                T *__restrict__ vals_out) {
  __shared__ T part[kWaves];
  const int tid = threadIdx.x;
  const unsigned long long nlist = long_list[0];
  for (unsigned long long b = blockIdx.x; b < nlist; b += gridDim.x) {  // uniform over the block
    const int64_t start = (int64_t)long_list[1 + 2 * b], slot = (int64_t)long_list[2 + 2 * b];
    const K k = keys[start];
    // end of the run: gallop, then bisect (every thread the same loads)
    int64_t lo = start + kLongRun, step = kLongRun;  // keys[lo] == k is known
    while (lo + step < count && keys[lo + step] == k) { lo += step; step *= 2; }
    int64_t hi = lo + step < count ? lo + step : count;  // keys[hi] != k or hi == count
    while (hi - lo > 1) {
      const int64_t mid = lo + (hi - lo) / 2;
      if (keys[mid] == k) lo = mid; else hi = mid;
    }
    const int64_t end = hi;
    T acc[4] = {(T)0, (T)0, (T)0, (T)0};
    int64_t q = start + tid;
    for (; q + 3 * kBlock < end; q += 4 * kBlock) {
      T v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = vals_in[perm ? (int64_t)perm[q + u * kBlock] : q + u * kBlock];
#pragma unroll
      for (int u = 0; u < 4; u++) acc[u] = (T)(acc[u] + v[u]);
    }
    for (int u = 0; q < end; q += kBlock, u++) acc[u] = (T)(acc[u] + vals_in[perm ? (int64_t)perm[q] : q]);
    T sum = (T)((T)(acc[0] + acc[1]) + (T)(acc[2] + acc[3]));
    sum = wave_reduce(sum, OpAdd());
    __syncthreads();  // part[] of the previous run has been read
    if ((tid & 63) == 0) part[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
      T total = part[0];
#pragma unroll
      for (int w = 1; w < kWaves; w++) total = (T)(total + part[w]);
      vals_out[slot] = total;
    }
  }
}

// Queues fold_long_runs_kernel for a list filled by an emission kernel; n_known >= 0: the list's length is already on
// the host (nothing is launched for an empty one), < 0: the kernel reads it from the list itself.
template <class K, class T>
int fold_long_runs(futhark_context *ctx, const K *keys, const uint32_t *perm, const T *vals_in, int64_t count,
                   const unsigned long long *long_list, int64_t n_known, T *vals_out) {
  if (n_known == 0) return 0;
  int64_t grid = n_known > 0 ? n_known : count / kLongRun;
  const int64_t cap = (int64_t)ctx->num_cu * 4;
  if (grid > cap) grid = cap;
  if (grid < 1) return 0;
  hipLaunchKernelGGL((fold_long_runs_kernel<K, T>), dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, keys, perm, vals_in,
                     count, long_list, vals_out);
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

// INDIRECT: vals_in is in source order and perm[i] is the source of sorted entry i
// (nullptr = identity); otherwise vals_in is already in key order.
template <class K, class T, class I, bool INDIRECT>
__global__ __launch_bounds__(kBlock) void merge_emit_kernel(const K *__restrict__ keys,
                                                            const uint32_t *__restrict__ perm,
                                                            const uint32_t *__restrict__ slot,  // inclusive head count
                                                            const T *__restrict__ vals_in, int64_t count,
                                                            int minor_bits, K minor_mask, I *__restrict__ col_out,
                                                            T *__restrict__ vals_out, int64_t *__restrict__ row_ptr,
                                                            unsigned long long *__restrict__ long_list) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= count) return;
  const K k = keys[i];
  const bool head = (i == 0) || keys[i - 1] != k;
  const uint32_t s = slot[i];
  const bool has_next = i + 1 < count;
  const K knext = has_next ? keys[i + 1] : (K)0;
  if (head) {
    auto src = [&](int64_t q) -> int64_t { return (INDIRECT && perm) ? (int64_t)perm[q] : q; };
    T acc = vals_in[src(i)];
    if (has_next && knext == k) {
      if (long_list && i + kLongRun < count && keys[i + kLongRun] == k) {  // sorted keys: the run is longer than kLongRun
        const unsigned long long slot = atomicAdd(long_list, 1ull);
        long_list[1 + 2 * slot] = (unsigned long long)i;
        long_list[2 + 2 * slot] = (unsigned long long)(s - 1);
      } else {
        for (int64_t j = i + 1; j < count && keys[j] == k; j++) acc = (T)(acc + vals_in[src(j)]);
      }
    }
    col_out[s - 1] = (I)(k & minor_mask);
    vals_out[s - 1] = acc;
  }
  const K major = shr_safe<K>(k, minor_bits);
  if (!has_next || shr_safe<K>(knext, minor_bits) != major) row_ptr[(int64_t)major + 1] = (int64_t)s;
}

// Head flags, their scan, the fold and the emission in ONE read of the sorted pairs: a workgroup takes a tile of
// kFoldTile consecutive sorted pairs (tiles by ticket), counts its run heads, learns how many outputs all earlier
// tiles hold from a decoupled look-back over one status word per tile (rl_lookback, as the in-LDS sort's emission
// does) and writes col_idx / vals / row ends.  Replaces scan_reduce + scan_apply + merge_emit_kernel, which between
// them read the keys four times and wrote and read a 4-byte slot per pair (C3: 3.3 of the assembly's 18.9 ms), and
// the host synchronisation that sized the output arrays between them (they are sized for the input here and cut
// down afterwards if few pairs survive).  A run is folded by the thread that holds its first pair, left to right
// (the reference's order), wherever the run ends.
constexpr int kFoldItems = 16;
constexpr int kFoldTile = kBlock * kFoldItems;

template <class K, class T, class I, bool INDIRECT>
__global__ __launch_bounds__(kBlock) void fold_emit_kernel(const K *__restrict__ keys, const uint32_t *__restrict__ perm,
                                                           const T *__restrict__ vals_in, uint32_t count, int minor_bits,
                                                           K minor_mask, I *__restrict__ col_out, T *__restrict__ vals_out,
                                                           int64_t *__restrict__ row_ptr,
                                                           unsigned long long *__restrict__ status,
                                                           uint32_t *__restrict__ ticket,
                                                           unsigned long long *__restrict__ merged_out,
                                                           int *__restrict__ err, int lab,
                                                           unsigned long long *__restrict__ long_list,
                                                           unsigned long long *__restrict__ rows_seen) {  // += rows that end in this tile (every row ends once)
  __shared__ uint32_t sh[kWaves];
  __shared__ uint32_t s_tile;
  __shared__ unsigned long long s_off;
  __shared__ int s_fail;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    s_tile = atomicAdd(ticket, 1u);
    s_fail = 0;
  }
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint32_t ntiles = (count + kFoldTile - 1) / kFoldTile;
  const uint32_t base = tile * (uint32_t)kFoldTile + (uint32_t)wave * (kFoldItems * 64) + (uint32_t)lane;  // layout rule of scan.hpp
  K k[kFoldItems], kn[kFoldItems];
  T val[kFoldItems];
  uint32_t v[kFoldItems];
  uint32_t headmask = 0, longmask = 0;  // longmask: heads whose run goes to fold_long_runs_kernel
#pragma unroll
  for (int j = 0; j < kFoldItems; j++) {
    const uint32_t i = base + j * 64;
    const bool ok = i < count;
    const uint32_t ic = ok ? i : count - 1;  // clamped, never predicated
    k[j] = keys[ic];
    const K kp = keys[ic > 0 ? ic - 1 : 0];
    kn[j] = keys[ic + 1 < count ? ic + 1 : ic];
    // every pair's value is requested here, with the keys and ahead of the scan and the look-back: fetched where it
    // is used (behind the `head` branch, one row of the tile after the other) each load was waited for on its own
    val[j] = vals_in[(INDIRECT && perm) ? (int64_t)perm[ic] : (int64_t)ic];
    const bool head = ok && (i == 0 || kp != k[j]);
    headmask |= head ? (1u << j) : 0u;
    v[j] = head ? 1u : 0u;
  }
  uint32_t total;
  block_scan_striped<uint32_t, OpAdd, kFoldItems>(v, 0u, OpAdd(), sh, total);  // v[j]: heads up to and including pair j
  // wave 0 puts out the tile's count and requests its predecessors' status words; it looks at them after the fold
  unsigned long long lb_words = 0;
  if (tid < 64 && !(lab & 1)) lb_words = rl_publish(status, tile, total);
  // ---- fold: a run's first pair adds the rest of the run, left to right.  The 64 lanes of a wavefront hold 64
  // consecutive pairs (same j), so a run goes on in the lanes to the right: its pairs come by lane shifts, one step
  // for the whole wavefront per pair of the longest run in these 64 (no step at all where nothing repeats); only a run
  // that reaches past lane 63 goes on from global memory.  (Round 2 took the second pair by a shift and the rest
  // from memory: on a power-law input, whose hot cells repeat dozens of times, nearly every one of the tile's rows of
  // 64 pairs then had a lane waiting for dependent loads -- 0.74 ms of the kernel's 2.65 on C3.)
#pragma unroll
  for (int j = 0; j < kFoldItems; j++) {
    const uint32_t i = base + j * 64;
    bool go = (headmask & (1u << j)) && !(lab & 2) && i + 1 < count && kn[j] == k[j];
    if (__ballot(go) == 0ull) continue;
    T acc = val[j];
    bool beyond = false;  // the run reaches past lane 63
    for (int t = 1; __ballot(go) != 0ull; t++) {
      const int srcl = lane + t;
      const T vv = wshfl(val[j], srcl & 63);
      const K kk = wshfl(k[j], srcl & 63);
      if (go) {
        if (srcl > 63) {
          beyond = (uint64_t)i + (uint64_t)t < (uint64_t)count;  // (the pair in lane 63 was the run's: the next one may be too)
          go = false;
        } else if ((uint64_t)i + (uint64_t)t < (uint64_t)count && kk == k[j]) {
          acc = (T)(acc + vv);
        } else {
          go = false;
        }
      }
    }
    if (beyond) {
      auto src = [&](uint32_t q) -> int64_t { return (INDIRECT && perm) ? (int64_t)perm[q] : (int64_t)q; };
      if ((uint64_t)i + kLongRun < (uint64_t)count && keys[(uint64_t)i + kLongRun] == k[j]) {  // more than kLongRun pairs
        longmask |= 1u << j;
      } else {
        for (uint32_t q = i + (uint32_t)(64 - lane); q < count && keys[q] == k[j]; q++) acc = (T)(acc + vals_in[src(q)]);
      }
    }
    if (headmask & (1u << j)) val[j] = acc;  // (behind the loop: the lanes to the left read val[j] as it was loaded)
  }
  if (tid < 64) {  // wave 0: the tile's first output slot
    bool fail = false;
    const unsigned long long off = (lab & 1) ? (unsigned long long)tile * kFoldTile : rl_finish(status, tile, total, lb_words, &fail);
    if (tid == 0) {
      if (fail) {
        *err = 1;
        s_fail = 1;
      }
      s_off = off;
      if (!fail && tile + 1 == ntiles) *merged_out = off + total;
    }
  }
  __syncthreads();
  if (s_fail) return;
  const unsigned long long off = s_off;
  uint32_t row_ends = 0;
#pragma unroll
  for (int j = 0; j < kFoldItems; j++) {
    const uint32_t i = base + j * 64;
    if (i >= count) continue;
    const unsigned long long s = off + v[j];  // inclusive: outputs up to and including this pair's run
    if (headmask & (1u << j)) {
      col_out[s - 1] = (I)(k[j] & minor_mask);
      if (!(lab & 4)) vals_out[s - 1] = val[j];
      if (longmask & (1u << j)) {  // summed afterwards by a workgroup; the value just stored is overwritten
        const unsigned long long e = atomicAdd(long_list, 1ull);
        long_list[1 + 2 * e] = (unsigned long long)i;
        long_list[2 + 2 * e] = s - 1;
      }
    }
    const K major = shr_safe<K>(k[j], minor_bits);
    if (i + 1 >= count || shr_safe<K>(kn[j], minor_bits) != major) {
      row_ptr[(int64_t)major + 1] = (int64_t)s;
      row_ends++;
    }
  }
  row_ends = wave_reduce(row_ends, OpAdd());
  if (lane == 0 && row_ends) atomicAdd(rows_seen + ((tile & (kSkipCounters - 1)) * kSkipStride), (unsigned long long)row_ends);
}

// The same for keys known to be distinct (a record being transposed): slot = position, nothing to fold, no head
// flags to scan.
template <class K, class T, class I, bool AOS = false>  // AOS: keys points at (key, value) pairs side by side (RsPair), vals_in / perm are not used
__global__ __launch_bounds__(kBlock) void emit_unique_kernel(const K *__restrict__ keys, const uint32_t *__restrict__ perm,
                                                             const T *__restrict__ vals_in, int64_t count,
                                                             int minor_bits, K minor_mask, I *__restrict__ col_out,
                                                             T *__restrict__ vals_out, int64_t *__restrict__ row_ptr) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= count) return;
  typedef typename BitsOf<T>::type VB;
  K k, knext = 0;
  if constexpr (AOS) {
    const RsPair<K, VB> *pairs = reinterpret_cast<const RsPair<K, VB> *>(keys);
    const RsPair<K, VB> pr = pairs[i];
    k = pr.k;
    T v;
    __builtin_memcpy(&v, &pr.p, sizeof(T));
    vals_out[i] = v;
    if (i + 1 < count) knext = pairs[i + 1].k;
  } else {
    k = keys[i];
    vals_out[i] = vals_in[perm ? (int64_t)perm[i] : i];
    if (i + 1 < count) knext = keys[i + 1];
  }
  col_out[i] = (I)(k & minor_mask);
  const K major = shr_safe<K>(k, minor_bits);
  if (i + 1 == count || shr_safe<K>(knext, minor_bits) != major) row_ptr[(int64_t)major + 1] = i + 1;
}

// ---------------------------------------------------------------------- plan
// tile_lb[ntiles + 1] and [ntiles + 2] (zeroed by the caller) receive two statistics of the plan:
// the longest first spill (the largest number of entries by which a row runs past the end of the
// tile it starts in) and the largest number of rows that start in one tile.
__device__ __forceinline__ int64_t plan_lower_bound(const int64_t *__restrict__ row_ptr, int64_t n, int64_t target) {
  int64_t lo = 0, hi = n;  // first r in [0,n) with row_ptr[r] >= target, else n
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (row_ptr[mid] >= target) hi = mid; else lo = mid + 1;
  }
  return lo;
}

__global__ __launch_bounds__(kBlock) void plan_kernel(const int64_t *__restrict__ row_ptr, int64_t n, int64_t tile,
                                                      int64_t ntiles, int64_t *__restrict__ tile_lb) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t > ntiles) return;
  if (t == ntiles) { tile_lb[t] = n; return; }
  const int64_t target = t * tile;
  const int64_t lo = plan_lower_bound(row_ptr, n, target);
  tile_lb[t] = lo;
  if (t > 0 && lo > 0) {
    const int64_t rs = row_ptr[lo - 1], re = row_ptr[lo];
    if (re > target && rs >= target - tile)  // row lo-1 starts in tile t-1 and runs into tile t
      atomicMax(reinterpret_cast<unsigned long long *>(&tile_lb[ntiles + 1]), (unsigned long long)(re - target));
  }
  const int64_t next = t + 1 == ntiles ? n : plan_lower_bound(row_ptr, n, target + tile);
  atomicMax(reinterpret_cast<unsigned long long *>(&tile_lb[ntiles + 2]),
            (unsigned long long)(next - (lo > 0 ? lo - 1 : 0)));
}

constexpr int64_t kPlanTile = 1024;  // entries per tile; matches the SpMV kernels' TILE

template <class T>
int build_plan(futhark_context *ctx, CsrBody<T> *b) {
  b->tile = kPlanTile;
  b->ntiles = cdiv(b->nnz, b->tile);
  SAMD_TRY(dalloc(ctx, &b->tile_lb, b->ntiles + 3));
  SAMD_TRY(dalloc(ctx, &b->head, b->ntiles));
  SAMD_HIP(ctx, hipMemsetAsync(b->tile_lb + b->ntiles + 1, 0, 2 * sizeof(int64_t), ctx->stream));
  b->spill_max = -1;
  b->max_tile_rows = -1;
  hipLaunchKernelGGL(plan_kernel, dim3((unsigned)cdiv(b->ntiles + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                     (const int64_t *)b->row_ptr, b->n, b->tile, b->ntiles, b->tile_lb);
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

inline bool use_col32(futhark_context *ctx, int64_t m) { return !ctx->opt_wide_col && m <= 0x7fffffffLL; }

// ------------------------------------------------------------------- the core
// keys: count packed keys (device, consumed as scratch).  Sort covers key bits
// [begin_bit, end_bit); bits below begin_bit must already be in order within
// equal upper bits (the CSR->CSC conversion uses that).  err_flag (device int,
// may be null) is checked at the one host synchronisation point.
// allow_msd / allow_retry: the caller can pack the keys again; the sort may then try radix_sort_msd.hpp, and if a
// chunk did not fit or a look-back wait timed out this returns kAssembleAgain with nothing built (the keys are
// spent): pack again and call with false, false.
constexpr int kAssembleAgain = -77;
template <class K, class T>
int assemble_core(futhark_context *ctx, Tmp<K> &keys, int64_t count, int begin_bit, int end_bit, int minor_bits,
                  const T *vals_in, int64_t n_major, int64_t n_minor, const int *err_flag, CsrBody<T> **out,
                  bool allow_msd = false, bool allow_retry = false, bool unique_keys = false,
                  int64_t sorted_runs_split = -1) {  // >= 0: keys[0, split) and keys[split, count) are each in order
  CsrBody<T> *b = new CsrBody<T>();
  CsrGuard<T> guard(ctx, b);
  b->n = n_major;
  b->m = n_minor;
  SAMD_TRY(dalloc(ctx, &b->row_ptr, n_major + 1));
  SAMD_HIP(ctx, hipMemsetAsync(b->row_ptr, 0, sizeof(int64_t) * (size_t)(n_major + 1), ctx->stream));

  int64_t merged = 0;
  if (count > 0) {
    // Payload (radix_sort.hpp): 4-byte values travel with their keys.  8-byte values either travel too (16-byte
    // pairs through every pass) or are represented by their 4-byte source index and gathered once at the end.
    // Measured both ways on MI355X: the gather is one random 64-byte line per value when the sorted order has
    // nothing to do with the input order (C2 shuffled: merge_emit 2.5 ms of 7.8; C5's CSR->CSC: 28 of 65 ms) and
    // nearly free when every value ends up near where it started (row-major input, conversions of banded
    // records); carrying costs 4 extra bytes per pair and pass (C2: 0.24 ms per pass).  So: values travel when
    // the input is large and a sample of pairs sits far from its sorted place (mean distance beyond a 1 MB window
    // of values; one small read-back).  Option sort_indirect: 1 always by index, -1 never.
    typedef typename BitsOf<T>::type VB;
    bool indirect = sizeof(T) > 4;
    if (sizeof(T) > 4 && ctx->opt_sort_indirect < 0) indirect = false;
    const bool sample_far = sorted_runs_split < 0 && sizeof(T) > 4 && ctx->opt_sort_indirect == 0 && count >= (1 << 20) && end_bit - begin_bit > 16;
    const bool presampled = ctx->pre_sample && ctx->pre_sample_n == count && ctx->pre_sample_end == end_bit && begin_bit == 0;
    if (sample_far && !presampled) {
      SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small + 8, 0, sizeof(int64_t), ctx->stream));
      const double keyspace = ldexp((double)(n_major > 0 ? n_major : 1), minor_bits);
      const unsigned long long cap = (unsigned long long)count;
      hipLaunchKernelGGL((displacement_sample_kernel<KeyArr<K>>), dim3(16), dim3(kBlock), 0, ctx->stream, KeyArr<K>{keys.p},
                         count, 1.0 / keyspace, cap, reinterpret_cast<unsigned long long *>(ctx->d_small + 8));
      SAMD_LAUNCH_CHECK(ctx);
    }
    // fewer trips through HBM when the key's top bits cut the input into small even buckets (radix_sort_msd.hpp):
    // top bits by global passes, then one kernel that sorts the rest inside LDS and writes the record itself.
    // Its sample and the one above share one read-back.
    int msd_tb = -1;
    uint32_t msd_ch = 0;
    bool msd_small = false;
    int64_t far = -1;
    if constexpr (sizeof(K) == 8) {
      if (allow_msd && begin_bit == 0 && end_bit <= 64 - kRlTagBits)
        SAMD_TRY(msd_choose<K>(ctx, keys.p, count, begin_bit, end_bit, RlEmitCfg<T>::kCap, RlEmitCfg<T>::kCapSmall, &msd_tb,
                               &msd_ch, &msd_small, &far));
      if (msd_tb > 38) msd_tb = -1;  // composite = ordinal (<= 12 bits) | low bits | 13-bit tag, in 64 bits
    }
    if (sample_far) {
      if (far < 0 && presampled) {
        SAMD_HIP(ctx, hipEventSynchronize(ctx->ev_pre));
        far = ctx->h_small[12];
      }
      if (far < 0) SAMD_TRY(read_small(ctx, ctx->d_small + 8, 1, &far));
      const double mean = (double)far / (16.0 * kBlock);
      indirect = mean * sizeof(T) <= (double)(1 << 20);
    }
    Tmp<K> keys_b(ctx);
    Tmp<uint32_t> pay_a(ctx), pay_b(ctx), slot(ctx);
    Tmp<VB> val_a(ctx), val_b(ctx);
    // 8-byte values that travel through the top-bit passes do so side by side with their keys, in buffers
    // msd_sort_emit takes itself: no second key array and no value arrays then
    const bool msd_pairs = msd_tb >= 0 && !indirect && sizeof(T) == 8 && sizeof(K) == 8 && ctx->opt_sort_pairs >= 0 &&
                           ctx->opt_sort_algo == 0 && !ctx->sort_force_scan;
    const bool lsd_pairs = msd_tb < 0 && unique_keys && sorted_runs_split < 0 && !indirect && sizeof(T) == 8 && sizeof(K) == 8 &&
                           ctx->opt_sort_pairs >= 0 && ctx->opt_sort_algo == 0 && !ctx->sort_force_scan &&
                           end_bit > begin_bit && end_bit - begin_bit <= 8 * kRsMaxPasses;
    Tmp<RsPair<K, VB>> aos_a(ctx), aos_b(ctx);
    bool aos_used = false;
    if (!msd_pairs && !lsd_pairs) SAMD_TRY(keys_b.get(count));
    K *ks = nullptr;
    const uint32_t *perm = nullptr;
    const T *vs = nullptr;
    ctx->last_sort_msd = msd_tb >= 0 ? end_bit - msd_tb : 0;
    SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small + 2, 0, 2 * sizeof(int64_t), ctx->stream));
    int *sort_err = (int *)(ctx->d_small + 2), *oversize = (int *)(ctx->d_small + 3);
    if constexpr (sizeof(K) == 8) {
      if (msd_tb >= 0) {
        SAMD_TRY(dalloc(ctx, &b->vals, count));
        SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t) * 2, ctx->stream));
        unsigned long long *merged_dev = reinterpret_cast<unsigned long long *>(ctx->d_small);
        // rows that hold something, counted by the emission: when that is every row, the row ends are final as
        // written and the max-scan below (which fills the empty rows in) leaves at once (C2: 0.09 of 3.9 ms)
        Tmp<unsigned long long> rows_cnt(ctx);  // (kSkipCounters counters, a line each: scan.hpp)
        SAMD_TRY(rows_cnt.get(kSkipCounters * kSkipStride));
        unsigned long long *rows_seen = rows_cnt.p;
        SAMD_HIP(ctx, hipMemsetAsync(rows_seen, 0, sizeof(int64_t) * kSkipCounters * kSkipStride, ctx->stream));
        const bool c32 = use_col32(ctx, n_minor);
        if (c32) SAMD_TRY(dalloc(ctx, &b->col32, count));
        else SAMD_TRY(dalloc(ctx, &b->col64, count));
        if (indirect) {
          SAMD_TRY(pay_a.get(count));
          SAMD_TRY(pay_b.get(count));
          if (c32)
            SAMD_TRY((msd_sort_emit<T, int32_t, true>(ctx, keys.p, keys_b.p, (const uint32_t *)nullptr, pay_a.p, pay_b.p,
                                                       vals_in, count, msd_tb, end_bit, msd_ch, msd_small, minor_bits, b->col32,
                                                       b->vals, b->row_ptr, merged_dev, sort_err, oversize, rows_seen)));
          else
            SAMD_TRY((msd_sort_emit<T, int64_t, true>(ctx, keys.p, keys_b.p, (const uint32_t *)nullptr, pay_a.p, pay_b.p,
                                                       vals_in, count, msd_tb, end_bit, msd_ch, msd_small, minor_bits, b->col64,
                                                       b->vals, b->row_ptr, merged_dev, sort_err, oversize, rows_seen)));
        } else {
          if (!msd_pairs) {
            SAMD_TRY(val_a.get(count));
            SAMD_TRY(val_b.get(count));
          }
          if (c32)
            SAMD_TRY((msd_sort_emit<T, int32_t, false>(ctx, keys.p, keys_b.p, reinterpret_cast<const VB *>(vals_in),
                                                        val_a.p, val_b.p, (const T *)nullptr, count, msd_tb, end_bit,
                                                        msd_ch, msd_small, minor_bits, b->col32, b->vals, b->row_ptr, merged_dev,
                                                        sort_err, oversize, rows_seen)));
          else
            SAMD_TRY((msd_sort_emit<T, int64_t, false>(ctx, keys.p, keys_b.p, reinterpret_cast<const VB *>(vals_in),
                                                        val_a.p, val_b.p, (const T *)nullptr, count, msd_tb, end_bit,
                                                        msd_ch, msd_small, minor_bits, b->col64, b->vals, b->row_ptr, merged_dev,
                                                        sort_err, oversize, rows_seen)));
        }
        if (err_flag)
          SAMD_HIP(ctx, hipMemcpyAsync(ctx->d_small + 1, err_flag, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
        // the row ends' max-scan needs nothing from the host: queued before the read-back so that it runs while the
        // host waits (wasted, harmlessly, if the assembly is repeated)
        if (n_major > 0)
          SAMD_TRY((device_scan<int64_t, OpMax, PtrIn<int64_t>, false>(ctx, PtrIn<int64_t>{b->row_ptr + 1}, n_major,
                                                                        (int64_t)0, OpMax(), b->row_ptr + 1,
                                                                        (int64_t *)nullptr, rows_seen,
                                                                        (unsigned long long)n_major)));
        int64_t host[4];
        SAMD_TRY(read_small(ctx, ctx->d_small, 4, host));
        if (ctx->opt_sort_inject && allow_retry) host[2] = 1;
        if ((int)(host[2] & 0xffffffff) != 0) {  // a look-back wait ran into its limit (GPU shared or preempted)
          ctx->last_sort_msd = -2;
          return kAssembleAgain;
        }
        if ((int)(host[1] & 0xffffffff) != 0)
          return set_error(ctx, "Error: index out of bounds in sparse (assertion 0 <= r < n && 0 <= c < m failed)");
        if ((int)(host[3] & 0xffffffff) != 0) {  // a chunk did not fit its workgroup: nothing of this is final
          ctx->last_sort_msd = -1;
          return kAssembleAgain;
        }
        merged = host[0];
        b->nnz = merged;
        if (merged * 4 < count * 3) {  // many duplicates: do not keep count-sized arrays for the record's lifetime
          T *v2 = nullptr;
          SAMD_TRY(dalloc(ctx, &v2, merged));
          SAMD_HIP(ctx, hipMemcpyAsync(v2, b->vals, sizeof(T) * (size_t)merged, hipMemcpyDeviceToDevice, ctx->stream));
          dfree(ctx, b->vals);
          b->vals = v2;
          if (c32) {
            int32_t *c2 = nullptr;
            SAMD_TRY(dalloc(ctx, &c2, merged));
            SAMD_HIP(ctx, hipMemcpyAsync(c2, b->col32, sizeof(int32_t) * (size_t)merged, hipMemcpyDeviceToDevice, ctx->stream));
            dfree(ctx, b->col32);
            b->col32 = c2;
          } else {
            int64_t *c2 = nullptr;
            SAMD_TRY(dalloc(ctx, &c2, merged));
            SAMD_HIP(ctx, hipMemcpyAsync(c2, b->col64, sizeof(int64_t) * (size_t)merged, hipMemcpyDeviceToDevice, ctx->stream));
            dfree(ctx, b->col64);
            b->col64 = c2;
          }
        }
        SAMD_TRY(build_plan(ctx, b));
        *out = guard.release();
        return 0;
      }
    }
    if (sorted_runs_split >= 0) {  // two runs in order: a stable merge is the stable sort
      SAMD_TRY(pay_a.get(count));
      hipLaunchKernelGGL((merge_runs_kernel<K>), dim3((unsigned)cdiv(count, kMergeTile)), dim3(kBlock), 0, ctx->stream,
                         (const K *)keys.p, sorted_runs_split, count - sorted_runs_split, keys_b.p, pay_a.p);
      SAMD_LAUNCH_CHECK(ctx);
      ks = keys_b.p;
      perm = pay_a.p;
      vs = vals_in;
    } else if (indirect) {
      const uint32_t *sorted_pay = nullptr;
      SAMD_TRY(pay_a.get(count));
      SAMD_TRY(pay_b.get(count));
      SAMD_TRY((radix_sort_pairs<K, uint32_t>(ctx, keys.p, keys_b.p, (const uint32_t *)nullptr, pay_a.p, pay_b.p, count,
                                               begin_bit, end_bit, &ks, &sorted_pay, sort_err)));
      perm = sorted_pay;
      vs = vals_in;
    } else if (lsd_pairs) {
      // a record being transposed whose 8-byte values travel with their keys: side by side, as between the top-bit
      // passes of radix_sort_msd.hpp (a pass 8-13 % faster); emit_unique_kernel reads the pairs
      const VB *sorted_val = nullptr;
      SAMD_TRY(aos_a.get(count));
      if (end_bit - begin_bit > 8) SAMD_TRY(aos_b.get(count));
      SAMD_TRY((radix_sort_pairs<K, VB>(ctx, keys.p, (K *)nullptr, reinterpret_cast<const VB *>(vals_in), (VB *)nullptr,
                                        (VB *)nullptr, count, begin_bit, end_bit, &ks, &sorted_val, sort_err, aos_a.p,
                                        aos_b.p ? aos_b.p : aos_a.p, &aos_used)));
      if (!aos_used) return set_error(ctx, "internal: side-by-side pairs were not taken");
    } else {
      const VB *sorted_val = nullptr;
      SAMD_TRY(val_a.get(count));
      SAMD_TRY(val_b.get(count));
      SAMD_TRY((radix_sort_pairs<K, VB>(ctx, keys.p, keys_b.p, reinterpret_cast<const VB *>(vals_in), val_a.p, val_b.p,
                                        count, begin_bit, end_bit, &ks, &sorted_val, sort_err)));
      vs = reinterpret_cast<const T *>(sorted_val);
    }

    if (unique_keys) {  // no two keys are equal (conversion of a record): positions are slots
      b->nnz = count;
      SAMD_TRY(dalloc(ctx, &b->vals, count));
      const K mmask = minor_bits >= (int)(sizeof(K) * 8) ? (K)~(K)0 : (K)(((K)1 << minor_bits) - 1);
      const unsigned ugrid = (unsigned)cdiv(count, kBlock);
      if (use_col32(ctx, n_minor)) {
        SAMD_TRY(dalloc(ctx, &b->col32, count));
        if (aos_used)
          hipLaunchKernelGGL((emit_unique_kernel<K, T, int32_t, true>), dim3(ugrid), dim3(kBlock), 0, ctx->stream, (const K *)ks,
                             perm, vs, count, minor_bits, mmask, b->col32, b->vals, b->row_ptr);
        else
          hipLaunchKernelGGL((emit_unique_kernel<K, T, int32_t>), dim3(ugrid), dim3(kBlock), 0, ctx->stream, (const K *)ks,
                             perm, vs, count, minor_bits, mmask, b->col32, b->vals, b->row_ptr);
      } else {
        SAMD_TRY(dalloc(ctx, &b->col64, count));
        if (aos_used)
          hipLaunchKernelGGL((emit_unique_kernel<K, T, int64_t, true>), dim3(ugrid), dim3(kBlock), 0, ctx->stream, (const K *)ks,
                             perm, vs, count, minor_bits, mmask, b->col64, b->vals, b->row_ptr);
        else
          hipLaunchKernelGGL((emit_unique_kernel<K, T, int64_t>), dim3(ugrid), dim3(kBlock), 0, ctx->stream, (const K *)ks,
                             perm, vs, count, minor_bits, mmask, b->col64, b->vals, b->row_ptr);
      }
      SAMD_LAUNCH_CHECK(ctx);
      if (n_major > 0)
        SAMD_TRY((device_scan<int64_t, OpMax, PtrIn<int64_t>, false>(ctx, PtrIn<int64_t>{b->row_ptr + 1}, n_major,
                                                                      (int64_t)0, OpMax(), b->row_ptr + 1,
                                                                      (int64_t *)nullptr)));
      int64_t flags[1];
      SAMD_TRY(read_small(ctx, ctx->d_small + 2, 1, flags));  // the sort's look-back flag; also keeps the temporaries alive until the work is done
      if ((int)(flags[0] & 0xffffffff) != 0) return set_error(ctx, "internal: radix sort look-back timed out");
      SAMD_TRY(build_plan(ctx, b));
      *out = guard.release();
      return 0;
    }
    const K minor_mask_f = minor_bits >= (int)(sizeof(K) * 8) ? (K)~(K)0 : (K)(((K)1 << minor_bits) - 1);
    if (!ctx->sort_force_scan && ctx->opt_sort_algo == 0 && ctx->opt_fold_emit >= 0 && count < 0xffffffffLL) {
      // one kernel: head flags, their scan (look-back over tiles), fold, emission (fold_emit_kernel)
      const bool c32 = use_col32(ctx, n_minor);
      SAMD_TRY(dalloc(ctx, &b->vals, count));
      if (c32) SAMD_TRY(dalloc(ctx, &b->col32, count));
      else SAMD_TRY(dalloc(ctx, &b->col64, count));
      const int64_t ftiles = cdiv(count, kFoldTile);
      Tmp<unsigned long long> fstatus(ctx);
      Tmp<uint32_t> fticket(ctx);
      SAMD_TRY(fstatus.get(ftiles));
      SAMD_TRY(fticket.get(1));
      SAMD_HIP(ctx, hipMemsetAsync(fstatus.p, 0, sizeof(unsigned long long) * (size_t)ftiles, ctx->stream));
      SAMD_HIP(ctx, hipMemsetAsync(fticket.p, 0, sizeof(uint32_t), ctx->stream));
      SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t) * 2, ctx->stream));
      unsigned long long *merged_dev = reinterpret_cast<unsigned long long *>(ctx->d_small);
      Tmp<unsigned long long> rows_cnt(ctx);  // as in the in-LDS emission
      SAMD_TRY(rows_cnt.get(kSkipCounters * kSkipStride));
      unsigned long long *rows_seen = rows_cnt.p;
      SAMD_HIP(ctx, hipMemsetAsync(rows_seen, 0, sizeof(int64_t) * kSkipCounters * kSkipStride, ctx->stream));
      Tmp<unsigned long long> longl(ctx);  // [0] = runs listed, then (first pair, output slot) of each
      SAMD_TRY(longl.get(1 + 2 * (count / kLongRun + 1)));
      SAMD_HIP(ctx, hipMemsetAsync(longl.p, 0, sizeof(unsigned long long), ctx->stream));
      if (c32)
        hipLaunchKernelGGL((fold_emit_kernel<K, T, int32_t, true>), dim3((unsigned)ftiles), dim3(kBlock), 0, ctx->stream,
                           (const K *)ks, perm, vs, (uint32_t)count, minor_bits, minor_mask_f, b->col32, b->vals, b->row_ptr,
                           fstatus.p, fticket.p, merged_dev, sort_err, (int)((ctx->opt_ablate >> 20) & 7), longl.p, rows_seen);
      else
        hipLaunchKernelGGL((fold_emit_kernel<K, T, int64_t, true>), dim3((unsigned)ftiles), dim3(kBlock), 0, ctx->stream,
                           (const K *)ks, perm, vs, (uint32_t)count, minor_bits, minor_mask_f, b->col64, b->vals, b->row_ptr,
                           fstatus.p, fticket.p, merged_dev, sort_err, (int)((ctx->opt_ablate >> 20) & 7), longl.p, rows_seen);
      SAMD_LAUNCH_CHECK(ctx);
      if (err_flag)
        SAMD_HIP(ctx, hipMemcpyAsync(ctx->d_small + 1, err_flag, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
      SAMD_HIP(ctx, hipMemcpyAsync(ctx->d_small + 3, longl.p, sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
      int64_t host[4];
      SAMD_TRY(read_small(ctx, ctx->d_small, 4, host));
      if (ctx->opt_sort_inject && allow_retry) host[2] = 1;
      if ((int)(host[2] & 0xffffffff) != 0) {  // a look-back wait ran into its limit (GPU shared or preempted)
        if (!allow_retry) return set_error(ctx, "internal: radix sort look-back timed out");
        ctx->last_sort_msd = -2;
        return kAssembleAgain;
      }
      if ((int)(host[1] & 0xffffffff) != 0)
        return set_error(ctx, "Error: index out of bounds in sparse (assertion 0 <= r < n && 0 <= c < m failed)");
      merged = host[0];
      b->nnz = merged;
      ctx->last_long_runs = host[3];
      SAMD_TRY((fold_long_runs<K, T>(ctx, ks, perm, vs, count, longl.p, host[3], b->vals)));  // before vals may be cut down
      if (merged * 4 < count * 3) {  // many duplicates: do not keep count-sized arrays for the record's lifetime
        T *v2 = nullptr;
        SAMD_TRY(dalloc(ctx, &v2, merged));
        SAMD_HIP(ctx, hipMemcpyAsync(v2, b->vals, sizeof(T) * (size_t)merged, hipMemcpyDeviceToDevice, ctx->stream));
        dfree(ctx, b->vals);
        b->vals = v2;
        if (c32) {
          int32_t *c2 = nullptr;
          SAMD_TRY(dalloc(ctx, &c2, merged));
          SAMD_HIP(ctx, hipMemcpyAsync(c2, b->col32, sizeof(int32_t) * (size_t)merged, hipMemcpyDeviceToDevice, ctx->stream));
          dfree(ctx, b->col32);
          b->col32 = c2;
        } else {
          int64_t *c2 = nullptr;
          SAMD_TRY(dalloc(ctx, &c2, merged));
          SAMD_HIP(ctx, hipMemcpyAsync(c2, b->col64, sizeof(int64_t) * (size_t)merged, hipMemcpyDeviceToDevice, ctx->stream));
          dfree(ctx, b->col64);
          b->col64 = c2;
        }
      }
      if (n_major > 0)
        SAMD_TRY((device_scan<int64_t, OpMax, PtrIn<int64_t>, false>(ctx, PtrIn<int64_t>{b->row_ptr + 1}, n_major,
                                                                      (int64_t)0, OpMax(), b->row_ptr + 1,
                                                                      (int64_t *)nullptr, rows_seen,
                                                                      (unsigned long long)n_major)));
      SAMD_TRY(build_plan(ctx, b));
      *out = guard.release();
      return 0;
    }
    // head flags -> slots; total -> host (the only synchronisation of an assembly)
    SAMD_TRY(slot.get(count));
    uint32_t *slots = slot.p;
    SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small, 0, sizeof(int64_t) * 2, ctx->stream));
    SAMD_TRY((device_scan<uint32_t, OpAdd, HeadFlagIn<K>, false>(ctx, HeadFlagIn<K>{ks}, count, 0u, OpAdd(), slots,
                                                                  (uint32_t *)ctx->d_small)));
    if (err_flag)
      SAMD_HIP(ctx, hipMemcpyAsync(ctx->d_small + 1, err_flag, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    int64_t host[4];
    SAMD_TRY(read_small(ctx, ctx->d_small, 4, host));
    if (ctx->opt_sort_inject && allow_retry) host[2] = 1;
    if ((int)(host[2] & 0xffffffff) != 0) {  // a look-back wait ran into its limit (GPU shared or preempted)
      if (!allow_retry) return set_error(ctx, "internal: radix sort look-back timed out");
      ctx->last_sort_msd = -2;
      return kAssembleAgain;
    }
    if ((int)(host[1] & 0xffffffff) != 0)
      return set_error(ctx, "Error: index out of bounds in sparse (assertion 0 <= r < n && 0 <= c < m failed)");
    merged = (int64_t)(uint32_t)(host[0] & 0xffffffffLL);

    b->nnz = merged;
    SAMD_TRY(dalloc(ctx, &b->vals, merged));
    const K minor_mask = minor_bits >= (int)(sizeof(K) * 8) ? (K)~(K)0 : (K)(((K)1 << minor_bits) - 1);
    const unsigned grid = (unsigned)cdiv(count, kBlock);
    Tmp<unsigned long long> longl(ctx);
    const bool may_be_long = count > kLongRun;
    if (may_be_long) {
      SAMD_TRY(longl.get(1 + 2 * (count / kLongRun + 1)));
      SAMD_HIP(ctx, hipMemsetAsync(longl.p, 0, sizeof(unsigned long long), ctx->stream));
    }
    if (use_col32(ctx, n_minor)) {
      SAMD_TRY(dalloc(ctx, &b->col32, merged));
      hipLaunchKernelGGL((merge_emit_kernel<K, T, int32_t, true>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         (const K *)ks, perm, (const uint32_t *)slots, vs, count, minor_bits, minor_mask, b->col32,
                         b->vals, b->row_ptr, longl.p);
    } else {
      SAMD_TRY(dalloc(ctx, &b->col64, merged));
      hipLaunchKernelGGL((merge_emit_kernel<K, T, int64_t, true>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         (const K *)ks, perm, (const uint32_t *)slots, vs, count, minor_bits, minor_mask, b->col64,
                         b->vals, b->row_ptr, longl.p);
    }
    SAMD_LAUNCH_CHECK(ctx);
    if (may_be_long) SAMD_TRY((fold_long_runs<K, T>(ctx, ks, perm, vs, count, longl.p, -1, b->vals)));
    if (n_major > 0)
      SAMD_TRY((device_scan<int64_t, OpMax, PtrIn<int64_t>, false>(ctx, PtrIn<int64_t>{b->row_ptr + 1}, n_major,
                                                                    (int64_t)0, OpMax(), b->row_ptr + 1,
                                                                    (int64_t *)nullptr)));
  } else {
    if (err_flag) {  // nothing to check, nothing stored
    }
    SAMD_TRY(dalloc(ctx, &b->vals, 0));
    if (use_col32(ctx, n_minor)) SAMD_TRY(dalloc(ctx, &b->col32, 0));
    else SAMD_TRY(dalloc(ctx, &b->col64, 0));
  }
  SAMD_TRY(build_plan(ctx, b));
  *out = guard.release();
  return 0;
}

// What the pack step leaves in the context for the sort that follows (digit histograms, early samples, the
// scan-passes-only switch) is cleared on EVERY way out of the functions that set it, error returns included: a stale
// sample would otherwise steer a later assembly of the same size and key width.
struct PreStateGuard {
  futhark_context *ctx;
  explicit PreStateGuard(futhark_context *c) : ctx(c) {}
  ~PreStateGuard() {
    ctx->pre_hist = nullptr;
    ctx->pre_sample = false;
    ctx->sort_force_scan = false;
  }
};

// sr.sparse n_major n_minor (zip3 major minor vals)
template <class T>
int assemble_coo(futhark_context *ctx, int64_t n_major, int64_t n_minor, const int64_t *major, const int64_t *minor,
                 const T *vals, int64_t count, CsrBody<T> **out) {
  if (n_major < 0 || n_minor < 0) return set_error(ctx, "sparse: negative dimension");
  const int rbits = bits_for(n_major), cbits = bits_for(n_minor);
  if (rbits + cbits > 64) return set_error(ctx, "sparse: n*m exceeds 2^64 key space");
  Tmp<int> err(ctx);
  SAMD_TRY(err.get(1));
  SAMD_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(int), ctx->stream));
  const unsigned grid = (unsigned)cdiv(count > 0 ? count : 1, kBlock);
  if (rbits + cbits <= 32) {
    Tmp<uint32_t> keys(ctx);
    SAMD_TRY(keys.get(count));
    if (count > 0) {
      hipLaunchKernelGGL((pack_coo_kernel<uint32_t>), dim3(grid), dim3(kBlock), 0, ctx->stream, major, minor, count,
                         n_major, n_minor, cbits, keys.p, err.p);
      SAMD_LAUNCH_CHECK(ctx);
    }
    return assemble_core<uint32_t, T>(ctx, keys, count, 0, rbits + cbits, cbits, vals, n_major, n_minor, err.p, out);
  }
  Tmp<uint64_t> keys(ctx);
  Tmp<uint32_t> hist3(ctx);
  PreStateGuard pre_guard(ctx);
  SAMD_TRY(keys.get(count));
  for (int attempt = 0; attempt < 2; attempt++) {
    // large inputs whose sort may go the top-bits-first way (first attempt only): digit counts taken while packing
    const bool with_hist = attempt == 0 && count >= kMsdMinCount && count < 0xffffffffLL && ctx->opt_sort_algo == 0;
    if (count > 0 && with_hist && ctx->opt_sort_msd == 0 && rbits + cbits >= 16 && rbits + cbits <= 64 - kRlTagBits) {
      // The samples the sort's choice rests on (top 16 key bits; distance from the sorted place) are taken from the
      // triples BEFORE the keys are packed and sent to the host behind an event: the host waits for them, decides
      // and queues the sort while the pack kernel runs, instead of leaving the GPU idle while it does (~0.2 ms on C2).
      const KeyCoo kc{major, minor, n_major, n_minor, cbits};
      SAMD_TRY(msd_launch_samples(ctx, kc, count, rbits + cbits));
      SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small + 8, 0, sizeof(int64_t), ctx->stream));
      if (sizeof(T) > 4) {
        const double keyspace = ldexp((double)(n_major > 0 ? n_major : 1), cbits);
        hipLaunchKernelGGL((displacement_sample_kernel<KeyCoo>), dim3(16), dim3(kBlock), 0, ctx->stream, kc, count,
                           1.0 / keyspace, (unsigned long long)count, reinterpret_cast<unsigned long long *>(ctx->d_small + 8));
        SAMD_LAUNCH_CHECK(ctx);
      }
      SAMD_HIP(ctx, hipMemcpyAsync(ctx->h_small + 8, ctx->d_small + 4, sizeof(int64_t) * 5, hipMemcpyDeviceToHost, ctx->stream));
      SAMD_HIP(ctx, hipEventRecord(ctx->ev_pre, ctx->stream));
      ctx->pre_sample = true;
      ctx->pre_sample_n = count;
      ctx->pre_sample_end = rbits + cbits;
    }
    if (count > 0 && with_hist) {
      const int kb = rbits + cbits;
      PackDigits dg;
      dg.n = 0;
      for (int sh = 0; sh < kb && dg.n < kPackDigitsMax; sh += 8) {  // the LSD sort's digits
        dg.shift[dg.n] = sh;
        dg.bits[dg.n++] = kb - sh < 8 ? kb - sh : 8;
      }
      for (int p = 1; p <= 3 && kb % 8 != 0 && kb - 8 * p >= 0 && dg.n < kPackDigitsMax; p++) {  // top-aligned ones
        dg.shift[dg.n] = kb - 8 * p;
        dg.bits[dg.n++] = 8;
      }
      if (!hist3.p) SAMD_TRY(hist3.get(kPackDigitsMax * 256));
      SAMD_HIP(ctx, hipMemsetAsync(hist3.p, 0, sizeof(uint32_t) * kPackDigitsMax * 256, ctx->stream));
      const int64_t cap = (int64_t)ctx->num_cu * 16;
      hipLaunchKernelGGL(pack_coo_hist_kernel, dim3((unsigned)((int64_t)grid < cap ? (int64_t)grid : cap)), dim3(kBlock), 0,
                         ctx->stream, major, minor, count, n_major, n_minor, cbits, keys.p, err.p, dg, hist3.p);
      SAMD_LAUNCH_CHECK(ctx);
      ctx->pre_hist = hist3.p;
      ctx->pre_hist_keys = keys.p;
      ctx->pre_hist_n = count;
      ctx->pre_digits = dg.n;
      for (int d = 0; d < dg.n; d++) {
        ctx->pre_shift[d] = dg.shift[d];
        ctx->pre_bits[d] = dg.bits[d];
      }
    } else if (count > 0) {
      hipLaunchKernelGGL((pack_coo_kernel<uint64_t>), dim3(grid), dim3(kBlock), 0, ctx->stream, major, minor, count,
                         n_major, n_minor, cbits, keys.p, err.p);
      SAMD_LAUNCH_CHECK(ctx);
    }
    // second attempt (a chunk of the in-LDS sort did not fit, or a look-back wait timed out): the plain LSD sort
    // with histogram + scan passes, which wait for nothing
    ctx->sort_force_scan = attempt == 1;
    const int rc = assemble_core<uint64_t, T>(ctx, keys, count, 0, rbits + cbits, cbits, vals, n_major, n_minor, err.p,
                                              out, attempt == 0, attempt == 0);
    ctx->pre_hist = nullptr;
    ctx->pre_sample = false;
    ctx->sort_force_scan = false;
    if (attempt == 1) ctx->last_sort_msd = -1;  // tried, sorted again
    if (rc != kAssembleAgain) return rc;
  }
  return set_error(ctx, "internal: assembly asked to sort again twice");
}

template <class K, class T, bool SWAP>
int pack_csr(futhark_context *ctx, const CsrBody<T> *a, int minor_bits, K *keys) {
  if (a->nnz == 0) return 0;
  if (a->tile == 4 * kBlock && a->n < 0x7fffffffLL) {  // whole tiles per block; row numbers relative to the tile fit 32 bits
    if (a->col32)
      hipLaunchKernelGGL((pack_csr_tile_kernel<K, int32_t, SWAP>), dim3((unsigned)a->ntiles), dim3(kBlock), 0, ctx->stream,
                         (const int64_t *)a->row_ptr, (const int32_t *)a->col32, a->nnz, (const int64_t *)a->tile_lb,
                         minor_bits, keys);
    else
      hipLaunchKernelGGL((pack_csr_tile_kernel<K, int64_t, SWAP>), dim3((unsigned)a->ntiles), dim3(kBlock), 0, ctx->stream,
                         (const int64_t *)a->row_ptr, (const int64_t *)a->col64, a->nnz, (const int64_t *)a->tile_lb,
                         minor_bits, keys);
    SAMD_LAUNCH_CHECK(ctx);
    return 0;
  }
  const unsigned grid = (unsigned)cdiv(a->nnz, kBlock);
  if (a->col32)
    hipLaunchKernelGGL((pack_csr_kernel<K, int32_t, SWAP>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)a->row_ptr, (const int32_t *)a->col32, a->nnz, (const int64_t *)a->tile_lb,
                       a->tile, minor_bits, keys);
  else
    hipLaunchKernelGGL((pack_csr_kernel<K, int64_t, SWAP>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                       (const int64_t *)a->row_ptr, (const int64_t *)a->col64, a->nnz, (const int64_t *)a->tile_lb,
                       a->tile, minor_bits, keys);
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

// Physical conversion: record of A (n x m) -> record of A^T (m x n)
// = sr.sparse m n (map swap (sr.coo A))  (compressed.fut:166-177, :211-212).
// Stored entries are already (row, col)-sorted, so a stable sort on the column
// bits alone reproduces the full (col, row) order; nothing can merge.
template <class T>
int transpose_physical(futhark_context *ctx, const CsrBody<T> *a, CsrBody<T> **out) {
  const int rbits = bits_for(a->n), cbits = bits_for(a->m);
  if (rbits + cbits <= 32) {
    Tmp<uint32_t> keys(ctx);
    SAMD_TRY(keys.get(a->nnz));
    SAMD_TRY((pack_csr<uint32_t, T, true>(ctx, a, rbits, keys.p)));
    return assemble_core<uint32_t, T>(ctx, keys, a->nnz, rbits, rbits + cbits, rbits, a->vals, a->m, a->n, nullptr,
                                      out, false, false, true);
  }
  Tmp<uint64_t> keys(ctx);
  Tmp<uint32_t> hist(ctx);
  PreStateGuard pre_guard(ctx);
  SAMD_TRY(keys.get(a->nnz));
  const int passes = (cbits + 7) / 8;
  if (a->nnz >= kMsdMinCount && a->nnz < 0xffffffffLL && a->tile == 4 * kBlock && a->n < 0x7fffffffLL && ctx->opt_sort_algo == 0 &&
      passes <= kPackDigitsMax) {
    // large record: the column digits are counted while the keys are packed (the sort's own histogram pass drops out)
    PackDigits dg;
    dg.n = 0;
    for (int sh = rbits; sh < rbits + cbits; sh += 8) {
      dg.shift[dg.n] = sh;
      dg.bits[dg.n++] = rbits + cbits - sh < 8 ? rbits + cbits - sh : 8;
    }
    SAMD_TRY(hist.get(kPackDigitsMax * 256));
    SAMD_HIP(ctx, hipMemsetAsync(hist.p, 0, sizeof(uint32_t) * kPackDigitsMax * 256, ctx->stream));
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    const unsigned grid = (unsigned)(a->ntiles < cap ? a->ntiles : cap);
    if (a->col32)
      hipLaunchKernelGGL((pack_csr_tile_hist_kernel<uint64_t, int32_t>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         (const int64_t *)a->row_ptr, (const int32_t *)a->col32, a->nnz, a->ntiles, (const int64_t *)a->tile_lb,
                         rbits, keys.p, dg, hist.p);
    else
      hipLaunchKernelGGL((pack_csr_tile_hist_kernel<uint64_t, int64_t>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         (const int64_t *)a->row_ptr, (const int64_t *)a->col64, a->nnz, a->ntiles, (const int64_t *)a->tile_lb,
                         rbits, keys.p, dg, hist.p);
    SAMD_LAUNCH_CHECK(ctx);
    ctx->pre_hist = hist.p;
    ctx->pre_hist_keys = keys.p;
    ctx->pre_hist_n = a->nnz;
    ctx->pre_digits = dg.n;
    for (int d = 0; d < dg.n; d++) {
      ctx->pre_shift[d] = dg.shift[d];
      ctx->pre_bits[d] = dg.bits[d];
    }
  } else {
    SAMD_TRY((pack_csr<uint64_t, T, true>(ctx, a, rbits, keys.p)));
  }
  const int rc = assemble_core<uint64_t, T>(ctx, keys, a->nnz, rbits, rbits + cbits, rbits, a->vals, a->m, a->n, nullptr, out,
                                            false, false, true);
  ctx->pre_hist = nullptr;
  return rc;
}

// sr.+ / sr.- : (coo a ++ coo (scale -1 b)) |> sparse n m  (compressed.fut:179-183)
template <class T>
int add_sub(futhark_context *ctx, const CsrBody<T> *a, const CsrBody<T> *b, bool sub, CsrBody<T> **out) {
  if (a->n != b->n || a->m != b->m) return set_error(ctx, "add/sub: shape mismatch");
  const int64_t tot = a->nnz + b->nnz;
  const int rbits = bits_for(a->n), cbits = bits_for(a->m);
  // the two halves of the concatenation are in key order: merged, not sorted (option add_sort = 1: the sort, for comparison)
  const int64_t split = (ctx->opt_add_sort || tot >= 0xffffffffLL) ? -1 : a->nnz;
  Tmp<T> vals(ctx);
  SAMD_TRY(vals.get(tot));
  if (tot > 0) {
    hipLaunchKernelGGL((concat_vals_kernel<T>), dim3((unsigned)cdiv(tot, kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const T *)a->vals, a->nnz, (const T *)b->vals, b->nnz, sub, vals.p);
    SAMD_LAUNCH_CHECK(ctx);
  }
  if (rbits + cbits <= 32) {
    Tmp<uint32_t> keys(ctx);
    SAMD_TRY(keys.get(tot));
    SAMD_TRY((pack_csr<uint32_t, T, false>(ctx, a, cbits, keys.p)));
    SAMD_TRY((pack_csr<uint32_t, T, false>(ctx, b, cbits, keys.p + a->nnz)));
    return assemble_core<uint32_t, T>(ctx, keys, tot, 0, rbits + cbits, cbits, vals.p, a->n, a->m, nullptr, out, false,
                                      false, false, split);
  }
  Tmp<uint64_t> keys(ctx);
  SAMD_TRY(keys.get(tot));
  SAMD_TRY((pack_csr<uint64_t, T, false>(ctx, a, cbits, keys.p)));
  SAMD_TRY((pack_csr<uint64_t, T, false>(ctx, b, cbits, keys.p + a->nnz)));
  return assemble_core<uint64_t, T>(ctx, keys, tot, 0, rbits + cbits, cbits, vals.p, a->n, a->m, nullptr, out, false, false,
                                    false, split);
}

}  // namespace samd
