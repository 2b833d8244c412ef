// scan.hpp -- wavefront/block/device-wide scans (add, max) for 32/64-bit values.
//
// Layout rule shared by every kernel that scans or ranks a tile: a block of 256
// threads (4 wavefronts of 64) owns ITEMS*256 consecutive elements; wavefront w
// owns the contiguous span [w*ITEMS*64, (w+1)*ITEMS*64) and reads it in ITEMS
// rounds of 64 consecutive elements, so element order == (wave, round, lane)
// order and every global access is a fully coalesced 64-lane row.
#pragma once

#include <type_traits>

#include "common.hpp"

namespace samd {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kScanItems = 16;
constexpr int kScanTile = kBlock * kScanItems;

struct OpAdd {
  template <class V>
  __device__ __forceinline__ V operator()(V a, V b) const { return a + b; }
};
struct OpMin {
  template <class V>
  __device__ __forceinline__ V operator()(V a, V b) const { return a < b ? a : b; }
};
struct OpMax {
  template <class V>
  __device__ __forceinline__ V operator()(V a, V b) const { return a > b ? a : b; }
};

template <class V>
struct PtrIn {
  const V *p;
  __device__ __forceinline__ V operator()(int64_t i) const { return p[i]; }
};

template <class V>
__device__ __forceinline__ V wshfl_up(V v, int d) { return __shfl_up(v, d, 64); }
template <class V>
__device__ __forceinline__ V wshfl(V v, int src) { return __shfl(v, src, 64); }
template <class V>
__device__ __forceinline__ V wshfl_xor(V v, int m) { return __shfl_xor(v, m, 64); }
template <class V>
__device__ __forceinline__ V wshfl_down(V v, int d) { return __shfl_down(v, d, 64); }

// Inclusive add-scan of one 32-bit value per lane with DPP moves (row shifts inside the rows of 16 lanes, then the
// two row broadcasts): six VALU-speed steps.  __shfl_up goes through the LDS crossbar (ds_bpermute, ~100 cycles a
// step), which is what the generic scan above pays; the in-LDS sort calls this once per pass and chunk.
__device__ __forceinline__ uint32_t wave_inclusive_add_dpp(uint32_t x) {
  // update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lanes without a source keep `old` (0: adds nothing)
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return x;
}

// the 32-bit value of the lane below (lane 0 keeps its own): one DPP move (wave_shr:1) instead of a trip through LDS
__device__ __forceinline__ uint32_t wave_prev_lane_dpp(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x138, 0xf, 0xf, false);
}

template <class V, class Op>
__device__ __forceinline__ V wave_inclusive_scan(V x, Op op, int lane) {
  if constexpr (std::is_same<V, uint32_t>::value && std::is_same<Op, OpAdd>::value) return wave_inclusive_add_dpp(x);
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    V y = wshfl_up(x, d);
    if (lane >= d) x = op(y, x);
  }
  return x;
}

template <class V, class Op>
__device__ __forceinline__ V wave_reduce(V x, Op op) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x = op(x, wshfl_xor(x, d));
  return x;
}

// Inclusive scan of the block's ITEMS*256 striped items (layout rule above).
// sh: at least kWaves elements of shared memory.  Ends with a barrier so sh can
// be reused by the caller.
template <class V, class Op, int ITEMS>
__device__ __forceinline__ void block_scan_striped(V (&v)[ITEMS], V ident, Op op, V *sh, V &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  V carry = ident;
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    V x = wave_inclusive_scan(v[k], op, lane);
    x = op(carry, x);
    v[k] = x;
    carry = wshfl(x, 63);
  }
  if (lane == 63) sh[wave] = carry;
  __syncthreads();
  V off = ident, tot = ident;
#pragma unroll
  for (int w = 0; w < kWaves; w++) {
    V s = sh[w];
    if (w < wave) off = op(off, s);
    tot = op(tot, s);
  }
#pragma unroll
  for (int k = 0; k < ITEMS; k++) v[k] = op(off, v[k]);
  total = tot;
  __syncthreads();
}

// One value per thread: exclusive add-scan over the block (+ total).
template <class V>
__device__ __forceinline__ V block_exclusive_add(V x, V *sh, V &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  V inc = wave_inclusive_scan(x, OpAdd(), lane);
  if (lane == 63) sh[wave] = inc;
  __syncthreads();
  V off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kWaves; w++) {
    V s = sh[w];
    if (w < wave) off += s;
    tot += s;
  }
  total = tot;
  __syncthreads();
  return off + inc - x;
}

// skip (optional, all three kernels of a scan alike): kSkipCounters counters, one per 128-byte line (kernels that
// count into one address from every workgroup queue up at that address: 262 K adds cost the in-LDS sort 2 ms on C2).
// Their sum == skip_when means the caller's array needs no scan -- the kernels leave at once (the row ends of an
// assembly in which every row holds something are final as written).
constexpr int kSkipCounters = 64, kSkipStride = 16;  // in 8-byte words
__device__ __forceinline__ bool scan_skips(const unsigned long long *skip, unsigned long long skip_when) {
  if (skip == nullptr) return false;  // uniform
  __shared__ int s_skip;
  if (threadIdx.x < 64) {
    unsigned long long v = threadIdx.x < kSkipCounters ? skip[threadIdx.x * kSkipStride] : 0ull;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (threadIdx.x == 0) s_skip = v == skip_when ? 1 : 0;
  }
  __syncthreads();
  return s_skip != 0;
}
template <class V, class Op, class In>
__global__ __launch_bounds__(kBlock) void scan_reduce_kernel(In in, int64_t n, V ident, Op op, V *sums,
                                                            const unsigned long long *skip = nullptr,
                                                            unsigned long long skip_when = 0) {
  __shared__ V sh[kWaves];
  if (scan_skips(skip, skip_when)) return;
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  V acc = ident;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    int64_t i = base + k * kBlock + threadIdx.x;
    if (i < n) acc = op(acc, in(i));
  }
  acc = wave_reduce(acc, op);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    V t = ident;
    for (int w = 0; w < kWaves; w++) t = op(t, sh[w]);
    sums[blockIdx.x] = t;
  }
}

// out may alias the input array: every element is read into registers before
// the block's first barrier and written only after it.
template <class V, class Op, class In, bool EXCLUSIVE_ADD>
__global__ __launch_bounds__(kBlock) void scan_apply_kernel(In in, int64_t n, V ident, Op op,
                                                           const V *sums_incl, V *out, V *d_total,
                                                           const unsigned long long *skip = nullptr,
                                                           unsigned long long skip_when = 0) {
  __shared__ V sh[kWaves];
  if (scan_skips(skip, skip_when)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)wave * (kScanItems * 64) + lane;
  V x[kScanItems], v[kScanItems];
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    int64_t i = base + k * 64;
    x[k] = i < n ? in(i) : ident;
    v[k] = x[k];
  }
  V total;
  block_scan_striped<V, Op, kScanItems>(v, ident, op, sh, total);
  const V off = (sums_incl != nullptr && blockIdx.x > 0) ? sums_incl[blockIdx.x - 1] : ident;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    int64_t i = base + k * 64;
    if (i < n) {
      V r = op(off, v[k]);
      out[i] = EXCLUSIVE_ADD ? (V)(r - x[k]) : r;
    }
  }
  if (d_total != nullptr && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *d_total = op(off, total);
}

// Device-wide scan (reduce-then-scan, three launches per level).
//   EXCLUSIVE_ADD=false: inclusive scan under op.
//   EXCLUSIVE_ADD=true : exclusive prefix sum (op must be OpAdd).
// d_total (optional, device) receives the reduction of all n inputs.
template <class V, class Op, class In, bool EXCLUSIVE_ADD>
int device_scan(futhark_context *ctx, In in, int64_t n, V ident, Op op, V *out, V *d_total,
                const unsigned long long *skip = nullptr, unsigned long long skip_when = 0) {
  if (n <= 0) {
    if (d_total) SAMD_HIP(ctx, hipMemsetAsync(d_total, 0, sizeof(V), ctx->stream));  // ident is 0 for every use here
    return 0;
  }
  const int64_t nblocks = cdiv(n, kScanTile);
  if (nblocks > 0x7fffffffLL) return set_error(ctx, "scan: input too large");
  if (nblocks == 1) {
    hipLaunchKernelGGL((scan_apply_kernel<V, Op, In, EXCLUSIVE_ADD>), dim3(1), dim3(kBlock), 0, ctx->stream, in, n,
                       ident, op, (const V *)nullptr, out, d_total, skip, skip_when);
    SAMD_LAUNCH_CHECK(ctx);
    return 0;
  }
  Tmp<V> sums(ctx);
  SAMD_TRY(sums.get(nblocks));
  hipLaunchKernelGGL((scan_reduce_kernel<V, Op, In>), dim3((unsigned)nblocks), dim3(kBlock), 0, ctx->stream, in, n,
                     ident, op, sums.p, skip, skip_when);
  SAMD_LAUNCH_CHECK(ctx);
  SAMD_TRY((device_scan<V, Op, PtrIn<V>, false>(ctx, PtrIn<V>{sums.p}, nblocks, ident, op, sums.p, (V *)nullptr, skip,
                                                skip_when)));
  hipLaunchKernelGGL((scan_apply_kernel<V, Op, In, EXCLUSIVE_ADD>), dim3((unsigned)nblocks), dim3(kBlock), 0,
                     ctx->stream, in, n, ident, op, (const V *)sums.p, out, d_total, skip, skip_when);
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

}  // namespace samd
