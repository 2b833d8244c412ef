// radix_sort.hpp -- stable LSD radix sort of (key, u32 payload) pairs on a bit range.
//
// Replaces merge_sort at compressed.fut:73 (sorts 0.6.0, un-vendored).  Keys are
// (row << cbits) | col packed by the caller, so the reference comparator
// r1 < r2 || (r1 == r2 && c1 <= c2) is plain unsigned order; LSD passes are
// stable, so equal (row, col) keep input order and duplicates are later summed
// in input order (deterministic, and equal to the oracle's stable sort).
//
// One pass = 8-bit digit:
//   rs_hist    per-tile digit histogram            -> table[digit][tile]
//   device_scan exclusive prefix sum of the table   (digit-major = global order)
//   rs_scatter re-read the tile, rank every element inside its wavefront with
//              ballot-based match-any, combine wavefront counters through LDS,
//              reorder the tile in LDS so each digit's run leaves as one
//              contiguous burst, write to table offsets.
// Tile = 256 threads x 16 items, read in the (wave, round, lane) order of
// scan.hpp so every load is a coalesced 64-lane row.
#pragma once

#include "scan.hpp"

namespace samd {

constexpr int kRsItems = 16;
constexpr int kRsTile = kBlock * kRsItems;  // 4096
constexpr int kRsRadix = 256;

static __global__ void fill_iota_u32(uint32_t *p, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

template <class K>
__global__ __launch_bounds__(kBlock) void rs_hist(const K *__restrict__ keys, uint32_t n, int shift, uint32_t mask,
                                                  uint32_t *__restrict__ table, uint32_t ntiles) {
  __shared__ uint32_t h[kRsRadix];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * (uint32_t)kRsTile;
#pragma unroll
  for (int k = 0; k < kRsItems; k++) {
    uint32_t i = base + k * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  table[threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

template <class K>
__global__ __launch_bounds__(kBlock) void rs_scatter(const K *__restrict__ kin, const uint32_t *__restrict__ pin,
                                                     K *__restrict__ kout, uint32_t *__restrict__ pout, uint32_t n,
                                                     int shift, uint32_t mask, const uint32_t *__restrict__ table,
                                                     uint32_t ntiles) {
  __shared__ uint32_t cnt[kWaves][kRsRadix];  // per-wave digit counters, then per-wave bases
  __shared__ uint32_t lstart[kRsRadix];       // start of each digit's run inside the sorted tile
  __shared__ uint32_t goff[kRsRadix];         // global offset of the run minus lstart
  __shared__ uint32_t shs[kWaves];
  __shared__ K skey[kRsTile];
  __shared__ uint32_t spay[kRsTile];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int d = tid; d < kWaves * kRsRadix; d += kBlock) (&cnt[0][0])[d] = 0;
  __syncthreads();

  const uint32_t tile_base = blockIdx.x * (uint32_t)kRsTile;
  const uint32_t base = tile_base + wave * (kRsItems * 64) + lane;
  K key[kRsItems];
  uint32_t pay[kRsItems];
  uint32_t rank[kRsItems];
#pragma unroll
  for (int k = 0; k < kRsItems; k++) {
    uint32_t i = base + k * 64;
    bool ok = i < n;
    key[k] = ok ? kin[i] : (K)~(K)0;  // padding sorts to the very end of the tile
    pay[k] = ok ? (pin ? pin[i] : i) : 0u;
  }
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int k = 0; k < kRsItems; k++) {
    const uint32_t d = (uint32_t)(key[k] >> shift) & mask;
    uint64_t peers = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const uint32_t prev = cnt[wave][d];
    const uint32_t r = (uint32_t)__popcll(peers & lt_mask);
    __builtin_amdgcn_wave_barrier();
    if (r == 0) cnt[wave][d] = prev + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    rank[k] = prev + r;
  }
  __syncthreads();
  // thread d: exclusive scan of digit d over the wavefronts, then over digits
  uint32_t tot = 0;
#pragma unroll
  for (int w = 0; w < kWaves; w++) {
    uint32_t c = cnt[w][tid];
    cnt[w][tid] = tot;
    tot += c;
  }
  uint32_t block_total;
  const uint32_t ex = block_exclusive_add<uint32_t>(tot, shs, block_total);
  lstart[tid] = ex;
  goff[tid] = table[(uint32_t)tid * ntiles + blockIdx.x] - ex;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kRsItems; k++) {
    const uint32_t d = (uint32_t)(key[k] >> shift) & mask;
    const uint32_t p = lstart[d] + cnt[wave][d] + rank[k];
    skey[p] = key[k];
    spay[p] = pay[k];
  }
  __syncthreads();
  const uint32_t valid = (n - tile_base) < (uint32_t)kRsTile ? (n - tile_base) : (uint32_t)kRsTile;
#pragma unroll
  for (int k = 0; k < kRsItems; k++) {
    const uint32_t i = k * kBlock + tid;
    if (i < valid) {
      const K kk = skey[i];
      const uint32_t d = (uint32_t)(kk >> shift) & mask;
      const uint32_t g = goff[d] + i;
      kout[g] = kk;
      pout[g] = spay[i];
    }
  }
}

// Sorts n (< 2^32) pairs by key bits [begin_bit, end_bit).  keys_a is the input;
// keys_b/pay_a/pay_b are scratch of the same length.  The initial payload is the
// element index.  On return *keys_out / *pay_out point at whichever buffers hold
// the result.
template <class K>
int radix_sort_pairs(futhark_context *ctx, K *keys_a, K *keys_b, uint32_t *pay_a, uint32_t *pay_b, int64_t n,
                     int begin_bit, int end_bit, K **keys_out, uint32_t **pay_out) {
  if (n >= (int64_t)0xffff0000LL) return set_error(ctx, "sort: more than 2^32 entries on one device");
  const uint32_t ntiles = (uint32_t)cdiv(n, kRsTile);
  K *kin = keys_a, *kout = keys_b;
  uint32_t *pin = nullptr, *pout = pay_a, *pspare = pay_b;
  if (n > 0 && end_bit > begin_bit) {
    Tmp<uint32_t> table(ctx);
    const int64_t tlen = (int64_t)kRsRadix * ntiles;
    SAMD_TRY(table.get(tlen));
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
      const int nb = end_bit - shift < 8 ? end_bit - shift : 8;
      const uint32_t mask = (1u << nb) - 1u;
      hipLaunchKernelGGL((rs_hist<K>), dim3(ntiles), dim3(kBlock), 0, ctx->stream, (const K *)kin, (uint32_t)n,
                         shift, mask, table.p, ntiles);
      SAMD_LAUNCH_CHECK(ctx);
      SAMD_TRY((device_scan<uint32_t, OpAdd, PtrIn<uint32_t>, true>(ctx, PtrIn<uint32_t>{table.p}, tlen, 0u, OpAdd(),
                                                                     table.p, (uint32_t *)nullptr)));
      hipLaunchKernelGGL((rs_scatter<K>), dim3(ntiles), dim3(kBlock), 0, ctx->stream, (const K *)kin,
                         (const uint32_t *)pin, kout, pout, (uint32_t)n, shift, mask, (const uint32_t *)table.p,
                         ntiles);
      SAMD_LAUNCH_CHECK(ctx);
      K *tk = kin; kin = kout; kout = tk;
      uint32_t *np = pout;
      pout = (pin == nullptr) ? pspare : pin;
      pin = np;
    }
  }
  if (pin == nullptr) {  // no pass ran: payload is the identity
    pin = pay_a;
    if (n > 0) {
      hipLaunchKernelGGL(fill_iota_u32, dim3((unsigned)cdiv(n, kBlock)), dim3(kBlock), 0, ctx->stream, pin,
                         (uint32_t)n);
      SAMD_LAUNCH_CHECK(ctx);
    }
  }
  *keys_out = kin;
  *pay_out = pin;
  return 0;
}

}  // namespace samd
