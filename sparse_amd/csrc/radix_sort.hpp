// radix_sort.hpp -- stable LSD radix sort of (key, payload) pairs on a bit range.
//
// Replaces merge_sort at compressed.fut:73 (sorts 0.6.0, un-vendored).  Keys are
// (row << cbits) | col packed by the caller, so the reference comparator
// r1 < r2 || (r1 == r2 && c1 <= c2) is plain unsigned order; LSD passes are
// stable, so equal (row, col) keep input order and duplicates are later summed
// in input order (deterministic, and equal to the oracle's stable sort).
//
// Payload: 4-byte values travel with their key (a final gather vals[perm[i]] is one random
// line fetch per entry); 8-byte values are sorted by 4-byte source index and gathered once
// at the end -- measured on MI355X, 16-byte pairs through every pass cost more than the
// gather (C2 f64: 11.0 ms vs 9.6 ms; C3 f32 with value payload: 23.8 ms vs 29.5 ms).
//
// One pass = 8-bit digit:
//   rs_hist    per-tile digit histogram            -> table[digit][tile]
//   device_scan exclusive prefix sum of the table   (digit-major = global order)
//   rs_scatter re-read the tile, rank every element inside its wavefront with
//              ballot-based match-any, combine wavefront counters through LDS,
//              reorder the tile in LDS so each digit's run leaves as one
//              contiguous burst, write to table offsets.
// Tile = 256 threads x 16 items (8 when key + payload exceed 12 bytes), read in the (wave, round, lane) order of
// scan.hpp so every load is a coalesced 64-lane row.
#pragma once

#include "scan.hpp"

namespace samd {

constexpr int kRsRadix = 256;

// items per thread: the tile's keys + payloads are re-ordered in LDS (<= 48 KiB of it)
template <class K, class P>
struct RsCfg {
  static constexpr int kItems = (sizeof(K) + sizeof(P)) <= 12 ? 16 : 8;
  static constexpr int kTile = kBlock * kItems;
};

template <class K, int ITEMS>
__global__ __launch_bounds__(kBlock) void rs_hist(const K *__restrict__ keys, uint32_t n, int shift, uint32_t mask,
                                                  uint32_t *__restrict__ table, uint32_t ntiles) {
  __shared__ uint32_t h[kRsRadix];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * (uint32_t)(kBlock * ITEMS);
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    uint32_t i = base + k * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  table[threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

template <class K, class P, int ITEMS, bool IOTA>
__global__ __launch_bounds__(kBlock) void rs_scatter(const K *__restrict__ kin, const P *__restrict__ pin,
                                                     K *__restrict__ kout, P *__restrict__ pout, uint32_t n,
                                                     int shift, uint32_t mask, const uint32_t *__restrict__ table,
                                                     uint32_t ntiles) {
  constexpr int TILE = kBlock * ITEMS;
  __shared__ uint32_t cnt[kWaves][kRsRadix];  // per-wave digit counters, then per-wave bases
  __shared__ uint32_t lstart[kRsRadix];       // start of each digit's run inside the sorted tile
  __shared__ uint32_t goff[kRsRadix];         // global offset of the run minus lstart
  __shared__ uint32_t shs[kWaves];
  __shared__ K skey[TILE];
  __shared__ P spay[TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int d = tid; d < kWaves * kRsRadix; d += kBlock) (&cnt[0][0])[d] = 0;
  __syncthreads();

  const uint32_t tile_base = blockIdx.x * (uint32_t)TILE;
  const uint32_t base = tile_base + wave * (ITEMS * 64) + lane;
  K key[ITEMS];
  P pay[ITEMS];
  uint32_t rank[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    uint32_t i = base + k * 64;
    bool ok = i < n;
    key[k] = ok ? kin[i] : (K)~(K)0;  // padding sorts to the very end of the tile
    pay[k] = ok ? (IOTA ? (P)i : pin[i]) : (P)0;
  }
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
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
  for (int k = 0; k < ITEMS; k++) {
    const uint32_t d = (uint32_t)(key[k] >> shift) & mask;
    const uint32_t p = lstart[d] + cnt[wave][d] + rank[k];
    skey[p] = key[k];
    spay[p] = pay[k];
  }
  __syncthreads();
  const uint32_t valid = (n - tile_base) < (uint32_t)TILE ? (n - tile_base) : (uint32_t)TILE;
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
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

// Sorts n (< 2^32) (key, payload) pairs by key bits [begin_bit, end_bit), stable.
// keys_a / pay_in are the inputs (pay_in is only read; nullptr = the payload is the
// element's position, P = uint32_t); keys_b, pay_a, pay_b are scratch of the same length.
// On return *keys_out / *pay_out point at whichever buffers hold the result (pay_in itself
// when no pass ran: nullptr then still means "identity").
template <class K, class P>
int radix_sort_pairs(futhark_context *ctx, K *keys_a, K *keys_b, const P *pay_in, P *pay_a, P *pay_b, int64_t n,
                     int begin_bit, int end_bit, K **keys_out, const P **pay_out) {
  constexpr int ITEMS = RsCfg<K, P>::kItems;
  constexpr int TILE = RsCfg<K, P>::kTile;
  if (n >= (int64_t)0xffff0000LL) return set_error(ctx, "sort: more than 2^32 entries on one device");
  const uint32_t ntiles = (uint32_t)cdiv(n, TILE);
  K *kin = keys_a, *kout = keys_b;
  const P *pin = pay_in;
  P *pout = pay_a, *pspare = pay_b;
  if (n > 0 && end_bit > begin_bit) {
    Tmp<uint32_t> table(ctx);
    const int64_t tlen = (int64_t)kRsRadix * ntiles;
    SAMD_TRY(table.get(tlen));
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
      const int nb = end_bit - shift < 8 ? end_bit - shift : 8;
      const uint32_t mask = (1u << nb) - 1u;
      hipLaunchKernelGGL((rs_hist<K, ITEMS>), dim3(ntiles), dim3(kBlock), 0, ctx->stream, (const K *)kin,
                         (uint32_t)n, shift, mask, table.p, ntiles);
      SAMD_LAUNCH_CHECK(ctx);
      SAMD_TRY((device_scan<uint32_t, OpAdd, PtrIn<uint32_t>, true>(ctx, PtrIn<uint32_t>{table.p}, tlen, 0u, OpAdd(),
                                                                     table.p, (uint32_t *)nullptr)));
      if (pin == nullptr)  // first pass of an index sort: the payload is the element's position
        hipLaunchKernelGGL((rs_scatter<K, P, ITEMS, true>), dim3(ntiles), dim3(kBlock), 0, ctx->stream,
                           (const K *)kin, pin, kout, pout, (uint32_t)n, shift, mask, (const uint32_t *)table.p,
                           ntiles);
      else
        hipLaunchKernelGGL((rs_scatter<K, P, ITEMS, false>), dim3(ntiles), dim3(kBlock), 0, ctx->stream,
                           (const K *)kin, pin, kout, pout, (uint32_t)n, shift, mask, (const uint32_t *)table.p,
                           ntiles);
      SAMD_LAUNCH_CHECK(ctx);
      K *tk = kin; kin = kout; kout = tk;
      P *written = pout;
      pout = (written == pay_a) ? pspare : pay_a;
      pin = written;
    }
  }
  *keys_out = kin;
  *pay_out = pin;
  return 0;
}

}  // namespace samd
