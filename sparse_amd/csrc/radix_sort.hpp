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
// Tile = 256 threads x 32 / 24 / 16 items for 8- / 12- / 16-byte pairs, read in the (wave, round, lane) order of
// scan.hpp so every load is a coalesced 64-lane row.
#pragma once

#include <vector>

#include "scan.hpp"

namespace samd {

constexpr int kRsRadix = 256;

// items per thread: the tile's keys + payloads are re-ordered in LDS
template <class K, class P>
struct RsCfg {
#ifdef SAMD_RS_ITEMS_LAB
  static constexpr int kItems = SAMD_RS_ITEMS_LAB;
#else
  // measured on MI355X (bench/sort_lab.hip, 8.4e7 pairs): 8-byte pairs 16 -> 32 items 2.80 -> 2.20 ms,
  // 12-byte pairs 16 -> 24 items 5.11 -> 4.20 ms (28 and 32 are slower again): 64-72 KB of LDS per block
  static constexpr int kItems = (sizeof(K) + sizeof(P)) <= 8 ? 32 : ((sizeof(K) + sizeof(P)) <= 12 ? 24 : 16);
#endif
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
  // (tried: one buffer for the re-ordered keys and then the payloads -- 39 instead of 71 KB for 16-byte pairs, three
  // blocks per CU by registers instead of two by LDS, two more barriers: the pass 0.81 -> 0.765 ms on C2, the
  // assembly as a whole unchanged within run-to-run noise and C4's slower; not kept)
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

// ---------------------------------------------------------------- one-sweep variant
// One read of the keys builds the digit histograms of ALL passes (rs_hist_all); each pass is then a
// single kernel: a tile learns where its digit runs go by looking back over the tiles before it
// (decoupled look-back on status[tile][digit] words = 2 flag bits + 62-bit count) instead of a
// per-tile histogram pass and a device-wide scan.  Tiles are handed out through a ticket counter so
// that every tile a block waits for has already started; a wait that lasts longer than
// kRsSpinLimit (2 s of the 100 MHz wall clock) poisons the chain and raises *err, so every wave
// reaches its exit whatever happens.
constexpr int kRsMaxPasses = 8;
constexpr unsigned long long kRsFlagAgg = 1ull << 62, kRsFlagIncl = 2ull << 62, kRsFlagPoison = 3ull << 62;
constexpr unsigned long long kRsValMask = (1ull << 62) - 1;
constexpr unsigned long long kRsSpinLimit = 200000000ull;

template <class K, int ITEMS>
__global__ __launch_bounds__(kBlock) void rs_hist_all(const K *__restrict__ keys, uint32_t n, int begin_bit,
                                                      int end_bit, uint32_t *__restrict__ ghist) {
  __shared__ uint32_t h[kRsMaxPasses][kRsRadix];
  for (int d = threadIdx.x; d < kRsMaxPasses * kRsRadix; d += kBlock) (&h[0][0])[d] = 0;
  __syncthreads();
  const int passes = (end_bit - begin_bit + 7) / 8;
  constexpr uint32_t TILE = kBlock * ITEMS;
  const uint32_t ntiles = (n + TILE - 1) / TILE;
  for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint32_t base = t * TILE;
    // One add per run of equal digits among neighbouring lanes: keys that are already in some order (a row-sorted
    // record being transposed, sorted input) put whole wavefronts on one counter, and 64 adds to one LDS address
    // are done one after the other (rs_hist_all on C2's CSR->CSC: 0.50 -> 0.2 ms).
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const uint32_t i = base + k * kBlock + threadIdx.x;
      const bool ok = i < n;
      const K key = ok ? keys[i] : (K)0;
      for (int p = 0; p < passes; p++) {
        const int shift = begin_bit + 8 * p;
        const int nb = end_bit - shift < 8 ? end_bit - shift : 8;
        const uint32_t d = ok ? ((uint32_t)(key >> shift) & ((1u << nb) - 1u)) : 0xffffffffu;
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
  }
  __syncthreads();
  for (int p = 0; p < passes; p++) {
    const uint32_t v = h[p][threadIdx.x];
    if (v) atomicAdd(&ghist[p * kRsRadix + threadIdx.x], v);
  }
}

// exclusive scan of each pass's 256 counters, in place (grid = passes)
__global__ __launch_bounds__(kBlock) void rs_scan_hist(uint32_t *__restrict__ ghist) {
  __shared__ uint32_t shs[kWaves];
  uint32_t total;
  const uint32_t v = ghist[blockIdx.x * kRsRadix + threadIdx.x];
  const uint32_t ex = block_exclusive_add<uint32_t>(v, shs, total);
  ghist[blockIdx.x * kRsRadix + threadIdx.x] = ex;
}

// Where a tile's digit runs go = the digit's global base + the counts of that digit in all earlier tiles.  The
// earlier tiles' counts come from a decoupled look-back, laid out for what it costs on this machine: device-scope
// loads go past the per-XCD L2, and with one 8-byte status word per (tile, digit) the walk (about 50 tiles back
// before an inclusive prefix is out, never an unpublished word: bench/sweep_lab.sh) read as many bytes as the tile
// itself -- 0.92 ms per pass on C2's 16-byte pairs with it, 0.50 without.  So:
//   agg[tile][digit]   u16  the tile's own count | 0x8000, published right after a plain LDS histogram of the keys
//                           (before the expensive ranking); 0 = not there yet, 0xffff = poisoned
//   incl[group][digit] u32  1 + count of the digit in all tiles up to the end of group `group` (16 tiles), published
//                           by the group's last tile only; 0 = not there yet, 0xffffffff = poisoned
// and a tile adds up, group by group going back, the 2-byte counts of the tiles before it until it meets a group
// whose inclusive prefix is out: one round trip per group, ~70 bytes per digit instead of ~460.  Collected as late
// as possible (the tile is already re-ordered in LDS), so slower predecessors have had time.
// Tiles are handed out through a ticket counter so that every tile a block waits for has already started; a wait
// that lasts longer than kRsSpinLimit poisons the chain and raises *err, so every wave reaches its exit.
constexpr int kRsGroup = 16;
constexpr uint16_t kRsAggReady = 0x8000, kRsAggPoison = 0xffff;
constexpr uint32_t kRsInclPoison = 0xffffffffu;

// IN_AOS / OUT_AOS: key and payload side by side in ONE array of pairs (kin / kout point at it, pin / pout are not
// used).  A digit's run then leaves the tile as one burst (16 pairs of 16 bytes: 256 B) instead of one of 128 B into
// each of two arrays, most of which straddle two lines: 0.817 -> 0.748 ms per pass on C2's pairs.  Used between the
// global passes of radix_sort_msd.hpp and into its in-LDS kernel, for 8-byte keys with 8-byte payloads.
template <class K, class P>
struct alignas((sizeof(K) + sizeof(P)) == 16 ? 16 : 8) RsPair {  // (only the 8 + 8 byte form is used; the others merely have to compile)
  K k;
  P p;
};

template <class K, class P, int ITEMS, bool IOTA, bool IN_AOS = false, bool OUT_AOS = false>
__global__ __launch_bounds__(kBlock) void rs_onesweep(const K *__restrict__ kin, const P *__restrict__ pin,
                                                      K *__restrict__ kout, P *__restrict__ pout, uint32_t n,
                                                      int shift, uint32_t mask, const uint32_t *__restrict__ gbase,
                                                      uint16_t *__restrict__ agg, uint32_t *__restrict__ incl,
                                                      uint32_t *__restrict__ ticket, int *__restrict__ err, int lab) {
  constexpr int TILE = kBlock * ITEMS;
  static_assert(TILE < 0x8000, "a tile's digit count must fit 15 bits");
  __shared__ uint32_t cnt[kWaves][kRsRadix];
  __shared__ uint32_t hcnt[kRsRadix];
  __shared__ uint32_t lstart[kRsRadix];
  __shared__ uint32_t goff[kRsRadix];
  __shared__ uint32_t shs[kWaves];
  __shared__ uint32_t s_tile;
  __shared__ int s_fail;
  // (tried: one buffer for the re-ordered keys and then the payloads -- 39 instead of 71 KB for 16-byte pairs, three
  // blocks per CU by registers instead of two by LDS, two more barriers: the pass 0.81 -> 0.765 ms on C2, the
  // assembly as a whole unchanged within run-to-run noise and C4's slower; not kept)
  __shared__ K skey[TILE];
  __shared__ P spay[TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    s_tile = atomicAdd(ticket, 1u);
    s_fail = 0;
  }
  for (int d = tid; d < kWaves * kRsRadix; d += kBlock) (&cnt[0][0])[d] = 0;
  hcnt[tid] = 0;
  __syncthreads();
  const uint32_t tile = s_tile;

  const uint32_t tile_base = tile * (uint32_t)TILE;
  const uint32_t valid = (n - tile_base) < (uint32_t)TILE ? (n - tile_base) : (uint32_t)TILE;
  const uint32_t base = tile_base + wave * (ITEMS * 64) + lane;
  K key[ITEMS];
  P pay[ITEMS];
  uint32_t rank[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    uint32_t i = base + k * 64;
    bool ok = i < n;
    if constexpr (IN_AOS) {
      RsPair<K, P> pr;
      pr.k = (K)~(K)0;
      pr.p = (P)0;
      if (ok) pr = reinterpret_cast<const RsPair<K, P> *>(kin)[i];
      key[k] = pr.k;
      pay[k] = pr.p;
    } else {
      key[k] = ok ? kin[i] : (K)~(K)0;  // padding sorts to the very end of the tile
      pay[k] = ok ? (IOTA ? (P)i : pin[i]) : (P)0;
    }
  }
  // ---- the tile's digit counts, and out with them (one add per run of equal neighbouring digits, see rs_hist_all)
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    const uint32_t d = (uint32_t)(key[k] >> shift) & mask;
    const uint32_t before = wave_prev_lane_dpp(d);
    const bool head = lane == 0 || d != before;
    const uint64_t hm = __ballot(head);
    if (head) {
      const uint64_t rest = lane == 63 ? 0ull : (hm >> (lane + 1));
      atomicAdd(&hcnt[d], rest ? (uint32_t)__builtin_ctzll(rest) + 1u : (uint32_t)(64 - lane));
    }
  }
  __syncthreads();
  // padding keys of the last tile were counted under digit `mask` of an all-ones key: take them out of what is
  // published (they are not written either)
  uint32_t pub = hcnt[tid];
  if ((uint32_t)tid == mask && valid < (uint32_t)TILE) pub -= (uint32_t)TILE - valid;
  uint16_t *mine = agg + (size_t)tile * kRsRadix + tid;
  __hip_atomic_store(mine, (uint16_t)(kRsAggReady | pub), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- rank every pair inside its wavefront (ballot match-any), wavefront counters in LDS
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    const uint32_t d = (uint32_t)(key[k] >> shift) & mask;
    uint64_t peers = ~0ull;
#pragma unroll
    for (int b = 0; b < ((lab & 0x2000) ? 0 : 8); b++) {  // lab 0x2000: no match-any (wrong ranks)
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
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    const uint32_t d = (uint32_t)(key[k] >> shift) & mask;
    uint32_t p = lstart[d] + cnt[wave][d] + rank[k];
    if (lab & 0x2000) p = (uint32_t)(wave * (ITEMS * 64) + k * 64 + lane);  // lab: ranks are not computed, keep the slot in range
    skey[p] = key[k];
    spay[p] = pay[k];
  }

  // ---- look back, for digit `tid`
  uint32_t excl = 0;
  if (!(lab & 0x8000)) {  // lab 0x8000 (with 0x4000: the offsets are not used): no look-back
    const uint32_t grp = tile / kRsGroup;
    const unsigned long long t0 = wall_clock64();
    bool fail = false, done = false;
    uint32_t rounds = 0, idle = 0;
    // round r: the counts of the tiles of group grp - r that lie before this one, and the inclusive prefix of group
    // grp - r - 1, requested together (one round trip, 36 bytes per digit; a walk is 3.8 rounds on average on C2).
    // Measured and worse, because the cost follows the bytes these loads and stores move, not the round trips: four
    // groups ahead per round trip (0.82 -> 0.95 ms per pass), 64 eight-byte words per round trip in the first layout
    // (0.92 -> 1.26), every walking tile putting out the prefixes it has just learnt (3.8 -> 3.1 rounds, 0.82 -> 0.87).
    for (uint32_t r = 0; !done && !fail; r++) {
      const uint32_t g = grp - r;
      const uint32_t first_t = g * kRsGroup;
      const uint32_t end_t = r == 0 ? tile : first_t + kRsGroup;  // exclusive
      while (true) {
        rounds++;
        uint16_t w[kRsGroup];
#pragma unroll
        for (int j = 0; j < kRsGroup; j++)
          w[j] = first_t + j < end_t ? __hip_atomic_load(agg + (size_t)(first_t + j) * kRsRadix + tid, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT)
                                     : kRsAggReady;
        const uint32_t iw = g > 0 ? __hip_atomic_load(incl + (size_t)(g - 1) * kRsRadix + tid, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT)
                                  : 1u;  // before group 0: an inclusive prefix of nothing
        bool all = true;
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < kRsGroup; j++) {
          if (w[j] == 0) all = false;
          if (w[j] == kRsAggPoison) fail = true;
          sum += w[j] & 0x7fffu;
        }
        if (iw == kRsInclPoison) fail = true;
        if (fail) break;
        if (all) {
          excl += sum;
          if (iw != 0) {
            excl += iw - 1u;
            done = true;
          }
          break;
        }
        idle++;
        if (wall_clock64() - t0 > kRsSpinLimit) { fail = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if ((lab & 0x40000) && tid == 0) {  // lab: how long the walks are (printed by the host)
      atomicAdd(ticket + 1, rounds);
      atomicAdd(ticket + 2, idle);
      atomicMax(ticket + 3, rounds);
    }
    const bool closes = tile % kRsGroup == kRsGroup - 1;  // the group's last tile publishes the group's inclusive prefix
    if (fail) {
      s_fail = 1;
      *err = 1;
      __hip_atomic_store(mine, kRsAggPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (closes) __hip_atomic_store(incl + (size_t)grp * kRsRadix + tid, kRsInclPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (closes) {
      __hip_atomic_store(incl + (size_t)grp * kRsRadix + tid, excl + pub + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  goff[tid] = gbase[tid] + excl - ex;
  __syncthreads();
  if (s_fail) return;  // uniform: nothing of this tile is written
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    const uint32_t i = k * kBlock + tid;
    if (i < valid) {
      const K kk = skey[i];
      const uint32_t d = (uint32_t)(kk >> shift) & mask;
      // lab 0x4000: contiguous stores (also forced by 0x2000 and 0x8000, whose offsets mean nothing)
      const uint32_t g = (lab & 0xe000) ? tile_base + i : goff[d] + i;
      if constexpr (OUT_AOS) {
        RsPair<K, P> pr;
        pr.k = kk;
        pr.p = spay[i];
        reinterpret_cast<RsPair<K, P> *>(kout)[g] = pr;
      } else {
        kout[g] = kk;
        pout[g] = spay[i];
      }
    }
  }
}

// aos_a / aos_b (both or neither): buffers of n pairs each.  The passes then write pairs side by side into them in turn
// (the first pass still reads keys_a / pay_in), keys_b / pay_a / pay_b are not touched, and *keys_out points at the
// buffer of pairs that holds the result (*pay_out = nullptr).
template <class K, class P>
int radix_sort_pairs_onesweep(futhark_context *ctx, K *keys_a, K *keys_b, const P *pay_in, P *pay_a, P *pay_b,
                              int64_t n, int begin_bit, int end_bit, K **keys_out, const P **pay_out, int *err,
                              RsPair<K, P> *aos_a = nullptr, RsPair<K, P> *aos_b = nullptr) {
  constexpr int ITEMS = RsCfg<K, P>::kItems;
  constexpr int TILE = RsCfg<K, P>::kTile;
  const uint32_t ntiles = (uint32_t)cdiv(n, TILE);
  const int passes = (end_bit - begin_bit + 7) / 8;
  Tmp<uint32_t> ghist(ctx);                 // [passes][256] digit bases, then one ticket counter per pass
  Tmp<uint16_t> agg(ctx);                   // [ntiles][256] tile counts, cleared before every pass
  Tmp<uint32_t> incl(ctx);                  // [groups][256] inclusive prefixes at group ends, likewise
  const int64_t gcount = (int64_t)passes * kRsRadix + 4 * passes;  // per pass: ticket + three lab counters
  SAMD_TRY(ghist.get(gcount));
  const int64_t ngroups = cdiv((int64_t)ntiles, kRsGroup);
  SAMD_TRY(agg.get((int64_t)ntiles * kRsRadix));
  SAMD_TRY(incl.get(ngroups * kRsRadix));
  SAMD_HIP(ctx, hipMemsetAsync(ghist.p, 0, sizeof(uint32_t) * (size_t)gcount, ctx->stream));
  {
    // the kernel that packed these keys may have counted their top digits already (assembly.hpp: pack_coo_hist_kernel)
    int pre_idx[kRsMaxPasses];
    bool pre = ctx->pre_hist != nullptr && ctx->pre_hist_keys == (const void *)keys_a && ctx->pre_hist_n == n;
    for (int p = 0; pre && p < passes; p++) {
      const int shift = begin_bit + 8 * p, nb = end_bit - shift < 8 ? end_bit - shift : 8;
      pre_idx[p] = -1;
      for (int d = 0; d < ctx->pre_digits; d++)
        if (ctx->pre_shift[d] == shift && ctx->pre_bits[d] == nb) pre_idx[p] = d;
      if (pre_idx[p] < 0) pre = false;
    }
    if (pre) {
      for (int p = 0; p < passes; p++)
        SAMD_HIP(ctx, hipMemcpyAsync(ghist.p + p * kRsRadix, ctx->pre_hist + pre_idx[p] * kRsRadix,
                                     sizeof(uint32_t) * kRsRadix, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
      const int64_t cap = (int64_t)ctx->num_cu * 8;
      const unsigned grid = (unsigned)(ntiles < cap ? ntiles : cap);
      hipLaunchKernelGGL((rs_hist_all<K, ITEMS>), dim3(grid), dim3(kBlock), 0, ctx->stream, (const K *)keys_a,
                         (uint32_t)n, begin_bit, end_bit, ghist.p);
      SAMD_LAUNCH_CHECK(ctx);
    }
    hipLaunchKernelGGL(rs_scan_hist, dim3(passes), dim3(kBlock), 0, ctx->stream, ghist.p);
    SAMD_LAUNCH_CHECK(ctx);
  }
  K *kin = keys_a, *kout = keys_b;
  const P *pin = pay_in;
  P *pout = pay_a, *pspare = pay_b;
  for (int p = 0; p < passes; p++) {
    const int shift = begin_bit + 8 * p;
    const int nb = end_bit - shift < 8 ? end_bit - shift : 8;
    const uint32_t mask = (1u << nb) - 1u;
    SAMD_HIP(ctx, hipMemsetAsync(agg.p, 0, sizeof(uint16_t) * (size_t)ntiles * kRsRadix, ctx->stream));
    SAMD_HIP(ctx, hipMemsetAsync(incl.p, 0, sizeof(uint32_t) * (size_t)ngroups * kRsRadix, ctx->stream));
    const uint32_t *gbase = ghist.p + p * kRsRadix;
    uint32_t *ticket = ghist.p + passes * kRsRadix + 4 * p;
    if (aos_a != nullptr) {
      if constexpr (sizeof(K) == 8 && sizeof(P) == 8) {
        RsPair<K, P> *dst = (p & 1) ? aos_b : aos_a;
        const RsPair<K, P> *src = (p & 1) ? aos_a : aos_b;  // written by pass p - 1
        if (p == 0)
          hipLaunchKernelGGL((rs_onesweep<K, P, ITEMS, false, false, true>), dim3(ntiles), dim3(kBlock), 0, ctx->stream,
                             (const K *)keys_a, pay_in, reinterpret_cast<K *>(dst), (P *)nullptr, (uint32_t)n, shift, mask, gbase,
                             agg.p, incl.p, ticket, err, (int)ctx->opt_ablate);
        else
          hipLaunchKernelGGL((rs_onesweep<K, P, ITEMS, false, true, true>), dim3(ntiles), dim3(kBlock), 0, ctx->stream,
                             reinterpret_cast<const K *>(src), (const P *)nullptr, reinterpret_cast<K *>(dst), (P *)nullptr,
                             (uint32_t)n, shift, mask, gbase, agg.p, incl.p, ticket, err, (int)ctx->opt_ablate);
        SAMD_LAUNCH_CHECK(ctx);
        kin = reinterpret_cast<K *>(dst);
        pin = nullptr;
        continue;
      } else {
        return set_error(ctx, "internal: side-by-side pairs are for 8-byte keys and payloads");
      }
    }
    if (pin == nullptr)
      hipLaunchKernelGGL((rs_onesweep<K, P, ITEMS, true>), dim3(ntiles), dim3(kBlock), 0, ctx->stream, (const K *)kin,
                         pin, kout, pout, (uint32_t)n, shift, mask, gbase, agg.p, incl.p, ticket, err, (int)ctx->opt_ablate);
    else
      hipLaunchKernelGGL((rs_onesweep<K, P, ITEMS, false>), dim3(ntiles), dim3(kBlock), 0, ctx->stream, (const K *)kin,
                         pin, kout, pout, (uint32_t)n, shift, mask, gbase, agg.p, incl.p, ticket, err, (int)ctx->opt_ablate);
    SAMD_LAUNCH_CHECK(ctx);
    K *tk = kin; kin = kout; kout = tk;
    P *written = pout;
    pout = (written == pay_a) ? pspare : pay_a;
    pin = written;
  }
  if (ctx->opt_ablate & 0x40000) {  // lab: look-back statistics of every pass
    std::vector<uint32_t> h((size_t)4 * passes);
    SAMD_HIP(ctx, hipMemcpyAsync(h.data(), ghist.p + passes * kRsRadix, sizeof(uint32_t) * h.size(), hipMemcpyDeviceToHost,
                                 ctx->stream));
    SAMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int p = 0; p < passes; p++)
      fprintf(stderr, "look-back pass %d: %u tiles, %.2f rounds per tile (max %u), %.2f idle rounds per tile\n", p, h[4 * p],
              (double)h[4 * p + 1] / ntiles, h[4 * p + 3], (double)h[4 * p + 2] / ntiles);
  }
  *keys_out = kin;
  *pay_out = pin;
  return 0;
}

// Sorts n (< 2^32) (key, payload) pairs by key bits [begin_bit, end_bit), stable.
// keys_a / pay_in are the inputs (pay_in is only read; nullptr = the payload is the
// element's position, P = uint32_t); keys_b, pay_a, pay_b are scratch of the same length.
// On return *keys_out / *pay_out point at whichever buffers hold the result (pay_in itself
// when no pass ran: nullptr then still means "identity").  sort_err: device int the one-sweep
// variant raises if a look-back wait timed out (the caller reads it at its next synchronisation).
// aos_a / aos_b / aos_used (all or none): see radix_sort_pairs_onesweep; *aos_used says whether the result is a buffer
// of pairs (the histogram + scan variant and an empty bit range leave it false and use the separate arrays).
template <class K, class P>
int radix_sort_pairs(futhark_context *ctx, K *keys_a, K *keys_b, const P *pay_in, P *pay_a, P *pay_b, int64_t n,
                     int begin_bit, int end_bit, K **keys_out, const P **pay_out, int *sort_err,
                     RsPair<K, P> *aos_a = nullptr, RsPair<K, P> *aos_b = nullptr, bool *aos_used = nullptr) {
  constexpr int ITEMS = RsCfg<K, P>::kItems;
  constexpr int TILE = RsCfg<K, P>::kTile;
  if (n >= (int64_t)0xffff0000LL) return set_error(ctx, "sort: more than 2^32 entries on one device");
  const uint32_t ntiles = (uint32_t)cdiv(n, TILE);
  if (aos_used) *aos_used = false;
  if (ctx->opt_sort_algo == 0 && !ctx->sort_force_scan && n > 0 && end_bit > begin_bit && end_bit - begin_bit <= 8 * kRsMaxPasses) {
    const bool aos = aos_used != nullptr && aos_a != nullptr && aos_b != nullptr && pay_in != nullptr;
    if (aos) *aos_used = true;
    return radix_sort_pairs_onesweep<K, P>(ctx, keys_a, keys_b, pay_in, pay_a, pay_b, n, begin_bit, end_bit, keys_out,
                                           pay_out, sort_err, aos ? aos_a : nullptr, aos ? aos_b : nullptr);
  }
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
