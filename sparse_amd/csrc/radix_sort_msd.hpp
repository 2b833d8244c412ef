// radix_sort_msd.hpp -- assembly with fewer trips through HBM: the TOP key bits by global passes, everything
// below them inside LDS, and the record written straight from there.
//
// An LSD sort of b key bits moves every pair through HBM b/8 times (C2: 48 bits, six passes, 5.5 of the
// assembly's 7.5 ms), and head-flag scan + merge_emit read the sorted pairs twice more.  Here the top T bits (16 or
// 24) are sorted first by the ordinary stable passes of radix_sort.hpp, which leaves the pairs grouped into 2^T
// buckets, input order kept inside each bucket.  When the buckets are small, one more kernel finishes the job: a
// workgroup takes a chunk of whole consecutive buckets, sorts it by the remaining low bits inside LDS, folds the
// duplicates and writes col_idx / vals / row ends (rs_local_sort_emit).  Two or three global passes plus one read
// of the pairs and one write of the record.  Measured: C2 7.5 -> 5.3 ms, C5 85 -> 65 ms, C4 2.25 -> 2.0 ms.
//
// Which T, if any, is decided from a sampled histogram of the top 16 bits (msd_choose): skewed keys (power-law
// rows, C3) keep the LSD sort.  The estimate can be wrong; a chunk that does not fit its workgroup emits nothing and
// raises *oversize, which the caller reads at its one synchronisation and answers by assembling again without
// this file (assembly.hpp: kAssembleAgain).
#pragma once

#include <type_traits>

#include "radix_sort.hpp"

namespace samd {

template <class T> struct RlBits;
template <> struct RlBits<double> { typedef uint64_t type; };
template <> struct RlBits<float> { typedef uint32_t type; };
template <> struct RlBits<int32_t> { typedef uint32_t type; };
template <class T>
__device__ __forceinline__ T rl_from_bits(typename RlBits<T>::type v) {
  T out;
  __builtin_memcpy(&out, &v, sizeof(T));
  return out;
}

constexpr int kRlDigitBits = 5;  // in-LDS passes: 32 digits, 16 counter words per thread (6 bits: one pass fewer on C2, same time)
constexpr int kMsdBins = 1 << 16;
constexpr int64_t kMsdMinCount = 1 << 22;    // below this the passes are launch- and ramp-bound anyway
constexpr int64_t kMsdSamples = 1 << 21;

// histogram of bits [shift, shift + 16) of every stride-th key.  gate (may be null): the largest bin of an earlier,
// much smaller sample; if that already shows heavy skew the full sample is skipped -- on power-law keys
// its adds would pile up on a few addresses (RMAT scale 24: 1.4 ms for 2M samples) and the answer is known.
// 2^15 samples over the bins in use: even keys stay below their mean + a few deviations
__host__ __device__ inline bool msd_gate_skewed(unsigned long long mx, unsigned long long used) {
  return used == 0 || mx * used > 32ull * used + 4ull * 32768ull;
}
// where a sampled key comes from: the packed array, or the triples it will be packed from
template <class K>
struct KeyArr {
  const K *k;
  __device__ __forceinline__ K operator()(int64_t i) const { return k[i]; }
};
struct KeyCoo {
  const int64_t *major, *minor;
  int64_t n_major, n_minor;
  int minor_bits;
  __device__ __forceinline__ uint64_t operator()(int64_t i) const {
    const int64_t r = major[i], c = minor[i];
    if (r < 0 || r >= n_major || c < 0 || c >= n_minor) return 0;  // as pack_coo_kernel stores it (the assembly fails later)
    return (minor_bits >= 64 ? 0ull : ((uint64_t)r << minor_bits)) | (uint64_t)c;
  }
};

template <class KF>
__global__ __launch_bounds__(kBlock) void msd_sample_kernel(KF keys, int64_t count, int64_t stride,
                                                            int shift, uint32_t *__restrict__ hist,
                                                            const unsigned long long *__restrict__ gate) {
  if (gate && msd_gate_skewed(gate[0], gate[1])) return;
  const int64_t total = (int64_t)gridDim.x * kBlock;
  const int lane = threadIdx.x & 63;
  for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s * stride < count; s += total) {
    const uint32_t bin = (uint32_t)(keys(s * stride) >> shift) & (uint32_t)(kMsdBins - 1);
    // sorted inputs put a whole wavefront into one bin: one add for all of them
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)bin);
    const uint64_t active = __ballot(true), same = __ballot(bin == first);
    if (same == active) {
      if (lane == __builtin_ctzll(active)) atomicAdd(&hist[bin], (uint32_t)__popcll(active));
    } else {
      atomicAdd(&hist[bin], 1u);
    }
  }
}

// out[0] = largest bin, out[1] = number of bins in use
__global__ __launch_bounds__(kBlock) void msd_max_kernel(const uint32_t *__restrict__ hist,
                                                         unsigned long long *__restrict__ out) {
  unsigned long long mx = 0, used = 0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < kMsdBins; i += gridDim.x * kBlock) {
    const uint32_t h = hist[i];
    mx = h > mx ? h : mx;
    used += h ? 1 : 0;
  }
  mx = wave_reduce(mx, OpMax());
  used = wave_reduce(used, OpAdd());
  if ((threadIdx.x & 63) == 0) {
    atomicMax(out, mx);
    if (used) atomicAdd(out + 1, used);
  }
}

// The two samples of the top 16 key bits (d_small[4 .. 8) = [max, used, gate max, gate used]); no synchronisation.
template <class KF>
int msd_launch_samples(futhark_context *ctx, KF keys, int64_t count, int end_bit) {
  Tmp<uint32_t> hist(ctx);
  SAMD_TRY(hist.get(2 * kMsdBins));
  SAMD_HIP(ctx, hipMemsetAsync(hist.p, 0, sizeof(uint32_t) * 2 * kMsdBins, ctx->stream));
  SAMD_HIP(ctx, hipMemsetAsync(ctx->d_small + 4, 0, 4 * sizeof(int64_t), ctx->stream));
  unsigned long long *res = reinterpret_cast<unsigned long long *>(ctx->d_small + 4);
  const int64_t stride = cdiv(count, kMsdSamples);
  const int64_t samples = cdiv(count, stride);
  // a small sample first; the full one runs only if that one is even
  const int64_t gstride = cdiv(count, (int64_t)1 << 15);
  hipLaunchKernelGGL((msd_sample_kernel<KF>), dim3((unsigned)cdiv(cdiv(count, gstride), kBlock)), dim3(kBlock), 0,
                     ctx->stream, keys, count, gstride, end_bit - 16, hist.p + kMsdBins, (const unsigned long long *)nullptr);
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(msd_max_kernel, dim3(64), dim3(kBlock), 0, ctx->stream, (const uint32_t *)(hist.p + kMsdBins), res + 2);
  SAMD_LAUNCH_CHECK(ctx);
  const int64_t want = cdiv(samples, kBlock), capg = (int64_t)ctx->num_cu * 16;
  hipLaunchKernelGGL((msd_sample_kernel<KF>), dim3((unsigned)(want < capg ? want : capg)), dim3(kBlock), 0, ctx->stream,
                     keys, count, stride, end_bit - 16, hist.p, (const unsigned long long *)(res + 2));
  SAMD_LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(msd_max_kernel, dim3(64), dim3(kBlock), 0, ctx->stream, (const uint32_t *)hist.p, res);
  SAMD_LAUNCH_CHECK(ctx);
  return 0;  // (hist goes back to the pool here; whatever takes it next runs behind these kernels on the same stream)
}

// Chooses the number of top bits the global passes take (*tb = first of them; -1 = keep the LSD sort) and the
// nominal chunk length.  One small read-back.
template <class K>
int msd_choose(futhark_context *ctx, const K *keys, int64_t count, int begin_bit, int end_bit, int cap, int cap_small,
               int *tb, uint32_t *ch, bool *small, int64_t *slot8 = nullptr) {  // slot8: receives d_small[8] if this call reads back
  *tb = -1;
  *ch = 0;
  *small = false;
  const int total = end_bit - begin_bit;
  if (ctx->opt_sort_msd < 0 || count <= 0) return 0;
  if (ctx->opt_sort_msd > 0) {  // tests and labs: this many top bits, whatever the input
    int t = (int)ctx->opt_sort_msd;
    if (t >= total) t = total - 1;
    if (t > 32) t = 32;  // top_tab keeps a bucket's top bits in 32
    if (t < 1) return 0;
    *tb = end_bit - t;
    uint32_t c = ctx->opt_sort_msd_chunk > 0 ? (uint32_t)ctx->opt_sort_msd_chunk : (uint32_t)cap / 2;
    if (c > (uint32_t)cap) c = (uint32_t)cap;
    *ch = c;
    *small = ctx->opt_sort_msd_chunk > 0 && c <= (uint32_t)cap_small / 2;  // short chunks: the small-chunk kernel
    return 0;
  }
  if (count < kMsdMinCount || total <= 24 || end_bit < 16) return 0;
  int64_t h[5];
  const int64_t stride = cdiv(count, kMsdSamples);
  const int64_t samples = cdiv(count, stride);
  if (ctx->pre_sample && ctx->pre_sample_n == count && ctx->pre_sample_end == end_bit && begin_bit == 0) {
    // taken from the triples before they were packed: the words are on their way (or there) while the pack kernel runs
    SAMD_HIP(ctx, hipEventSynchronize(ctx->ev_pre));
    for (int i = 0; i < 5; i++) h[i] = ctx->h_small[8 + i];
  } else {
    SAMD_TRY(msd_launch_samples(ctx, KeyArr<K>{keys}, count, end_bit));
    SAMD_TRY(read_small(ctx, ctx->d_small + 4, 5, h));
  }
  if (slot8) *slot8 = h[4];
  if (msd_gate_skewed((unsigned long long)h[2], (unsigned long long)h[3]) || h[1] <= 0) return 0;  // skewed at first sight
  const double per_sample = (double)count / (double)samples;
  const double max16 = (double)h[0] * per_sample;            // largest 16-bit bucket, estimated (from above)
  const double avg16 = (double)count / (double)h[1];
  if (max16 > 4.0 * avg16 + 64.0 * per_sample) return 0;     // skewed: the buckets below would not be even either
  // Fewest global passes first; within that, the smaller chunk (more workgroups per CU: C5 shard 3.0 -> 2.3 ms)
  // if the buckets leave it at least half full.
  double maxb = max16;
  int t = 16;
  if (maxb > 0.5 * cap) {
    // the next 8 bits split a bucket into up to 256; allow the finer level to be 4 times less even
    if (total <= 32) return 0;
    t = 24;
    maxb = max16 / 256.0 * 4.0 + 64.0;
    if (maxb > 0.5 * cap) return 0;
  }
  if (ctx->opt_ablate & 0x80000)
    fprintf(stderr, "msd_choose: samples %lld, largest bin %lld, bins in use %lld, max16 %.0f, avg16 %.0f, T %d, maxb %.0f\n",
            (long long)samples, (long long)h[0], (long long)h[1], max16, avg16, t, maxb);
  // The small chunk as long as the largest bucket leaves a quarter of it free: nominal spacing = capacity - largest
  // bucket, so no chunk can outgrow its workgroup; spacings shorter than the buckets leave empty chunks, which cost
  // next to nothing (msd_mark_empty_kernel).  C2 (buckets of 1280): 5.1 -> 4.8 ms with the small chunk.
  const int use_cap = maxb <= 0.75 * cap_small ? cap_small : cap;
  uint32_t c = (uint32_t)use_cap - (uint32_t)maxb;
  c &= ~255u;
  if (c < 256u) c = 256u;
  *tb = end_bit - t;
  *ch = c;
  *small = use_cap == cap_small;
  return 0;
}

// chunk c = [a[c], a[c+1]): a[c] = the first bucket boundary at or after c * ch
template <class K>
__global__ __launch_bounds__(kBlock) void msd_chunk_plan_kernel(const K *__restrict__ keys_base, uint32_t n, int tb,
                                                                uint32_t ch, uint32_t nchunks,
                                                                uint32_t *__restrict__ a, int kstride) {  // kstride 2: keys inside pairs (RsPair)
  auto keys = [&](uint64_t i) -> K { return keys_base[i * (uint64_t)kstride]; };
  const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
  if (c > nchunks) return;
  if (c == nchunks) { a[c] = n; return; }
  if (c == 0) { a[0] = 0; return; }
  const uint64_t s = (uint64_t)c * ch;  // < n
  const K top0 = (K)(keys(s - 1) >> tb);
  // smallest i in [s, n] with i == n or top(keys[i]) != top0: gallop, then bisect
  uint64_t f = s - 1, t, step = 1;
  while (true) {
    const uint64_t probe = f + step;
    if (probe >= n) { t = n; break; }
    if ((K)(keys(probe) >> tb) != top0) { t = probe; break; }
    f = probe;
    step *= 2;
  }
  while (t - f > 1) {
    const uint64_t mid = f + (t - f) / 2;
    if ((K)(keys(mid) >> tb) != top0) t = mid; else f = mid;
  }
  a[c] = (uint32_t)t;
}

// Chunks that hold nothing (the nominal spacing is shorter than the buckets) are taken out of the list before the
// sort kernel runs: even a workgroup that leaves at once holds its slot for a ticket and two loads, and at two empty
// chunks in three that was 0.3 ms on C2.  Flag = chunk holds something; exclusive scan = its place in the short list.
struct ChunkHoldsIn {
  const uint32_t *a;
  __device__ __forceinline__ uint32_t operator()(int64_t i) const { return a[i] != a[i + 1] ? 1u : 0u; }
};
__global__ __launch_bounds__(kBlock) void msd_chunk_compact_kernel(const uint32_t *__restrict__ a, uint32_t nchunks,
                                                                   const uint32_t *__restrict__ place, uint32_t n,
                                                                   uint32_t *__restrict__ out) {
  const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
  if (c < nchunks && a[c] != a[c + 1]) out[place[c]] = a[c];
  if (c == nchunks) out[place[nchunks]] = n;  // place[nchunks] = number of chunks kept (written by the host-side copy below)
}

// Addresses are padded by one slot per 32 so that blocked reads (lane stride = items per thread) stay off each
// other's banks.
__device__ __forceinline__ uint32_t rl_pad(uint32_t i) { return i + (i >> 5); }

// ------------------------------------------------------------------ in-LDS sort fused with the record's emission
// The assembly's own version of the kernel above (assembly.hpp calls it; begin_bit = 0): the chunk is sorted as
// 8-byte composites (bucket ordinal | low key bits | 13-bit position in the chunk) -- payloads stay where they
// were loaded and are fetched by position at the end, so a pass moves 8 bytes per pair instead of 18 -- and the
// sorted chunk never goes back to HBM as pairs: duplicates (equal composites above the tag; they cannot span
// chunks, a chunk being whole buckets) are folded left to right, the chunk's first output slot comes from a
// decoupled look-back over the chunks before it (one status word per chunk, tickets as in rs_onesweep), and the
// workgroup writes col_idx / vals / row ends directly.  Replaces head-flag scan + merge_emit (C2: 0.95 ms) and the
// 16-byte write + 20-byte read between them.
constexpr int kRlTagBits = 13;

// Look-back by one whole wavefront (all 64 lanes call it): 64 predecessors per step instead of one -- the first
// 512 chunks of a launch start together and finish together, and a lane-by-lane walk over them was 0.25 ms of the
// kernel on C2.  Returns the number of entries emitted by all earlier chunks; every lane gets it.
// In two halves, so that the first round trip is in flight while the caller does something else: rl_publish puts
// out this chunk's count and requests the status words of the 64 chunks before it; rl_finish (later, same wavefront)
// starts from those words.  Device-scope loads take ~2 us under load, and a chunk that sat waiting for them kept its
// workgroup's slot for nothing (C2: 0.49 of the in-LDS kernel's 1.67 ms went to the look-back).
__device__ __forceinline__ unsigned long long rl_publish(unsigned long long *__restrict__ status, uint32_t chunk,
                                                         unsigned long long mine_count) {
  const int lane = threadIdx.x & 63;
  unsigned long long *mine = status + chunk;
  if (lane == 0)
    __hip_atomic_store(mine, (chunk == 0 ? kRsFlagIncl : kRsFlagAgg) | mine_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int64_t t = (int64_t)chunk - 1 - lane;
  return t >= 0 ? __hip_atomic_load(status + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kRsFlagIncl;
}

__device__ __forceinline__ unsigned long long rl_finish(unsigned long long *__restrict__ status, uint32_t chunk,
                                                        unsigned long long mine_count, unsigned long long first_words,
                                                        bool *fail) {
  const int lane = threadIdx.x & 63;
  unsigned long long *mine = status + chunk;
  *fail = false;
  if (chunk == 0) return 0;
  unsigned long long excl = 0;
  int64_t top = (int64_t)chunk - 1;  // lane l looks at chunk top - l
  const unsigned long long t0 = wall_clock64();
  bool bad = false, first = true;
  while (true) {
    const int64_t t = top - lane;
    unsigned long long w = kRsFlagIncl;  // before chunk 0: an inclusive prefix of nothing
    if (first) w = first_words;
    else if (t >= 0) w = __hip_atomic_load(status + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    first = false;
    const unsigned long long f = w >> 62;
    const uint64_t incl = __ballot(f == 2), pending = __ballot(f == 0), poison = __ballot(f == 3);
    // the nearest inclusive prefix, and everything nearer than it, must be there
    const int stop = incl ? __builtin_ctzll(incl) : 64;
    const uint64_t nearer = stop >= 64 ? ~0ull : ((1ull << stop) - 1ull);
    if (poison & (nearer | (stop < 64 ? (1ull << stop) : 0ull))) { bad = true; break; }
    if (pending & nearer) {
      if (wall_clock64() - t0 > kRsSpinLimit) { bad = true; break; }
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    unsigned long long v = (lane <= stop && lane < 64) ? (w & kRsValMask) : 0ull;
    if (lane > stop) v = 0;
    excl += wave_reduce(v, OpAdd());
    if (stop < 64) break;
    top -= 64;
  }
  *fail = bad;
  if (lane == 0) {
    if (bad) __hip_atomic_store(mine, kRsFlagPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(mine, kRsFlagIncl | (excl + mine_count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return excl;
}

__device__ __forceinline__ unsigned long long rl_lookback(unsigned long long *__restrict__ status, uint32_t chunk,
                                                          unsigned long long mine_count, bool *fail) {
  const unsigned long long w = rl_publish(status, chunk, mine_count);
  return rl_finish(status, chunk, mine_count, w, fail);
}

template <class T, class I, bool INDIRECT, int ITEMS, int DB, bool AOS = false>  // AOS: kin points at pairs side by side (RsPair), pin is not used
__global__ __launch_bounds__(kBlock) void rs_local_sort_emit(
    const uint64_t *__restrict__ kin, const typename std::conditional<INDIRECT, uint32_t, typename RlBits<T>::type>::type *__restrict__ pin,
    const T *__restrict__ vals_src, const uint32_t *__restrict__ chunk_a, const uint32_t *__restrict__ nchunks_dev, int tb,
    int minor_bits,
    I *__restrict__ col_out, T *__restrict__ vals_out, int64_t *__restrict__ row_ptr, int row_atomic,
    unsigned long long *__restrict__ status, uint32_t *__restrict__ ticket, unsigned long long *__restrict__ merged_out,
    int *__restrict__ oversize, int *__restrict__ err, int lab, unsigned long long *__restrict__ rows_seen) {  // rows_seen (may be null; kSkipCounters counters of scan.hpp): += rows that end in this chunk
  typedef typename std::conditional<INDIRECT, uint32_t, typename RlBits<T>::type>::type P;
  constexpr int CAP = kBlock * ITEMS;
  static_assert(CAP <= (1 << kRlTagBits), "position tag too narrow");
  constexpr int CAPP = CAP + CAP / 32 + 1;
  constexpr int NBH = (1 << DB) / 2;
  constexpr size_t kR1 = ((size_t)CAPP * 8 + 15) & ~(size_t)15;  // composites / counters / emitted columns
  constexpr size_t kPV = sizeof(P) > sizeof(T) ? sizeof(P) : sizeof(T);
  constexpr size_t kR2 = ((size_t)CAPP * kPV + 15) & ~(size_t)15;  // payloads by position / emitted values
  static_assert((size_t)(NBH * kBlock + NBH * kBlock / 32 + 1) * 4 <= kR1, "counters do not fit the composite buffer");
  __shared__ __align__(16) unsigned char raw[kR1 + kR2];
  constexpr int kRlMaxBuckets = CAP / 2;  // bucket ordinals per chunk; more = the chunk is refused like an oversize one
  __shared__ uint32_t top_tab[kRlMaxBuckets];
  __shared__ uint32_t shs[kWaves];
  __shared__ uint32_t s_chunk;
  __shared__ unsigned long long s_off;
  __shared__ int s_fail;
  __shared__ unsigned long long s_fmin, s_fmax;
  uint64_t *scomp = reinterpret_cast<uint64_t *>(raw);
  P *spay = reinterpret_cast<P *>(raw + kR1);
  uint32_t *C = reinterpret_cast<uint32_t *>(raw);
  I *scol = reinterpret_cast<I *>(raw);
  T *sval = reinterpret_cast<T *>(raw + kR1);

  const int tid = threadIdx.x;
  const uint32_t nchunks = *nchunks_dev;  // chunks that hold something; the grid was sized before this was known
  if (blockIdx.x >= nchunks) return;
  if (tid == 0) {
    s_chunk = atomicAdd(ticket, 1u);
    s_fail = 0;
    s_fmin = ~0ull;
    s_fmax = 0ull;
  }
  __syncthreads();
  const uint32_t chunk = s_chunk;
  if (chunk >= nchunks) return;
  const uint32_t a = chunk_a[chunk], b = chunk_a[chunk + 1];
  const uint32_t L = b - a;
  const bool last_chunk = chunk + 1 == nchunks;
  if (L == 0 || L > (uint32_t)CAP) {  // nothing here, or it does not fit: pass the count on, say so
    if (tid < 64) {
      bool fail;
      const unsigned long long off = rl_lookback(status, chunk, 0, &fail);
      if (tid == 0) {
        if (L) *oversize = 1;
        if (fail) *err = 1;
        else if (last_chunk) *merged_out = off;
      }
    }
    return;
  }
  const int R = (int)((L + kBlock - 1) / kBlock);
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    const uint32_t i = (uint32_t)k * kBlock + tid;
    if (k < R && i < L) {
      if constexpr (AOS) {
        const RsPair<uint64_t, P> pr = reinterpret_cast<const RsPair<uint64_t, P> *>(kin)[a + i];
        scomp[rl_pad(i)] = pr.k;
        spay[rl_pad(i)] = pr.p;
      } else {
        scomp[rl_pad(i)] = kin[a + i];
        spay[rl_pad(i)] = pin[a + i];
      }
    }
  }
  __syncthreads();
  uint64_t comp[ITEMS];
  uint32_t rank[ITEMS];
  const uint32_t pbase = (uint32_t)tid * (uint32_t)R;
  // The low tb bits of a key are (row bits below the buckets | column), or just the column's low bits when the
  // buckets already split rows.  Inside a chunk the column part F spans far less than its field on banded and stencil
  // records (C2: ~9000 of 2^24), so the chunk sorts F - min(F) in as many bits as its span needs: C2 34 -> 24 sorted
  // bits, five in-LDS passes instead of seven.
  const int fbits = tb < minor_bits ? tb : minor_bits;
  const uint64_t fmask = (1ull << fbits) - 1ull;  // fbits <= tb < 64
  const uint64_t lowmask_in = (1ull << tb) - 1ull;
  uint32_t heads = 0;
  {
    uint64_t before = (pbase > 0 && pbase < L) ? scomp[rl_pad(pbase - 1)] : 0ull;
    uint64_t fmn = ~0ull, fmx = 0ull;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const uint32_t p = pbase + k;
      const bool live = k < R && p < L;
      comp[k] = live ? scomp[rl_pad(p)] : ~0ull;  // still the key
      const bool head = live && (p == 0 || (comp[k] >> tb) != (before >> tb));
      heads += head ? 1u : 0u;
      rank[k] = head ? (heads | 0x80000000u) : heads;  // inclusive count in the thread, top bit = bucket head
      before = comp[k];
      if (live) {
        const uint64_t f = comp[k] & fmask;
        fmn = f < fmn ? f : fmn;
        fmx = f > fmx ? f : fmx;
      }
    }
    fmn = wave_reduce(fmn, OpMin());
    fmx = wave_reduce(fmx, OpMax());
    if ((tid & 63) == 0) {
      atomicMin(&s_fmin, (unsigned long long)fmn);
      atomicMax(&s_fmax, (unsigned long long)fmx);
    }
  }
  uint32_t nbk;
  const uint32_t hex = block_exclusive_add<uint32_t>(heads, shs, nbk);  // ends with a barrier: every key is in registers
  const uint64_t fmin = s_fmin, fspan = s_fmax - fmin;  // L > 0: some thread held a live pair
  const int fb2 = fspan == 0 ? 0 : 64 - __builtin_clzll(fspan);  // bits of F - fmin
  const int lowbits = (tb - fbits) + fb2;                         // what the chunk sorts below the bucket ordinal
  const uint64_t lowmask = (1ull << lowbits) - 1ull, fmask2 = (1ull << fb2) - 1ull;
  if (nbk > (uint32_t)kRlMaxBuckets) {
    if (tid < 64) {
      bool fail;
      const unsigned long long off = rl_lookback(status, chunk, 0, &fail);
      if (tid == 0) {
        *oversize = 1;
        if (fail) *err = 1;
        else if (last_chunk) *merged_out = off;
      }
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    const uint32_t p = pbase + k;
    if (k < R && p < L) {
      const uint32_t o = hex + (rank[k] & 0x7fffffffu) - 1u;
      if (rank[k] & 0x80000000u) top_tab[o] = (uint32_t)(comp[k] >> tb);
      const uint64_t low = (((comp[k] & lowmask_in) >> fbits) << fb2) | ((comp[k] & fmask) - fmin);
      comp[k] = ((((uint64_t)o << lowbits) | low) << kRlTagBits) | p;
    }
  }
  const int ordbits = nbk > 1 ? 32 - __builtin_clz(nbk - 1) : 0;
  int passes = (lowbits + ordbits + DB - 1) / DB;
  if ((lab & 0xff) > 0 && passes > (lab & 0xff) - 1) passes = (lab & 0xff) - 1;  // lab: at most this many passes (wrong result)
  for (int pass = 0; pass < passes; pass++) {
    const int shift = kRlTagBits + DB * pass;
#pragma unroll
    for (int j = 0; j < NBH; j++) C[rl_pad((uint32_t)j * kBlock + tid)] = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      if (k < R) {
        const uint32_t d = (uint32_t)(comp[k] >> shift) & (uint32_t)((1 << DB) - 1);
        const uint32_t old = atomicAdd(&C[rl_pad((d & (NBH - 1)) * kBlock + tid)], d >= NBH ? 65536u : 1u);
        rank[k] = d >= NBH ? old >> 16 : old & 0xffffu;
      }
    }
    __syncthreads();
    {
      uint32_t w[NBH], sum = 0;
#pragma unroll
      for (int q = 0; q < NBH; q++) {
        w[q] = C[rl_pad((uint32_t)tid * NBH + q)];
        sum += w[q];
      }
      uint32_t total;
      uint32_t run = block_exclusive_add<uint32_t>(sum, shs, total);
      run += total << 16;
#pragma unroll
      for (int q = 0; q < NBH; q++) {
        C[rl_pad((uint32_t)tid * NBH + q)] = run;
        run += w[q];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      if (k < R) {
        const uint32_t d = (uint32_t)(comp[k] >> shift) & (uint32_t)((1 << DB) - 1);
        const uint32_t word = C[rl_pad((d & (NBH - 1)) * kBlock + tid)];
        rank[k] += d >= NBH ? word >> 16 : word & 0xffffu;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; k++)
      if (k < R) scomp[rl_pad(rank[k])] = comp[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; k++)
      if (k < R) comp[k] = scomp[rl_pad(pbase + k)];
    if (pass + 1 < passes) __syncthreads();  // the next pass's counters overwrite the buffer
  }
  // ---- emission.  scomp holds the sorted composites (dead ones, all ones, behind the L live ones); thread t has
  // those of positions pbase .. pbase + R - 1 in registers.
  uint32_t hc = 0, headmask = 0;
  {
    uint64_t prev_id = (pbase > 0 && pbase <= L) ? (scomp[rl_pad(pbase - 1)] >> kRlTagBits) : 0ull;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const uint32_t p = pbase + k;
      const bool live = k < R && p < L;
      const uint64_t id = comp[k] >> kRlTagBits;
      const bool head = live && (p == 0 || id != prev_id);
      prev_id = id;
      hc += head ? 1u : 0u;
      headmask |= head ? (1u << k) : 0u;
      rank[k] = hc;  // inclusive count of run heads in the thread
    }
  }
  uint32_t U;
  const uint32_t uex = block_exclusive_add<uint32_t>(hc, shs, U);
  // wave 0 puts out the chunk's count and requests its predecessors' status words; it looks at them after the fold
  unsigned long long lb_words = 0;
  if (tid < 64 && !(lab & 0x400)) lb_words = rl_publish(status, chunk, U);
  auto rl_key = [&](uint64_t id) -> uint64_t {  // the key a sorted composite (tag shifted out) stands for
    const uint64_t low = id & lowmask;
    return ((uint64_t)top_tab[(uint32_t)(id >> lowbits)] << tb) | ((low >> fb2) << fbits) | ((low & fmask2) + fmin);
  };
  I ocol[ITEMS];
  T oval[ITEMS];
  uint32_t endmask = 0;
  uint64_t orow[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    const uint32_t p = pbase + k;
    ocol[k] = 0;
    oval[k] = (T)0;
    orow[k] = 0;
    if (k < R && p < L) {
      const uint64_t id = comp[k] >> kRlTagBits;
      const uint64_t key = rl_key(id);
      if ((headmask & (1u << k)) && !(lab & 0x800)) {
        auto value_at = [&](uint64_t c) -> T {
          const P pv = spay[rl_pad((uint32_t)c & ((1u << kRlTagBits) - 1u))];
          if constexpr (INDIRECT) return vals_src[pv];
          else return rl_from_bits<T>(pv);
        };
        T acc = value_at(comp[k]);
        for (uint32_t j = p + 1; j < L; j++) {  // the rest of the run, in input order (the sort is stable)
          const uint64_t cj = scomp[rl_pad(j)];
          if ((cj >> kRlTagBits) != id) break;
          acc = (T)(acc + value_at(cj));
        }
        ocol[k] = (I)(key & ((minor_bits >= 64) ? ~0ull : ((1ull << minor_bits) - 1ull)));
        oval[k] = acc;
      }
      // last pair of its row inside the chunk?
      bool rend = p + 1 >= L;
      if (!rend) {
        const uint64_t nid = scomp[rl_pad(p + 1)] >> kRlTagBits;
        const uint64_t nkey = rl_key(nid);
        rend = (minor_bits >= 64 ? 0ull : (nkey >> minor_bits)) != (minor_bits >= 64 ? 0ull : (key >> minor_bits));
      }
      if (rend && !(lab & 0x200)) {
        endmask |= 1u << k;
        orow[k] = minor_bits >= 64 ? 0ull : (key >> minor_bits);
      }
    }
  }
  if (rows_seen != nullptr) {  // (rows lie inside one chunk when this is asked for: every row is counted once)
    const uint32_t ends = wave_reduce((uint32_t)__popc(endmask), OpAdd());
    if ((tid & 63) == 0 && ends) atomicAdd(rows_seen + ((chunk & (kSkipCounters - 1)) * kSkipStride), (unsigned long long)ends);
  }
  if (tid < 64) {  // wave 0: the chunk's first output slot
    bool fail = false;
    const unsigned long long off = (lab & 0x400) ? (unsigned long long)a : rl_finish(status, chunk, U, lb_words, &fail);
    if (tid == 0) {
      if (fail) {
        *err = 1;
        s_fail = 1;
      }
      s_off = off;
      if (!fail && last_chunk) *merged_out = off + U;
    }
  }
  __syncthreads();  // composites and payloads are spent; s_off is there
  if (s_fail) return;
  const unsigned long long off = s_off;
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    if (k < R) {
      const uint32_t slot = uex + rank[k];  // inclusive: runs up to and including this pair's
      if (headmask & (1u << k)) {
        scol[slot - 1] = ocol[k];
        sval[slot - 1] = oval[k];
      }
      if (endmask & (1u << k)) {
        const int64_t v = (int64_t)(off + slot);
        if (row_atomic) atomicMax(reinterpret_cast<unsigned long long *>(row_ptr + orow[k] + 1), (unsigned long long)v);
        else row_ptr[orow[k] + 1] = v;
      }
    }
  }
  __syncthreads();
  if (lab & 0x100) return;
  for (uint32_t u = tid; u < U; u += kBlock) {
    col_out[off + u] = scol[u];
    vals_out[off + u] = sval[u];
  }
}

// Assembly through the fused kernel: global passes on key bits [tb, end_bit), then one launch that sorts the rest
// per chunk and writes the record.  col_out / vals_out hold n slots; *merged_out (device) receives the number of
// stored entries; row_ptr (zeroed by the caller) receives the row ends (the caller's max-scan fills empty rows).
template <class T>
struct RlEmitCfg {
  // two chunk sizes: 2 workgroups per CU for buckets of up to ~2000 pairs, 4 per CU when the buckets are small
  static constexpr int kItems = sizeof(T) > 4 ? 16 : 20;
  static constexpr int kCap = kBlock * kItems;
  static constexpr int kItemsSmall = sizeof(T) > 4 ? 8 : 10;
  static constexpr int kCapSmall = kBlock * kItemsSmall;
};

template <class T, class I, bool INDIRECT>
int msd_sort_emit(futhark_context *ctx, uint64_t *keys_a, uint64_t *keys_b,
                  const typename std::conditional<INDIRECT, uint32_t, typename RlBits<T>::type>::type *pay_in,
                  typename std::conditional<INDIRECT, uint32_t, typename RlBits<T>::type>::type *pay_a,
                  typename std::conditional<INDIRECT, uint32_t, typename RlBits<T>::type>::type *pay_b,
                  const T *vals_src, int64_t n, int tb, int end_bit, uint32_t ch, bool small, int minor_bits, I *col_out, T *vals_out,
                  int64_t *row_ptr, unsigned long long *merged_out, int *sort_err, int *oversize,
                  unsigned long long *rows_seen = nullptr) {  // rows_seen: zeroed by the caller; receives the number of rows that hold something (only when every row lies inside one bucket)
  typedef typename std::conditional<INDIRECT, uint32_t, typename RlBits<T>::type>::type P;
  uint64_t *ks = nullptr;
  const P *ps = nullptr;
  // 8-byte values that travel with their keys: the passes write key and value side by side (radix_sort.hpp: RsPair)
  constexpr bool kPairs = !INDIRECT && sizeof(P) == 8;
  Tmp<RsPair<uint64_t, P>> aos_a(ctx), aos_b(ctx);
  bool aos = false;
  if (kPairs && ctx->opt_sort_pairs >= 0) {
    SAMD_TRY(aos_a.get(n));
    if (end_bit - tb > 8) SAMD_TRY(aos_b.get(n));  // (one pass writes only the first)
    SAMD_TRY((radix_sort_pairs<uint64_t, P>(ctx, keys_a, keys_b, pay_in, pay_a, pay_b, n, tb, end_bit, &ks, &ps, sort_err,
                                            aos_a.p, aos_b.p ? aos_b.p : aos_a.p, &aos)));
  } else {
    SAMD_TRY((radix_sort_pairs<uint64_t, P>(ctx, keys_a, keys_b, pay_in, pay_a, pay_b, n, tb, end_bit, &ks, &ps, sort_err)));
  }
  if (!aos && (ps == nullptr || ps == pay_in)) return set_error(ctx, "internal: msd sort without a global pass");
  const uint32_t nchunks = (uint32_t)cdiv(n, (int64_t)ch);  // nominal; the ones that hold nothing drop out below
  Tmp<uint32_t> ca(ctx), place(ctx), cb(ctx);
  Tmp<unsigned long long> status(ctx);
  SAMD_TRY(ca.get((int64_t)nchunks + 2));
  SAMD_TRY(place.get((int64_t)nchunks + 2));
  SAMD_TRY(cb.get((int64_t)nchunks + 3));
  SAMD_TRY(status.get(nchunks));
  uint32_t *ticket = cb.p + nchunks + 2;
  SAMD_HIP(ctx, hipMemsetAsync(ticket, 0, sizeof(uint32_t), ctx->stream));
  SAMD_HIP(ctx, hipMemsetAsync(status.p, 0, sizeof(unsigned long long) * (size_t)nchunks, ctx->stream));
  hipLaunchKernelGGL((msd_chunk_plan_kernel<uint64_t>), dim3((unsigned)cdiv((int64_t)nchunks + 1, kBlock)), dim3(kBlock), 0,
                     ctx->stream, (const uint64_t *)ks, (uint32_t)n, tb, ch, nchunks, ca.p, aos ? 2 : 1);
  SAMD_LAUNCH_CHECK(ctx);
  // place[c] = chunks kept before c; place[nchunks] = chunks kept
  SAMD_TRY((device_scan<uint32_t, OpAdd, ChunkHoldsIn, true>(ctx, ChunkHoldsIn{ca.p}, nchunks, 0u, OpAdd(), place.p,
                                                             place.p + nchunks)));
  hipLaunchKernelGGL(msd_chunk_compact_kernel, dim3((unsigned)cdiv((int64_t)nchunks + 1, kBlock)), dim3(kBlock), 0,
                     ctx->stream, (const uint32_t *)ca.p, nchunks, (const uint32_t *)place.p, (uint32_t)n, cb.p);
  SAMD_LAUNCH_CHECK(ctx);
  const uint32_t *kept = place.p + nchunks;
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(nchunks), dim3(kBlock), 0, ctx->stream, (const uint64_t *)ks, ps, vals_src,
                       (const uint32_t *)cb.p, kept, tb, minor_bits, col_out, vals_out, row_ptr, tb < minor_bits ? 1 : 0,
                       status.p, ticket, merged_out, oversize, sort_err, (int)ctx->opt_ablate,
                       tb < minor_bits ? (unsigned long long *)nullptr : rows_seen);
  };
  if constexpr (kPairs) {
    if (aos && small) launch(rs_local_sort_emit<T, I, INDIRECT, RlEmitCfg<T>::kItemsSmall, kRlDigitBits, true>);
    else if (aos) launch(rs_local_sort_emit<T, I, INDIRECT, RlEmitCfg<T>::kItems, kRlDigitBits, true>);
  }
  if (!aos && small) launch(rs_local_sort_emit<T, I, INDIRECT, RlEmitCfg<T>::kItemsSmall, kRlDigitBits, false>);
  else if (!aos) launch(rs_local_sort_emit<T, I, INDIRECT, RlEmitCfg<T>::kItems, kRlDigitBits, false>);
  SAMD_LAUNCH_CHECK(ctx);
  return 0;
}

}  // namespace samd
