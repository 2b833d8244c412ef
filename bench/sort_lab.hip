// sort_lab.hip -- the assembly's radix sort beside rocPRIM's on the same keys (calibration only;
// rocPRIM is a comparator here, the library does not use it).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 bench/sort_lab.hip -o bench/sort_lab
//   bench/sort_lab [count] [key_bits] [rounds] [ablate]
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cstdio>
#include <vector>

#include "../sparse_amd/csrc/radix_sort.hpp"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);  \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

using namespace samd;

template <class K>
__global__ void gen_keys(K *k, uint32_t n, int bits) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t z = (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  k[i] = (K)(bits >= 64 ? z : (z & ((1ull << bits) - 1)));
}

template <class K>
__global__ void check_sorted(const K *k, const uint32_t *p, const K *orig, uint32_t n, int *bad) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (orig[p[i]] != k[i]) atomicAdd(bad, 1);
  if (i + 1 < n && (k[i] > k[i + 1] || (k[i] == k[i + 1] && p[i] > p[i + 1]))) atomicAdd(bad, 1);
}

template <class K>
void run(int64_t n, int bits, int rounds, long ablate) {
  futhark_context ctx;
  ctx.opt_ablate = ablate;
  CK(hipStreamCreate(&ctx.stream));
  ctx.own_stream = ctx.stream;
  CK(hipHostMalloc((void **)&ctx.h_small, 128));
  CK(hipMalloc((void **)&ctx.d_small, 128));
  K *orig, *ka, *kb, *kout;
  uint32_t *pa, *pb, *iota;
  int *bad;
  CK(hipMalloc(&orig, sizeof(K) * n));
  CK(hipMalloc(&ka, sizeof(K) * n));
  CK(hipMalloc(&kb, sizeof(K) * n));
  CK(hipMalloc(&pa, 4 * n));
  CK(hipMalloc(&pb, 4 * n));
  CK(hipMalloc(&iota, 4 * n));
  CK(hipMalloc(&bad, 4));
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(gen_keys<K>, dim3(grid), dim3(256), 0, ctx.stream, orig, (uint32_t)n, bits);
  {
    std::vector<uint32_t> h(n);
    for (int64_t i = 0; i < n; i++) h[i] = (uint32_t)i;
    CK(hipMemcpy(iota, h.data(), 4 * n, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // ---- ours: algo 1 (histogram pass per digit) then algo 0 (one-sweep)
  std::vector<float> ours, theirs, classic;
  CK(hipMemset(ctx.d_small, 0, 128));
  for (int algo = 1; algo >= 0; algo--) {
  ctx.opt_sort_algo = algo;
  ours.clear();
  for (int r = 0; r < rounds; r++) {
    CK(hipMemcpyAsync(ka, orig, sizeof(K) * n, hipMemcpyDeviceToDevice, ctx.stream));
    CK(hipEventRecord(e0, ctx.stream));
    const uint32_t *pout;
    if (radix_sort_pairs<K, uint32_t>(&ctx, ka, kb, nullptr, pa, pb, n, 0, bits, &kout, &pout, (int *)ctx.d_small) != 0) {
      fprintf(stderr, "ours failed: %s\n", ctx.err.c_str());
      exit(1);
    }
    CK(hipEventRecord(e1, ctx.stream));
    CK(hipStreamSynchronize(ctx.stream));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ours.push_back(ms);
    if (r == 0) {
      CK(hipMemsetAsync(bad, 0, 4, ctx.stream));
      hipLaunchKernelGGL(check_sorted<K>, dim3(grid), dim3(256), 0, ctx.stream, (const K *)kout, pout, (const K *)orig,
                         (uint32_t)n, bad);
      int hb;
      CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      int he;
      CK(hipMemcpy(&he, ctx.d_small, 4, hipMemcpyDeviceToHost));
      printf("ours algo %d: %s%s\n", algo, hb ? "WRONG" : "sorted, stable", he ? " LOOK-BACK TIMEOUT" : "");
    }
  }
  if (algo == 1) classic = ours;
  }
  // ---- rocPRIM
  size_t tmp_bytes = 0;
  void *tmp = nullptr;
  CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const K *)orig, kb, (const uint32_t *)iota, pb, (size_t)n, 0u,
                               (unsigned)bits, ctx.stream));
  CK(hipMalloc(&tmp, tmp_bytes));
  for (int r = 0; r < rounds; r++) {
    CK(hipEventRecord(e0, ctx.stream));
    CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, (const K *)orig, kb, (const uint32_t *)iota, pb, (size_t)n, 0u,
                                 (unsigned)bits, ctx.stream));
    CK(hipEventRecord(e1, ctx.stream));
    CK(hipStreamSynchronize(ctx.stream));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    theirs.push_back(ms);
    if (r == 0) {
      CK(hipMemsetAsync(bad, 0, 4, ctx.stream));
      hipLaunchKernelGGL(check_sorted<K>, dim3(grid), dim3(256), 0, ctx.stream, (const K *)kb, (const uint32_t *)pb,
                         (const K *)orig, (uint32_t)n, bad);
      int hb;
      CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      printf("rocprim: %s (temp %zu MB)\n", hb ? "WRONG" : "sorted, stable", tmp_bytes >> 20);
    }
  }
  std::sort(ours.begin(), ours.end());
  std::sort(classic.begin(), classic.end());
  printf("classic %.3f ms   ", classic[classic.size() / 2]);
  std::sort(theirs.begin(), theirs.end());
  printf("n=%lld key=%zuB bits=%d  ours %.3f ms (%.2f Gpairs/s)   rocprim %.3f ms (%.2f Gpairs/s)\n", (long long)n,
         sizeof(K), bits, ours[ours.size() / 2], n / ours[ours.size() / 2] / 1e6, theirs[theirs.size() / 2],
         n / theirs[theirs.size() / 2] / 1e6);
}

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 83869696;
  const int bits = argc > 2 ? atoi(argv[2]) : 48;
  const int rounds = argc > 3 ? atoi(argv[3]) : 5;
  const long ablate = argc > 4 ? atol(argv[4]) : 0;  // the library's lab switches (0x10000: tiles in ticket order)
  if (bits <= 32) run<uint32_t>(n, bits, rounds, ablate);
  else run<uint64_t>(n, bits, rounds, ablate);
  return 0;
}
