// bw_lab.hip -- what does this MI355X actually sustain for read-mostly streams?
// (the SpMV roofline denominator is the 8 TB/s spec; this measures the reachable ceiling)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o bench/bw_lab bench/bw_lab.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <functional>
#include <vector>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e = (x);                                                              \
    if (e != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

typedef float float4v __attribute__((ext_vector_type(4)));

// grid-stride read, UNROLL independent 16-B loads in flight per thread
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_gs(const float4v *__restrict__ in, int64_t n16, float *__restrict__ out) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nth = (int64_t)gridDim.x * 256;
  float4v acc = {0, 0, 0, 0};
  int64_t i = tid;
  for (; i + (UNROLL - 1) * nth < n16; i += UNROLL * nth) {
    float4v v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(&in[i + u * nth]) : in[i + u * nth];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc += v[u];
  }
  for (; i < n16; i += nth) acc += in[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[tid] = acc.x;
}

// one tile per block (no loop): block b reads [b*TILE16, (b+1)*TILE16)
template <int PER_THREAD, bool NT>
__global__ __launch_bounds__(256) void read_tile(const float4v *__restrict__ in, int64_t n16, float *__restrict__ out) {
  const int64_t base = (int64_t)blockIdx.x * (256 * PER_THREAD);
  float4v v[PER_THREAD];
#pragma unroll
  for (int u = 0; u < PER_THREAD; u++) {
    const int64_t i = base + u * 256 + threadIdx.x;
    v[u] = i < n16 ? (NT ? __builtin_nontemporal_load(&in[i]) : in[i]) : float4v{0, 0, 0, 0};
  }
  float4v acc = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < PER_THREAD; u++) acc += v[u];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[threadIdx.x] = acc.x;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void copy_gs(const float4v *__restrict__ in, float4v *__restrict__ out, int64_t n16) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nth = (int64_t)gridDim.x * 256;
  int64_t i = tid;
  for (; i + (UNROLL - 1) * nth < n16; i += UNROLL * nth) {
    float4v v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = in[i + u * nth];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) out[i + u * nth] = v[u];
  }
  for (; i < n16; i += nth) out[i] = in[i];
}

__global__ __launch_bounds__(256) void write_gs(float4v *__restrict__ out, int64_t n16) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nth = (int64_t)gridDim.x * 256;
  for (int64_t i = tid; i < n16; i += nth) out[i] = float4v{1, 2, 3, 4};
}


// SpMV-like mix: each block reads a contiguous tile (PER_THREAD x 16 B per thread) and
// writes ROWS doubles (ROWS*8 B) at out + block*ROWS: 8-B stores (one row per lane),
// 16-B stores (two rows per lane), optionally non-temporal.
template <int PER_THREAD, int ROWS, int WIDTH, bool NT, bool NTL = false>
__global__ __launch_bounds__(256) void read_tile_write(const float4v *__restrict__ in, int64_t n16,
                                                       double *__restrict__ out) {
  const int64_t base = (int64_t)blockIdx.x * (256 * PER_THREAD);
  float4v v[PER_THREAD];
#pragma unroll
  for (int u = 0; u < PER_THREAD; u++) {
    const int64_t i = base + u * 256 + threadIdx.x;
    v[u] = i < n16 ? (NTL ? __builtin_nontemporal_load(&in[i]) : in[i]) : float4v{0, 0, 0, 0};
  }
  float4v acc = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < PER_THREAD; u++) acc += v[u];
  const double r = (double)(acc.x + acc.y + acc.z + acc.w);
  double *o = out + (int64_t)blockIdx.x * ROWS;
  if (WIDTH == 8) {
    if (threadIdx.x < ROWS) { if (NT) __builtin_nontemporal_store(r, &o[threadIdx.x]); else o[threadIdx.x] = r; }
  } else {
    typedef double v2d __attribute__((ext_vector_type(2)));
    if (threadIdx.x * 2 + 1 < ROWS) {
      v2d w = {r, r};
      if (NT) __builtin_nontemporal_store(w, reinterpret_cast<v2d *>(&o[threadIdx.x * 2]));
      else *reinterpret_cast<v2d *>(&o[threadIdx.x * 2]) = w;
    }
  }
}

struct V {
  const char *name;
  double bytes;
  std::function<void()> run;
  std::vector<float> ms;
};

int main(int argc, char **argv) {
  const int64_t bytes = (argc > 1 ? atoll(argv[1]) : 1408) * (1ll << 20);
  const int64_t n16 = bytes / 16;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float4v *a, *b;
  float *out;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMalloc(&out, 4 * 1024 * 1024 * 16));
  CK(hipMemsetAsync(a, 1, bytes, st));
  CK(hipMemsetAsync(b, 1, bytes, st));
  std::vector<V> vs;
#define RGS(U, NT, G) vs.push_back({"read grid-stride U=" #U " nt=" #NT " grid=" #G, (double)bytes, [&] { hipLaunchKernelGGL((read_gs<U, NT>), dim3(G), dim3(256), 0, st, a, n16, out); }, {}})
  RGS(1, false, 2048);
  RGS(4, false, 2048);
  RGS(8, false, 2048);
  RGS(4, true, 2048);
  RGS(4, false, 1024);
  RGS(4, false, 4096);
  RGS(4, false, 8192);
  RGS(8, false, 1024);
#define RT(P, NT) vs.push_back({"read one tile/block per-thread=" #P " nt=" #NT, (double)bytes, [&] { hipLaunchKernelGGL((read_tile<P, NT>), dim3((unsigned)((n16 + 256 * P - 1) / (256 * P))), dim3(256), 0, st, a, n16, out); }, {}})
  RT(4, false);
  RT(8, false);
  RT(16, false);
  RT(8, true);
  vs.push_back({"copy grid-stride U=4 grid=2048 (r+w bytes)", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_gs<4>), dim3(2048), dim3(256), 0, st, a, b, n16); }, {}});
  vs.push_back({"copy grid-stride U=4 grid=8192 (r+w bytes)", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_gs<4>), dim3(8192), dim3(256), 0, st, a, b, n16); }, {}});
  vs.push_back({"write grid-stride grid=2048", (double)bytes, [&] { hipLaunchKernelGGL(write_gs, dim3(2048), dim3(256), 0, st, b, n16); }, {}});
  vs.push_back({"hipMemcpyAsync D2D (r+w bytes)", 2.0 * bytes, [&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, st)); }, {}});


  // 12 KB read per block (3 x 16 B x 256 threads) + ROWS*8 B written: the C2 SpMV mix
#define RTW(ROWS, W, NT) vs.push_back({"tile read 12KB + write " #ROWS " rows, " #W "B stores nt=" #NT, (double)bytes + (double)((n16 + 767) / 768) * ROWS * 8, [&] { hipLaunchKernelGGL((read_tile_write<3, ROWS, W, NT>), dim3((unsigned)((n16 + 767) / 768)), dim3(256), 0, st, a, n16, (double *)b); }, {}})
  RTW(205, 8, false);
  RTW(204, 8, false);
  RTW(208, 8, false);
  RTW(256, 8, false);
  RTW(208, 8, true);
  RTW(208, 16, false);
  RTW(208, 16, true);
  RTW(256, 16, false);


#define RTW2(ROWS, W, NT) vs.push_back({"NT-LOAD tile read 12KB + write " #ROWS " rows, " #W "B stores nt=" #NT, (double)bytes + (double)((n16 + 767) / 768) * ROWS * 8, [&] { hipLaunchKernelGGL((read_tile_write<3, ROWS, W, NT, true>), dim3((unsigned)((n16 + 767) / 768)), dim3(256), 0, st, a, n16, (double *)b); }, {}})
  RTW2(205, 8, false);
  RTW2(205, 8, true);
  RTW2(208, 16, true);
  RT(3, false);
  RT(3, true);
  // reads only, but ROWS=0 through the same kernel
  vs.push_back({"same kernel, no rows written", (double)bytes, [&] { hipLaunchKernelGGL((read_tile_write<3, 0, 8, false>), dim3((unsigned)((n16 + 767) / 768)), dim3(256), 0, st, a, n16, (double *)b); }, {}});

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int rounds = 5, launches = 10;
  for (int r = 0; r < rounds; r++)
    for (auto &v : vs) {
      v.run();
      CK(hipEventRecord(e0, st));
      for (int l = 0; l < launches; l++) v.run();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      v.ms.push_back(ms / launches);
    }
  printf("buffer %.0f MiB\n", bytes / 1048576.0);
  for (auto &v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    const double med = v.ms[v.ms.size() / 2];
    printf("%-52s %8.2f us  %7.1f GB/s\n", v.name, med * 1e3, v.bytes / (med * 1e-3) / 1e9);
  }
  return 0;
}
