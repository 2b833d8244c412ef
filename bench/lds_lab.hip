// lds_lab.hip -- kernel lab: what an LDS accumulation costs on MI355X.  One 1024-thread workgroup per CU
// (128 KB slice of floats, as spmv_pb's pass 2), every lane adds ITER values into the slice at
//   pattern 0: random addresses (distinct within a wavefront with high probability)
//   pattern 1: ascending addresses with random gaps (a piece sorted by row)
//   pattern 2: the same address for the whole wavefront
// with  op 0: ds_add_f32 (atomicAdd float)   op 1: ds_add_u32 (atomicAdd int)   op 2: plain read-add-write
//       op 3: ds_add_f64 (atomicAdd double, 16 K slice)
// Build: hipcc -O3 --offload-arch=gfx950 -o bench/lds_lab bench/lds_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kThreads = 1024;
constexpr int kSlice = 32768;

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int OP>
__global__ __launch_bounds__(kThreads) void lab(float *out, int iters, int pattern) {
  __shared__ float ys[kSlice];
  for (int j = threadIdx.x; j < kSlice; j += kThreads) ys[j] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t s = mix(blockIdx.x * 4096u + threadIdx.x + 1u);
  uint32_t base = s & (kSlice - 1);
  for (int it = 0; it < iters; it++) {
    s = mix(s + it);
    uint32_t a;
    if (pattern == 0) a = s & (kSlice - 1);
    else if (pattern == 1) { base = (base + 64 * 61) & (kSlice - 1); a = (base + lane * 61 + (s & 31)) & (kSlice - 1); }
    else a = __shfl((int)(s & (kSlice - 1)), 0, 64);
    const float v = (float)(s >> 8) * 1e-9f;
    if (OP == 0) atomicAdd(&ys[a], v);
    else if (OP == 1) atomicAdd(reinterpret_cast<int *>(ys) + a, (int)(s >> 20));
    else if (OP == 2) ys[a] = ys[a] + v;
    else atomicAdd(reinterpret_cast<double *>(ys) + (a >> 1), (double)v);
  }
  __syncthreads();
  float acc = 0.f;
  for (int j = threadIdx.x; j < kSlice; j += kThreads) acc += ys[j];
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  float *out;
  CK(hipMalloc(&out, 4));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate / 1e6;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 4096;
  const char *opn[4] = {"ds_add_f32", "ds_add_u32", "read-add-write", "ds_add_f64"};
  const char *pn[3] = {"random", "ascending", "same address"};
  for (int op = 0; op < 4; op++)
    for (int pat = 0; pat < 3; pat++) {
      auto launch = [&]() {
        if (op == 0) hipLaunchKernelGGL(lab<0>, dim3(cus), dim3(kThreads), 0, 0, out, iters, pat);
        else if (op == 1) hipLaunchKernelGGL(lab<1>, dim3(cus), dim3(kThreads), 0, 0, out, iters, pat);
        else if (op == 2) hipLaunchKernelGGL(lab<2>, dim3(cus), dim3(kThreads), 0, 0, out, iters, pat);
        else hipLaunchKernelGGL(lab<3>, dim3(cus), dim3(kThreads), 0, 0, out, iters, pat);
      };
      launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double lanes = (double)iters * kThreads;  // per CU
      printf("%-15s %-13s %8.3f ms  %6.2f cycles per lane-op per CU (at %.2f GHz)  %7.1f G ops/s chip\n", opn[op], pn[pat], ms,
             ms * 1e-3 * ghz * 1e9 / lanes, ghz, lanes * cus / (ms * 1e-3) / 1e9);
    }
  return 0;
}
