// bwprobe.hpp -- what this card sustains for a streaming read/write mix, measured in the caller's process
// (sparse_amd_context_measure_bandwidth).  No counterpart in the reference: it exists so that a roofline fraction
// can be quoted against the measured ceiling of the kernel's own access mix next to the 8 TB/s of the data sheet
// ("% of both").  No arithmetic: a workgroup reads `rd` contiguous bytes with 128-bit
// non-temporal loads (the way the SpMV / SpMM kernels stream a tile) and writes `wr` contiguous bytes with
// non-temporal stores; rd = 0 is a pure write, wr = 0 a pure read.
#pragma once

#include "spmv.hpp"

namespace samd {

__global__ __launch_bounds__(kBlock) void bw_mix_kernel(const raw16 *__restrict__ in, raw16 *__restrict__ out, int rd16,
                                                        int wr16, unsigned *__restrict__ sink) {
  const int tid = threadIdx.x;
  const raw16 *ip = in + (int64_t)blockIdx.x * rd16;
  raw16 acc = {0u, 0u, 0u, 0u};
  int p = tid;
  for (; p + 3 * kBlock < rd16; p += 4 * kBlock) {
    raw16 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = ntload(ip + p + u * kBlock);
#pragma unroll
    for (int u = 0; u < 4; u++) acc ^= v[u];
  }
  for (; p < rd16; p += kBlock) acc ^= ntload(ip + p);
  raw16 *op = out + (int64_t)blockIdx.x * wr16;
  for (int q = tid; q < wr16; q += kBlock) __builtin_nontemporal_store(acc, op + q);
  if (wr16 == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[0] = acc.x;  // keeps the loads alive
}

// Median over `rounds` timings of `launches` back-to-back launches each; GB/s of (rd + wr) bytes per workgroup.
inline int measure_bandwidth(futhark_context *ctx, int64_t rd_bytes_per_group, int64_t wr_bytes_per_group, int64_t groups,
                             double *gbs, double *us) {
  if (rd_bytes_per_group < 0 || wr_bytes_per_group < 0 || groups <= 0 || groups > 0x7fffffffLL ||
      (rd_bytes_per_group % 16) || (wr_bytes_per_group % 16) || rd_bytes_per_group + wr_bytes_per_group == 0 ||
      rd_bytes_per_group > (1 << 24) || wr_bytes_per_group > (1 << 24))
    return set_error(ctx, "measure_bandwidth: bytes per workgroup must be multiples of 16 below 16 MiB, not both 0; groups in 1..2^31");
  Tmp<raw16> in(ctx), out(ctx);
  Tmp<unsigned> sink(ctx);
  const int64_t rd16 = rd_bytes_per_group / 16, wr16 = wr_bytes_per_group / 16;
  SAMD_TRY(in.get(rd16 * groups));
  SAMD_TRY(out.get(wr16 * groups));
  SAMD_TRY(sink.get(1));
  if (rd16) SAMD_HIP(ctx, hipMemsetAsync(in.p, 1, (size_t)(rd16 * groups) * 16, ctx->stream));
  hipEvent_t e0, e1;
  SAMD_HIP(ctx, hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return set_error(ctx, "measure_bandwidth: hipEventCreate failed"); }
  const int rounds = 7, launches = 5;
  float ms[rounds];
  int rc = 0;
  for (int r = 0; r < rounds + 1 && rc == 0; r++) {  // round 0 warms up
    hipError_t e = hipEventRecord(e0, ctx->stream);
    for (int l = 0; l < launches; l++)
      hipLaunchKernelGGL(bw_mix_kernel, dim3((unsigned)groups), dim3(kBlock), 0, ctx->stream, (const raw16 *)in.p, out.p,
                         (int)rd16, (int)wr16, sink.p);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float t = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
    if (e != hipSuccess) { (void)hipGetLastError(); rc = set_error(ctx, "measure_bandwidth: %s", hipGetErrorString(e)); }
    if (r > 0) ms[r - 1] = t / launches;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc) return rc;
  for (int i = 1; i < rounds; i++)  // insertion sort, median
    for (int j = i; j > 0 && ms[j] < ms[j - 1]; j--) { const float t = ms[j]; ms[j] = ms[j - 1]; ms[j - 1] = t; }
  const double med = ms[rounds / 2];
  if (us) *us = med * 1e3;
  if (gbs) *gbs = (double)(rd_bytes_per_group + wr_bytes_per_group) * (double)groups / (med * 1e-3) / 1e9;
  return 0;
}

}  // namespace samd
