// How fast can ONE compute unit stream from HBM / L2?  Each workgroup reads its own contiguous chunk with D independent
// 16-B loads per thread in flight (unrolled), varying the number of workgroups (1 .. 1024), threads and D.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o tools/probes/stream_probe && tools/probes/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int D>
__global__ void stream_kernel(const float4* __restrict__ src, float* __restrict__ out, long per_wg_vec, int reps) {
  const float4* p = src + (long)blockIdx.x * per_wg_vec;
  float acc = 0.f;
  for (int r = 0; r < reps; ++r) {
    for (long i = threadIdx.x; i + (long)(D - 1) * blockDim.x < per_wg_vec; i += (long)D * blockDim.x) {
      float4 v[D];
#pragma unroll
      for (int d = 0; d < D; ++d) v[d] = p[i + (long)d * blockDim.x];
#pragma unroll
      for (int d = 0; d < D; ++d) acc += v[d].x + v[d].y + v[d].z + v[d].w;
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}
template <int D>
float run(const float4* src, float* out, int wgs, int threads, long per_wg_bytes, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const long vec = per_wg_bytes / 16;
  stream_kernel<D><<<wgs, threads>>>(src, out, vec, reps);
  hipDeviceSynchronize();
  hipEventRecord(a);
  stream_kernel<D><<<wgs, threads>>>(src, out, vec, reps);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  const long total = 2L << 30;  // 2 GB
  float4* src; float* out;
  hipMalloc(&src, total); hipMalloc(&out, 64);
  hipMemset(src, 0, total);
  printf("%6s %7s %3s %10s | %9s %12s\n", "wgs", "threads", "D", "MB/wg", "us", "GB/s per wg");
  for (int wgs : {1, 8, 64, 256, 512, 1024}) {
    for (int threads : {256, 512}) {
      const long per = wgs <= 64 ? (4L << 20) : (1L << 20);  // distinct data per workgroup, beyond the caches in total
      float ms;
      for (int D : {1, 4, 8, 16}) {
        if (D == 1) ms = run<1>(src, out, wgs, threads, per, 1);
        else if (D == 4) ms = run<4>(src, out, wgs, threads, per, 1);
        else if (D == 8) ms = run<8>(src, out, wgs, threads, per, 1);
        else ms = run<16>(src, out, wgs, threads, per, 1);
        printf("%6d %7d %3d %10.1f | %9.1f %12.1f   (total %.0f GB/s)\n", wgs, threads, D, per / 1e6, ms * 1e3,
               per / (ms * 1e-3) / 1e9, per * (double)wgs / (ms * 1e-3) / 1e9);
      }
    }
  }
  return 0;
}
