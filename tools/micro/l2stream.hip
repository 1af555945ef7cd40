// dev micro-benchmark: how fast can ONE workgroup per CU stream a small L2-resident buffer (the WaveNet's 3.2 MB of weights) that
// every workgroup reads in the same order?  Sets the floor of any kernel whose workgroups each need all the weights.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/l2stream.hip -o gpurun_out/l2stream && gpurun_out/l2stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int U, int LDSDMA>
__global__ __launch_bounds__(512) void stream_kernel(const uint4* __restrict__ w, size_t n16, uint4* out)
{
  // every wave reads a different 1/nwaves of each U-KB block, fragments of 1 KB (64 lanes x 16 B), like the MFMA-A fragment stream
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const size_t frags = n16 / 64;
  for (size_t f = wave * U; f + U <= frags; f += (size_t)nw * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = w[(f + u) * 64 + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
  }
  if (acc.x == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int U>
float run(const uint4* w, size_t bytes, int nwg, int threads, uint4* out, int reps)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<U, 0>), dim3(nwg), dim3(threads), 0, 0, w, bytes / 16, out);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<U, 0>), dim3(nwg), dim3(threads), 0, 0, w, bytes / 16, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

int main()
{
  const size_t bytes = 3276800;   // 3.2 MB
  uint4 *w, *out;
  hipMalloc(&w, bytes); hipMalloc(&out, 1 << 22);
  hipMemset(w, 1, bytes);
  for (int nwg : {32, 128, 180, 256, 512})
    for (int threads : {256, 512}) {
      float t8 = run<8>(w, bytes, nwg, threads, out, 20), t16 = run<16>(w, bytes, nwg, threads, out, 20), t32 = run<32>(w, bytes, nwg, threads, out, 20);
      printf("wgs %4d x %3d thr: U=8 %6.1f us (%5.1f GB/s/WG)  U=16 %6.1f us (%5.1f)  U=32 %6.1f us (%5.1f)\n", nwg, threads,
             t8, bytes / t8 * 1e-3, t16, bytes / t16 * 1e-3, t32, bytes / t32 * 1e-3);
    }
  return 0;
}
