// Microbenchmark 2: the recursion kernel's memory behaviour, feature by feature.  One wavefront per stream
// (utterance), 12 per workgroup, non-persistent grid; per step 25 row loads of 384 B (consecutive rows), then
//   bit 0: one 384-byte store per step into a second array [stream][step][48]
//   bit 1: two more such stores into a third and fourth array
//   bit 2: one wave-uniform 8-byte store per step ([stream][step])
//   bit 3: a 200-byte "row maxima" load per step from a [stream][rows] array
//   bit 4: the wait for the loads is preceded by ~2.7 us of s_sleep
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k_fill(double* p, size_t n, int mode) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long x = i * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    p[i] = mode ? (double)(x >> 11) * (1.0 / 9007199254740992.0) : 0.0;
  }
}

__global__ __launch_bounds__(768) void k(const double* __restrict__ buf, size_t region_rows, int steps, int n_streams,
                                         double* __restrict__ o1, double* __restrict__ o2, double* __restrict__ o3,
                                         double* __restrict__ sc, const double* __restrict__ smx, int feat) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int strm = blockIdx.x * (blockDim.x >> 6) + wave;
  if (strm >= n_streams) return;
  const double* p = buf + (size_t)strm * region_rows * 48;
  const int lc = lane < 48 ? lane : 47;
  double a = 0.0;
  for (int t = 0; t < steps; t++) {
    double es[25];
#pragma unroll
    for (int d = 0; d < 25; d++) es[d] = p[((size_t)t * 25 + d) * 48 + lc];
    double m = 0.0;
    if (feat & 8) m = smx[(size_t)strm * region_rows + (size_t)t * 25 + (lane < 25 ? lane : 0)];
    if (feat & 16) __builtin_amdgcn_s_sleep(100);
    double q = m;
#pragma unroll
    for (int d = 0; d < 25; d++) q += es[d];
    a = a * 0.5 + q;
    const size_t vo = ((size_t)strm * steps + t) * 48 + lc;
    if (feat & 1) o1[vo] = a;
    if (feat & 2) { o2[vo] = a + 1.0; o3[vo] = a + 2.0; }
    if (feat & 4) sc[(size_t)strm * steps + t] = __shfl(a, 0);
  }
  if (a == 12345.678) sc[0] = a;
}

int main(int argc, char** argv) {
  const int n_streams = 8192, steps = 288;
  const size_t region_rows = 7200;
  const size_t bytes = (size_t)n_streams * region_rows * 48 * 8;
  const size_t vbytes = (size_t)n_streams * steps * 48 * 8;
  double *buf, *o1, *o2, *o3, *sc, *smx;
  if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
  hipMalloc(&o1, vbytes); hipMalloc(&o2, vbytes); hipMalloc(&o3, vbytes);
  hipMalloc(&sc, (size_t)n_streams * steps * 8);
  hipMalloc(&smx, (size_t)n_streams * region_rows * 8);
  hipMemset(buf, 0, bytes);
  hipMemset(smx, 0, (size_t)n_streams * region_rows * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb = (double)n_streams * steps * 25 * 384 / 1e9;
  for (int fill : {0, 1})
  for (int waves : {12, 8}) {
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, bytes / 8, fill);
    hipDeviceSynchronize();
    printf("-- array filled with %s\n", fill ? "random doubles in [0,1)" : "zeros");
    for (int feat : {0, 8}) {
      const int blocks = (n_streams + waves - 1) / waves;
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(waves * 64), 0, 0, buf, region_rows, steps, n_streams, o1, o2, o3, sc, smx, feat);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double rounds = (double)blocks / 256.0;
      printf("waves/wg %2d feat %2d: %7.3f ms  %6.0f GB/s (loads only)  %.2f rounds, %.2f us per step at ceil(rounds)\n", waves, feat, best,
             gb / best * 1e3, rounds, best * 1e3 / (steps * (double)(int)(rounds + 0.999)));
    }
  }
  return 0;
}
