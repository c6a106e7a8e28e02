// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on this chip (the MFMA roofline the
// contraction kernels are priced against).  hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  v4f64 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (v4f64){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wpb_blocks, int iters) {
  double* d;
  hipMalloc(&d, sizeof(double) * 256 * wpb_blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(wpb_blocks), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(wpb_blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)wpb_blocks * 4 * iters * NACC * 2048.0;
  printf("NACC=%d blocks=%d: %.2f ms, %.1f TFLOP/s\n", NACC, wpb_blocks, ms, flop / ms / 1e9);
  hipFree(d);
}
int main() {
  run<12>(256, 20000);
  run<12>(512, 20000);
  run<12>(1024, 10000);
  run<4>(1024, 30000);
  run<1>(1024, 60000);
  return 0;
}
