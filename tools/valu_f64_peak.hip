// Microbenchmark: sustained v_fma_f64 rate, alone and next to f64 MFMA wavefronts on the same CUs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
// mode 0: all waves VALU; 1: all waves MFMA; 2: even waves MFMA, odd waves VALU
__global__ __launch_bounds__(512) void k(double* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 1 || (mode == 2 && (wave & 1) == 0);
  double a = threadIdx.x * 1e-3 + 1.0, b = 0.999999;
  double s = 0;
  if (do_mfma) {
    v4f64 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (v4f64){0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double x[16];
    for (int i = 0; i < 16; i++) x[i] = a + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = fma(x[i], b, a);
    }
    for (int i = 0; i < 16; i++) s += x[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  double* d;
  hipMalloc(&d, sizeof(double) * 512 * 2048);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; mode++) {
    const int blocks = 1024, iters = 4000;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, 10, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, iters, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)blocks * 8;
    double f_valu = (mode == 1 ? 0 : (mode == 2 ? 0.5 : 1.0)) * waves * iters * 128.0 * 64 * 2;
    double f_mfma = (mode == 0 ? 0 : (mode == 2 ? 0.5 : 1.0)) * waves * iters * 8 * 2048.0;
    printf("mode %d: %.2f ms  VALU %.1f TF  MFMA %.1f TF  total %.1f TF\n", mode, ms, f_valu / ms / 1e9,
           f_mfma / ms / 1e9, (f_valu + f_mfma) / ms / 1e9);
  }
  return 0;
}
