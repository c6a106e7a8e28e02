// Sustained v_mfma_f64_16x16x4_f64 rate as a function of resident wavefronts per SIMD and of
// independent accumulators per wavefront (sets the occupancy targets of the contraction kernels).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, int iters, int lds_pad) {
  extern __shared__ double pad[];
  v4f64 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (v4f64){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3 + 1.0, b = 0.999;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (lds_pad < 0) pad[threadIdx.x] = s;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int waves_per_simd, int mult = 1) {
  // one 256-thread block = 1 wave per SIMD of a CU; LDS sized so exactly waves_per_simd blocks fit a CU
  const int blocks = 256 * waves_per_simd, iters = mult * 60000 / NACC;
  size_t lds = (160 * 1024) / waves_per_simd - 512;
  if (lds > 65536) hipFuncSetAttribute((const void*)k<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  double* d; hipMalloc(&d, sizeof(double) * 256 * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), lds, 0, d, 10, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), lds, 0, d, iters, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("NACC=%2d waves/SIMD=%d: %7.2f ms %6.1f TFLOP/s\n", NACC, waves_per_simd, ms,
         (double)blocks * 4 * iters * NACC * 2048.0 / ms / 1e9);
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 3, 4, 6, 8}) run<3>(w);
  for (int w : {1, 2, 3, 4}) run<12>(w);
  // duration sweep: the same loop run 1x .. 32x longer (sustained rate under the power cap)
  for (int m : {1, 2, 4, 8, 16, 32}) run<12>(4, m);
  for (int m : {1, 4, 16}) run<12>(1, m);
  return 0;
}
