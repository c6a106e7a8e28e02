// Do matrix and vector instructions of DIFFERENT wavefronts on one SIMD run side by side?
// A 512-thread workgroup (2 wavefronts per SIMD): wavefronts 0-3 issue a chain-free MFMA stream,
// wavefronts 4-7 a vector stream; each stream is also timed with the other half idle.
//   mode: bit0 = MFMA half active, bit1 = VALU half active
//   MK: 0 = v_mfma_f64_16x16x4_f64, 1 = v_mfma_f32_16x16x4_f32;  VK: 0 = f32 fma, 1 = f64 fma, 2 = int mad
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
template <int MK, int VK>
__global__ __launch_bounds__(512) void k(double* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  double s = 0;
  if (wave < 4) {
    if (mode & 1) {
      v4f64 a64[6]; v4f32 a32[6];
      for (int i = 0; i < 6; i++) { a64[i] = (v4f64){0, 0, 0, 0}; a32[i] = (v4f32){0, 0, 0, 0}; }
      const double a = threadIdx.x * 1e-3 + 1.0, b = 0.999;
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 6; i++) {
          if (MK == 0) a64[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a64[i], 0, 0, 0);
          else a32[i] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a, (float)b, a32[i], 0, 0, 0);
        }
      }
      for (int i = 0; i < 6; i++) s += a64[i][0] + a32[i][0];
    }
  } else if (mode & 2) {
    float f[8]; double d[8]; int n[8];
    for (int i = 0; i < 8; i++) { f[i] = threadIdx.x * 0.001f + i; d[i] = f[i]; n[i] = threadIdx.x + i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
          if (VK == 0) f[i] = fmaf(f[i], 0.9999f, 0.5f);
          else if (VK == 1) d[i] = fma(d[i], 0.9999, 0.5);
          else n[i] = n[i] * 3 + 7;
        }
    }
    for (int i = 0; i < 8; i++) s += f[i] + d[i] + n[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MK, int VK>
float run(int mode) {
  const int blocks = 256, iters = 20000;
  double* d; hipMalloc(&d, sizeof(double) * 512 * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MK, VK>), dim3(blocks), dim3(512), 0, 0, d, 10, mode);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MK, VK>), dim3(blocks), dim3(512), 0, 0, d, iters, mode);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  return ms;
}
template <int MK, int VK>
void report(const char* name) {
  const float m = run<MK, VK>(1), v = run<MK, VK>(2), b = run<MK, VK>(3);
  printf("%-34s mfma alone %6.2f ms, valu alone %6.2f ms, together %6.2f ms (sum %6.2f, max %6.2f)\n", name, m, v, b, m + v,
         m > v ? m : v);
}
int main() {
  // per iteration and wavefront: 6 MFMAs (f64: 384 cycles, f32: 192) against 48 vector instructions
  report<0, 0>("mfma f64 + f32 fma");
  report<0, 1>("mfma f64 + f64 fma");
  report<0, 2>("mfma f64 + int mad");
  report<1, 0>("mfma f32 + f32 fma");
  report<1, 1>("mfma f32 + f64 fma");
  report<1, 2>("mfma f32 + int mad");
  return 0;
}
