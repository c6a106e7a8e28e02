// Same question inside ONE wavefront: independent vector instructions placed between MFMAs.
//   mode 1 = MFMAs only, 2 = vector only, 3 = interleaved (1 MFMA, then NV vector instructions)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
template <int MK, int VK, int NV, int WAVES, int mode>
__global__ __launch_bounds__(WAVES * 64) void k(double* out, int iters) {
  v4f64 a64[4]; v4f32 a32[4];
  for (int i = 0; i < 4; i++) { a64[i] = (v4f64){0, 0, 0, 0}; a32[i] = (v4f32){0, 0, 0, 0}; }
  const double a = threadIdx.x * 1e-3 + 1.0, b = 0.999;
  float f[NV]; double d[NV];
  for (int i = 0; i < NV; i++) { f[i] = threadIdx.x * 0.001f + i; d[i] = f[i]; }
  const float cf = 0.9999f; const double cd = 0.9999; const float af = (float)a, bf = (float)b;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (mode & 1) {
        if (MK == 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(a64[i]) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a32[i]) : "v"(af), "v"(bf));
      }
      if (mode & 2) {
#pragma unroll
        for (int j = 0; j < NV; j++) {
          if (VK == 0) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(f[j]) : "v"(cf));
          else asm volatile("v_fma_f64 %0, %0, %1, 0.5" : "+v"(d[j]) : "v"(cd));
        }
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 4; i++) s += a64[i][0] + a32[i][0];
  for (int i = 0; i < NV; i++) s += f[i] + d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MK, int VK, int NV, int WAVES, int mode>
float run() {
  const int blocks = 256, iters = 20000;
  double* d; hipMalloc(&d, sizeof(double) * 64 * WAVES * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MK, VK, NV, WAVES, mode>), dim3(blocks), dim3(64 * WAVES), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MK, VK, NV, WAVES, mode>), dim3(blocks), dim3(64 * WAVES), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  return ms;
}
template <int MK, int VK, int NV, int WAVES>
void report(const char* name) {
  const float m = run<MK, VK, NV, WAVES, 1>(), v = run<MK, VK, NV, WAVES, 2>(), b = run<MK, VK, NV, WAVES, 3>();
  printf("%-44s mfma %6.2f  valu %6.2f  interleaved %6.2f (sum %6.2f)\n", name, m, v, b, m + v);
}
int main() {
  report<0, 0, 8, 4>("f64 mfma + 8 f32 fma, 1 wave/SIMD");
  report<0, 1, 8, 4>("f64 mfma + 8 f64 fma, 1 wave/SIMD");
  report<0, 1, 4, 4>("f64 mfma + 4 f64 fma, 1 wave/SIMD");
  report<1, 0, 6, 4>("f32 mfma + 6 f32 fma, 1 wave/SIMD");
  report<1, 1, 4, 4>("f32 mfma + 4 f64 fma, 1 wave/SIMD");
  report<0, 0, 8, 8>("f64 mfma + 8 f32 fma, 2 waves/SIMD");
  report<0, 1, 8, 8>("f64 mfma + 8 f64 fma, 2 waves/SIMD");
  report<1, 1, 4, 8>("f32 mfma + 4 f64 fma, 2 waves/SIMD");
  return 0;
}
