// What does one wavefront per SIMD sustain in a k-loop shaped like fe_mfma_tile (scrf_fused.hip): per k-step 7
// v_mfma_f64_16x16x4_f64 fed by 3 ds_read_b64 (A) and 7 ds_read_b32 + v_cvt_f64_f32 (B)?  Variants strip the
// conversions and / or the LDS reads to price them.  Prints cycles per MFMA (64 = the pipe's rate).
//   mode bit0: B operands converted from float (else stored as doubles);  bit1: operands from LDS (else registers)
//   waves per SIMD: 1 or 2 (blockDim 256 or 512)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define NKS 19
#define XS 144
template <int MODE>
__global__ void k(double* out, int tiles) {
  __shared__ float Xs[(4 * NKS) * XS];
  __shared__ double Xd[(4 * NKS) * 80];
  __shared__ double Rs[(4 * NKS) * 48];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  for (int i = tid; i < 4 * NKS * XS; i += blockDim.x) Xs[i] = 1.0f + i * 1e-6f;
  for (int i = tid; i < 4 * NKS * 80; i += blockDim.x) Xd[i] = 1.0 + i * 1e-6;
  for (int i = tid; i < 4 * NKS * 48; i += blockDim.x) Rs[i] = 0.5 + i * 1e-7;
  __syncthreads();
  v4f64 acc[7];
  for (int j = 0; j < 7; j++) acc[j] = (v4f64){0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; t++) {
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
      double a[3], b[7];
#pragma unroll
      for (int i = 0; i < 3; i++) a[i] = (MODE & 2) ? Rs[(ks * 4 + lk) * 48 + i * 16 + li] : (double)(lane + i + ks);
#pragma unroll
      for (int j = 0; j < 7; j++) {
        if (MODE & 2) b[j] = (MODE & 1) ? (double)Xs[(ks * 4 + lk) * XS + li + ((wave & 3) + j) * 16] : Xd[(ks * 4 + lk) * 80 + li + (j % 4) * 16];
        else { float f = (float)(lane * 3 + j + ks + t); asm volatile("" : "+v"(f)); b[j] = (MODE & 1) ? (double)f : (double)(lane + j); }
      }
#pragma unroll
      for (int j = 0; j < 7; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j % 3], b[j], acc[j], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int j = 0; j < 7; j++) s += acc[j][0] + acc[j][3];
  out[blockIdx.x * blockDim.x + tid] = s + (double)(t1 - t0);
  if (tid == 0) out[gridDim.x * blockDim.x + blockIdx.x] = (double)(t1 - t0);
}
template <int MODE>
static void run(const char* what, int waves) {
  const int blocks = 256, tiles = 400;
  double* d;
  hipMalloc(&d, sizeof(double) * (blocks * 64 * waves * 4 + blocks));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256 * waves), 0, 0, d, tiles);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256 * waves), 0, 0, d, tiles);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ticks;
  hipMemcpy(&ticks, d + blocks * 256 * waves, sizeof(double), hipMemcpyDeviceToHost);
  const double nm = (double)tiles * NKS * 7;
  printf("%-44s %d wave(s)/SIMD: %7.1f s_memtime ticks per MFMA per wave, %6.2f ms, %5.1f TFLOP/s\n", what, waves, ticks / nm, ms,
         blocks * 4.0 * waves * nm * 2048 / (ms * 1e-3) / 1e12);
  hipFree(d);
}
int main() {
  for (int w = 1; w <= 2; w++) {
    run<0>("registers, no conversion", w);
    run<1>("registers, v_cvt_f64_f32 per B operand", w);
    run<2>("LDS operands (B as doubles)", w);
    run<3>("LDS operands, B floats + v_cvt_f64_f32", w);
  }
  return 0;
}
