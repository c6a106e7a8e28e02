// Microbenchmark: HBM read bandwidth of the recursion kernels' access pattern -- many concurrent wavefront-private
// streams over a large array of [rows][48] doubles (384-byte rows), 25 rows per step.
//   mode 0: 25 consecutive rows per step (the forward sweep, 9.6 KB contiguous per wavefront and step)
//   mode 1: 25 rows at a stride of 26 rows (the pull-form backward sweep)
//   mode 2: consecutive rows, 16 bytes per lane (dwordx4; 24 lanes cover a row, 2 rows + 2/3 per instruction)
//   mode 3: whole-wave 1 KB loads over the same 9.6 KB (64 lanes x 16 B, contiguous): 9.4 instructions per step
// `waves` wavefronts per workgroup, one workgroup per CU-slot; `ahead` = steps of loads kept in flight.
// Streams are `region` rows apart (utterance stride).  Output: GB/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE, int AHEAD>
__global__ __launch_bounds__(768) void k(const double* __restrict__ buf, size_t region_rows, int steps, int n_streams, double* out, int delay) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  double s = 0.0;
  for (int strm = blockIdx.x * nw + wave; strm < n_streams; strm += gridDim.x * nw) {
    const double* p = buf + (size_t)strm * region_rows * 48;
    if (MODE == 0 || MODE == 1) {
      const int lc = lane < 48 ? lane : 47;
      double cur[AHEAD][25];
#pragma unroll
      for (int a = 0; a < AHEAD; a++)
#pragma unroll
        for (int d = 0; d < 25; d++) cur[a][d] = 0.0;
      for (int t = 0; t < steps + AHEAD; t++) {
        double nxt[25];
        if (t < steps) {
          const size_t base = (size_t)t * 25;
#pragma unroll
          for (int d = 0; d < 25; d++) {
            const size_t row = MODE == 0 ? base + d : base + (size_t)d * 26;   // node t+1+d, slot d
            nxt[d] = p[row * 48 + lc];
          }
        } else {
#pragma unroll
          for (int d = 0; d < 25; d++) nxt[d] = 0.0;
        }
        // `delay` x 64 clocks of "compute" between issuing this step's loads and consuming the oldest ones
        for (int i = 0; i < delay; i += 100) __builtin_amdgcn_s_sleep(100);
#pragma unroll
        for (int d = 0; d < 25; d++) s += cur[0][d];
#pragma unroll
        for (int a = 0; a + 1 < AHEAD; a++)
#pragma unroll
          for (int d = 0; d < 25; d++) cur[a][d] = cur[a + 1][d];
#pragma unroll
        for (int d = 0; d < 25; d++) cur[AHEAD - 1][d] = nxt[d];
      }
    } else {
      typedef double v2 __attribute__((ext_vector_type(2)));
      const v2* q = (const v2*)p;
      for (int t = 0; t < steps; t++) {
        const size_t base = (size_t)t * 600;   // 25 rows x 24 double2
        v2 x[10];
#pragma unroll
        for (int j = 0; j < 10; j++) {
          const int e = j * 64 + lane;
          x[j] = q[base + (e < 600 ? e : 599)];
        }
#pragma unroll
        for (int j = 0; j < 10; j++) s += x[j].x + x[j].y;
      }
    }
  }
  if (s == 12345.678) out[0] = s;
}

int main(int argc, char** argv) {
  const int n_streams = argc > 1 ? atoi(argv[1]) : 8192;
  const size_t region_rows = 7200;
  const int steps = 260;   // last row touched: 25*259 + 24*26 = 7099 < region_rows
  const size_t bytes = (size_t)n_streams * region_rows * 48 * 8;
  double *buf, *out;
  hipMalloc(&buf, bytes);
  hipMalloc(&out, 64);
  hipMemset(buf, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb = (double)n_streams * steps * 25 * 384 / 1e9;
  int delay = 0;
  auto run = [&](const char* name, auto kern, int waves, int blocks) {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), 0, 0, buf, region_rows, steps, n_streams, out, delay);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), 0, 0, buf, region_rows, steps, n_streams, out, delay);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/wg %2d blocks %4d delay %4d: %7.3f ms  %6.0f GB/s  (%.2f us per step)\n", name, waves, blocks, delay, ms, gb / ms * 1e3,
           ms * 1e3 / (((double)n_streams / (waves * blocks)) * steps));
  };
  if (argc > 2) {   // the recursion's shape: loads, then `delay` x 64 clocks of work before they are consumed
    for (int dl : {0, 25, 50, 100, 200}) {
      delay = dl;
      for (int waves : {8, 12}) {
        run("fwd rows, 1 step in flight", k<0, 1>, waves, 256);
        run("fwd rows, 2 steps in flight", k<0, 2>, waves, 256);
        run("fwd rows, 3 steps in flight", k<0, 3>, waves, 256);
        run("bwd rows, 1 step in flight", k<1, 1>, waves, 256);
        run("bwd rows, 2 steps in flight", k<1, 2>, waves, 256);
      }
    }
    return 0;
  }
  for (int waves : {4, 8, 12}) {
    for (int blocks : {256, 512, 1024}) {
      run("fwd rows, 1 step in flight", k<0, 1>, waves, blocks);
      run("fwd rows, 2 steps in flight", k<0, 2>, waves, blocks);
      run("bwd rows (stride 26), 1 step", k<1, 1>, waves, blocks);
      run("bwd rows (stride 26), 2 steps", k<1, 2>, waves, blocks);
      run("fwd 16 B/lane whole-wave", k<2, 1>, waves, blocks);
    }
  }
  return 0;
}
