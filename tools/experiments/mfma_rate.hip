// experiment: issue rate of v_mfma_f32_32x32x2_f32 with 1 / 2 / 4 accumulator chains, one wave per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int n) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0; for (int c = 0; c < CH; ++c) s += acc[c][0];
  asm volatile("" :: "v"(s));
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH> void run(float* out, unsigned long long* cyc, int grid) {
  int n = 64;
  hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, out, cyc, n);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, out, cyc, n);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  double nm = (double)n * 8 * CH;
  printf("chains=%d grid=%d: %.1f cycles/MFMA (wave0), kernel %.1f us, %.1f TFLOP/s\n", CH, grid, h / nm, ms * 1e3,
         nm * 4096 * 4 * grid / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 64);
  for (int grid : {1, 256}) { run<1>(out, cyc, grid); run<2>(out, cyc, grid); run<4>(out, cyc, grid); }
  return 0;
}
