// experiment: issue rate of v_mfma_f32_16x16x4_f32 (the Winograd consumers' instruction) against v_mfma_f32_32x32x2_f32,
// one wave per SIMD, CH independent accumulator chains, operands taken from rotating registers like the kernel's fragments
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256, 1) void k16(float* out, unsigned long long* cyc, int n) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 a = {threadIdx.x * 1e-3f, 1.f, 2.f, 3.f}, b = {1.0f + threadIdx.x * 1e-4f, .5f, .25f, .125f};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + c) & 3], b[(u + c) & 3], acc[c], 0, 0, 0);
  }
  float s = 0; for (int c = 0; c < CH; ++c) s += acc[c][0];
  asm volatile("" :: "v"(s));
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH>
__global__ __launch_bounds__(256, 1) void k32(float* out, unsigned long long* cyc, int n) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  f32x4 a = {threadIdx.x * 1e-3f, 1.f, 2.f, 3.f}, b = {1.0f + threadIdx.x * 1e-4f, .5f, .25f, .125f};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + c) & 3], b[(u + c) & 3], acc[c], 0, 0, 0);
  }
  float s = 0; for (int c = 0; c < CH; ++c) s += acc[c][0];
  asm volatile("" :: "v"(s));
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K> void run(const char* name, K kern, int ch, double flop, float* out, unsigned long long* cyc, int grid) {
  int n = 64;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, cyc, n);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, cyc, n);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  double nm = (double)n * 8 * ch;
  printf("%s chains=%d grid=%d: %.1f cycles/MFMA (wave0), kernel %.1f us, %.1f TFLOP/s\n", name, ch, grid, h / nm, ms * 1e3,
         nm * flop * 4 * grid / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 64);
  for (int grid : {1, 256}) {
    run("16x16x4", k16<1>, 1, 2048, out, cyc, grid); run("16x16x4", k16<2>, 2, 2048, out, cyc, grid);
    run("16x16x4", k16<4>, 4, 2048, out, cyc, grid); run("16x16x4", k16<16>, 16, 2048, out, cyc, grid);
    run("32x32x2", k32<1>, 1, 4096, out, cyc, grid); run("32x32x2", k32<2>, 2, 4096, out, cyc, grid);
    run("32x32x2", k32<4>, 4, 4096, out, cyc, grid);
  }
  return 0;
}
