// experiment: cycles per "group" (2 ds_read_b128 + [4 v_cndmask] + 4 fp32 MFMA) in the conv main loop shape
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MASK, int SB, int CHAINS>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 256) ((float*)smem)[i] = i * 1e-4f;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const char* wb = smem + lane * 16;
  const char* xb = smem + 32768 + lane * 16;
  const bool kill = (lane & 15) == 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
    f32x4 wv = *(const f32x4*)(wb), xv = *(const f32x4*)(xb), wn, xn;
#pragma unroll
    for (int g = 0; g < 18; ++g) {
      if (MASK && (g % 3) != 1) { xv.x = kill ? 0.f : xv.x; xv.y = kill ? 0.f : xv.y; xv.z = kill ? 0.f : xv.z; xv.w = kill ? 0.f : xv.w; }
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, xv.x, acc0, 0, 0, 0);
      if (g + 1 < 18) { wn = *(const f32x4*)(wb + (g + 1) * 1024); xn = *(const f32x4*)(xb + (g + 1) * 256); }
      if (CHAINS == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, xv.y, acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, xv.y, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, xv.z, acc0, 0, 0, 0);
      if (CHAINS == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, xv.w, acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, xv.w, acc0, 0, 0, 0);
      wv = wn; xv = xn;
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      if (SB) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = acc0[0] + acc1[0];
  asm volatile("" :: "v"(s));
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MASK, int SB, int CHAINS> void run(float* out, unsigned long long* cyc) {
  int n = 16;
  hipFuncSetAttribute((const void*)k<MASK, SB, CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<MASK, SB, CHAINS>), dim3(256), dim3(256), 65536, 0, out, cyc, n); hipDeviceSynchronize(); }
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("mask=%d sched_barrier=%d chains=%d: %.1f cycles/group (ideal 260)\n", MASK, SB, CHAINS, (double)h / (n * 18));
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 64);
  run<0, 0, 1>(out, cyc); run<0, 0, 2>(out, cyc); run<1, 0, 2>(out, cyc); run<1, 1, 2>(out, cyc); run<0, 1, 2>(out, cyc); run<1, 1, 1>(out, cyc);
  return 0;
}
