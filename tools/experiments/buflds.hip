// experiment: does buffer_load ... lds zero-fill LDS for out-of-range lanes?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* out, int nbytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  // poison LDS
  ((f32x4*)smem)[lane] = f32x4{-7.f, -7.f, -7.f, -7.f};
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  // lanes >= 32 read out of range
  int voff = lane < 32 ? lane * 16 : 0x40000000 + lane * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x4 v = ((f32x4*)smem)[lane];
  out[lane * 4 + 0] = v.x; out[lane * 4 + 1] = v.y; out[lane * 4 + 2] = v.z; out[lane * 4 + 3] = v.w;
}
int main() {
  float *src, *out; float h[256];
  hipMalloc(&src, 1024); hipMalloc(&out, 1024);
  for (int i = 0; i < 256; ++i) h[i] = i + 1;
  hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, src, out, 1024);
  hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
  printf("lane0: %g %g  lane31: %g  lane32: %g %g  lane63: %g\n", h[0], h[1], h[31 * 4], h[32 * 4], h[32 * 4 + 1], h[63 * 4 + 3]);
  return 0;
}
