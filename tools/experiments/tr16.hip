#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i;
  __syncthreads();
  // per 16-lane group: block of 4 rows x 16 columns; lane 4q+p supplies the address of row q, columns 4p..4p+3
  const int lane = threadIdx.x, g = lane >> 4, l = lane & 15, q = l >> 2, p = l & 3;
  const int row_stride = 64;  // elements
  __attribute__((address_space(3))) s16x4* addr = (__attribute__((address_space(3))) s16x4*)(lds + (g * 4 + q) * row_stride + 4 * p);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(addr);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int lane = 0; lane < 64; lane += 1) if (lane < 20 || lane % 16 == 0) printf("lane %2d: %d %d %d %d\n", lane, h[lane*4], h[lane*4+1], h[lane*4+2], h[lane*4+3]);
  return 0;
}
