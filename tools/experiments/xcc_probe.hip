// Which XCD does workgroup i of a 256 x 512-thread, 160 KiB-LDS launch run on?  (HW_REG_XCC_ID, hwreg id 20.)
// Also times a same-XCD hand-off: workgroup A stores + bumps a counter, workgroup B polls it (relaxed agent-scope atomics).
// hipcc --offload-arch=gfx950 -O2 xcc_probe.hip -o xcc_probe && ./xcc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(512, 1) void probe(unsigned* out) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0) {
    out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    smem[0] = 1;
  }
}

// ping-pong between two workgroups `a` and `b`: n round trips through one counter
__global__ __launch_bounds__(64, 1) void pingpong(unsigned* ctr, int a, int b, int n, unsigned long long* cycles, int sleep) {
  extern __shared__ char smem[];
  if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
  const unsigned me = (int)blockIdx.x == a ? 0u : 1u;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    const unsigned want = 2u * i + me;  // a moves on even values, b on odd
    int guard = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
      if (sleep) __builtin_amdgcn_s_sleep(2);
      if (++guard > (1 << 22)) return;
    }
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0 && me == 0) *cycles = __builtin_amdgcn_s_memrealtime() - t0;
  if (threadIdx.x == 0) smem[0] = 1;
}

int main() {
  unsigned* out;
  hipMalloc(&out, 256 * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(probe, dim3(256), dim3(512), 160 * 1024, 0, out);
  unsigned h[256];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int rr = 1;
  for (int i = 0; i < 256; ++i) rr &= ((h[i] & 15) == (unsigned)(i & 7));
  printf("xcc ids of blocks 0..15:");
  for (int i = 0; i < 16; ++i) printf(" %u", h[i]);
  printf("\nround-robin (block i on XCD i %% 8) for all 256 blocks: %s\n", rr ? "yes" : "NO");
  unsigned* ctr;
  unsigned long long* cyc;
  hipMalloc(&ctr, 4);
  hipMalloc(&cyc, 8);
  hipFuncSetAttribute((const void*)pingpong, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int pairs[3][2] = {{0, 8}, {0, 1}, {0, 16}};
  for (int sl = 0; sl < 2; ++sl)
    for (auto& p : pairs) {
      hipMemset(ctr, 0, 4);
      int n = 1000;
      hipMemset(cyc, 0, 8);
      hipLaunchKernelGGL(pingpong, dim3(256), dim3(64), 160 * 1024, 0, ctr, p[0], p[1], n, cyc, sl);  // one workgroup per CU: all resident
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) printf("launch failed: %s\n", hipGetErrorString(e));
      unsigned long long c = 0;
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("ping-pong blocks %d <-> %d (xcd %u, %u), sleep=%d: %.1f ns per one-way hand-off\n", p[0], p[1], h[p[0]] & 15, h[p[1]] & 15, sl,
             c * 10.0 / (2.0 * n));
    }
  return 0;
}
