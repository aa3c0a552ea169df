"""In-kernel timeline of wgrad32_wino5_kernel (csrc/wgrad_wino5.hip): builds an INSTRUMENTED COPY of the kernel source as a standalone
HIP program (s_memtime stamps by lane 0 of every wave of workgroup (0, 0): start of a unit, after the transforms, after the first
barrier, after the multiply phase, after the second barrier) and runs it on synthetic data (B = 64, 10 frames, esplit 4: the
model's launch).  The product library is not touched.  Variants (compile-time): --define W5_NOISSUE (no activation loads),
W5_NOREAD (no fragment reads).
  python tools/experiments/w5_stamps.py build      (here: hipcc cross-compiles to gpurun_ab/w5s/w5_stamps)
  gpurun_ab/w5s/w5_stamps                          (on the GPU box)
What it showed (round 3): the 30 scattered 8-byte loads of the next unit take a wave 2400 - 3800 cycles to ISSUE; the K loop alone
runs at 44 cycles per MFMA (32 is the matrix core's rate with two waves per SIMD: mfma_rate in the same directory's history);
transforms 2400 - 3400 cycles; a unit 13,300 cycles -> 12,000 with the two waves of a SIMD taking turns at loads and MFMAs; raising
the priority of waves 4-7 (s_setprio 1 / 3) only swaps which group is fast (63.4 vs 60.5 us)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "ode-rl_amd", "csrc", "wgrad_wino5.hip")
OUT = os.path.join(ROOT, "gpurun_ab", "w5s")

MAIN = r'''
#include <stdio.h>
#include <vector>
namespace odehip { int hip_fail(hipError_t, const char*) { return 1; } void launch_slab_sum4(const float*, int, int, int, float*, hipStream_t) {} }
using namespace odehip;
int main() {
  const int B = 64, T = 10, esplit = 4;
  const size_t q4 = (size_t)B * 32 * 256 * 4;   // 128 channels: 32 quads x 256 px x 4
  float *g, *a, *slabs; WgradPair* tab;
  hipMalloc(&g, q4 * 4 * T); hipMalloc(&a, q4 * 4 * T); hipMalloc(&tab, sizeof(WgradPair) * T);
  hipMalloc(&slabs, ((size_t)B * esplit + 1) * kWgradSlabFloats * 4);
  std::vector<float> h(q4);
  for (size_t i = 0; i < q4; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
  for (int t = 0; t < T; ++t) { hipMemcpy(g + q4 * t, h.data(), q4 * 4, hipMemcpyHostToDevice); hipMemcpy(a + q4 * t, h.data(), q4 * 4, hipMemcpyHostToDevice); }
  std::vector<WgradPair> ht(T);
  for (int t = 0; t < T; ++t) { ht[t].g = g + q4 * t; ht[t].a = a + q4 * t; ht[t].scale = 1.0f; }
  hipMemcpy(tab, ht.data(), sizeof(WgradPair) * T, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)wgrad32_wino5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    for (int k = 0; k < 20; ++k)
      hipLaunchKernelGGL(wgrad32_wino5_kernel, dim3(B, esplit), dim3(512), kW5Lds, 0, tab, T, esplit, slabs, kWgradSlabFloats, 0, 32, 0, 32);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("avg launch %.1f us\n", ms * 1000 / 20);
  }
  unsigned long long st[8][64];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_w5_stamps), sizeof(st));
  for (int w = 0; w < 8; w += 4) {
    double tr = 0, w1 = 0, mu = 0, w2 = 0;
    for (int it = 1; it < 9; ++it) {
      const unsigned long long* p = &st[w][1 + 5 * it];
      tr += p[1] - p[0]; w1 += p[2] - p[1]; mu += p[3] - p[2]; w2 += p[4] - p[3];
    }
    printf("wave %d (shader cycles per unit): transform %.0f wait %.0f loads + multiply %.0f wait %.0f | slab %llu total %llu\n", w, tr / 8, w1 / 8,
           mu / 8, w2 / 8, st[w][57] - st[w][56], st[w][57] - st[w][0]);
  }
  return 0;
}
'''


def sub(s, old, new):
    assert s.count(old) == 1, old
    return s.replace(old, new, 1)


def build(defines):
    s = open(SRC).read()
    s = sub(s, '#include "conv_common.h"', '#include "conv_common.h"\n__device__ unsigned long long g_w5_stamps[8][64];\n'
            '#define W5_STAMP(i) do { if (b == 0 && es == 0 && lane == 0 && (i) < 64) g_w5_stamps[wave][(i)] = __builtin_amdgcn_s_memtime(); } while (0)')
    s = sub(s, "  if (n_it > 0) {\n    issue(0);", "  W5_STAMP(0);\n  if (n_it > 0) {\n    issue(0);")
    s = sub(s, "    transform();\n    __builtin_amdgcn_s_barrier();  // W and V of this chunk are in LDS\n",
            "    W5_STAMP(1 + 5 * it);\n    transform();\n    W5_STAMP(2 + 5 * it);\n    __builtin_amdgcn_s_barrier();\n    W5_STAMP(3 + 5 * it);\n")
    s = sub(s, "    if (wave >= 4 && it + 1 < n_it) issue(it + 1);", "#ifndef W5_NOISSUE\n    if (wave >= 4 && it + 1 < n_it) issue(it + 1);\n#endif")
    s = sub(s, "    if (wave < 4 && it + 1 < n_it) issue(it + 1);\n", "#ifndef W5_NOISSUE\n    if (wave < 4 && it + 1 < n_it) issue(it + 1);\n#endif\n    W5_STAMP(4 + 5 * it);\n")
    s = sub(s, "    __builtin_amdgcn_s_barrier();  // every wave is done reading before the next chunk is written\n  }\n",
            "    __builtin_amdgcn_s_barrier();\n    W5_STAMP(5 + 5 * it);\n  }\n  W5_STAMP(56);\n")
    s = sub(s, "    if (colhalf == 0 && cp == 0 && tl == 0) slab[36 * 32 * 32 + 4 * quad + c] = v;\n  }\n}",
            "    if (colhalf == 0 && cp == 0 && tl == 0) slab[36 * 32 * 32 + 4 * quad + c] = v;\n  }\n  W5_STAMP(57);\n}")
    s = sub(s, "        const float af = *(const float*)(fw + p * kW5Plane + off_a);",
            "#ifdef W5_NOREAD\n        const float af = __builtin_bit_cast(float, off_a + p);\n#else\n        const float af = *(const float*)(fw + p * kW5Plane + off_a);\n#endif")
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "w5_stamps.hip")
    open(path, "w").write(s + MAIN)
    csrc = os.path.join(ROOT, "ode-rl_amd", "csrc")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-value", "-mllvm", "-amdgpu-kernarg-preload-count=8",
           "-I" + csrc, "-I" + os.path.join(ROOT, "include")] + ["-D" + d for d in defines] + [path, "-o", os.path.join(OUT, "w5_stamps")]
    subprocess.check_call(cmd)
    print("built", os.path.join(OUT, "w5_stamps"))


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "build":
        build([a for a in sys.argv[2:]])
    else:
        print(__doc__)
