// conv_f43.hip -- EXPERIMENTAL (not on the product path yet): 3x3 conv 64 -> 64 on 16x16 maps by Winograd F(4x4,3x3) in exact
// fp32 arithmetic, with the input transform OUTSIDE the matrix kernel (the candidate for the next round).
//
//   F(4x4,3x3): a 16x16 map is 4x4 tiles of 4x4 outputs; each tile needs 36 instead of 144 multiplies per channel pair
//   (F(2x2,3x3): 64).  fp32 error vs an fp64 direct conv: 1.7e-6 rel-L2 per layer (tools/experiments/winograd_f43_error.py).
//
//   odehip_f43_transform_input : raw Q4 (B,64,16,16) -> V[b][slice 4][xi 36][lane 64][j 4]   (V = B^T d B per 6x6 patch;
//                                lane = k*16 + tile, input channel = 16*slice + 4*j + k: the B-operand image of the MFMA)
//   odehip_pack_conv_weight_f43: OIHW -> U[co-tile 4][slice 4][xi 36][lane 64][j 4]            (U = G g G^T; lane = k*16 + co)
//   odehip_conv_f43            : workgroup = (sample, 16 output channels, whole image), 256 workgroups at B = 64.  The four
//       waves split K: wave s multiplies the 16 input channels of slice s for all 36 positions (144 v_mfma_f32_16x16x4_f32),
//       streaming its own U and V slices through a PRIVATE 3-stage LDS ring (6 positions = 12 KiB per stage) -- no barrier in
//       the MFMA loop, no VALU work either (fp32 MFMA shares the VALU).  Each wave applies the linear output transform
//       A^T M A to its K-partial result in registers, the four partials meet once in LDS, then bias / ReLU and a transposing
//       pass through LDS for coalesced Q4 stores.
#include "conv_common.h"

namespace odehip {

// ---- transform matrices of F(4x4,3x3) (Lavin & Gray), applied as fixed instruction sequences
// B^T rows: {4,0,-5,0,1,0}, {0,-4,-4,1,1,0}, {0,4,-4,-1,1,0}, {0,-2,-1,2,1,0}, {0,2,-1,-2,1,0}, {0,4,0,-5,0,1}
__device__ __forceinline__ void bt6(const float (&d)[6], float (&o)[6]) {
  o[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
  o[1] = -4.0f * (d[1] + d[2]) + d[3] + d[4];
  o[2] = 4.0f * (d[1] - d[2]) - d[3] + d[4];
  o[3] = -2.0f * d[1] - d[2] + 2.0f * d[3] + d[4];
  o[4] = 2.0f * d[1] - d[2] - 2.0f * d[3] + d[4];
  o[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
}
// A^T rows: {1,1,1,1,1,0}, {0,1,-1,2,-2,0}, {0,1,1,4,4,0}, {0,1,-1,8,-8,1}
__device__ __forceinline__ void at6(const float (&m)[6], float (&o)[4]) {
  const float s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
  o[0] = m[0] + s1 + s2;
  o[1] = d1 + 2.0f * d2;
  o[2] = s1 + 4.0f * s2;
  o[3] = d1 + 8.0f * d2 + m[5];
}

// raw Q4 -> V.  One workgroup = (sample, 16-channel slice); thread = (channel c = tid >> 4, tile n = tid & 15).
__global__ __launch_bounds__(256) void f43_transform_input_kernel(const float* __restrict__ src, float* __restrict__ v, int quads) {
  __shared__ __attribute__((aligned(16))) float raw[4 * 324 * 4];  // [quad 4][18 x 18][4]
  __shared__ __attribute__((aligned(16))) float stage[36 * 256];    // [xi][lane 64][j 4]
  const int tid = threadIdx.x, s = blockIdx.x, b = blockIdx.y;
  for (int i = tid; i < 4 * 324; i += 256) *(f32x4*)(raw + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const f32x4* g = (const f32x4*)(src + ((size_t)b * quads + s * 4) * kPix * 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = i * 256 + tid, q = idx >> 8, p = idx & 255;
    *(f32x4*)(raw + (q * 324 + ((p >> 4) + 1) * 18 + (p & 15) + 1) * 4) = g[idx];
  }
  __syncthreads();
  const int c = tid >> 4, n = tid & 15, ty = n >> 2, tx = n & 3;
  const float* base = raw + ((c >> 2) * 324 + (4 * ty) * 18 + 4 * tx) * 4 + (c & 3);
  float t[6][6];
#pragma unroll
  for (int col = 0; col < 6; ++col) {  // B^T d: down the columns
    float d[6], o[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) d[r] = base[(r * 18 + col) * 4];
    bt6(d, o);
#pragma unroll
    for (int r = 0; r < 6; ++r) t[r][col] = o[r];
  }
  const int lane_img = (c & 3) * 16 + n, j = c >> 2;
#pragma unroll
  for (int r = 0; r < 6; ++r) {  // (B^T d) B: along the rows
    float o[6];
    bt6(t[r], o);
#pragma unroll
    for (int col = 0; col < 6; ++col) stage[((r * 6 + col) * 64 + lane_img) * 4 + j] = o[col];
  }
  __syncthreads();
  f32x4* dst = (f32x4*)(v + ((size_t)(b * 4 + s) * 36) * 256);
#pragma unroll
  for (int i = 0; i < 9; ++i) dst[i * 256 + tid] = *(const f32x4*)(stage + (i * 256 + tid) * 4);
}

// U[ct][s][xi][lane = k*16 + m][j] = (G g G^T)[xi] of W[co = 16 ct + m][ci = 16 s + 4 j + k]
__global__ __launch_bounds__(256) void f43_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int transpose_flip) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // 4*4*36*64*4 = 147456
  if (idx >= 147456) return;
  int r = idx;
  const int j = r & 3; r >>= 2;
  const int lane = r & 63; r >>= 6;
  const int xi = r % 36; r /= 36;
  const int s = r & 3;
  const int ct = r >> 2;
  const int co = ct * 16 + (lane & 15), ci = s * 16 + 4 * j + (lane >> 4);
  float g[3][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
      g[ky][kx] = transpose_flip ? w[((size_t)ci * 64 + co) * 9 + (2 - ky) * 3 + (2 - kx)] : w[((size_t)co * 64 + ci) * 9 + ky * 3 + kx];
  // G rows: {1/4,0,0}, {-1/6,-1/6,-1/6}, {-1/6,1/6,-1/6}, {1/24,1/12,1/6}, {1/24,-1/12,1/6}, {0,0,1}
  const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                         {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
  const int ur = xi / 6, uc = xi % 6;
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) acc += G[ur][ky] * g[ky][kx] * G[uc][kx];
  u[idx] = acc;
}

constexpr int kF43Group = 6;                       // positions per ring stage
constexpr int kF43Stage = 2 * kF43Group * 1024;    // U group + V group
constexpr int kF43Ring = 3 * kF43Stage;            // per wave
constexpr int kF43Lds = 4 * kF43Ring;              // 147,456 B; reused by the reduction (64 KiB) and the store pass (16 KiB)

__global__ __launch_bounds__(256, 1) void conv_f43_kernel(const float* __restrict__ vin, const float* __restrict__ u,
                                                          const float* __restrict__ bias, float* __restrict__ dst, int relu) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ct = blockIdx.x & 3, b = blockIdx.x >> 2;
  char* const ring = smem + wave * kF43Ring;
  const __amdgpu_buffer_rsrc_t ru = make_rsrc((const char*)u + (size_t)((ct * 4 + wave) * 36) * 1024, 36 * 1024);
  const __amdgpu_buffer_rsrc_t rv = make_rsrc((const char*)vin + (size_t)((b * 4 + wave) * 36) * 1024, 36 * 1024);
  const int vo = lane * 16;
  auto issue = [&](int g) {
    char* st = ring + (g % 3) * kF43Stage;
#pragma unroll
    for (int x = 0; x < kF43Group; ++x) {
      dma16(ru, st + x * 1024, vo, (g * kF43Group + x) * 1024);
      dma16(rv, st + (kF43Group + x) * 1024, vo, (g * kF43Group + x) * 1024);
    }
  };
  issue(0);
  issue(1);
  issue(2);
  f32x4 acc[36];
#pragma unroll
  for (int g = 0; g < 6; ++g) {
    // this wave's own DMAs: 12 per group, groups g+1, g+2 may stay in flight
    if (g <= 3) wait_vmcnt<24>(); else if (g == 4) wait_vmcnt<12>(); else wait_vmcnt<0>();
    const char* st = ring + (g % 3) * kF43Stage + vo;
#pragma unroll
    for (int x = 0; x < kF43Group; ++x) {
      const f32x4 a4 = *(const f32x4*)(st + x * 1024);
      const f32x4 b4 = *(const f32x4*)(st + (kF43Group + x) * 1024);
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, c, 0, 0, 0);
      acc[g * kF43Group + x] = c;
    }
    if (g + 3 < 6) issue(g + 3);  // the stage just read is free: the MFMAs above consumed its fragments
  }
  // ---- output transform of this wave's K-partial: lane = (tile n = lane & 15, co group lane >> 4), acc[xi][r] = M_xi[co 4(lane>>4)+r][n]
  __syncthreads();  // every wave is out of its ring: LDS is reused below
  float* red = (float*)smem;  // [wave 4][value 64][lane 64]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float tmp[4][6];
#pragma unroll
    for (int col = 0; col < 6; ++col) {
      float m[6], o[4];
#pragma unroll
      for (int row = 0; row < 6; ++row) m[row] = acc[row * 6 + col][r];
      at6(m, o);
#pragma unroll
      for (int i = 0; i < 4; ++i) tmp[i][col] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o[4];
      at6(tmp[i], o);
#pragma unroll
      for (int jx = 0; jx < 4; ++jx) red[(wave * 64 + r * 16 + i * 4 + jx) * 64 + lane] = o[jx];
    }
  }
  __syncthreads();
  // wave w finishes channel component r = w of every lane's co group: sum of the four K-partials, bias, ReLU
  float* img = (float*)(smem + 4 * 64 * 64 * 4);  // [quad 4][pixel 256][4] = 16 KiB behind the 64 KiB reduction buffer
  {
    const int r = wave, n = lane & 15, cg = lane >> 4;
    const float bv = bias ? bias[ct * 16 + cg * 4 + r] : 0.0f;
#pragma unroll
    for (int v16 = 0; v16 < 16; ++v16) {
      float y = bv;
#pragma unroll
      for (int w = 0; w < 4; ++w) y += red[(w * 64 + r * 16 + v16) * 64 + lane];
      if (relu) y = fmaxf(y, 0.0f);
      const int py = (n >> 2) * 4 + (v16 >> 2), px = (n & 3) * 4 + (v16 & 3);
      img[(cg * 256 + py * 16 + px) * 4 + r] = y;
    }
  }
  __syncthreads();
  f32x4* out = (f32x4*)(dst + ((size_t)b * 16 + ct * 4) * kPix * 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i * 256 + threadIdx.x] = *(const f32x4*)(img + (i * 256 + threadIdx.x) * 4);
}

}  // namespace odehip

using namespace odehip;

extern "C" size_t odehip_f43_weight_floats(void) { return 147456; }
extern "C" size_t odehip_f43_input_floats(int batch) { return (size_t)batch * 4 * 36 * 256; }

extern "C" int odehip_pack_conv_weight_f43(const float* w_oihw, float* u, int transpose_flip, void* stream) {
  ODEHIP_REQUIRE(w_oihw && u, "pack_conv_weight_f43: null pointer");
  hipLaunchKernelGGL(f43_pack_kernel, dim3(576), dim3(256), 0, (hipStream_t)stream, w_oihw, u, transpose_flip);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

extern "C" int odehip_f43_transform_input(const float* src_q4, float* v, int batch, void* stream) {
  ODEHIP_REQUIRE(src_q4 && v && batch > 0, "f43_transform_input: bad argument");
  hipLaunchKernelGGL(f43_transform_input_kernel, dim3(4, batch), dim3(256), 0, (hipStream_t)stream, src_q4, v, 16);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

// n back-to-back launches (n = 1: the layer itself); dst: Q4 (B,64,16,16)
extern "C" int odehip_conv_f43(const float* v, const float* u, const float* bias, float* dst_q4, int batch, int relu, int n,
                               void* stream) {
  ODEHIP_REQUIRE(v && u && dst_q4 && batch > 0 && n >= 1, "conv_f43: bad argument");
  static bool attr_set = false;
  if (!attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)conv_f43_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  for (int i = 0; i < n; ++i)
    hipLaunchKernelGGL(conv_f43_kernel, dim3(4 * batch), dim3(256), kF43Lds, (hipStream_t)stream, v, u, bias, dst_q4, relu);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}
