// wgrad_wino5.hip -- weight gradient of the ConvGRU cell's 5x5 convolutions (/root/reference/modules/ConvGRUCell.py:40-50) in the
// Winograd F(2x2,5x5) domain (gfx950, exact fp32 MFMA).
//
//   dg[co][ci] (5x5) = G^T [ sum over evaluations e, samples b, tiles  (A dY A^T)[co] .* (B^T d B)[ci] ] G
// with the transforms of conv_wino5.hip (interpolation points 0, +-1, +-2, inf): the forward tile Y = A^T[(G g G^T) .* (B^T d B)]A is
// linear in g, so the 36 multiplies of the forward tile become 36 multiplies of the gradient tile -- against 100 per 2x2 outputs in
// the direct sum (wgrad_tile_kernel<5, ..>: three launches per 64x64 channel tile at 0.5 of the fp32 MFMA peak).
// tools/experiments/winograd_f25_wgrad_check.py: identity 6.6e-15 in fp64; fp32 error 3.6e-6 rel-L2 (direct sum 5.3e-7).
//
// Same contract as the other batched weight-gradient kernels: one launch per layer and channel tile, workgroup (sample b, split s)
// walks its share of the evaluations with the whole gradient tile in MFMA accumulators, one slab per workgroup at the end, a
// fixed-order sum over the slabs (launch_slab_sum4, wgrad.hip: bitwise reproducible, no float atomics), then G^T . G per channel pair.  36 positions x 64 x 64
// accumulators do not fit a workgroup, so the channel tile here is 32 x 32 (four launches per 64x64 tile of the caller).
//
// Per chunk of 16 tiles (two tile rows): four threads share a (tile, channel quad) -- the activation transform V = B^T d B as in the
// forward kernel ((3 of the 6 V columns) x (2 of the 4 channels) each: 30 zero-padded 8-byte loads straight from global memory, 72
// packed-fp32 instructions, 18 LDS writes), the gradient transform W = A dY A^T by three of the four (two W rows each) -- into LDS
// ([xi 36][tile 16][quad 8] x 16 B per operand = 144 KiB, quads XOR-swizzled by the tile: conflict-free fragment reads); then wave w
// multiplies positions 9 (w >> 1) .. +8 for output-channel half (w & 1): per K-step of 4 tiles 27 ds_read_b32 feed 18
// v_mfma_f32_16x16x4_f32.  The loads of the next chunk are issued before the MFMAs of the current one.
#include <stdlib.h>

#include "conv_common.h"

namespace odehip {

constexpr int kW5Plane = 16 * 8 * 16;          // one position of one operand: [tile 16][quad 8] x 16 B
constexpr int kW5Operand = 36 * kW5Plane;      // 72 KiB
constexpr int kW5Lds = 2 * kW5Operand;         // W | V
constexpr int kW5Slab = 36 * 32 * 32 + 32;     // floats per workgroup slab: dM [xi 36][co 32][ci 32] + db (32)

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 w5_fma2(float c, f32x2 a, f32x2 b) { return f32x2{__builtin_fmaf(c, a.x, b.x), __builtin_fmaf(c, a.y, b.y)}; }

// out[i] = sum_k BT[i][k] in[k],  BT = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void w5_bt6(const f32x2* in, f32x2* out) {
  out[0] = w5_fma2(4.0f, in[0], w5_fma2(-5.0f, in[2], in[4]));
  const f32x2 a = w5_fma2(-4.0f, in[2], in[4]), b = w5_fma2(-4.0f, in[1], in[3]);
  out[1] = a + b;
  out[2] = a - b;
  const f32x2 c = in[4] - in[2], s = in[3] - in[1];
  out[3] = w5_fma2(2.0f, s, c);
  out[4] = w5_fma2(-2.0f, s, c);
  out[5] = w5_fma2(4.0f, in[1], w5_fma2(-5.0f, in[3], in[5]));
}

template <int V> struct W5Tag { static constexpr int value = V; };

// quad swizzle by the tile: the 8 tiles x 16-byte writes of a lane group and the fragment reads of a K-step (tiles 4s .. 4s+3, one
// 16-channel block) both spread over the banks
__device__ __forceinline__ int w5_swz(int tile) { return ((tile & 1) << 2) | ((tile >> 1) & 3); }

__global__ __launch_bounds__(512, 1) void wgrad32_wino5_kernel(const WgradPair* __restrict__ table, int n_eval, int esplit,
                                                               float* __restrict__ slabs, int slab_stride, int g_quad0, int g_quads,
                                                               int a_quad0, int a_quads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, es = blockIdx.y;
  // ---- transform role.  Wave w takes channel quads 2 (w >> 1), 2 (w >> 1) + 1 and V-column half (w & 1) -- WAVE-UNIFORM, so the two
  // forms of the column transform are a uniform branch (as lane-dependent code every lane ran both) --; lane = tile | quad select << 4 |
  // channel pair << 5.  The gradient transform's six W rows go to sub-tasks 0, 1, 2 of sub = column half + 2 channel pair.
  const int colhalf = wave & 1;
  const int tl = lane & 15, qsel = (lane >> 4) & 1, cp = lane >> 5;
  const int quad = 2 * (wave >> 1) + qsel, sub = colhalf + 2 * cp;
  const int tyl = tl >> 3, tx = tl & 7;
  // (tried: rotating the slot by the position as well, so that lanes writing different positions in one instruction spread over the
  // banks -- 84.5 instead of 79.0 us per launch: the extra address arithmetic of the fragment reads sits in the MFMA loop)
  char* const wr_w = smem + tl * 128 + ((quad ^ w5_swz(tl)) * 16);                       // + xi * kW5Plane
  char* const wr_v = smem + kW5Operand + tl * 128 + ((quad ^ w5_swz(tl)) * 16) + cp * 8;  // + xi * kW5Plane
  // ---- multiply role
  const int pg = wave >> 1, cohalf = wave & 1;
  const int m = lane & 15, kq = lane >> 4;
  f32x4 acc[9][2];
#pragma unroll
  for (int p = 0; p < 9; ++p) acc[p][0] = acc[p][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  // work units are (evaluation, chunk) pairs, chunk c = tile rows 2c, 2c + 1; split s takes units s, s + esplit, ... (10 evaluations
  // over 4 splits: 10 units each -- whole evaluations would be 12 | 12 | 8 | 8 and the launch as long as the longest)
  const int n_units = 4 * n_eval;
  const int n_it = es < n_units ? (n_units - es + esplit - 1) / esplit : 0;
  f32x2 ra[6][5];               // activation patch: rows 0..5, columns colhalf .. colhalf + 4 (loaded a chunk ahead)
  float esc_raw = 0.0f;
  const float* g_next = nullptr;
  int ty_next = 0;
  auto issue = [&](int it) {
    const int u = es + it * esplit, e = u >> 2, c = u & 3;
    const WgradPair pr = table[e];
    esc_raw = pr.scale;
    const int ty = 2 * c + tyl;
    g_next = pr.g;
    ty_next = ty;
    const __amdgpu_buffer_rsrc_t rsa = make_rsrc(pr.a + ((size_t)b * a_quads + a_quad0) * 4 * kPix, 32 * kPix * 4);
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int l = 0; l < 5; ++l) {
        const int row = 2 * ty - 2 + k, col = 2 * tx - 2 + colhalf + l;
        const bool in = row >= 0 && row < kHW && col >= 0 && col < kHW;
        ra[k][l] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsa, in ? quad * 4096 + (row * 16 + col) * 16 + cp * 8 : kOobOffset, 0, 0));
      }
  };
  // V columns 3 CH .. 3 CH + 2 of this thread's (tile, channel pair): (d B)[k][3 CH + jj] row by row, then B^T down the column
  auto act_transform = [&](auto ch_tag) {
    constexpr int CH = decltype(ch_tag)::value;
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      f32x2 W[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const f32x2* d = ra[k];   // d[l] = patch column CH + l
        if (CH == 0) {            // columns 0, 1, 2 of d B from patch columns 0 .. 4
          if (jj == 0) {
            W[k] = w5_fma2(4.0f, d[0], w5_fma2(-5.0f, d[2], d[4]));
          } else {
            const f32x2 aa = w5_fma2(-4.0f, d[2], d[4]), bb = w5_fma2(-4.0f, d[1], d[3]);
            W[k] = jj == 1 ? aa + bb : aa - bb;
          }
        } else {                  // columns 3, 4, 5 from patch columns 1 .. 5 (d[l] = column 1 + l)
          if (jj == 2) {
            W[k] = w5_fma2(4.0f, d[0], w5_fma2(-5.0f, d[2], d[4]));
          } else {
            const f32x2 cc = d[3] - d[1], ss = d[2] - d[0];
            W[k] = w5_fma2(jj == 0 ? 2.0f : -2.0f, ss, cc);
          }
        }
      }
      f32x2 o[6];
      w5_bt6(W, o);
#pragma unroll
      for (int i = 0; i < 6; ++i) *(f32x2*)(wr_v + (6 * i + 3 * CH + jj) * kW5Plane) = o[i];
    }
  };
  // gradient loads (4 coalesced 16-byte loads): issued BEHIND the multiply phase of the previous chunk -- they fly during the barrier
  // and the activation transform and are not carried across a multiply phase, where every register counts
  f32x4 rg[4];
  auto issue_grad = [&]() {
    if (sub < 3) {
      const __amdgpu_buffer_rsrc_t rsg = make_rsrc(g_next + ((size_t)b * g_quads + g_quad0) * 4 * kPix, 32 * kPix * 4);
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int l = 0; l < 2; ++l)
          rg[2 * k + l] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsg, quad * 4096 + ((2 * ty_next + k) * 16 + 2 * tx + l) * 16, 0, 0));
    }
  };
  auto transform = [&]() {
    if (colhalf == 0) act_transform(W5Tag<0>{}); else act_transform(W5Tag<1>{});
    // ---- gradient: W = A dY A^T, A rows (1,0) (1,1) (1,-1) (1,2) (1,-2) (0,1); this thread writes W rows 2 sub, 2 sub + 1
    if (sub < 3) {
      const f32x4 d00 = rg[0] * esc_raw, d01 = rg[1] * esc_raw, d10 = rg[2] * esc_raw, d11 = rg[3] * esc_raw;
      if (sub == 0) bsum += (d00 + d01) + (d10 + d11);
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * sub + ii;
        // t = row i of A dY (a 2-vector over the tile's columns)
        const float a0 = i == 5 ? 0.0f : 1.0f;
        const float a1 = i == 0 ? 0.0f : (i == 1 ? 1.0f : (i == 2 ? -1.0f : (i == 3 ? 2.0f : (i == 4 ? -2.0f : 1.0f))));
        const f32x4 t0 = d00 * a0 + d10 * a1, t1 = d01 * a0 + d11 * a1;
        char* const w = wr_w + (6 * i) * kW5Plane;
        *(f32x4*)(w + 0 * kW5Plane) = t0;
        *(f32x4*)(w + 1 * kW5Plane) = t0 + t1;
        *(f32x4*)(w + 2 * kW5Plane) = t0 - t1;
        *(f32x4*)(w + 3 * kW5Plane) = t0 + t1 * 2.0f;
        *(f32x4*)(w + 4 * kW5Plane) = t0 - t1 * 2.0f;
        *(f32x4*)(w + 5 * kW5Plane) = t1;
      }
    }
  };

  if (n_it > 0) {
    issue(0);
    issue_grad();
  }
#pragma unroll 1
  for (int it = 0; it < n_it; ++it) {
    transform();
    __builtin_amdgcn_s_barrier();  // W and V of this chunk are in LDS
    // The 30 scattered 8-byte loads of the next unit take a wave 2400 - 3800 cycles to ISSUE (in-kernel stamps,
    // tools/experiments/w5_stamps.py: the address unit takes ~16 cycles per wave instruction and eight waves queue up), during which
    // it issues no MFMA.  So the two waves of a SIMD take turns: waves 4-7 issue their loads first while waves 0-3 have the matrix
    // core, then the other way round (66.7 -> 61.5 us per launch).
    if (wave >= 4 && it + 1 < n_it) issue(it + 1);
    // fragment of K-step s: tile 4 s + kq, channel 16 blk + m -> quad (4 blk + m / 4) ^ swz(tile), float m % 4
    const char* const fw = smem + (9 * pg) * kW5Plane + (m & 3) * 4;
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {   // (not unrolled: the 27 fragment reads of a K-step must not be hoisted over the previous one's MFMAs' registers)
      const int tile = 4 * s + kq;
      const int sw = w5_swz(tile);
      const int off_a = tile * 128 + (((4 * cohalf + (m >> 2)) ^ sw) * 16);
      const int off_b0 = tile * 128 + ((((m >> 2)) ^ sw) * 16);
      const int off_b1 = tile * 128 + (((4 + (m >> 2)) ^ sw) * 16);
#pragma unroll
      for (int p = 0; p < 9; ++p) {
        const float af = *(const float*)(fw + p * kW5Plane + off_a);
        const float b0 = *(const float*)(fw + kW5Operand + p * kW5Plane + off_b0);
        const float b1 = *(const float*)(fw + kW5Operand + p * kW5Plane + off_b1);
        acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, b0, acc[p][0], 0, 0, 0);
        acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, b1, acc[p][1], 0, 0, 0);
      }
    }
    if (wave < 4 && it + 1 < n_it) issue(it + 1);
    if (it + 1 < n_it) issue_grad();
    __builtin_amdgcn_s_barrier();  // every wave is done reading before the next chunk is written
  }

  // slab[(b * esplit + es)] = dM [xi 36][co 32][ci 32] followed by db (32)
  float* slab = slabs + (size_t)(b * esplit + es) * slab_stride;
#pragma unroll
  for (int p = 0; p < 9; ++p)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)  // D row = 4 kq + r -> co (within this wave's half), D col = m -> ci
        slab[((size_t)(9 * pg + p) * 32 + 16 * cohalf + 4 * kq + r) * 32 + 16 * j + m] = acc[p][j][r];
  // bias sums: the sub == 0 threads (column half 0, channel pair 0) of a wave cover the 16 tiles of every chunk of their channel quad
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float v = sub == 0 ? bsum[c] : 0.0f;
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    if (colhalf == 0 && cp == 0 && tl == 0) slab[36 * 32 * 32 + 4 * quad + c] = v;
  }
}

// dg = G^T dM G per channel pair; G rows: [1/4 0 0 0 0], -[1 1 1 1 1]/6, -[1 -1 1 -1 1]/6, [1 2 4 8 16]/24, [1 -2 4 -8 16]/24, [0 0 0 0 1]
__global__ __launch_bounds__(256) void wgrad_wino5_finish_kernel(const float* __restrict__ sum, float* __restrict__ dw, float* __restrict__ db,
                                                                 int cin_total, int co0, int ci0, int write_bias) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // co * 32 + ci
  const int co = idx >> 5, ci = idx & 31;
  float M[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) M[i][j] = sum[(size_t)(6 * i + j) * 1024 + idx];
  // R = G^T M (5 x 6): R[a][j] = sum_i G[i][a] M[i][j]
  float R[5][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float m0 = M[0][j], m1 = M[1][j], m2 = M[2][j], m3 = M[3][j], m4 = M[4][j], m5 = M[5][j];
    const float s12 = -(m1 + m2) * (1.0f / 6.0f), d12 = -(m1 - m2) * (1.0f / 6.0f);
    const float s34 = (m3 + m4) * (1.0f / 24.0f), d34 = (m3 - m4) * (1.0f / 24.0f);
    R[0][j] = m0 * 0.25f + s12 + s34;
    R[1][j] = d12 + 2.0f * d34;
    R[2][j] = s12 + 4.0f * s34;
    R[3][j] = d12 + 8.0f * d34;
    R[4][j] = s12 + 16.0f * s34 + m5;
  }
  float* o = dw + ((size_t)(co0 + co) * cin_total + ci0 + ci) * 25;
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    const float r0 = R[a][0], r1 = R[a][1], r2 = R[a][2], r3 = R[a][3], r4 = R[a][4], r5 = R[a][5];
    const float s12 = -(r1 + r2) * (1.0f / 6.0f), d12 = -(r1 - r2) * (1.0f / 6.0f);
    const float s34 = (r3 + r4) * (1.0f / 24.0f), d34 = (r3 - r4) * (1.0f / 24.0f);
    o[5 * a + 0] = r0 * 0.25f + s12 + s34;
    o[5 * a + 1] = d12 + 2.0f * d34;
    o[5 * a + 2] = s12 + 4.0f * s34;
    o[5 * a + 3] = d12 + 8.0f * d34;
    o[5 * a + 4] = s12 + 16.0f * s34 + r5;
  }
  if (write_bias && ci == 0) db[co0 + co] = sum[36 * 32 * 32 + co];
}

// One 64 x 64 tile of a 5x5 weight gradient (the contract of launch_wgrad_tile, wgrad.hip) as four 32 x 32 launches in the Winograd
// domain.  Returns 1 if switched off (ODEHIP_WGRAD_WINO5=0: the direct kernel).  slabs: (batch * esplit + 1) * kWgradSlabFloats floats.
int launch_wgrad_wino5(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, float* db, int cin_total,
                       int co0, int ci0, int g_quads, int g_quad0, int a_quads, int a_quad0, bool write_bias, hipStream_t stream) {
  static const bool off = [] { const char* e = getenv("ODEHIP_WGRAD_WINO5"); return e && e[0] == '0'; }();
  if (off) return 1;
  static_assert(kW5Slab <= kWgradSlabFloats, "a slab of this kernel must fit the callers' slab allocation");
  static bool attr_set = false;
  if (!attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad32_wino5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  float* const sum = slabs + (size_t)batch * esplit * kWgradSlabFloats;  // the slab region is sized for one more slab
  for (int cs = 0; cs < 2; ++cs)
    for (int as = 0; as < 2; ++as) {
      hipLaunchKernelGGL(wgrad32_wino5_kernel, dim3(batch, esplit), dim3(512), kW5Lds, stream, table_dev, n_eval, esplit, slabs, kWgradSlabFloats,
                         g_quad0 + 8 * cs, g_quads, a_quad0 + 8 * as, a_quads);
      launch_slab_sum4(slabs, batch * esplit, kWgradSlabFloats, kW5Slab, sum, stream);
      hipLaunchKernelGGL(wgrad_wino5_finish_kernel, dim3(4), dim3(256), 0, stream, sum, dw, db, cin_total, co0 + 32 * cs, ci0 + 32 * as,
                         (int)(write_bias && as == 0));
    }
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

}  // namespace odehip
