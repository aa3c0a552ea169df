// wgrad.hip -- weight/bias gradient of the 3x3 conv layers of f for the whole backward sweep (gfx950, exact fp32).
//
//   dW[co][ci][tap] = sum over every evaluation e of f, sample b, pixel p of  GP_e[b][co][p] * A_e[b][ci][p + tap]
//   db[co]          = sum of GP_e[b][co][p]
// where A_e is the saved input of the layer and GP_e the gradient w.r.t. its output (both Q4, written by the forward
// pass with save_for_backward and by the dgrad sweep).  What autograd does through torchdiffeq's ops, restated as ONE
// launch per layer: workgroup (sample b, split s) walks its share of the evaluations and keeps the full 64x64x9
// gradient tile in MFMA accumulators, so nothing is reduced through HBM until the very end (one 147 KB slab per
// workgroup, summed in a fixed order by `wgrad_reduce_kernel`: bitwise reproducible, no float atomics).
//
// MFMA mapping (v_mfma_f32_16x16x4_f32, K = 4 consecutive pixels of an image row):
//   A operand: lane (i, kq) reads GP[quad i][pixel p0+kq] (one ds_read_b128 = channels 4i..4i+3); wave w uses channel
//              4i+w, so the 4 waves split the 64 output channels by channel-within-quad.
//   B operand: lane (j, kq) reads A[quad j][pixel p0+kq+tap] (b128 = channels 4j..4j+3 = the four N blocks).
//   One G read + nine A reads feed 36 MFMAs.  LDS planes are padded by 16 B per quad so the 16 quads of a lane
//   group fall in different bank slots.
#include <string.h>

#include "odehip_internal.h"

namespace odehip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ODEHIP_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))


constexpr int kGPlane = 4096 + 16;       // one quad plane of G in LDS (256 px * 16 B, padded)
constexpr int kAPlane = 5 * 1024 + 16;   // one quad plane of A in LDS: rows -1..18 (5 DMA pieces), padded
constexpr int kWgradLds = 16 * kGPlane + 16 * kAPlane;

// One 64(co) x 64(ci) tile of the layer: output channels 4*g_quad0.., input channels 4*a_quad0.. of tensors that
// have g_quads / a_quads channel quads per sample.
__global__ __launch_bounds__(256, 1) void wgrad64_kernel(const WgradPair* __restrict__ table, int n_eval, int esplit,
                                                         float* __restrict__ slabs, int g_quad0, int g_quads, int a_quad0,
                                                         int a_quads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const gl = smem;
  char* const al = smem + 16 * kGPlane;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x, es = blockIdx.y;
  const int i16 = lane & 15, kq = lane >> 4;

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  // per-lane DMA offsets of the A pieces: piece r covers image rows 4r-1 .. 4r+2; out-of-image rows -> zero fill
  int va[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const int row = 4 * r - 1 + (lane >> 4);
    va[r] = (row >= 0 && row < kHW) ? row * 256 + (lane & 15) * 16 : 0x7fff0000;
  }
  const int vg = lane * 16;

  for (int e = es; e < n_eval; e += esplit) {
    const WgradPair pr = table[e];
    const float esc = pr.scale;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(pr.g) + ((size_t)b * g_quads + g_quad0) * 4 * kPix, 0, 64 * kPix * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(pr.a) + ((size_t)b * a_quads + a_quad0) * 4 * kPix, 0, 64 * kPix * 4, 0x00020000);
    __builtin_amdgcn_s_barrier();  // every wave is done reading the previous evaluation's tiles
    // 16 quads x (4 G pieces + 5 A pieces) = 144 DMAs, 36 per wave: wave w loads quads 4w..4w+3
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int q = wave * 4 + qq;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, ODEHIP_LDS_PTR(gl + q * kGPlane + r * 1024), 16, vg, q * 4096 + r * 1024, 0, 0);
#pragma unroll
      for (int r = 0; r < 5; ++r)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, ODEHIP_LDS_PTR(al + q * kAPlane + r * 1024), 16, va[r], q * 4096, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const char* gbase = gl + i16 * kGPlane + kq * 16;
    const char* abase = al + i16 * kAPlane + kq * 16;  // LDS row 0 = image row -1
    // K-step ks = 4*y + s covers pixels (y, 4s .. 4s+3).  Fragments of step ks+1 are read (10 ds_read_b128) while the
    // 36 MFMAs of step ks run: explicit register double buffering.
    auto load_step = [&](int ks, f32x4& g, f32x4 (&av)[9]) {
      const int y = ks >> 2, s4 = (ks & 3) * 4;
      g = *(const f32x4*)(gbase + (y * 16 + s4) * 16);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        av[t] = *(const f32x4*)(abase + ((y + 1 + dy) * 16 + s4 + dx) * 16);
      }
    };
    f32x4 gbuf[2], abuf[2][9];  // ping-pong by K-step parity: no register copies (VALU cycles are MFMA cycles on fp32)
    load_step(0, gbuf[0], abuf[0]);
    for (int y = 0; y < kHW; ++y) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int ks = y * 4 + s;
        if (ks + 1 < 64) load_step(ks + 1, gbuf[(s + 1) & 1], abuf[(s + 1) & 1]);
        const f32x4 gv = gbuf[s & 1] * esc;
        bsum += gv;  // every wave keeps the bias sums (only wave 0 writes them): no branch in the MFMA stream
        const float ga = wave == 0 ? gv.x : (wave == 1 ? gv.y : (wave == 2 ? gv.z : gv.w));
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int dx = t % 3 - 1;
          f32x4 av = abuf[s & 1][t];
          if (dx < 0 && s == 0) {  // pixel x-1 of x = 0: only the kq = 0 lanes
            const bool kill = kq == 0;
            av.x = kill ? 0.f : av.x; av.y = kill ? 0.f : av.y; av.z = kill ? 0.f : av.z; av.w = kill ? 0.f : av.w;
          }
          if (dx > 0 && s == 3) {  // pixel x+1 of x = 15: only the kq = 3 lanes
            const bool kill = kq == 3;
            av.x = kill ? 0.f : av.x; av.y = kill ? 0.f : av.y; av.z = kill ? 0.f : av.z; av.w = kill ? 0.f : av.w;
          }
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, av.x, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, av.y, acc[t][1], 0, 0, 0);
          acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, av.z, acc[t][2], 0, 0, 0);
          acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, av.w, acc[t][3], 0, 0, 0);
        }
      }
    }
  }

  // slab[(b*esplit+es)] = dW tile in OIHW order (64*64*9) followed by db (64)
  float* slab = slabs + (size_t)(b * esplit + es) * (64 * 64 * 9 + 64);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 4 * (4 * kq + r) + wave;  // D row = 4*(lane>>4) + r  -> quad index -> channel 4*quad + wave
        const int ci = 4 * i16 + n;              // D col = lane & 15        -> quad index -> channel 4*quad + n
        slab[((size_t)co * 64 + ci) * 9 + t] = acc[t][n][r];
      }
  if (wave == 0) {
    // lane (i16, kq) holds the sums over its pixels of channels 4*i16..4*i16+3: fold the four kq lanes
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = bsum[c];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (kq == 0) slab[64 * 64 * 9 + 4 * i16 + c] = v;
    }
  }
}

// The slab sum of the Winograd-domain weight gradients (wgrad_wino.hip, wgrad_wino5.hip): 256 slabs of 144 - 256 KiB.  One block =
// 64 float4 outputs x 16 slab groups; a thread adds its group's slabs in 4 independent chains of 16-byte loads (the first version --
// 64 floats x 4 groups of 4-byte loads -- had 12 KiB in flight per CU and ran at 2.3 TB/s), the 16 partial sums through LDS.
__global__ __launch_bounds__(1024) void slab_sum4_kernel(const f32x4* __restrict__ slabs, int n_slabs, int stride4, int n4, f32x4* __restrict__ sum) {
  __shared__ f32x4 part[16][64];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0, p2 = p0, p3 = p0;
  if (i < n4) {
    for (int k = g; k < n_slabs; k += 64) {
      p0 += slabs[(size_t)k * stride4 + i];
      if (k + 16 < n_slabs) p1 += slabs[(size_t)(k + 16) * stride4 + i];
      if (k + 32 < n_slabs) p2 += slabs[(size_t)(k + 32) * stride4 + i];
      if (k + 48 < n_slabs) p3 += slabs[(size_t)(k + 48) * stride4 + i];
    }
  }
  part[g][o] = (p0 + p1) + (p2 + p3);
  __syncthreads();
  if (g == 0 && i < n4) {
    f32x4 s = part[0][o];
#pragma unroll
    for (int k = 1; k < 16; ++k) s += part[k][o];
    sum[i] = s;
  }
}

void launch_slab_sum4(const float* slabs, int n_slabs, int stride, int n_vals, float* sum, hipStream_t stream) {
  const int n4 = n_vals / 4;
  hipLaunchKernelGGL(slab_sum4_kernel, dim3((n4 + 63) / 64), dim3(1024), 0, stream, (const f32x4*)slabs, n_slabs, stride / 4, n4, (f32x4*)sum);
}

// out[i] = sum over slabs in a fixed order: thread (o, g) adds slabs g, g+4, ... of output o; the 4 partial sums meet in LDS
// The 64x64 tile lands at (co0, ci0) of the (cout, cin, 3, 3) gradient; db (only for ci0 == 0) at co0.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int slab_floats,
                                                           float* __restrict__ dw, float* __restrict__ db, int cin, int co0,
                                                           int ci0) {
  __shared__ float part[4][64];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;  // independent chains: the loads overlap
  if (i < slab_floats) {
    for (int k = g; k < n_slabs; k += 16) {
      p0 += slabs[(size_t)k * slab_floats + i];
      if (k + 4 < n_slabs) p1 += slabs[(size_t)(k + 4) * slab_floats + i];
      if (k + 8 < n_slabs) p2 += slabs[(size_t)(k + 8) * slab_floats + i];
      if (k + 12 < n_slabs) p3 += slabs[(size_t)(k + 12) * slab_floats + i];
    }
  }
  part[g][o] = (p0 + p1) + (p2 + p3);
  __syncthreads();
  if (g == 0 && i < slab_floats) {
    const float s = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
    if (i < 64 * 64 * 9) {
      const int co = i / (64 * 9), r = i - co * 64 * 9, ci = r / 9, t = r - ci * 9;
      dw[((size_t)(co0 + co) * cin + ci0 + ci) * 9 + t] = s;
    } else if (ci0 == 0) {
      db[co0 + i - 64 * 64 * 9] = s;
    }
  }
}

// ---- generic tile kernel for the other layer shapes of the encoder (5x5 ConvGRU convs, 1x1 head): same structure, taps
// (ty, tx) with ty in [TY0, TY0+NTY): a 5x5 layer takes three launches (tap rows 0-1, 2-3, 4) so the accumulators fit.
template <int KS, int TY0, int NTY>
__global__ __launch_bounds__(256, 1) void wgrad_tile_kernel(const WgradPair* __restrict__ table, int n_eval, int esplit,
                                                            float* __restrict__ slabs, int g_quad0, int g_quads, int a_quad0,
                                                            int a_quads) {
  constexpr int HALO = KS / 2, NT = NTY * KS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const gl = smem;
  char* const al = smem + 16 * kGPlane;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x, es = blockIdx.y;
  const int i16 = lane & 15, kq = lane >> 4;

  f32x4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  int va[5];  // piece r covers image rows 4r-HALO .. 4r-HALO+3; out-of-image rows -> zero fill
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const int row = 4 * r - HALO + (lane >> 4);
    va[r] = (row >= 0 && row < kHW) ? row * 256 + (lane & 15) * 16 : 0x7fff0000;
  }
  const int vg = lane * 16;

  for (int e = es; e < n_eval; e += esplit) {
    const WgradPair pr = table[e];
    const float esc = pr.scale;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(pr.g) + ((size_t)b * g_quads + g_quad0) * 4 * kPix, 0, 64 * kPix * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(pr.a) + ((size_t)b * a_quads + a_quad0) * 4 * kPix, 0, 64 * kPix * 4, 0x00020000);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int q = wave * 4 + qq;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, ODEHIP_LDS_PTR(gl + q * kGPlane + r * 1024), 16, vg, q * 4096 + r * 1024, 0, 0);
#pragma unroll
      for (int r = 0; r < 5; ++r)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, ODEHIP_LDS_PTR(al + q * kAPlane + r * 1024), 16, va[r], q * 4096, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const char* gbase = gl + i16 * kGPlane + kq * 16;
    const char* abase = al + i16 * kAPlane + kq * 16;  // LDS row 0 = image row -HALO
    auto load_step = [&](int ks, f32x4& g, f32x4 (&av)[NT]) {
      const int y = ks >> 2, s4 = (ks & 3) * 4;
      g = *(const f32x4*)(gbase + (y * 16 + s4) * 16);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int dy = TY0 + t / KS - HALO, dx = t % KS - HALO;
        av[t] = *(const f32x4*)(abase + ((y + HALO + dy) * 16 + s4 + dx) * 16);
      }
    };
    f32x4 gbuf[2], abuf[2][NT];
    load_step(0, gbuf[0], abuf[0]);
    for (int y = 0; y < kHW; ++y) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int ks = y * 4 + s;
        if (ks + 1 < 64) load_step(ks + 1, gbuf[(s + 1) & 1], abuf[(s + 1) & 1]);
        const f32x4 gv = gbuf[s & 1] * esc;
        bsum += gv;
        const float ga = wave == 0 ? gv.x : (wave == 1 ? gv.y : (wave == 2 ? gv.z : gv.w));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int dx = t % KS - HALO;
          f32x4 av = abuf[s & 1][t];
          if ((dx < 0 && s == 0) || (dx > 0 && s == 3)) {  // pixel x+dx outside the row: x = 4s + kq
            const bool kill = (dx < 0) ? (kq + dx < 0) : (kq + dx > 3);
            av.x = kill ? 0.f : av.x; av.y = kill ? 0.f : av.y; av.z = kill ? 0.f : av.z; av.w = kill ? 0.f : av.w;
          }
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, av.x, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, av.y, acc[t][1], 0, 0, 0);
          acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, av.z, acc[t][2], 0, 0, 0);
          acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, av.w, acc[t][3], 0, 0, 0);
        }
      }
    }
  }

  float* slab = slabs + (size_t)(b * esplit + es) * (64 * 64 * NT + 64);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 4 * (4 * kq + r) + wave;
        const int ci = 4 * i16 + n;
        slab[((size_t)co * 64 + ci) * NT + t] = acc[t][n][r];
      }
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = bsum[c];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (kq == 0) slab[64 * 64 * NT + 4 * i16 + c] = v;
    }
  }
}

// fixed-order sum of the slabs of one tile launch; taps [t0, t0+nt) of a layer with `taps` taps in all
__global__ __launch_bounds__(256) void wgrad_tile_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int nt, int t0, int taps,
                                                                float* __restrict__ dw, float* __restrict__ db, int cin, int co0,
                                                                int ci0, int write_bias) {
  __shared__ float part[4][64];
  const int slab_floats = 64 * 64 * nt + 64;
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float p = 0.f;
  if (i < slab_floats)
    for (int k = g; k < n_slabs; k += 4) p += slabs[(size_t)k * slab_floats + i];
  part[g][o] = p;
  __syncthreads();
  if (g == 0 && i < slab_floats) {
    const float s = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
    if (i < 64 * 64 * nt) {
      const int co = i / (64 * nt), r = i - co * 64 * nt, ci = r / nt, t = r - ci * nt;
      dw[((size_t)(co0 + co) * cin + ci0 + ci) * taps + t0 + t] = s;
    } else if (write_bias) {
      db[co0 + i - 64 * 64 * nt] = s;
    }
  }
}

template <int KS, int TY0, int NTY>
static int launch_tile_part(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, float* db,
                            int cin_total, int co0, int ci0, int g_quads, int g_quad0, int a_quads, int a_quad0, bool write_bias,
                            hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_tile_kernel<KS, TY0, NTY>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024));
    attr_set = true;
  }
  constexpr int NT = NTY * KS;
  hipLaunchKernelGGL((wgrad_tile_kernel<KS, TY0, NTY>), dim3(batch, esplit), dim3(256), kWgradLds, stream, table_dev, n_eval, esplit,
                     slabs, g_quad0, g_quads, a_quad0, a_quads);
  const int sf = 64 * 64 * NT + 64;
  hipLaunchKernelGGL(wgrad_tile_reduce_kernel, dim3((sf + 63) / 64), dim3(256), 0, stream, slabs, batch * esplit, NT, TY0 * KS,
                     KS * KS, dw, db, cin_total, co0, ci0, (int)(write_bias && TY0 == 0));
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

// One 64 x 64 tile of dW (cout_total, cin_total, ks, ks): output channels co0.. from G tensors with g_quads quads per sample
// (tile at quad g_quad0), input channels ci0.. of the WEIGHT from A tensors with a_quads quads per sample (tile at a_quad0) --
// the A tensor may be one half of a concatenated conv input.  slabs: batch*esplit*(64*64*10+64) floats (ks == 5:
// (batch * wgrad_esplit_max(batch) + 1) * kWgradSlabFloats).
int launch_wgrad_tile(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, float* db, int ks,
                      int cin_total, int co0, int ci0, int g_quads, int g_quad0, int a_quads, int a_quad0, bool write_bias,
                      hipStream_t stream) {
  int rc;
  if (ks == 5) {   // fp32 5x5: the Winograd-domain kernel (36 instead of 100 multiplies per 2x2 outputs, wgrad_wino5.hip) unless switched off
    // its work units are (evaluation, chunk of 16 tiles) pairs: up to 4 n_eval splits are useful (small batches; the 5x5 callers' slab
    // areas hold batch * wgrad_esplit_max(batch) + 1 slabs)
    const int e5 = wgrad_esplit(batch, 4 * n_eval);
    rc = launch_wgrad_wino5(table_dev, n_eval, batch, e5 > esplit ? e5 : esplit, slabs, dw, db, cin_total, co0, ci0, g_quads, g_quad0, a_quads, a_quad0, write_bias,
                            stream);
    if (rc != 1) return rc;
  }
  if (ks == 1) return launch_tile_part<1, 0, 1>(table_dev, n_eval, batch, esplit, slabs, dw, db, cin_total, co0, ci0, g_quads, g_quad0,
                                                a_quads, a_quad0, write_bias, stream);
  if (ks == 3) return launch_tile_part<3, 0, 3>(table_dev, n_eval, batch, esplit, slabs, dw, db, cin_total, co0, ci0, g_quads, g_quad0,
                                                a_quads, a_quad0, write_bias, stream);
  ODEHIP_REQUIRE(ks == 5, "wgrad: kernel size %d unsupported", ks);
  rc = launch_tile_part<5, 0, 2>(table_dev, n_eval, batch, esplit, slabs, dw, db, cin_total, co0, ci0, g_quads, g_quad0, a_quads, a_quad0,
                                 write_bias, stream);
  if (rc != ODEHIP_OK) return rc;
  rc = launch_tile_part<5, 2, 2>(table_dev, n_eval, batch, esplit, slabs, dw, db, cin_total, co0, ci0, g_quads, g_quad0, a_quads, a_quad0,
                                 write_bias, stream);
  if (rc != ODEHIP_OK) return rc;
  return launch_tile_part<5, 4, 1>(table_dev, n_eval, batch, esplit, slabs, dw, db, cin_total, co0, ci0, g_quads, g_quad0, a_quads,
                                   a_quad0, write_bias, stream);
}

int launch_wgrad_bf16(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, float* db, int cout,
                      int cin, hipStream_t stream);

int launch_wgrad(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, float* db, int cout,
                 int cin, hipStream_t stream, bool bf16) {
  if (bf16) return launch_wgrad_bf16(table_dev, n_eval, batch, esplit, slabs, dw, db, cout, cin, stream);
  {  // fp32: the Winograd-domain kernel (2.25x fewer multiplies, wgrad_wino.hip) unless switched off
    const int rw = launch_wgrad_wino(table_dev, n_eval, batch, esplit, slabs, dw, db, cout, cin, stream);
    if (rw != 1) return rw;
  }
  static bool attr_set = false;
  if (!attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int sf = 64 * 64 * 9 + 64;
  for (int co0 = 0; co0 < cout; co0 += 64)
    for (int ci0 = 0; ci0 < cin; ci0 += 64) {
      hipLaunchKernelGGL(wgrad64_kernel, dim3(batch, esplit), dim3(256), kWgradLds, stream, table_dev, n_eval, esplit, slabs,
                         co0 / 4, cout / 4, ci0 / 4, cin / 4);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((sf + 63) / 64), dim3(256), 0, stream, slabs, batch * esplit, sf, dw, db, cin,
                         co0, ci0);
    }
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}


// =====================================================================================================================
// bf16 variant (BASELINE.json configs[4]: bf16 compute): dW = sum G^T A with G and A rounded to bf16, fp32 accumulation.
// The contraction index of a weight gradient is the PIXEL, but the activations are pixel-major ([pixel][channel], what the
// forward kernels want), so both MFMA operands are column reads of their LDS tiles: gfx950's transposing LDS read
// (ds_read_b64_tr_b16: a 4 x 16 block delivered column-major) makes them directly, no transposed copy exists.
//   LDS: G tile [256 px][64 co] and A tile [18][18][64 ci] (zero border = the conv padding), bf16, 192 B per pixel (the
//   128 data bytes + 64: the four pixel rows of a transposed read then fall on disjoint banks).
//   One K-step = one image row (16 pixels); wave w owns the 32(co) x 32(ci) block (w & 1, w >> 1) of all 9 taps:
//   2 + 18 transposed reads feed 9 v_mfma_f32_32x32x16_bf16.  The next evaluation's 128 KiB of fp32 activations are loaded
//   into registers while the current one is multiplied (global latency hidden behind the MFMAs), then rounded into LDS.
// Same slab output and fixed-order reduction as the fp32 kernel: deterministic.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
typedef float f32x16w __attribute__((ext_vector_type(16)));

constexpr int kWS = 192;                       // bytes per pixel in the bf16 tiles
constexpr int kWG = 256 * kWS;                 // G tile
constexpr int kWA = 18 * 18 * kWS;             // A tile
constexpr int kWgradBf16Lds = kWG + kWA;

__device__ __forceinline__ unsigned pkw(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ bf16x8w tr_pair(const char* p) {  // two transposed reads: 8 consecutive pixel rows of this lane's column
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * kWS));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8w, v);
}

__global__ __launch_bounds__(256, 1) void wgrad64_bf16_kernel(const WgradPair* __restrict__ table, int n_eval, int esplit,
                                                              float* __restrict__ slabs, int g_quad0, int g_quads, int a_quad0,
                                                              int a_quads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const gt = smem;
  char* const at = smem + kWG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, es = blockIdx.y;
  const int mb = wave & 1, nb = wave >> 1;

  // zero border of the A tile (68 pixels x 128 data bytes)
  for (int i = tid; i < 68 * 8; i += 256) {
    const int p = i >> 3, c16 = i & 7;
    int row, col;
    if (p < 18) { row = 0; col = p; }
    else if (p < 36) { row = 17; col = p - 18; }
    else if (p < 52) { row = p - 36 + 1; col = 0; }
    else { row = p - 52 + 1; col = 17; }
    *(f32x4*)(at + (row * 18 + col) * kWS + c16 * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  f32x16w acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  f32x4 bsum[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bsum[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane bases of the transposed reads: lane 4q+p of a 16-lane group addresses pixel row q, channels 4p..4p+3 of the group's 16
  const int grp = lane >> 4, l16 = lane & 15, q = l16 >> 2, p4 = l16 & 3, h = lane >> 5;
  const char* gbase = gt + (8 * h + q) * kWS + (mb * 32 + 16 * (grp & 1) + 4 * p4) * 2;
  const char* abase = at + (8 * h + q) * kWS + (nb * 32 + 16 * (grp & 1) + 4 * p4) * 2;  // + (row*18 + col offset) * kWS per tap

  f32x4 gv[16], av[16];
  float esc = 0.0f;
  auto prefetch = [&](int e) {
    const WgradPair pr = table[e];
    esc = pr.scale;
    const f32x4* g = (const f32x4*)(pr.g + ((size_t)b * g_quads + g_quad0) * 4 * kPix) + tid;
    const f32x4* a = (const f32x4*)(pr.a + ((size_t)b * a_quads + a_quad0) * 4 * kPix) + tid;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      gv[i] = g[i * 256];
      av[i] = a[i * 256];
    }
  };
  if (es < n_eval) prefetch(es);
  for (int e = es; e < n_eval; e += esplit) {
    __syncthreads();  // every wave is done with the previous evaluation's tiles
    {
      const int prow = tid >> 4, pcol = tid & 15;
#pragma unroll
      for (int i = 0; i < 16; ++i) {   // quad i of pixel tid
        const f32x4 g = gv[i] * esc;
        bsum[i] += g;
        *(u32x2w*)(gt + tid * kWS + i * 8) = u32x2w{pkw(g.x, g.y), pkw(g.z, g.w)};
        *(u32x2w*)(at + ((prow + 1) * 18 + pcol + 1) * kWS + i * 8) = u32x2w{pkw(av[i].x, av[i].y), pkw(av[i].z, av[i].w)};
      }
    }
    __syncthreads();
    if (e + esplit < n_eval) prefetch(e + esplit);  // in flight while this evaluation is multiplied
#pragma unroll 2
    for (int y = 0; y < 16; ++y) {
      const bf16x8w gf = tr_pair(gbase + y * 16 * kWS);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = t / 3, dx = t % 3;  // tile coordinates: row y + dy, column x + dx (border included)
        const bf16x8w af = tr_pair(abase + ((y + dy) * 18 + dx) * kWS);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf, af, acc[t], 0, 0, 0);
      }
    }
  }

  float* slab = slabs + (size_t)(b * esplit + es) * (64 * 64 * 9 + 64);
  {
    const int n = lane & 31;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int ci = nb * 32 + n;
        slab[((size_t)co * 64 + ci) * 9 + t] = acc[t][r];
      }
  }
  // bias gradient: thread = pixel, bsum[i] = its four channels of quad i; fold the 256 pixels
  __syncthreads();
  float* red = (float*)smem;  // 4 waves x 64 channels
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = bsum[i][c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) red[wave * 64 + 4 * i + c] = v;
    }
  __syncthreads();
  if (tid < 64) slab[64 * 64 * 9 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
}

// ---- the same with operands that are already bf16 (Q4h), for the whole-trajectory bf16 training path
__global__ __launch_bounds__(256, 1) void wgrad64_q4h_kernel(const WgradPair* __restrict__ table, int n_eval, int esplit,
                                                              float* __restrict__ slabs, int g_quad0, int g_quads, int a_quad0,
                                                              int a_quads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const gt = smem;
  char* const at = smem + kWG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, es = blockIdx.y;
  const int mb = wave & 1, nb = wave >> 1;

  // zero border of the A tile (68 pixels x 128 data bytes)
  for (int i = tid; i < 68 * 8; i += 256) {
    const int p = i >> 3, c16 = i & 7;
    int row, col;
    if (p < 18) { row = 0; col = p; }
    else if (p < 36) { row = 17; col = p - 18; }
    else if (p < 52) { row = p - 36 + 1; col = 0; }
    else { row = p - 52 + 1; col = 17; }
    *(f32x4*)(at + (row * 18 + col) * kWS + c16 * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  f32x16w acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  // Tile layout: 192 B per pixel slot (conflict-free transposed reads), and inside a slot the four 16-channel groups are XOR-swizzled
  // with f(slot) = (slot >> 2) & 3.  Why: the tiles are FILLED from Q4h ([quad][pixel], 8 B), and with one pixel per thread the 64
  // lanes of a ds_write_b64 are 192 B apart = only 4 distinct bank groups -- a 16-way conflict that cost 40 % of the kernel
  // (ablation: 534 -> 316 us without the fill).  Filling 16 consecutive pixels x 4 quads per wave keeps the global loads in 128-B
  // runs, and the swizzle spreads pixels p, p+4, p+8, p+12 (same bank group at a 192-B stride) over the four channel groups:
  // 2 accesses per bank, the minimum for 512 B.
  const int grp = lane >> 4, l16 = lane & 15, q = l16 >> 2, p4 = l16 & 3, h = lane >> 5;
  auto chan_off = [&](int group16, int slot) { return (4 * (group16 ^ ((slot >> 2) & 3)) + p4) * 8; };   // byte offset of this lane's quad
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4w;
  auto tr_two = [&](const char* p0, const char* p1) {   // two transposed reads (pixels 8h+q and 8h+q+4 of a 16-pixel run)
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4w*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4w*)p1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8w, v);
  };
  // G: slot = pixel; row y starts at slot 16 y, so f only depends on the lane: (2h) & 3 for the first read, (2h + 1) & 3 for the second
  const int gs0 = 8 * h + q;
  const char* gbase0 = gt + gs0 * kWS + chan_off(2 * mb + (grp & 1), gs0);
  const char* gbase1 = gt + (gs0 + 4) * kWS + chan_off(2 * mb + (grp & 1), gs0 + 4);

  // operands are ALREADY bf16 ("Q4h": [sample][quad][pixel] x 4 bf16 = 8 bytes; written by ftraj_bf16_kernel<RK4, SAVE> and
  // btraj_bf16_rk4_kernel): half the HBM bytes of the fp32 kernel above and no conversion.  Thread (wave w, lane) moves quad
  // 4 w + (lane >> 4) of pixel 16 j + (lane & 15) in iteration j.
  const int fq = 4 * wave + (lane >> 4), fx = lane & 15;
  u32x2w gvA[16], avA[16];
  auto prefetch = [&](const WgradPair& pr, u32x2w (&gv)[16], u32x2w (&av)[16]) {
    const u32x2w* g = (const u32x2w*)pr.g + ((size_t)b * g_quads + g_quad0 + fq) * kPix + fx;
    const u32x2w* a = (const u32x2w*)pr.a + ((size_t)b * a_quads + a_quad0 + fq) * kPix + fx;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      gv[j] = g[j * 16];
      av[j] = a[j * 16];
    }
  };
  auto fill = [&](const u32x2w (&gv)[16], const u32x2w (&av)[16]) {
    const int gq = (fq ^ (4 * ((fx >> 2) & 3))) * 8;       // G: slot = 16 j + fx, f = (fx >> 2) & 3 for every j
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      *(u32x2w*)(gt + (j * 16 + fx) * kWS + gq) = gv[j];
      const int s = (j + 1) * 18 + fx + 1;                 // A: image pixel (j, fx) sits in tile slot (j + 1, fx + 1)
      *(u32x2w*)(at + s * kWS + (fq ^ (4 * ((s >> 2) & 3))) * 8) = av[j];
    }
  };
  auto multiply = [&]() {
  // The three taps of a kernel row are the SAME image row shifted by one pixel along K (= the pixel index of the MFMA), and tile
  // row r serves output rows r, r-1, r-2 (dy = 0, 1, 2): every tile row is read ONCE (2 transposed reads) -- K element j of lane
  // half h = tile column 1 + 8h + j -- and the dx = 0 / 2 operands are derived in registers: a 16-bit funnel shift of the lane's 8
  // values, the value that crosses the half boundary comes from lane ^ 32, the one that crosses the row's end is the zero
  // padding.  18 x 2 transposed reads + 18 lane exchanges per evaluation instead of 16 x 18 x 2 reads.
  // Same operands and the same accumulation order per tap (y ascending) as wgrad64_bf16_kernel: bit-identical results.
  typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
  bf16x8w gprev[2];   // G fragments of output rows r-1, r-2
#pragma unroll 6
  for (int r = 0; r < 18; ++r) {
    const int as0 = r * 18 + 1 + 8 * h + q;
    const bf16x8w ac = tr_two(at + as0 * kWS + chan_off(2 * nb + (grp & 1), as0), at + (as0 + 4) * kWS + chan_off(2 * nb + (grp & 1), as0 + 4));
    const u32x4w d = __builtin_bit_cast(u32x4w, ac);
    const unsigned recv = (unsigned)__shfl_xor((int)(h ? d[0] : d[3]), 32, 64);   // h = 0 sends its element 7, h = 1 its element 0
    const unsigned xr = h ? 0u : recv;   // low 16 bits: element 0 of the upper half = the pixel right of this lane's eight
    const unsigned xl = h ? recv : 0u;   // high 16 bits: element 7 of the lower half = the pixel left of this lane's eight
    const u32x4w dl = {__builtin_amdgcn_alignbit(d[0], xl, 16), __builtin_amdgcn_alignbit(d[1], d[0], 16),
                       __builtin_amdgcn_alignbit(d[2], d[1], 16), __builtin_amdgcn_alignbit(d[3], d[2], 16)};   // tile column k + 0
    const u32x4w dr = {__builtin_amdgcn_alignbit(d[1], d[0], 16), __builtin_amdgcn_alignbit(d[2], d[1], 16),
                       __builtin_amdgcn_alignbit(d[3], d[2], 16), __builtin_amdgcn_alignbit(xr, d[3], 16)};     // tile column k + 2
    const bf16x8w al = __builtin_bit_cast(bf16x8w, dl), ar = __builtin_bit_cast(bf16x8w, dr);
    bf16x8w gcur = ac;
    if (r < 16) gcur = tr_two(gbase0 + r * 16 * kWS, gbase1 + r * 16 * kWS);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {       // output row y = r - dy
      const int y = r - dy;
      if (y < 0 || y > 15) continue;
      const bf16x8w gf = dy == 0 ? gcur : gprev[dy - 1];
      acc[3 * dy] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf, al, acc[3 * dy], 0, 0, 0);
      acc[3 * dy + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf, ac, acc[3 * dy + 1], 0, 0, 0);
      acc[3 * dy + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf, ar, acc[3 * dy + 2], 0, 0, 0);
    }
    gprev[1] = gprev[0];
    gprev[0] = gcur;
  }
  };
  // The table entry of an evaluation is fetched one evaluation BEFORE its operands are requested: read where it is needed it puts a
  // dependent trip to memory (entry -> addresses -> operands) in front of every operand load
  WgradPair pr_next;
  pr_next.g = pr_next.a = nullptr;
  if (es < n_eval) {
    prefetch(table[es], gvA, avA);
    if (es + esplit < n_eval) pr_next = table[es + esplit];
  }
  for (int e = es; e < n_eval; e += esplit) {
    __syncthreads();  // every wave is done with the previous evaluation's tiles
    fill(gvA, avA);
    __syncthreads();
    if (e + esplit < n_eval) {
      prefetch(pr_next, gvA, avA);                                   // in flight while this evaluation is multiplied
      if (e + 2 * esplit < n_eval) pr_next = table[e + 2 * esplit];
    }
    __builtin_amdgcn_sched_barrier(0);                               // the loads are issued before the multiply, not sunk below it
    multiply();
  }

  float* slab = slabs + (size_t)(b * esplit + es) * (64 * 64 * 9 + 64);
  {
    const int n = lane & 31;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int ci = nb * 32 + n;
        slab[((size_t)co * 64 + ci) * 9 + t] = acc[t][r];
      }
  }
  if (tid < 64) slab[64 * 64 * 9 + tid] = 0.0f;   // the bias gradient of this path comes from the backward sweep (unrounded fp32)
}


// ---- the same for the 5x5 convs of the ConvGRU cell: tap rows [TY0, TY0+NTY) per launch (10 + 10 + 5 taps) so the accumulators
// fit; A tile [20][20][64] with a 2-pixel zero border.  Slab layout and reduction of wgrad_tile_kernel.
template <int KS, int TY0, int NTY>
__global__ __launch_bounds__(256, 1) void wgrad_tile_bf16_kernel(const WgradPair* __restrict__ table, int n_eval, int esplit,
                                                                 float* __restrict__ slabs, int g_quad0, int g_quads, int a_quad0,
                                                                 int a_quads) {
  constexpr int HALO = KS / 2, NT = NTY * KS, W = 16 + 2 * HALO;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const gt = smem;
  char* const at = smem + kWG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, es = blockIdx.y;
  const int mb = wave & 1, nb = wave >> 1;

  for (int i = tid; i < W * W * 8; i += 256) {  // zero the tile's data bytes once; the interior is rewritten per evaluation
    const int p = i >> 3, c16 = i & 7;
    *(f32x4*)(at + p * kWS + c16 * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  f32x16w acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  f32x4 bsum[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bsum[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int grp = lane >> 4, l16 = lane & 15, q = l16 >> 2, p4 = l16 & 3, h = lane >> 5;
  const char* gbase = gt + (8 * h + q) * kWS + (mb * 32 + 16 * (grp & 1) + 4 * p4) * 2;
  const char* abase = at + (8 * h + q) * kWS + (nb * 32 + 16 * (grp & 1) + 4 * p4) * 2;

  f32x4 gv[16], av[16];
  float esc = 0.0f;
  auto prefetch = [&](int e) {
    const WgradPair pr = table[e];
    esc = pr.scale;
    const f32x4* g = (const f32x4*)(pr.g + ((size_t)b * g_quads + g_quad0) * 4 * kPix) + tid;
    const f32x4* a = (const f32x4*)(pr.a + ((size_t)b * a_quads + a_quad0) * 4 * kPix) + tid;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      gv[i] = g[i * 256];
      av[i] = a[i * 256];
    }
  };
  if (es < n_eval) prefetch(es);
  for (int e = es; e < n_eval; e += esplit) {
    __syncthreads();
    {
      const int prow = tid >> 4, pcol = tid & 15;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f32x4 g = gv[i] * esc;
        if (TY0 == 0) bsum[i] += g;
        *(u32x2w*)(gt + tid * kWS + i * 8) = u32x2w{pkw(g.x, g.y), pkw(g.z, g.w)};
        *(u32x2w*)(at + ((prow + HALO) * W + pcol + HALO) * kWS + i * 8) = u32x2w{pkw(av[i].x, av[i].y), pkw(av[i].z, av[i].w)};
      }
    }
    __syncthreads();
    if (e + esplit < n_eval) prefetch(e + esplit);
    for (int y = 0; y < 16; ++y) {
      const bf16x8w gf = tr_pair(gbase + y * 16 * kWS);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int dy = TY0 + t / KS, dx = t % KS;  // tile coordinates (border included)
        const bf16x8w af = tr_pair(abase + ((y + dy) * W + dx) * kWS);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf, af, acc[t], 0, 0, 0);
      }
    }
  }

  float* slab = slabs + (size_t)(b * esplit + es) * (64 * 64 * NT + 64);
  {
    const int n = lane & 31;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int ci = nb * 32 + n;
        slab[((size_t)co * 64 + ci) * NT + t] = acc[t][r];
      }
  }
  __syncthreads();
  float* red = (float*)smem;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = bsum[i][c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) red[wave * 64 + 4 * i + c] = v;
    }
  __syncthreads();
  if (tid < 64) slab[64 * 64 * NT + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
}

template <int KS, int TY0, int NTY>
static int launch_tile_bf16_part(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, float* db,
                                 int cin_total, int co0, int ci0, int g_quads, int g_quad0, int a_quads, int a_quad0, bool write_bias,
                                 hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_tile_bf16_kernel<KS, TY0, NTY>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024));
    attr_set = true;
  }
  constexpr int NT = NTY * KS, W = 16 + 2 * (KS / 2);
  hipLaunchKernelGGL((wgrad_tile_bf16_kernel<KS, TY0, NTY>), dim3(batch, esplit), dim3(256), kWG + W * W * kWS, stream, table_dev, n_eval,
                     esplit, slabs, g_quad0, g_quads, a_quad0, a_quads);
  const int sf = 64 * 64 * NT + 64;
  hipLaunchKernelGGL(wgrad_tile_reduce_kernel, dim3((sf + 63) / 64), dim3(256), 0, stream, slabs, batch * esplit, NT, TY0 * KS,
                     KS * KS, dw, db, cin_total, co0, ci0, (int)(write_bias && TY0 == 0));
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

// bf16 variant of launch_wgrad_tile for 5x5 layers; slabs: batch*esplit*(64*64*10+64) floats
int launch_wgrad_tile_bf16_5x5(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, float* db,
                               int cin_total, int co0, int ci0, int g_quads, int g_quad0, int a_quads, int a_quad0, bool write_bias,
                               hipStream_t stream) {
  int rc = launch_tile_bf16_part<5, 0, 2>(table_dev, n_eval, batch, esplit, slabs, dw, db, cin_total, co0, ci0, g_quads, g_quad0, a_quads,
                                          a_quad0, write_bias, stream);
  if (rc != ODEHIP_OK) return rc;
  rc = launch_tile_bf16_part<5, 2, 2>(table_dev, n_eval, batch, esplit, slabs, dw, db, cin_total, co0, ci0, g_quads, g_quad0, a_quads,
                                      a_quad0, write_bias, stream);
  if (rc != ODEHIP_OK) return rc;
  return launch_tile_bf16_part<5, 4, 1>(table_dev, n_eval, batch, esplit, slabs, dw, db, cin_total, co0, ci0, g_quads, g_quad0, a_quads,
                                        a_quad0, write_bias, stream);
}

int launch_wgrad_bf16(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, float* db, int cout,
                      int cin, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad64_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int sf = 64 * 64 * 9 + 64;
  for (int co0 = 0; co0 < cout; co0 += 64)
    for (int ci0 = 0; ci0 < cin; ci0 += 64) {
      hipLaunchKernelGGL(wgrad64_bf16_kernel, dim3(batch, esplit), dim3(256), kWgradBf16Lds, stream, table_dev, n_eval, esplit, slabs,
                         co0 / 4, cout / 4, ci0 / 4, cin / 4);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((sf + 63) / 64), dim3(256), 0, stream, slabs, batch * esplit, sf, dw, db, cin,
                         co0, ci0);
    }
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

// dw (+)= sum of the slabs' weight part, fixed order (the bias part of a Q4h slab is zero: the reverse sweep sums the bias gradients)
__global__ __launch_bounds__(256) void wgrad_reduce_acc_kernel(const float* __restrict__ slabs, int n_slabs, int slab_floats,
                                                               float* __restrict__ dw, int accumulate) {
  __shared__ float part[4][64];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
  if (i < 64 * 64 * 9) {
    for (int k = g; k < n_slabs; k += 16) {
      p0 += slabs[(size_t)k * slab_floats + i];
      if (k + 4 < n_slabs) p1 += slabs[(size_t)(k + 4) * slab_floats + i];
      if (k + 8 < n_slabs) p2 += slabs[(size_t)(k + 8) * slab_floats + i];
      if (k + 12 < n_slabs) p3 += slabs[(size_t)(k + 12) * slab_floats + i];
    }
  }
  part[g][o] = (p0 + p1) + (p2 + p3);
  __syncthreads();
  if (g == 0 && i < 64 * 64 * 9) {
    const float s = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
    dw[i] = accumulate ? dw[i] + s : s;   // a 64 x 64 layer: the slab's (co, ci, tap) order IS the OIHW order
  }
}

// operands in Q4h (bf16), one 64 -> 64 layer; dw = (accumulate ? dw : 0) + sum over the table's evaluations.  The bias gradient is
// NOT produced here (the reverse sweep sums it from the unrounded gradients).
int launch_wgrad_q4h(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, int accumulate,
                     hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad64_q4h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int sf = 64 * 64 * 9 + 64;
  hipLaunchKernelGGL(wgrad64_q4h_kernel, dim3(batch, esplit), dim3(256), kWgradBf16Lds, stream, table_dev, n_eval, esplit, slabs, 0, 16, 0, 16);
  hipLaunchKernelGGL(wgrad_reduce_acc_kernel, dim3((64 * 64 * 9 + 63) / 64), dim3(256), 0, stream, slabs, batch * esplit, sf, dw, accumulate);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

}  // namespace odehip
