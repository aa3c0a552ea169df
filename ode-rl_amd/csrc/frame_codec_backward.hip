// frame_codec_backward.hip -- backward of the fused frame encoder / decoder (frame_codec.hip):
// the gradients a training step of /root/reference/models/ODEConvGRU.py:101-140 needs from
//
//   Encoder  Conv2d(1, 16, 3, 2, 1) -> LeakyReLU -> Conv2d(16, C, 3, 2, 1) -> LeakyReLU          dW1 db1 dW2 db2   (frames carry no gradient)
//   Decoder  ConvTranspose2d(C, 32, 4, 2, 1) -> LeakyReLU -> ConvTranspose2d(32, 1, 4, 2, 1) -> sigmoid     d latents, dW1 db1 dW2 db2
//
// for C = 32 or 64 latent channels and one frame channel.  Under autograd these were eight library launches plus layout
// transposes and elementwise kernels (1.2 ms + ~0.5 ms of a 13.5 ms training step at B = 64, 10 + 10 frames); here (0.43 ms):
//
//   frame_decode_bwd_mid_kernel   per (image, quarter): the 32-channel intermediate as an LDS tile -- read back from the copy the training
//                                 forward saved (odehip_frame_decode_train; 161 -> 49 us per 640 images), or RECOMPUTED from the latents
//                                 by the forward's own routine (dec_mid_to_lds) when nothing but inputs and outputs is to be kept --,
//                                 g = dL/d(pre-sigmoid) staged as an [18][66] tile; each thread owns one intermediate
//                                 pixel: its gradient through the 4x4 window of g (VALU, weights as scalar loads), LeakyReLU mask,
//                                 written once to HBM as channel quads; dW2 as an MFMA product (k = pixels; a third block with A = 1
//                                 yields db2).
//   frame_decode_bwd_lat_kernel   per half image: that gradient as an [18][34][36] LDS tile + the latents' eight rows; d latents =
//                                 a stride-2 4x4 convolution on the MFMA (k = channels, quad operand trick of frame_codec.hip), dW1 on
//                                 the MFMA with k = pixels (wave w: taps 2w, 2w + 1; the four centre taps with A = 1 yield db1).
//   frame_encode_bwd_kernel       per frame: conv1 recomputed in LDS (enc_conv1_to_lds), g2 = dL/d(conv2 output) masked into LDS;
//                                 dW2 on the MFMA (k = pixels), d(intermediate) on the MFMA output-stationary per parity class
//                                 (k = channels), mask, dW1 / db1 on the VALU.
//
// All three are persistent (<= one or two workgroups per CU walk the images) with the weight gradients in registers across images;
// one slab per workgroup at the end and a fixed-order sum (codec_slab_sum_kernel): bitwise reproducible, no float atomics.
#include "frame_codec.h"

namespace odehip {

typedef const __attribute__((address_space(4))) float CodecConstF;

// 0, but opaque to the compiler: added to the base of wave-uniform weight loads inside a persistent loop so that they are issued where
// they are used (hoisted out of the loop as loop invariants, hundreds of scalars would be live across it and spill)
__device__ __forceinline__ int opaque_zero() {
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
}

// out[i] = sum_k slabs[k * stride + i] in a fixed order; i < n0 goes to out0, the next n1 to out1, ... (the natural parameter layouts)
struct SlabSumArgs {
  const float* slabs;
  int n_slabs, stride;
  float* out[4];
  int n[4];
};

__global__ __launch_bounds__(1024) void codec_slab_sum_kernel(const SlabSumArgs a) {
  __shared__ float part[16][64];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  const int total = a.n[0] + a.n[1] + a.n[2] + a.n[3];
  float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
  if (i < total) {
    for (int k = g; k < a.n_slabs; k += 64) {
      p0 += a.slabs[(size_t)k * a.stride + i];
      if (k + 16 < a.n_slabs) p1 += a.slabs[(size_t)(k + 16) * a.stride + i];
      if (k + 32 < a.n_slabs) p2 += a.slabs[(size_t)(k + 32) * a.stride + i];
      if (k + 48 < a.n_slabs) p3 += a.slabs[(size_t)(k + 48) * a.stride + i];
    }
  }
  part[g][o] = (p0 + p1) + (p2 + p3);
  __syncthreads();
  if (g == 0 && i < total) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += part[k][o];
    int j = i;
    if (j < a.n[0]) { a.out[0][j] = s; return; }
    j -= a.n[0];
    if (j < a.n[1]) { a.out[1][j] = s; return; }
    j -= a.n[1];
    if (j < a.n[2]) { a.out[2][j] = s; return; }
    a.out[3][j - a.n[2]] = s;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s, 64);
  return v;
}

// =============================================================================================================== decoder, part 1
struct DecBwdMidArgs {
  const float* pack;     // the forward's decoder pack
  const float* latents;  // (N, in_ch, 16, 16)
  const float* mid;      // SAVED: [N][quad 8][32][32] x 4, the intermediate after its LeakyReLU as odehip_frame_decode_train saved it
  const float* pred;     // (N, 1, 64, 64): the forward's output (after the sigmoid if `sigmoid`)
  const float* g_out;    // (N, 1, 64, 64): dL/d pred
  float* gmid;           // [N][quad 8][32][32] x 4: dL/d(intermediate before its LeakyReLU)
  float* slabs;          // per workgroup: dW2 [32][16] | db2
  int n_images, sigmoid, slab_stride;
  float slope;
};

constexpr int kGzW = 66;                 // g tile: rows 16q-1 .. 16q+16, columns -1 .. 64
constexpr int kDecBwdSlab1 = 576;        // >= 513 floats

// SAVED: the intermediate comes from HBM (this unit's eight rows as an [8][32][36] LDS tile: 41 KiB with g, three workgroups per CU);
// otherwise it is recomputed from the latents (z + [10][34][36] tile, two per CU).
template <int G, bool SAVED>
__global__ __launch_bounds__(kCodecThreads, 2) void frame_decode_bwd_mid_kernel(const DecBwdMidArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int in_ch = G * 16;
  // pixel (row pr of the unit's eight, column mx) of the LDS tile: mid + pix0 + (pr * pixw + mx) * kMPix
  constexpr int pixw = SAVED ? kHalf : kMW, pix0 = SAVED ? 0 : (kMW + 1) * kMPix;
  f32x4* const z = (f32x4*)smem;
  float* const mid = (float*)(smem + (SAVED ? (size_t)0 : (size_t)4 * G * kZRows * kZW * 16));
  float* const gz = mid + (SAVED ? 8 * kHalf : kMRows * kMW) * kMPix;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, kq = lane >> 4;
  CodecConstF* const w2 = (CodecConstF*)(a.pack + dec_off_w2(in_ch));  // [parity 4][tap 4][o = 1][ci 32]

  f32x4 accw[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, accb = {0.f, 0.f, 0.f, 0.f};
  const int n_units = a.n_images * 4;
  for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
    const int n = u >> 2, q = u & 3;
    if (u != (int)blockIdx.x) __syncthreads();  // the previous unit's readers are done with z, mid, gz
    // ---- g = dL/d(layer-2 output before the sigmoid): rows 16q-1 .. 16q+16, zero outside the frame
    for (int i = tid; i < 18 * kGzW; i += kCodecThreads) {
      const int r = i / kGzW, c = i - r * kGzW;
      const int oy = 16 * q - 1 + r, ox = c - 1;
      float v = 0.0f;
      if (oy >= 0 && oy < kFrame && ox >= 0 && ox < kFrame) {
        const size_t at = (size_t)n * kFrame * kFrame + oy * kFrame + ox;
        v = a.g_out[at];
        if (a.sigmoid) { const float p = a.pred[at]; v *= p * (1.0f - p); }
      }
      gz[i] = v;
    }
    if (SAVED) {
      const f32x4* const src = (const f32x4*)a.mid + (((size_t)n * 8) * kHalf + 8 * q) * kHalf;
#pragma unroll
      for (int k = 0; k < 8; ++k) {   // 8 quads x 256 pixels: element i = quad * 256 + pixel
        const int i = tid + k * kCodecThreads;
        *(f32x4*)(mid + (i & 255) * kMPix + 4 * (i >> 8)) = src[(size_t)(i >> 8) * kHalf * kHalf + (i & 255)];
      }
    } else {
      dec_mid_to_lds<G>(a.pack, a.latents + (size_t)n * in_ch * 256, q, a.slope, z, mid, tid, lane, wave);
    }
    __syncthreads();

    // ---- this thread's intermediate pixel (8q + pr, mx): gradient through the 4x4 window of g, mask, one write
    {
      const int pr = tid >> 5, mx = tid & 31;
      float win[16];
#pragma unroll
      for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) win[ky * 4 + kx] = gz[(2 * pr + ky) * kGzW + 2 * mx + kx];
      float gacc[kDecMid];
#pragma unroll
      for (int c = 0; c < kDecMid; ++c) gacc[c] = 0.0f;
#pragma unroll
      for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          // kernel index k -> (parity, tap) of the pack: 0 -> (1,0), 1 -> (0,0), 2 -> (1,1), 3 -> (0,1)  (convt_tap)
          const int pa = (ky & 1) ^ 1, ta = ky >> 1, pb = (kx & 1) ^ 1, tb = kx >> 1;
          CodecConstF* const w = w2 + opaque_zero() + ((pa * 2 + pb) * 4 + ta * 2 + tb) * kDecMid;
#pragma unroll
          for (int c = 0; c < kDecMid; ++c) gacc[c] = __builtin_fmaf(win[ky * 4 + kx], w[c], gacc[c]);
        }
      const float* const mp = mid + pix0 + (pr * pixw + mx) * kMPix;
      f32x4* const dst = (f32x4*)a.gmid + (((size_t)n * 8) * kHalf + 8 * q + pr) * kHalf + mx;
#pragma unroll
      for (int cq = 0; cq < 8; ++cq) {
        const f32x4 mv = *(const f32x4*)(mp + 4 * cq);
        f32x4 gv = {gacc[4 * cq], gacc[4 * cq + 1], gacc[4 * cq + 2], gacc[4 * cq + 3]};
        gv.x *= mv.x > 0.0f ? 1.0f : a.slope;
        gv.y *= mv.y > 0.0f ? 1.0f : a.slope;
        gv.z *= mv.z > 0.0f ? 1.0f : a.slope;
        gv.w *= mv.w > 0.0f ? 1.0f : a.slope;
        dst[(size_t)cq * kHalf * kHalf] = gv;
      }
    }
    // ---- dW2[ci][ky][kx] += sum over this wave's 64 pixels of act[ci][pixel] * g[2 my - 1 + ky][2 mx - 1 + kx]  (MFMA, k = pixels);
    // with A = 1 the same product sums g itself: the taps (1|2, 1|2) cover every output pixel of the quarter once -> db2
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      const int p = 4 * s + kq, pr = 2 * wave + (p >> 5), mx = p & 31;
      const float* const mp = mid + pix0 + (pr * pixw + mx) * kMPix + m16;
      const float bv = gz[(2 * pr + (m16 >> 2)) * kGzW + 2 * mx + (m16 & 3)];
      accw[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(mp[0], bv, accw[0], 0, 0, 0);
      accw[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(mp[16], bv, accw[1], 0, 0, 0);
      accb = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, bv, accb, 0, 0, 0);
    }
  }
  // ---- one slab per workgroup: the four waves' accumulators through LDS (z is free)
  __syncthreads();
  float* const red = (float*)smem;  // [wave 4][513]
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave * 520 + (16 * blk + 4 * kq + i) * 16 + m16] = accw[blk][i];
  if (kq == 0 && (m16 == 5 || m16 == 6 || m16 == 9 || m16 == 10)) red[wave * 520 + 512 + (m16 == 5 ? 0 : (m16 == 6 ? 1 : (m16 == 9 ? 2 : 3)))] = accb[0];
  __syncthreads();
  float* const slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
  for (int i = tid; i < 512; i += kCodecThreads) slab[i] = (red[i] + red[520 + i]) + (red[1040 + i] + red[1560 + i]);
  if (tid == 0) {
    float s = 0.0f;
    for (int w = 0; w < 4; ++w) s += (red[w * 520 + 512] + red[w * 520 + 513]) + (red[w * 520 + 514] + red[w * 520 + 515]);
    slab[512] = s;
  }
}

// =============================================================================================================== decoder, part 2
struct DecBwdLatArgs {
  const float* a1t;      // [tap 16][g 2][kq 4][ci][j 4] = w1[ci][16 g + 4 kq + j][ky][kx], tap = 4 ky + kx
  const float* latents;  // (N, in_ch, 16, 16)
  const float* gmid;     // [N][quad 8][32][32] x 4
  float* g_latents;      // (N, in_ch, 16, 16)
  float* slabs;          // per workgroup: dW1 [ci][co 32][tap 16] | db1 [32]
  int n_images, slab_stride;
};

constexpr int kLatStride = 132;          // floats per channel of the latent tile (8 rows x 16 + 4: a 16-lane 16-byte read is conflict-free)
constexpr int kDecBwdThreads2 = 512;

__global__ void pack_a1t_kernel(const float* __restrict__ w1, int in_ch, float* __restrict__ dst) {
  const int n = 16 * 2 * 4 * in_ch * 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int e = i;
    const int j = e & 3; e >>= 2;
    const int ci = e % in_ch; e /= in_ch;
    const int kq = e & 3; e >>= 2;
    const int g = e & 1, tap = e >> 1;
    dst[i] = w1[((size_t)ci * kDecMid + 16 * g + 4 * kq + j) * 16 + tap];
  }
}

template <int G>
__global__ __launch_bounds__(kDecBwdThreads2, 1) void frame_decode_bwd_lat_kernel(const DecBwdLatArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int in_ch = G * 16;
  constexpr int NR = G;                       // latent rows per wave in the d-latents product (8 waves: G channel blocks x 8 / G row groups)
  float* const gm = (float*)smem;             // [18][34][36]: rows 16h-1 .. 16h+16, columns -1 .. 32
  float* const lat = gm + 18 * kMW * kMPix;   // [ci][8][16] (+4)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 15, kq = lane >> 4;

  f32x4 accw[2][G][2], accdb[2][2];
#pragma unroll
  for (int ts = 0; ts < 2; ++ts) {
#pragma unroll
    for (int cib = 0; cib < G; ++cib) accw[ts][cib][0] = accw[ts][cib][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    accdb[ts][0] = accdb[ts][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int n_units = a.n_images * 2;
  for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
    const int n = u >> 1, h = u & 1;
    if (u != (int)blockIdx.x) __syncthreads();
    for (int i = tid; i < 8 * 18 * kMW; i += kDecBwdThreads2) {
      const int cq = i / (18 * kMW), e = i - cq * (18 * kMW), r = e / kMW, c = e - r * kMW;
      const int my = 16 * h - 1 + r, mx = c - 1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (my >= 0 && my < kHalf && mx >= 0 && mx < kHalf) v = ((const f32x4*)a.gmid)[(((size_t)n * 8 + cq) * kHalf + my) * kHalf + mx];
      *(f32x4*)(gm + (r * kMW + c) * kMPix + 4 * cq) = v;
    }
    for (int i = tid; i < in_ch * 32; i += kDecBwdThreads2) {
      const int ci = i >> 5, e = i & 31;  // e = row * 4 + column quad
      *(f32x4*)(lat + ci * kLatStride + 4 * e) = *(const f32x4*)(a.latents + ((size_t)n * in_ch + ci) * 256 + 8 * h * 16 + 4 * e);
    }
    __syncthreads();

    // ---- d latents[ci][iy][ix] = sum_{co, ky, kx} g[co][2 iy - 1 + ky][2 ix - 1 + kx] w1[ci][co][ky][kx]: k = channels (16 per step)
    {
      const int cib = wave % G, r0 = (wave / G) * NR;
      f32x4 acc[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4* const ap = (const f32x4*)a.a1t + kq * in_ch + 16 * cib + nn;
      f32x4 af = ap[0], afn = af;
#pragma unroll 1
      for (int t = 0; t < 32; ++t) {  // t = 2 tap + g
        if (t + 1 < 32) afn = ap[(size_t)(t + 1) * 4 * in_ch];
        const int tap = t >> 1, g = t & 1, ky = tap >> 2, kx = tap & 3;
        const float* const bp = gm + ((2 * r0 + ky) * kMW + 2 * nn + kx) * kMPix + 16 * g + 4 * kq;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const f32x4 bf = *(const f32x4*)(bp + 2 * r * kMW * kMPix);
          acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bf.x, acc[r], 0, 0, 0);
          acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bf.y, acc[r], 0, 0, 0);
          acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bf.z, acc[r], 0, 0, 0);
          acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bf.w, acc[r], 0, 0, 0);
        }
        af = afn;
      }
      float* const dst = a.g_latents + ((size_t)n * in_ch + 16 * cib + 4 * kq) * 256 + (8 * h + r0) * 16 + nn;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        dst[r * 16] = acc[r].x; dst[256 + r * 16] = acc[r].y; dst[512 + r * 16] = acc[r].z; dst[768 + r * 16] = acc[r].w;
      }
    }
    // ---- dW1[ci][co][ky][kx] += sum_{iy, ix} latents[ci][iy][ix] g[co][2 iy - 1 + ky][2 ix - 1 + kx]: k = pixels (4 columns per
    // MFMA, a row per 4 MFMAs); this wave: taps 2 wave, 2 wave + 1.  A = 1 on the centre taps (1|2, 1|2) sums g itself -> db1
#pragma unroll 1
    for (int iy = 0; iy < 8; ++iy) {
      f32x4 af[G];
#pragma unroll
      for (int cib = 0; cib < G; ++cib) af[cib] = *(const f32x4*)(lat + (16 * cib + nn) * kLatStride + iy * 16 + 4 * kq);
#pragma unroll
      for (int ts = 0; ts < 2; ++ts) {
        const int tap = 2 * wave + ts, ky = tap >> 2, kx = tap & 3;
        const bool centre = (ky == 1 || ky == 2) && (kx == 1 || kx == 2);
        const float* const bp = gm + ((2 * iy + ky) * kMW + 8 * kq + kx) * kMPix + nn;
#pragma unroll
        for (int cob = 0; cob < 2; ++cob) {
          const float b0 = bp[16 * cob], b1 = bp[2 * kMPix + 16 * cob], b2 = bp[4 * kMPix + 16 * cob], b3 = bp[6 * kMPix + 16 * cob];
#pragma unroll
          for (int cib = 0; cib < G; ++cib) {
            accw[ts][cib][cob] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cib].x, b0, accw[ts][cib][cob], 0, 0, 0);
            accw[ts][cib][cob] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cib].y, b1, accw[ts][cib][cob], 0, 0, 0);
            accw[ts][cib][cob] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cib].z, b2, accw[ts][cib][cob], 0, 0, 0);
            accw[ts][cib][cob] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cib].w, b3, accw[ts][cib][cob], 0, 0, 0);
          }
          if (centre) {
            accdb[ts][cob] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, b0, accdb[ts][cob], 0, 0, 0);
            accdb[ts][cob] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, b1, accdb[ts][cob], 0, 0, 0);
            accdb[ts][cob] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, b2, accdb[ts][cob], 0, 0, 0);
            accdb[ts][cob] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, b3, accdb[ts][cob], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- the slab: dW1 in the parameter's own layout (in_ch, 32, 4, 4); db1 = the four centre taps (waves 2 .. 5) through LDS
  float* const slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
#pragma unroll
  for (int ts = 0; ts < 2; ++ts)
#pragma unroll
    for (int cib = 0; cib < G; ++cib)
#pragma unroll
      for (int cob = 0; cob < 2; ++cob)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          slab[((size_t)(16 * cib + 4 * kq + i) * kDecMid + 16 * cob + nn) * 16 + 2 * wave + ts] = accw[ts][cib][cob][i];
  __syncthreads();
  float* const red = (float*)smem;  // [centre tap 4][co 32]
  if (kq == 0) {
#pragma unroll
    for (int ts = 0; ts < 2; ++ts) {
      const int tap = 2 * wave + ts, ky = tap >> 2, kx = tap & 3;
      if ((ky == 1 || ky == 2) && (kx == 1 || kx == 2)) {
        const int slot = (ky - 1) * 2 + (kx - 1);
        red[slot * 32 + nn] = accdb[ts][0][0];
        red[slot * 32 + 16 + nn] = accdb[ts][1][0];
      }
    }
  }
  __syncthreads();
  if (tid < kDecMid) slab[(size_t)in_ch * 512 + tid] = (red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]);
}

// ====================================================================================================================== encoder
struct EncBwdArgs {
  const float* pack;     // the forward's encoder pack (w1, b1 in front)
  const float* a2t;      // [tap 9][t out_ch/4][kq 4][ci 16] = w2[16 (t >> 2) + 4 kq + (t & 3)][ci][tap]
  const float* frames;   // (B, T, 1, 64, 64)
  const float* out;      // (T, B, out_ch, 16, 16): the forward's output (its sign is the second LeakyReLU's mask)
  const float* g_out;    // (T, B, out_ch, 16, 16)
  float* slabs;          // per workgroup: dW2 [co][ci 16][tap 9] | db2 [co] | dW1 [c 16][tap 9] | db1 [16]
  int batch, n_frames, slab_stride;
  float slope;
};

constexpr int kG2Stride = 260;  // floats per channel of the g2 tile (256 + 4): conflict-free for both products' reads

__global__ void pack_a2t_kernel(const float* __restrict__ w2, int out_ch, float* __restrict__ dst) {
  const int nt = out_ch / 4, n = 9 * nt * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int ci = i & 15, kq = (i >> 4) & 3, t = (i >> 6) % nt, tap = (i >> 6) / nt;
    dst[i] = w2[((size_t)(16 * (t >> 2) + 4 * kq + (t & 3)) * kEncMid + ci) * 9 + tap];
  }
}

constexpr int kEncBwdThreads = 512;  // 8 waves, two per SIMD: one wave's LDS / global latencies are covered by the other's MFMAs

template <int COB>  // out_ch / 16
__global__ __launch_bounds__(kEncBwdThreads, 1) void frame_encode_bwd_kernel(const EncBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int out_ch = COB * 16, NT = out_ch / 4;
  constexpr int RG = 8 / COB, ROWS = 16 / RG;  // dW2: wave = (channel block, one of RG groups of ROWS output rows)
  constexpr int kImgBytes = (kImgW * kImgW * 4 + 15) & ~15;
  float* const img = (float*)smem;
  f32x4* const mid = (f32x4*)(smem + kImgBytes);
  float* const g2 = (float*)(mid + 4 * kMidW * kMidW);  // [co][260]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 15, kq = lane >> 4;
  const int cb = wave % COB, rg = wave / COB;

  f32x4 acc1[9];            // dW2: output-channel block cb, rows of group rg, per tap
  float aw1[9][4], ab1[4];  // dW1 / db1 partials of channels 4 kq + i
  float ab2[COB * 2];       // db2 partials of channels wave + 8 k
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) aw1[t][i] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) ab1[i] = 0.0f;
#pragma unroll
  for (int k = 0; k < COB * 2; ++k) ab2[k] = 0.0f;

  const int n_units = a.batch * a.n_frames;
  for (int f = blockIdx.x; f < n_units; f += gridDim.x) {
    const int b = f / a.n_frames, t = f - b * a.n_frames;
    if (f != (int)blockIdx.x) __syncthreads();
    enc_stage_frame<kEncBwdThreads>(a.frames + (size_t)f * kFrame * kFrame, 1, img, mid, tid);
    {
      const size_t base = ((size_t)t * a.batch + b) * out_ch * 256;
#pragma unroll
      for (int k = 0; k < COB * 2; ++k) {
        const int i = tid + k * kEncBwdThreads, co = i >> 6, e = i & 63;  // co = wave + 8 k
        f32x4 g = *(const f32x4*)(a.g_out + base + (size_t)co * 256 + 4 * e);
        const f32x4 o = *(const f32x4*)(a.out + base + (size_t)co * 256 + 4 * e);
        g.x *= o.x > 0.0f ? 1.0f : a.slope;
        g.y *= o.y > 0.0f ? 1.0f : a.slope;
        g.z *= o.z > 0.0f ? 1.0f : a.slope;
        g.w *= o.w > 0.0f ? 1.0f : a.slope;
        *(f32x4*)(g2 + co * kG2Stride + 4 * e) = g;
        ab2[k] += (g.x + g.y) + (g.z + g.w);
      }
    }
    __syncthreads();
    enc_conv1_to_lds<kEncBwdThreads>(a.pack + opaque_zero(), 1, a.slope, img, mid, tid);
    __syncthreads();

    // ---- dW2[co][ci][ky][kx] += sum_{oy, ox} g2[co][oy][ox] act[ci][2 oy - 1 + ky][2 ox - 1 + kx]: k = pixels, a row per 4 MFMAs
    {
      const float* const mf = (const float*)mid + ((nn >> 2) * kMidW * kMidW) * 4 + (nn & 3);
#pragma unroll 1
      for (int oy = rg * ROWS; oy < (rg + 1) * ROWS; ++oy) {
        const f32x4 af = *(const f32x4*)(g2 + (16 * cb + nn) * kG2Stride + oy * 16 + 4 * kq);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int ky = tap / 3, kx = tap % 3;
          const float* const bp = mf + ((2 * oy + ky) * kMidW + 8 * kq + kx) * 4;
          acc1[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bp[0], acc1[tap], 0, 0, 0);
          acc1[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bp[8], acc1[tap], 0, 0, 0);
          acc1[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bp[16], acc1[tap], 0, 0, 0);
          acc1[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bp[24], acc1[tap], 0, 0, 0);
        }
      }
    }
    // ---- d act[ci][my][mx] = sum_{co, (ky, kx): my = 2 oy - 1 + ky, mx = 2 ox - 1 + kx} g2[co][oy][ox] w2[co][ci][ky][kx], output-
    // stationary: this wave owns rows 4 wave .. 4 wave + 3; block (row, column parity pb) = the 16 pixels mx = 2 b + pb; k = channels
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {  // two passes (4 blocks of accumulators each): even columns (kx = 1), odd columns (kx = 0, 2)
      f32x4 acc2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        if ((kx == 1 ? 0 : 1) != pb) continue;
        float af[NT];
        const float* const ap = a.a2t + opaque_zero() + (tap * NT * 4 + kq) * 16 + nn;  // (opaque: not to be hoisted out of the frame loop)
#pragma unroll
        for (int t4 = 0; t4 < NT; ++t4) af[t4] = ap[t4 * 64];
        const int dx = kx == 0 ? 1 : 0;
        const bool col_ok = nn + dx < 16;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int oy = 2 * wave + rr + (ky == 0 ? 1 : 0);
          const int mrow = ky == 1 ? 2 * rr : 2 * rr + 1;
          if (oy < 16) {  // wave-uniform
            const float* const bp = g2 + (4 * kq) * kG2Stride + oy * 16 + nn + dx;
#pragma unroll
            for (int t4 = 0; t4 < NT; ++t4) {
              const float bv = col_ok ? bp[(16 * (t4 >> 2) + (t4 & 3)) * kG2Stride] : 0.0f;
              acc2[mrow] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t4], bv, acc2[mrow], 0, 0, 0);
            }
          }
        }
      }
      // ---- mask of the first LeakyReLU, then dW1[c][ky][kx] += g[c][my][mx] frame[2 my - 1 + ky][2 mx - 1 + kx], db1[c] += g
#pragma unroll
      for (int mrow = 0; mrow < 4; ++mrow) {
        const int my = 4 * wave + mrow, mx = 2 * nn + pb;
        const f32x4 mv = mid[(kq * kMidW + my + 1) * kMidW + mx + 1];
        f32x4 g = acc2[mrow];
        g.x *= mv.x > 0.0f ? 1.0f : a.slope;
        g.y *= mv.y > 0.0f ? 1.0f : a.slope;
        g.z *= mv.z > 0.0f ? 1.0f : a.slope;
        g.w *= mv.w > 0.0f ? 1.0f : a.slope;
#pragma unroll
        for (int i = 0; i < 4; ++i) ab1[i] += g[i];
        const float* const ip = img + (2 * my) * kImgW + 2 * mx;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const float v = ip[(tap / 3) * kImgW + (tap % 3)];
#pragma unroll
          for (int i = 0; i < 4; ++i) aw1[tap][i] = __builtin_fmaf(g[i], v, aw1[tap][i]);
        }
      }
    }
  }
  // ---- the slab.  dW2: the RG row groups of a channel block through LDS (everything in LDS is dead now)
  float* const slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
  __syncthreads();
  float* const red2 = (float*)smem;  // [rg - 1][cb][tap 9][lane 64] x 4 floats
  if (rg > 0) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) *(f32x4*)(red2 + ((((rg - 1) * COB + cb) * 9 + tap) * 64 + lane) * 4) = acc1[tap];
  }
  __syncthreads();
  if (rg == 0) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      f32x4 v = acc1[tap];
#pragma unroll
      for (int r = 1; r < RG; ++r) v += *(const f32x4*)(red2 + ((((r - 1) * COB + cb) * 9 + tap) * 64 + lane) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) slab[((size_t)(16 * cb + 4 * kq + i) * kEncMid + nn) * 9 + tap] = v[i];
    }
  }
#pragma unroll
  for (int k = 0; k < COB * 2; ++k) {
    const float s = wave_sum(ab2[k]);
    if (lane == 0) slab[out_ch * 144 + wave + 8 * k] = s;
  }
  __syncthreads();
  float* const red = (float*)smem;  // [wave 8][kq 4][40]
#pragma unroll
  for (int tap = 0; tap < 10; ++tap)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = tap < 9 ? aw1[tap < 9 ? tap : 0][i] : ab1[i];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      if (nn == 0) red[(wave * 4 + kq) * 40 + tap * 4 + i] = v;
    }
  __syncthreads();
  if (tid < 160) {
    const int tap = tid / 16, c = tid % 16;  // tap 9 = the bias
    const int at = (c >> 2) * 40 + tap * 4 + (c & 3);
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red[w * 160 + at];
    if (tap < 9) slab[out_ch * 145 + c * 9 + tap] = s;
    else slab[out_ch * 145 + 144 + c] = s;
  }
}

static int round_up(int v, int m) { return (v + m - 1) / m * m; }
static int dec_slab2_floats(int in_ch) { return round_up(in_ch * 512 + kDecMid, 64); }
static int enc_slab_floats(int out_ch) { return round_up(out_ch * 145 + 160, 64); }
static int n_workgroups(long long units, int per_cu) {
  const long long cap = (long long)per_cu * 256;
  return (int)(units < cap ? units : cap);
}

static void launch_slab_sum(const float* slabs, int n_slabs, int stride, float* o0, int n0, float* o1, int n1, float* o2, int n2, float* o3, int n3,
                            hipStream_t stream) {
  SlabSumArgs s;
  s.slabs = slabs; s.n_slabs = n_slabs; s.stride = stride;
  s.out[0] = o0; s.out[1] = o1; s.out[2] = o2; s.out[3] = o3;
  s.n[0] = n0; s.n[1] = n1; s.n[2] = n2; s.n[3] = n3;
  const int total = n0 + n1 + n2 + n3;
  hipLaunchKernelGGL(codec_slab_sum_kernel, dim3((total + 63) / 64), dim3(1024), 0, stream, s);
}

}  // namespace odehip

using namespace odehip;

static int check_codec_backward_shape(const char* who, int frame_ch, int lat_ch) {
  ODEHIP_REQUIRE(frame_ch == 1, "%s: one frame channel (got %d) -- use the library's backward otherwise", who, frame_ch);
  ODEHIP_REQUIRE(lat_ch == 32 || lat_ch == 64, "%s: 32 or 64 latent channels (got %d) -- use the library's backward otherwise", who, lat_ch);
  return ODEHIP_OK;
}

extern "C" size_t odehip_frame_decode_backward_workspace_floats(int n_images, int in_ch, int out_ch) {
  if (n_images <= 0 || out_ch != 1 || (in_ch != 32 && in_ch != 64)) return 0;
  return (size_t)in_ch * 512 + (size_t)n_images * kDecMid * kHalf * kHalf + (size_t)n_workgroups(4LL * n_images, 2) * kDecBwdSlab1 +
         (size_t)n_workgroups(2LL * n_images, 1) * dec_slab2_floats(in_ch);
}

extern "C" int odehip_frame_decode_backward(const float* pack, const float* w1, const float* latents, const float* mid_saved,
                                            const float* pred, const float* g_out, int n_images, int in_ch, int out_ch, float negative_slope, int sigmoid_applied, float* g_latents,
                                            float* dw1, float* db1, float* dw2, float* db2, float* workspace, size_t workspace_floats,
                                            void* stream_) {
  ODEHIP_REQUIRE(pack && w1 && latents && pred && g_out && g_latents && dw1 && db1 && dw2 && db2 && workspace,
                 "frame_decode_backward: null pointer argument");
  ODEHIP_REQUIRE(n_images > 0, "frame_decode_backward: n_images must be positive");
  int rc = check_codec_backward_shape("frame_decode_backward", out_ch, in_ch);
  if (rc != ODEHIP_OK) return rc;
  ODEHIP_REQUIRE(workspace_floats >= odehip_frame_decode_backward_workspace_floats(n_images, in_ch, out_ch),
                 "frame_decode_backward: workspace of %zu floats, %zu needed", workspace_floats,
                 odehip_frame_decode_backward_workspace_floats(n_images, in_ch, out_ch));
  hipStream_t stream = (hipStream_t)stream_;
  const int p1 = n_workgroups(4LL * n_images, 2), p2 = n_workgroups(2LL * n_images, 1), s2 = dec_slab2_floats(in_ch);
  float* const a1t = workspace;
  float* const gmid = a1t + (size_t)in_ch * 512;
  float* const slabs1 = gmid + (size_t)n_images * kDecMid * kHalf * kHalf;
  float* const slabs2 = slabs1 + (size_t)p1 * kDecBwdSlab1;
  hipLaunchKernelGGL(pack_a1t_kernel, dim3(64), dim3(256), 0, stream, w1, in_ch, a1t);

  DecBwdMidArgs m;
  m.pack = pack; m.latents = latents; m.mid = mid_saved; m.pred = pred; m.g_out = g_out; m.gmid = gmid; m.slabs = slabs1; m.n_images = n_images;
  m.sigmoid = sigmoid_applied; m.slab_stride = kDecBwdSlab1; m.slope = negative_slope;
  DecBwdLatArgs l;
  l.a1t = a1t; l.latents = latents; l.gmid = gmid; l.g_latents = g_latents; l.slabs = slabs2; l.n_images = n_images; l.slab_stride = s2;
  const size_t lds1 = (mid_saved ? (size_t)8 * kHalf * kMPix * 4 : (size_t)(in_ch / 4) * kZRows * kZW * 16 + (size_t)kMRows * kMW * kMPix * 4) +
                      (size_t)18 * kGzW * 4;
  const size_t lds2 = (size_t)18 * kMW * kMPix * 4 + (size_t)in_ch * kLatStride * 4;
  static bool attr[2] = {false, false};
#define ODEHIP_DECB_LAUNCH(G, slot)                                                                                              \
  {                                                                                                                             \
    if (!attr[slot]) {                                                                                                          \
      ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)frame_decode_bwd_mid_kernel<G, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)frame_decode_bwd_mid_kernel<G, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)frame_decode_bwd_lat_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      attr[slot] = true;                                                                                                        \
    }                                                                                                                           \
    if (mid_saved) hipLaunchKernelGGL((frame_decode_bwd_mid_kernel<G, true>), dim3(p1), dim3(kCodecThreads), lds1, stream, m);  \
    else hipLaunchKernelGGL((frame_decode_bwd_mid_kernel<G, false>), dim3(p1), dim3(kCodecThreads), lds1, stream, m);           \
    hipLaunchKernelGGL(frame_decode_bwd_lat_kernel<G>, dim3(p2), dim3(kDecBwdThreads2), lds2, stream, l);                       \
  }
  if (in_ch == 32) ODEHIP_DECB_LAUNCH(2, 0)
  else ODEHIP_DECB_LAUNCH(4, 1)
#undef ODEHIP_DECB_LAUNCH
  launch_slab_sum(slabs1, p1, kDecBwdSlab1, dw2, kDecMid * 16, db2, 1, nullptr, 0, nullptr, 0, stream);
  launch_slab_sum(slabs2, p2, s2, dw1, in_ch * 512, db1, kDecMid, nullptr, 0, nullptr, 0, stream);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

extern "C" size_t odehip_frame_encode_backward_workspace_floats(int batch, int n_frames, int in_ch, int out_ch) {
  if (batch <= 0 || n_frames <= 0 || in_ch != 1 || (out_ch != 32 && out_ch != 64)) return 0;
  return (size_t)9 * out_ch * 16 + (size_t)n_workgroups((long long)batch * n_frames, 1) * enc_slab_floats(out_ch);
}

extern "C" int odehip_frame_encode_backward(const float* pack, const float* w2, const float* frames, const float* out_time_first,
                                            const float* g_out_time_first, int batch, int n_frames, int in_ch, int out_ch,
                                            float negative_slope, float* dw1, float* db1, float* dw2, float* db2, float* workspace,
                                            size_t workspace_floats, void* stream_) {
  ODEHIP_REQUIRE(pack && w2 && frames && out_time_first && g_out_time_first && dw1 && db1 && dw2 && db2 && workspace,
                 "frame_encode_backward: null pointer argument");
  ODEHIP_REQUIRE(batch > 0 && n_frames > 0, "frame_encode_backward: batch and n_frames must be positive");
  int rc = check_codec_backward_shape("frame_encode_backward", in_ch, out_ch);
  if (rc != ODEHIP_OK) return rc;
  ODEHIP_REQUIRE(workspace_floats >= odehip_frame_encode_backward_workspace_floats(batch, n_frames, in_ch, out_ch),
                 "frame_encode_backward: workspace of %zu floats, %zu needed", workspace_floats,
                 odehip_frame_encode_backward_workspace_floats(batch, n_frames, in_ch, out_ch));
  hipStream_t stream = (hipStream_t)stream_;
  const int p = n_workgroups((long long)batch * n_frames, 1), st = enc_slab_floats(out_ch);
  float* const a2t = workspace;
  float* const slabs = a2t + (size_t)9 * out_ch * 16;
  hipLaunchKernelGGL(pack_a2t_kernel, dim3(16), dim3(256), 0, stream, w2, out_ch, a2t);
  EncBwdArgs e;
  e.pack = pack; e.a2t = a2t; e.frames = frames; e.out = out_time_first; e.g_out = g_out_time_first; e.slabs = slabs; e.batch = batch;
  e.n_frames = n_frames; e.slab_stride = st; e.slope = negative_slope;
  const size_t lds = (size_t)((kImgW * kImgW * 4 + 15) & ~15) + (size_t)4 * kMidW * kMidW * 16 + (size_t)out_ch * kG2Stride * 4;
  static bool attr[2] = {false, false};
#define ODEHIP_ENCB_LAUNCH(COB, slot)                                                                                          \
  {                                                                                                                           \
    if (!attr[slot]) {                                                                                                        \
      ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)frame_encode_bwd_kernel<COB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      attr[slot] = true;                                                                                                      \
    }                                                                                                                         \
    hipLaunchKernelGGL(frame_encode_bwd_kernel<COB>, dim3(p), dim3(kEncBwdThreads), lds, stream, e);                           \
  }
  if (out_ch == 32) ODEHIP_ENCB_LAUNCH(2, 0)
  else ODEHIP_ENCB_LAUNCH(4, 1)
#undef ODEHIP_ENCB_LAUNCH
  launch_slab_sum(slabs, p, st, dw2, out_ch * 144, db2, out_ch, dw1, 144, db1, 16, stream);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}
