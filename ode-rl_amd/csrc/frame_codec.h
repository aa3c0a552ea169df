// frame_codec.h -- what the fused frame encoder / decoder (frame_codec.hip) and their backward kernels (frame_codec_backward.hip)
// share: the shapes of /root/reference/models/ODEConvGRU.py:101-140 (n_downs = n_ups = 2), the weight packs, and the two device
// routines that build the 32x32 intermediate in LDS (the backward kernels recompute it instead of reading it from HBM).
#pragma once
#include <stddef.h>

#include "conv_common.h"

namespace odehip {

constexpr int kFrame = 64;     // frames are 64x64: two stride-2 layers take them to the path's 16x16 latents
constexpr int kHalf = 32;      // the intermediate resolution
constexpr int kEncMid = 16;    // Encoder: chan = 16 (ODEConvGRU.py:105)
constexpr int kDecMid = 32;    // Decoder: chan = 32 (:131)
constexpr int kCodecThreads = 256;

// ConvTranspose2d(k = 4, stride 2, pad 1): output row 2 i + a receives input row i + d through kernel row k, for two (k, d) per
// parity a (o = 2 iy - 1 + k):  a = 0: (1, 0), (3, -1);   a = 1: (0, +1), (2, 0).  Same for columns.
__host__ __device__ __forceinline__ void convt_tap(int parity, int t, int& k, int& d) {
  if (parity == 0) { k = t == 0 ? 1 : 3; d = t == 0 ? 0 : -1; }
  else             { k = t == 0 ? 0 : 2; d = t == 0 ? 1 : 0; }
}

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.0f ? v : v * slope; }

// Encoder pack (floats): [w1 (16, in_ch, 3, 3) as is | b1 (16) | pad to x4 | A2 [tap 9][kq 4][co][j 4] = w2[co][4 kq + j][tap] | b2]
__host__ __device__ inline size_t enc_off_b1(int in_ch) { return (size_t)kEncMid * in_ch * 9; }
__host__ __device__ inline size_t enc_off_a2(int in_ch) { return (enc_off_b1(in_ch) + kEncMid + 3) & ~(size_t)3; }
__host__ __device__ inline size_t enc_off_b2(int in_ch, int out_ch) { return enc_off_a2(in_ch) + (size_t)9 * 16 * out_ch; }
__host__ __device__ inline size_t enc_pack_floats(int in_ch, int out_ch) { return enc_off_b2(in_ch, out_ch) + out_ch; }

// Decoder pack (floats): [A1 [parity 4][tap 4][g][kq 4][co 32][j 4] = w1[16 g + 4 kq + j][co][ky][kx] | b1 (32) |
//                         W2 [parity 4][tap 4][o][ci 32] = w2[ci][o][ky][kx] | b2 (out_ch)]      (ky, kx) = convt_tap(parity, tap)
__host__ __device__ inline size_t dec_off_b1(int in_ch) { return (size_t)in_ch * 512; }
__host__ __device__ inline size_t dec_off_w2(int in_ch) { return dec_off_b1(in_ch) + kDecMid; }
__host__ __device__ inline size_t dec_off_b2(int in_ch, int out_ch) { return dec_off_w2(in_ch) + (size_t)16 * out_ch * kDecMid; }
__host__ __device__ inline size_t dec_pack_floats(int in_ch, int out_ch) { return (dec_off_b2(in_ch, out_ch) + out_ch + 3) & ~(size_t)3; }

constexpr int kImgW = kFrame + 1;   // row / column 0 = the zero border at index -1 (stride 2, pad 1 never reaches index 64)
constexpr int kMidW = kHalf + 1;

// Encoder, stage 0: one frame `src` (in_ch, 64, 64) -> img [in_ch][65][65] floats with its zero border; the zero border of
// mid [quad 4][33][33] x 16 B.  No barrier inside.
template <int THREADS = kCodecThreads>
__device__ __forceinline__ void enc_stage_frame(const float* __restrict__ src, int in_ch, float* img, f32x4* mid, int tid) {
  for (int i = tid; i < in_ch * (2 * kImgW - 1); i += THREADS) {
    const int ic = i / (2 * kImgW - 1), e = i - ic * (2 * kImgW - 1);
    img[(size_t)ic * kImgW * kImgW + (e < kImgW ? e : (e - kImgW + 1) * kImgW)] = 0.0f;
  }
  for (int i = tid; i < 4 * (2 * kMidW - 1); i += THREADS) {
    const int q = i / (2 * kMidW - 1), e = i - q * (2 * kMidW - 1);
    mid[q * kMidW * kMidW + (e < kMidW ? e : (e - kMidW + 1) * kMidW)] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int i = tid; i < in_ch * kFrame * kFrame / 4; i += THREADS) {
    const int ic = i / (kFrame * kFrame / 4), e = i - ic * (kFrame * kFrame / 4), r = e / (kFrame / 4), c4 = e - r * (kFrame / 4);
    const f32x4 v = *(const f32x4*)(src + (size_t)i * 4);
    float* d = img + (size_t)ic * kImgW * kImgW + (r + 1) * kImgW + 4 * c4 + 1;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

// Encoder, stage 1: Conv2d(in_ch, 16, 3, 2, 1) + LeakyReLU on the VALU, img -> mid: 4 output pixels per thread, all 16 channels.
// `pack` = the encoder pack (w1, b1 in front).  Needs a barrier before (img complete) and after.
template <int THREADS = kCodecThreads>
__device__ __forceinline__ void enc_conv1_to_lds(const float* __restrict__ pack, int in_ch, float slope, const float* img, f32x4* mid, int tid) {
  typedef const __attribute__((address_space(4))) float ConstF;  // uniform reads of the small weights: scalar loads
  ConstF* const pk = (ConstF*)pack;
  for (int s = 0; s < kHalf * kHalf / THREADS; ++s) {
    const int p = tid + s * THREADS, oy = p / kHalf, ox = p - oy * kHalf;
    float acc[kEncMid];
#pragma unroll
    for (int c = 0; c < kEncMid; ++c) acc[c] = pk[enc_off_b1(in_ch) + c];
    for (int ic = 0; ic < in_ch; ++ic) {
      const float* ip = img + (size_t)ic * kImgW * kImgW + 2 * oy * kImgW + 2 * ox;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float v = ip[(k / 3) * kImgW + (k % 3)];
#pragma unroll
        for (int c = 0; c < kEncMid; ++c) acc[c] = __builtin_fmaf(pk[(c * in_ch + ic) * 9 + k], v, acc[c]);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      mid[(q * kMidW + oy + 1) * kMidW + ox + 1] =
          f32x4{leaky(acc[4 * q], slope), leaky(acc[4 * q + 1], slope), leaky(acc[4 * q + 2], slope), leaky(acc[4 * q + 3], slope)};
  }
}

constexpr int kZRows = 6, kZW = 18;          // latent rows 4q-1 .. 4q+4, columns -1 .. 16
constexpr int kMRows = 10, kMW = 34;         // intermediate rows 8q-1 .. 8q+8, columns -1 .. 32
constexpr int kMPix = 36;                    // floats per intermediate pixel (32 channels + 4: a 2-pixel lane stride is 72 words)

// Decoder, layer 1 for quarter q of an image: latent rows 4q-1 .. 4q+4 of `src` (in_ch, 16, 16) -> z [quad in_ch/4][6][18] x 16 B,
// then ConvTranspose2d(in_ch, 32, 4, 2, 1) + LeakyReLU on the MFMA -> mid [10][34][36] floats (intermediate rows 8q-1 .. 8q+8,
// zero outside the image).  ConvTranspose(k4, s2, p1) = four 2x2 convolutions, one per output parity: wave w takes parity
// (w >> 1, w & 1); its weights are one contiguous quarter of the pack (16-byte fragments straight from L2).  Two barriers inside;
// the caller adds the one after.
template <int G>  // in_ch / 16
__device__ __forceinline__ void dec_mid_to_lds(const float* __restrict__ pack, const float* __restrict__ src, int q, float slope, f32x4* z,
                                               float* mid, int tid, int lane, int wave) {
  constexpr int in_ch = G * 16;
  const int pa = wave >> 1, pb = wave & 1;
  // ---- phase 0: latent rows -> LDS as channel quads, zero outside the image; the intermediate's border columns
  for (int i = tid; i < 4 * G * kZRows * kZW; i += kCodecThreads) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < kMRows * 2; i += kCodecThreads) {
    float* m = mid + ((i >> 1) * kMW + (i & 1) * (kMW - 1)) * kMPix;
#pragma unroll
    for (int c = 0; c < kDecMid; c += 4) *(f32x4*)(m + c) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  for (int i = tid; i < in_ch * kZRows * 4; i += kCodecThreads) {
    const int ci = i / (kZRows * 4), e = i - ci * (kZRows * 4), rr = e >> 2, c4 = e & 3;
    const int zr = 4 * q - 1 + rr;
    if (zr < 0 || zr > 15) continue;
    const f32x4 v = *(const f32x4*)(src + (size_t)ci * 256 + zr * 16 + 4 * c4);
    float* d = (float*)(z + ((ci >> 2) * kZRows + rr) * kZW + 4 * c4 + 1) + (ci & 3);
    d[0] = v.x; d[4] = v.y; d[8] = v.z; d[12] = v.w;
  }
  __syncthreads();

  // ---- phase 1: ConvTranspose2d(in_ch, 32, 4, 2, 1) + LeakyReLU on the MFMA.  This wave: the five intermediate rows of parity pa
  // (local rows mr = 2 r + 1 - pa; row 8q-1 is odd), columns of parity pb (16 per row = one block), both 16-channel halves.
  {
    const int nn = lane & 15, kq = lane >> 4;
    f32x4 acc[2][5];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 5; ++r) acc[cb][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* const a1 = (const f32x4*)pack + (size_t)(pa * 2 + pb) * 4 * G * 4 * kDecMid + kq * kDecMid + nn;
    // the weight fragments come straight from L2: those of the next tap are requested before this tap's MFMAs
    f32x4 af[G][2], afn[G][2];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) af[g][cb] = a1[g * 4 * kDecMid + cb * 16];
#pragma unroll 1
    for (int tap = 0; tap < 4; ++tap) {
      int ky, kx, dy, dx;
      convt_tap(pa, tap >> 1, ky, dy);
      convt_tap(pb, tap & 1, kx, dx);
      (void)ky; (void)kx;
      if (tap + 1 < 4) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) afn[g][cb] = a1[((tap + 1) * G + g) * 4 * kDecMid + cb * 16];
      }
      // intermediate row my = 8q - 1 + mr = 2 i' + pa (mr = 2 r + 1 - pa)  ->  latent row i' + dy, local index r - pa + 1 + dy
      const f32x4* const zb = z + (kq * kZRows + 1 - pa + dy) * kZW + nn + dx + 1;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        f32x4 bf[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) bf[r] = zb[(4 * g * kZRows + r) * kZW];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 5; ++r) {
            acc[cb][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][cb].x, bf[r].x, acc[cb][r], 0, 0, 0);
            acc[cb][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][cb].y, bf[r].y, acc[cb][r], 0, 0, 0);
            acc[cb][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][cb].z, bf[r].z, acc[cb][r], 0, 0, 0);
            acc[cb][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][cb].w, bf[r].w, acc[cb][r], 0, 0, 0);
          }
      }
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) af[g][cb] = afn[g][cb];
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int c0 = cb * 16 + 4 * kq;
      const f32x4 bias = *(const f32x4*)(pack + dec_off_b1(in_ch) + c0);
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const int mr = 2 * r + 1 - pa, my = 8 * q - 1 + mr;
        f32x4 v = acc[cb][r] + bias;
        v = f32x4{leaky(v.x, slope), leaky(v.y, slope), leaky(v.z, slope), leaky(v.w, slope)};
        if (my < 0 || my >= kHalf) v = f32x4{0.f, 0.f, 0.f, 0.f};  // rows beyond the intermediate image contribute nothing
        *(f32x4*)(mid + (mr * kMW + 2 * nn + pb + 1) * kMPix + c0) = v;
      }
    }
  }
}

}  // namespace odehip
