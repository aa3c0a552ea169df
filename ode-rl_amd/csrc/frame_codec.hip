// frame_codec.hip -- the conv encoder / decoder either side of the path, each as ONE fused launch.
//
//   Encoder (/root/reference/models/ODEConvGRU.py:101-118, n_downs = 2):
//       Conv2d(in_ch, 16, 3, 2, 1) -> LeakyReLU(0.2) -> Conv2d(16, out_ch, 3, 2, 1) -> LeakyReLU(0.2)      64x64 -> 32x32 -> 16x16
//   Decoder (:121-140, n_ups = 2):
//       ConvTranspose2d(in_ch, 32, 4, 2, 1) -> LeakyReLU(0.2) -> ConvTranspose2d(32, out_ch, 4, 2, 1) [-> sigmoid, :85]
//                                                                                                       16x16 -> 32x32 -> 64x64
// Unfused these are four library launches that write the 32x32 intermediate (16 resp. 32 channels: 2x / 4x the bytes of the
// latent) to HBM and read it back.  Here the intermediate lives in LDS; the two layers with the arithmetic (16 -> out_ch and
// in_ch -> 32) run on the exact-fp32 MFMA, the two one-channel-sided layers on the VALU.  The encoder writes its result
// TIME-FIRST (T,B,C,16,16) -- what ODEConvGRUCell consumes (ODEConvGRU.py:68 permutes a view) -- and the decoder reads the
// solver's (T,B,C,16,16) as it lies: no layout copy on either side.
//
// MFMA operand trick used by both: activations sit in LDS as channel quads ([quad][y][x] x 4 floats).  Lane (n = pixel of a
// 16-pixel block, kq = lane / 16) reads ONE 16-byte quad `kq` of its (shifted) input pixel: component j of it is its B value for
// the j-th of four v_mfma_f32_16x16x4_f32, whose k index therefore runs over channels {j, 4 + j, 8 + j, 12 + j} of a 16-channel
// group -- any fixed permutation of k is fine as long as the weights are packed the same way, which the pack kernels below do.
#include "frame_codec.h"

namespace odehip {

// ------------------------------------------------------------------------------------------------------------------ packing
// Encoder pack (floats): [w1 (16, in_ch, 3, 3) as is | b1 (16) | pad to x4 | A2 [tap 9][kq 4][co][j 4] = w2[co][4 kq + j][tap] | b2]

__global__ void pack_frame_encoder_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                          const float* __restrict__ b2, int in_ch, int out_ch, float* __restrict__ dst) {
  const size_t n = enc_pack_floats(in_ch, out_ch);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = 0.0f;
    if (i < enc_off_b1(in_ch)) v = w1[i];
    else if (i < enc_off_b1(in_ch) + kEncMid) v = b1[i - enc_off_b1(in_ch)];
    else if (i < enc_off_a2(in_ch)) v = 0.0f;
    else if (i < enc_off_b2(in_ch, out_ch)) {
      const size_t e = i - enc_off_a2(in_ch);
      const int j = (int)(e & 3), co = (int)((e >> 2) % out_ch), kq = (int)((e >> 2) / out_ch & 3), tap = (int)((e >> 2) / out_ch >> 2);
      v = w2[((size_t)co * kEncMid + 4 * kq + j) * 9 + tap];
    } else v = b2[i - enc_off_b2(in_ch, out_ch)];
    dst[i] = v;
  }
}

// Decoder pack (floats): [A1 [parity 4][tap 4][g][kq 4][co 32][j 4] = w1[16 g + 4 kq + j][co][ky][kx] | b1 (32) |
//                         W2 [parity 4][tap 4][o][ci 32] = w2[ci][o][ky][kx] | b2 (out_ch)]      (ky, kx) = convt_tap(parity, tap)

__global__ void pack_frame_decoder_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                          const float* __restrict__ b2, int in_ch, int out_ch, float* __restrict__ dst) {
  const size_t n = dec_pack_floats(in_ch, out_ch);
  const int G = in_ch / 16;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = 0.0f;
    if (i < dec_off_b1(in_ch)) {
      size_t e = i;
      const int j = (int)(e & 3); e >>= 2;
      const int co = (int)(e % kDecMid); e /= kDecMid;
      const int kq = (int)(e & 3); e >>= 2;
      const int g = (int)(e % G); e /= G;
      const int tap = (int)(e & 3), par = (int)(e >> 2);
      int ky, kx, d;
      convt_tap(par >> 1, tap >> 1, ky, d);
      convt_tap(par & 1, tap & 1, kx, d);
      v = w1[(((size_t)(16 * g + 4 * kq + j) * kDecMid + co) * 4 + ky) * 4 + kx];
    } else if (i < dec_off_w2(in_ch)) v = b1[i - dec_off_b1(in_ch)];
    else if (i < dec_off_b2(in_ch, out_ch)) {
      size_t e = i - dec_off_w2(in_ch);
      const int ci = (int)(e % kDecMid); e /= kDecMid;
      const int o = (int)(e % out_ch); e /= out_ch;
      const int tap = (int)(e & 3), par = (int)(e >> 2);
      int ky, kx, d;
      convt_tap(par >> 1, tap >> 1, ky, d);
      convt_tap(par & 1, tap & 1, kx, d);
      v = w2[(((size_t)ci * out_ch + o) * 4 + ky) * 4 + kx];
    } else if (i < dec_off_b2(in_ch, out_ch) + out_ch) v = b2[i - dec_off_b2(in_ch, out_ch)];
    dst[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------ encoder
struct EncArgs {
  const float* pack;
  const float* frames;  // (B, T, in_ch, 64, 64)
  float* out;           // (T, B, out_ch, 16, 16)
  int batch, n_frames, in_ch;
  float slope;
};

// One workgroup per frame.  LDS: img [in_ch][65][65] floats | mid [quad 4][33][33] x 16 B | A2 [tap 9][kq 4][co] x 16 B.
template <int COB>  // out_ch / 16
__global__ __launch_bounds__(kCodecThreads) void frame_encode_kernel(const EncArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int out_ch = COB * 16;
  float* const img = (float*)smem;
  f32x4* const mid = (f32x4*)(smem + (((size_t)a.in_ch * kImgW * kImgW * 4 + 15) & ~(size_t)15));
  f32x4* const a2 = mid + 4 * kMidW * kMidW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x, b = n / a.n_frames, t = n - b * a.n_frames;

  // ---- phase 0: frame -> LDS (with its zero border), conv2's weights -> LDS
  enc_stage_frame(a.frames + (size_t)n * a.in_ch * kFrame * kFrame, a.in_ch, img, mid, tid);
  {
    const f32x4* const g = (const f32x4*)(a.pack + enc_off_a2(a.in_ch));
    for (int i = tid; i < 9 * 4 * out_ch; i += kCodecThreads) a2[i] = g[i];
  }
  __syncthreads();

  // ---- phase 1: Conv2d(in_ch, 16, 3, 2, 1) + LeakyReLU on the VALU (frame_codec.h)
  enc_conv1_to_lds(a.pack, a.in_ch, a.slope, img, mid, tid);
  __syncthreads();

  // ---- phase 2: Conv2d(16, out_ch, 3, 2, 1) on the MFMA: wave w owns output rows 4w..4w+3 (one 16-pixel block each) x all co
  const int nn = lane & 15, kq = lane >> 4;
  f32x4 acc[COB][4];
#pragma unroll
  for (int cb = 0; cb < COB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[cb][r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap % 3;
    f32x4 bf[4], af[COB];
#pragma unroll
    for (int r = 0; r < 4; ++r) bf[r] = mid[(kq * kMidW + 2 * (4 * wave + r) + ky) * kMidW + 2 * nn + kx];
#pragma unroll
    for (int cb = 0; cb < COB; ++cb) af[cb] = a2[(tap * 4 + kq) * out_ch + cb * 16 + nn];
#pragma unroll
    for (int cb = 0; cb < COB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[cb][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cb].x, bf[r].x, acc[cb][r], 0, 0, 0);
        acc[cb][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cb].y, bf[r].y, acc[cb][r], 0, 0, 0);
        acc[cb][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cb].z, bf[r].z, acc[cb][r], 0, 0, 0);
        acc[cb][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cb].w, bf[r].w, acc[cb][r], 0, 0, 0);
      }
  }
  // lane (nn, kq) holds channels cb*16 + 4 kq + i of pixel (row, nn); time-first destination
  float* const dst = a.out + ((size_t)t * a.batch + b) * out_ch * 256;
#pragma unroll
  for (int cb = 0; cb < COB; ++cb) {
    const int c0 = cb * 16 + 4 * kq;
    const f32x4 bias = *(const f32x4*)(a.pack + enc_off_b2(a.in_ch, out_ch) + c0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 v = acc[cb][r] + bias;
      float* o = dst + (size_t)c0 * 256 + (4 * wave + r) * 16 + nn;
      o[0] = leaky(v.x, a.slope); o[256] = leaky(v.y, a.slope); o[512] = leaky(v.z, a.slope); o[768] = leaky(v.w, a.slope);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ decoder
struct DecArgs {
  const float* pack;
  const float* latents;  // (N, in_ch, 16, 16)
  float* out;            // (N, out_ch, 64, 64)
  int n_images, in_ch, out_ch;
  float slope;
  int sigmoid;
  float* mid_out;        // optional: [N][quad 8][32][32] x 4, the intermediate after its LeakyReLU (the training forward saves it for
                         // frame_decode_bwd_mid_kernel, which would otherwise recompute it: 100 us of MFMA work against 25 us of stores)
};

// One workgroup per (image, quarter q of the output rows): output rows 16q .. 16q+15 need intermediate rows 8q-1 .. 8q+8, which
// need latent rows 4q-1 .. 4q+4.  LDS: z [quad in_ch/4][6][18] x 16 B | mid [10][34][36] floats: 76 KiB at in_ch = 64, two
// workgroups per CU.  ConvTranspose(k4, s2, p1) = four 2x2 convolutions, one per output parity: wave w takes parity (w >> 1, w & 1)
// in both layers, so its weights are one contiguous quarter of the pack (layer 1: 16-byte fragments straight from L2; layer 2:
// wave-uniform scalar loads).
template <int G>  // in_ch / 16
__global__ __launch_bounds__(kCodecThreads, 2) void frame_decode_kernel(const DecArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int in_ch = G * 16;
  f32x4* const z = (f32x4*)smem;
  float* const mid = (float*)(smem + (size_t)4 * G * kZRows * kZW * 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x >> 2, q = blockIdx.x & 3;
  const int pa = wave >> 1, pb = wave & 1;

  // ---- phases 0, 1: latent rows -> LDS, ConvTranspose2d(in_ch, 32, 4, 2, 1) + LeakyReLU on the MFMA -> mid (frame_codec.h)
  dec_mid_to_lds<G>(a.pack, a.latents + (size_t)n * in_ch * 256, q, a.slope, z, mid, tid, lane, wave);
  __syncthreads();
  if (a.mid_out) {  // this workgroup's own eight rows 8q .. 8q+7 (local rows 1 .. 8), one pixel per thread
    const int pr = tid >> 5, mx = tid & 31;
    const float* const mp = mid + ((pr + 1) * kMW + mx + 1) * kMPix;
    f32x4* const dst = (f32x4*)a.mid_out + (((size_t)n * 8) * kHalf + 8 * q + pr) * kHalf + mx;
#pragma unroll
    for (int cq = 0; cq < 8; ++cq) dst[(size_t)cq * kHalf * kHalf] = *(const f32x4*)(mp + 4 * cq);
  }

  // ---- phase 2: ConvTranspose2d(32, out_ch, 4, 2, 1) [+ sigmoid] on the VALU.  This wave: output pixels (16q + 2 il + pa,
  // 2 j + pb), il = 0..7, j = 0..31: four per lane.  Weights of its parity: [tap 4][o][ci 32], wave-uniform.
  {
    typedef const __attribute__((address_space(4))) float ConstF;
    ConstF* const w2 = (ConstF*)(a.pack + dec_off_w2(in_ch)) + (size_t)(pa * 2 + pb) * 4 * a.out_ch * kDecMid;
    ConstF* const b2 = (ConstF*)(a.pack + dec_off_b2(in_ch, a.out_ch));
    const int j = lane & 31, ih = lane >> 5;
    for (int o = 0; o < a.out_ch; ++o) {
      float accv[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) accv[s] = b2[o];
#pragma unroll 1
      for (int tap = 0; tap < 4; ++tap) {
        int ky, kx, dy, dx;
        convt_tap(pa, tap >> 1, ky, dy);
        convt_tap(pb, tap & 1, kx, dx);
        (void)ky; (void)kx;
        ConstF* const w = w2 + (tap * a.out_ch + o) * kDecMid;
        // output row 16q + 2 il + pa = 2 (8q + il) + pa  ->  intermediate row 8q + il + dy, local index il + dy + 1 (il = ih + 2 s)
        const float* const mb = mid + ((ih + dy + 1) * kMW + j + dx + 1) * kMPix;
#pragma unroll
        for (int c = 0; c < kDecMid; c += 4) {
          const float w0 = w[c], w1 = w[c + 1], w2v = w[c + 2], w3 = w[c + 3];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const f32x4 mv = *(const f32x4*)(mb + 2 * s * kMW * kMPix + c);
            accv[s] = __builtin_fmaf(mv.x, w0, accv[s]);
            accv[s] = __builtin_fmaf(mv.y, w1, accv[s]);
            accv[s] = __builtin_fmaf(mv.z, w2v, accv[s]);
            accv[s] = __builtin_fmaf(mv.w, w3, accv[s]);
          }
        }
      }
      float* const dst = a.out + ((size_t)n * a.out_ch + o) * kFrame * kFrame;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float v = accv[s];
        if (a.sigmoid) v = 1.0f / (1.0f + expf(-v));
        dst[(16 * q + 2 * (ih + 2 * s) + pa) * kFrame + 2 * j + pb] = v;
      }
    }
  }
}

static size_t enc_lds_bytes(int in_ch, int out_ch) {
  return (((size_t)in_ch * kImgW * kImgW * 4 + 15) & ~(size_t)15) + (size_t)4 * kMidW * kMidW * 16 + (size_t)9 * 4 * out_ch * 16;
}
static size_t dec_lds_bytes(int in_ch) { return (size_t)(in_ch / 4) * kZRows * kZW * 16 + (size_t)kMRows * kMW * kMPix * 4; }

template <typename K>
static int set_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return ODEHIP_OK;
}

}  // namespace odehip

using namespace odehip;

extern "C" size_t odehip_frame_encoder_pack_floats(int in_ch, int out_ch) {
  return (in_ch >= 1 && out_ch >= 16) ? enc_pack_floats(in_ch, out_ch) : 0;
}
extern "C" size_t odehip_frame_decoder_pack_floats(int in_ch, int out_ch) {
  return (in_ch >= 16 && out_ch >= 1) ? dec_pack_floats(in_ch, out_ch) : 0;
}

static int check_encoder_shape(const char* who, int in_ch, int out_ch) {
  ODEHIP_REQUIRE(in_ch >= 1 && in_ch <= 4, "%s: 1..4 frame channels (got %d)", who, in_ch);
  ODEHIP_REQUIRE(out_ch == 32 || out_ch == 64 || out_ch == 128, "%s: 32, 64 or 128 latent channels (got %d)", who, out_ch);
  ODEHIP_REQUIRE(enc_lds_bytes(in_ch, out_ch) <= 160 * 1024, "%s: in_ch %d with out_ch %d does not fit in LDS", who, in_ch, out_ch);
  return ODEHIP_OK;
}
static int check_decoder_shape(const char* who, int in_ch, int out_ch) {
  ODEHIP_REQUIRE(in_ch == 32 || in_ch == 64 || in_ch == 128, "%s: 32, 64 or 128 latent channels (got %d)", who, in_ch);
  ODEHIP_REQUIRE(out_ch >= 1 && out_ch <= 4, "%s: 1..4 frame channels (got %d)", who, out_ch);
  return ODEHIP_OK;
}

extern "C" int odehip_pack_frame_encoder(const float* w1, const float* b1, const float* w2, const float* b2, int in_ch, int out_ch,
                                         float* pack, void* stream) {
  ODEHIP_REQUIRE(w1 && b1 && w2 && b2 && pack, "pack_frame_encoder: null pointer argument");
  int rc = check_encoder_shape("pack_frame_encoder", in_ch, out_ch);
  if (rc != ODEHIP_OK) return rc;
  hipLaunchKernelGGL(pack_frame_encoder_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2, in_ch, out_ch, pack);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

extern "C" int odehip_pack_frame_decoder(const float* w1, const float* b1, const float* w2, const float* b2, int in_ch, int out_ch,
                                         float* pack, void* stream) {
  ODEHIP_REQUIRE(w1 && b1 && w2 && b2 && pack, "pack_frame_decoder: null pointer argument");
  int rc = check_decoder_shape("pack_frame_decoder", in_ch, out_ch);
  if (rc != ODEHIP_OK) return rc;
  hipLaunchKernelGGL(pack_frame_decoder_kernel, dim3(128), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2, in_ch, out_ch, pack);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

extern "C" int odehip_frame_encode(const float* pack, const float* frames, int batch, int n_frames, int in_ch, int out_ch,
                                   float negative_slope, float* out_time_first, void* stream) {
  ODEHIP_REQUIRE(pack && frames && out_time_first, "frame_encode: null pointer argument");
  ODEHIP_REQUIRE(batch > 0 && n_frames > 0, "frame_encode: batch and n_frames must be positive");
  int rc = check_encoder_shape("frame_encode", in_ch, out_ch);
  if (rc != ODEHIP_OK) return rc;
  EncArgs a;
  a.pack = pack; a.frames = frames; a.out = out_time_first; a.batch = batch; a.n_frames = n_frames; a.in_ch = in_ch; a.slope = negative_slope;
  const size_t lds = enc_lds_bytes(in_ch, out_ch);
  const dim3 grid((unsigned)(batch * n_frames));
  static bool attr[3] = {false, false, false};
#define ODEHIP_ENC_LAUNCH(COB, slot)                                                                       \
  {                                                                                                       \
    if (!attr[slot]) {                                                                                    \
      rc = set_lds(frame_encode_kernel<COB>, 160 * 1024);                                                 \
      if (rc != ODEHIP_OK) return rc;                                                                     \
      attr[slot] = true;                                                                                  \
    }                                                                                                     \
    hipLaunchKernelGGL(frame_encode_kernel<COB>, grid, dim3(kCodecThreads), lds, (hipStream_t)stream, a); \
  }
  if (out_ch == 32) ODEHIP_ENC_LAUNCH(2, 0)
  else if (out_ch == 64) ODEHIP_ENC_LAUNCH(4, 1)
  else ODEHIP_ENC_LAUNCH(8, 2)
#undef ODEHIP_ENC_LAUNCH
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

extern "C" int odehip_frame_decode_train(const float* pack, const float* latents, int n_images, int in_ch, int out_ch,
                                         float negative_slope, int apply_sigmoid, float* out, float* mid_save, void* stream);

extern "C" int odehip_frame_decode(const float* pack, const float* latents, int n_images, int in_ch, int out_ch, float negative_slope,
                                   int apply_sigmoid, float* out, void* stream) {
  return odehip_frame_decode_train(pack, latents, n_images, in_ch, out_ch, negative_slope, apply_sigmoid, out, nullptr, stream);
}

extern "C" int odehip_frame_decode_train(const float* pack, const float* latents, int n_images, int in_ch, int out_ch,
                                         float negative_slope, int apply_sigmoid, float* out, float* mid_save, void* stream) {
  ODEHIP_REQUIRE(pack && latents && out, "frame_decode: null pointer argument");
  ODEHIP_REQUIRE(n_images > 0, "frame_decode: n_images must be positive");
  int rc = check_decoder_shape("frame_decode", in_ch, out_ch);
  if (rc != ODEHIP_OK) return rc;
  DecArgs a;
  a.pack = pack; a.latents = latents; a.out = out; a.n_images = n_images; a.in_ch = in_ch; a.out_ch = out_ch; a.slope = negative_slope;
  a.sigmoid = apply_sigmoid;
  a.mid_out = mid_save;
  const size_t lds = dec_lds_bytes(in_ch);
  const dim3 grid((unsigned)n_images * 4);
  static bool attr[3] = {false, false, false};
#define ODEHIP_DEC_LAUNCH(G, slot)                                                                       \
  {                                                                                                     \
    if (!attr[slot]) {                                                                                  \
      rc = set_lds(frame_decode_kernel<G>, 160 * 1024);                                                 \
      if (rc != ODEHIP_OK) return rc;                                                                   \
      attr[slot] = true;                                                                                \
    }                                                                                                   \
    hipLaunchKernelGGL(frame_decode_kernel<G>, grid, dim3(kCodecThreads), lds, (hipStream_t)stream, a); \
  }
  if (in_ch == 32) ODEHIP_DEC_LAUNCH(2, 0)
  else if (in_ch == 64) ODEHIP_DEC_LAUNCH(4, 1)
  else ODEHIP_DEC_LAUNCH(8, 2)
#undef ODEHIP_DEC_LAUNCH
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}
