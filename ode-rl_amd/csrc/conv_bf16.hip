// conv_bf16.hip -- 3x3 convolution on 16x16 latent maps with bf16 operands and fp32 accumulation (gfx950 matrix cores:
// v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate).  BASELINE.json configs[4] ("ODEConvGRU bf16"): compute in bf16,
// solver state / stage derivatives / stage combines in fp32 -- what `torch.autocast(dtype=torch.bfloat16)` makes of the
// reference's nn.Conv2d layers (helpers/utils.py:167-177), except that conv OUTPUTS stay fp32 here.
//
// Same boundary as the fp32 kernels: Q4 fp32 activations in HBM, the same fused epilogues (bias, ReLU, Runge-Kutta stage
// combine, error-norm partials, ReLU-mask / reverse-sweep targets).  Only the inner product changes:
//   * weights are pre-rounded (RNE) and pre-packed in the exact A-operand image: [tap][16-channel block][lane][8 bf16],
//     1 KiB per (tap, block); the whole tile (9*cin/16 KiB <= 72 KiB) is LDS-DMA'd at kernel start;
//   * activations are read as fp32 quads (coalesced 16 B per lane), rounded with v_cvt_pk_bf16_f32 and written to an LDS tile
//     [10 rows][18 columns][cin] bf16 with zero borders (the conv padding needs no lane masks) and a pixel stride of
//     2*cin+16 B, which makes the B-operand ds_read_b128 (8 consecutive channels of 32 pixels) bank-conflict free;
//   * one workgroup = (sample, 32 output channels, 8 image rows), wave w owns image rows 2w, 2w+1: 9*cin/16 MFMAs on ONE
//     accumulator chain per wave (a single chain of this instruction issues back-to-back), two ds_read_b128 per MFMA.
// With the matrix work at ~0.6 us per launch the kernel is bound by its launch boundary, the staging of the input tile
// and the epilogue's HBM/L2 traffic, not by the matrix cores.
#include "conv_common.h"

namespace odehip {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};  // v_cvt_pk_bf16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, v);
}

// NCB = cin / 16
template <int NCB>
__global__ __launch_bounds__(256, 1) void conv3x3_bf16_kernel(const ConvArgs a) {
  constexpr int CIN = NCB * 16;
  constexpr int S = CIN * 2 + 16;            // bytes per pixel of the LDS activation tile
  constexpr int W_BYTES = 9 * NCB * 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wl = smem;
  char* const xl = smem + W_BYTES;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = xcd_block_id();
  const int ct_count = a.qout >> 3;
  const int rh = bid & 1;
  const int ct = (bid >> 1) % ct_count;
  const int b = (bid >> 1) / ct_count;
  const int r0 = rh * 8;
  if (a.skip && *a.skip) return;

  // ---- weights: 9*NCB pieces of 1 KiB, LDS-DMA, wave w takes pieces w, w+4, ...
  const __amdgpu_buffer_rsrc_t rw = make_rsrc((const char*)a.w_bf16 + (size_t)ct * W_BYTES, W_BYTES);
#pragma unroll
  for (int p = 0; p < (9 * NCB + 3) / 4; ++p) {
    const int piece = p * 4 + wave;
    if (piece < 9 * NCB) dma16(rw, wl + piece * 1024, lane * 16, piece * 1024);
  }

  // ---- activations: fp32 quads -> bf16 tile with zero borders
  {
    const f32x4* src = (const f32x4*)(a.src1 + (size_t)b * a.qin * kPix * 4);
    constexpr int NLOAD = 10 * 16 * (CIN / 4);  // f32x4 elements of the tile (rows r0-1 .. r0+8)
#pragma unroll
    for (int i = 0; i < (NLOAD + 255) / 256; ++i) {
      const int idx = i * 256 + (int)threadIdx.x;
      if (NLOAD % 256 != 0 && idx >= NLOAD) break;
      const int px = idx & 15, rr = (idx >> 4) % 10, q = idx / 160;
      const int row = r0 - 1 + rr;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row >= 0 && row < kHW) v = src[(size_t)q * kPix + row * 16 + px];
      const u32x2 o = {pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)};
      *(u32x2*)(xl + (rr * 18 + px + 1) * S + q * 8) = o;
    }
    // left / right zero columns: 10 rows x 2 columns x CIN/4 quads of 8 B
    constexpr int NZ = 10 * 2 * (CIN / 4);
    for (int idx = threadIdx.x; idx < NZ; idx += 256) {
      const int q = idx % (CIN / 4), rc = idx / (CIN / 4), rr = rc >> 1, col = (rc & 1) ? 17 : 0;
      *(u32x2*)(xl + (rr * 18 + col) * S + q * 8) = u32x2{0u, 0u};
    }
  }
  wait_vmcnt<0>();
  __syncthreads();

  // ---- 9*NCB MFMAs per wave: A = weights [co 32][k 16], B = activations [k 16][pixel 32]
  const int i32 = lane & 31, kq = lane >> 5;
  f32x16 acc = bias_init(a.bias, ct, kq);
  const int px = i32 & 15, pyl = i32 >> 4;
  const char* abase = wl + lane * 16;
  const char* bbase = xl + ((wave * 2 + pyl + 1) * 18 + px + 1) * S + kq * 16;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const bf16x8 wv = *(const bf16x8*)(abase + (tap * NCB + cb) * 1024);
      const bf16x8 xv = *(const bf16x8*)(bbase + (dy * 18 + dx) * S + cb * 32);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, xv, acc, 0, 0, 0);
    }
  }
  epilogue(a, acc, b, ct, (r0 + wave * 2) * 16 + i32, kq, wave);
}

// ---- KS x KS convs with a large K extent (the ConvGRU cell's 5x5 convs on cat(x, h), modules/ConvGRUCell.py:40-50): the whole
// input tile is converted once ([8+2H rows][16+2H cols][cin] bf16, zero borders), the weights stream through a 3-stage LDS ring,
// one 16-channel block (KS*KS KiB) per stage, with counted s_waitcnt vmcnt + one barrier per block.
// Weight image: [ct][cb][tap][lane][8 bf16] (odehip_pack_conv_weight_bf16_ks).
template <int KS>
__global__ __launch_bounds__(256, 1) void conv_bf16_ring_kernel(const ConvArgs a) {
  constexpr int HALO = KS / 2, TAPS = KS * KS, ROWS = 8 + 2 * HALO, COLS = 16 + 2 * HALO;
  constexpr int STAGE = TAPS * 1024, NSTAGE = 3;
  constexpr int G = (TAPS + 3) / 4;  // DMAs per wave per block
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wl = smem;
  char* const xl = smem + NSTAGE * STAGE;
  const int cin = a.qin * 4, ncb = cin / 16;
  const int S = cin * 2 + 16;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = xcd_block_id();
  const int ct_count = a.qout >> 3;
  const int rh = bid & 1;
  const int ct = (bid >> 1) % ct_count;
  const int b = (bid >> 1) / ct_count;
  const int r0 = rh * 8;
  if (a.skip && *a.skip) return;

  const __amdgpu_buffer_rsrc_t rw = make_rsrc((const char*)a.w_bf16 + (size_t)ct * ncb * STAGE, (unsigned)(ncb * STAGE));
  auto issue = [&](int cb) {
    char* stage = wl + (cb % NSTAGE) * STAGE;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      int piece = g * 4 + wave;
      if (piece > TAPS - 1) piece = TAPS - 1;  // surplus slots re-copy the last piece (same bytes): every wave issues G DMAs
      dma16(rw, stage + piece * 1024, lane * 16, cb * STAGE + piece * 1024);
    }
  };
  issue(0);
  if (ncb > 1) issue(1);

  // activations of both sources -> bf16 tile with zero borders
  {
    const int q2n = a.qin - a.q1;
    const f32x4* s1 = (const f32x4*)(a.src1 + (size_t)b * a.q1 * kPix * 4);
    const f32x4* s2 = a.src2 ? (const f32x4*)(a.src2 + (size_t)b * q2n * kPix * 4) : s1;
    const int nload = ROWS * 16 * a.qin;
    // 8 loads in flight per thread before the first conversion: the tile arrives in a few memory round trips, not one per element
    for (int base = threadIdx.x; base < nload; base += 256 * 8) {
      f32x4 v[8];
      int dst[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256;
        const int px = idx & 15, t = idx >> 4, rr = t % ROWS, q = t / ROWS;
        const int row = r0 - HALO + rr;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        dst[u] = idx < nload ? (rr * COLS + px + HALO) * S + q * 8 : -1;
        if (idx < nload && row >= 0 && row < kHW)
          v[u] = q < a.q1 ? s1[(size_t)q * kPix + row * 16 + px] : s2[(size_t)(q - a.q1) * kPix + row * 16 + px];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (dst[u] >= 0) *(u32x2*)(xl + dst[u]) = u32x2{pack_bf16(v[u].x, v[u].y), pack_bf16(v[u].z, v[u].w)};
    }
    const int nz = ROWS * 2 * HALO * a.qin;
    for (int idx = threadIdx.x; idx < nz; idx += 256) {
      const int q = idx % a.qin, rc = idx / a.qin, rr = rc / (2 * HALO), c = rc % (2 * HALO);
      const int col = c < HALO ? c : 16 + c;
      *(u32x2*)(xl + (rr * COLS + col) * S + q * 8) = u32x2{0u, 0u};
    }
  }

  const int i32 = lane & 31, kq = lane >> 5;
  f32x16 acc = bias_init(a.bias, ct, kq);
  const int px = i32 & 15, pyl = i32 >> 4;
  const char* bbase = xl + ((wave * 2 + pyl + HALO) * COLS + px + HALO) * S + kq * 16;
  for (int cb = 0; cb < ncb; ++cb) {
    // blocks cb (and cb+1 if issued) are in flight behind the activation loads, which have been consumed above
    if (cb + 1 < ncb) wait_vmcnt<G>(); else wait_vmcnt<0>();
    // raw barrier (a __syncthreads() would drain vmcnt to 0 and serialise the ring); lgkmcnt(0): this wave's staging writes
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // block cb landed for every wave; every wave is done with block cb-1 (its stage is free for cb+2)
    if (cb + 2 < ncb) issue(cb + 2);
    const char* abase = wl + (cb % NSTAGE) * STAGE + lane * 16;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int dy = tap / KS - HALO, dx = tap % KS - HALO;
      const bf16x8 wv = *(const bf16x8*)(abase + tap * 1024);
      const bf16x8 xv = *(const bf16x8*)(bbase + (dy * COLS + dx) * S + cb * 32);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, xv, acc, 0, 0, 0);
    }
  }
  epilogue(a, acc, b, ct, (r0 + wave * 2) * 16 + i32, kq, wave);
}

// out[ct][cb][tap][h][co32][j]: the block-major image of conv_bf16_ring_kernel
__global__ __launch_bounds__(256) void pack_weight_bf16_ks_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int cout, int cin,
                                                                  int ks, int transpose_flip, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int taps = ks * ks;
  int r = idx;
  const int j = r & 7; r >>= 3;
  const int co_l = r & 31; r >>= 5;
  const int h = r & 1; r >>= 1;
  const int tap = r % taps; r /= taps;
  const int ncb = cin / 16;
  const int cb = r % ncb;
  const int ct = r / ncb;
  const int co = ct * 32 + co_l, ci = cb * 16 + h * 8 + j;
  const float v = transpose_flip ? w[((size_t)ci * cout + co) * taps + (taps - 1 - tap)] : w[((size_t)co * cin + ci) * taps + tap];
  out[idx] = (__bf16)v;
}

template <int KS>
static int launch_bf16_ring(const ConvArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  constexpr int HALO = KS / 2, TAPS = KS * KS;
  const int cin = a.qin * 4;
  const size_t lds = (size_t)3 * TAPS * 1024 + (size_t)(8 + 2 * HALO) * (16 + 2 * HALO) * (cin * 2 + 16);
  if (lds > 160 * 1024) return 1;
  if (!attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)conv_bf16_ring_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_bf16_ring_kernel<KS>), dim3(a.batch * (a.qout / 8) * 2), dim3(256), lds, stream, a);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

// 5x5 layers (weights packed by odehip_pack_conv_weight_bf16_ks); returns 1 if the shape is not served
int launch_bf16_5x5(const ConvArgs& a, hipStream_t stream) {
  if (a.qin % 4 != 0) return 1;
  return launch_bf16_ring<5>(a, stream);
}

// out[ct][tap][cb][h][co32][j] = bf16( W[co = 32 ct + r][ci = 16 cb + 8 h + j][tap] )          (transpose_flip == 0)
//                              = bf16( W[ci][co][8 - tap] )                                     (input-gradient conv)
__global__ __launch_bounds__(256) void pack_weight_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int cout, int cin,
                                                               int transpose_flip, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  int r = idx;
  const int j = r & 7; r >>= 3;
  const int co_l = r & 31; r >>= 5;
  const int h = r & 1; r >>= 1;
  const int ncb = cin / 16;
  const int cb = r % ncb; r /= ncb;
  const int tap = r % 9;
  const int ct = r / 9;
  const int co = ct * 32 + co_l, ci = cb * 16 + h * 8 + j;
  const float v = transpose_flip ? w[((size_t)ci * cout + co) * 9 + (8 - tap)] : w[((size_t)co * cin + ci) * 9 + tap];
  out[idx] = (__bf16)v;
}

template <int NCB>
static int launch_bf16_n(const ConvArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  constexpr int CIN = NCB * 16;
  const size_t lds = (size_t)9 * NCB * 1024 + (size_t)10 * 18 * (CIN * 2 + 16);
  if (!attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_bf16_kernel<NCB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_bf16_kernel<NCB>), dim3(a.batch * (a.qout / 8) * 2), dim3(256), lds, stream, a);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

// returns 1 if the shape is not served by the bf16 kernel (caller falls through to the fp32 kernels)
int launch_bf16(const ConvArgs& a, hipStream_t stream) {
  if (a.q1 != a.qin || a.qin % 4 != 0) return 1;
  switch (a.qin / 4) {
    case 1: return launch_bf16_n<1>(a, stream);
    case 2: return launch_bf16_n<2>(a, stream);
    case 4: return launch_bf16_n<4>(a, stream);
    case 8: return launch_bf16_n<8>(a, stream);
    default: return 1;
  }
}

}  // namespace odehip

using namespace odehip;

extern "C" size_t odehip_bf16_weight_bytes(int cout, int cin) { return (size_t)cout * cin * 9 * 2; }

extern "C" int odehip_pack_conv_weight_bf16_ks(const float* w_oihw, void* w_bf16, int cout, int cin, int ks, int transpose_flip,
                                               void* stream) {
  ODEHIP_REQUIRE(w_oihw && w_bf16, "pack_conv_weight_bf16_ks: null pointer");
  ODEHIP_REQUIRE(ks == 5, "pack_conv_weight_bf16_ks: kernel size %d unsupported (5; 3x3 layers use odehip_pack_conv_weight_bf16)", ks);
  ODEHIP_REQUIRE(cout > 0 && cout % 32 == 0 && cin > 0 && cin % 16 == 0, "pack_conv_weight_bf16_ks: cout %% 32 and cin %% 16 must be 0");
  const int total = cout * cin * ks * ks;
  hipLaunchKernelGGL(pack_weight_bf16_ks_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, (__bf16*)w_bf16,
                     cout, cin, ks, transpose_flip, total);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

extern "C" int odehip_pack_conv_weight_bf16(const float* w_oihw, void* w_bf16, int cout, int cin, int transpose_flip, void* stream) {
  ODEHIP_REQUIRE(w_oihw && w_bf16, "pack_conv_weight_bf16: null pointer");
  ODEHIP_REQUIRE(cout > 0 && cout % 32 == 0, "pack_conv_weight_bf16: cout must be a multiple of 32 (got %d)", cout);
  ODEHIP_REQUIRE(cin > 0 && cin % 16 == 0, "pack_conv_weight_bf16: cin must be a multiple of 16 (got %d)", cin);
  const int total = cout * cin * 9;
  hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, (__bf16*)w_bf16, cout,
                     cin, transpose_flip, total);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}
