// conv_q4.hip -- implicit-GEMM convolution on 16x16 latent maps for gfx950 (MI355X), exact fp32.
//
// Computes one nn.Conv2d(cin, cout, KS, stride 1, pad KS/2) (+bias, +ReLU or the Runge-Kutta
// stage combine) of the dynamics f built by the reference's `create_convnet`
// (/root/reference/helpers/utils.py:158-183), or one of the ConvGRU cell's convs
// (/root/reference/modules/ConvGRUCell.py:40-50).
//
// Mapping:
//   * GEMM view: D[co][pixel] = sum_{tap,ci} W[co][ci][tap] * X[ci][pixel+tap], M = co, N = pixel.
//   * one workgroup = 4 waves = (sample b, 32 output channels, 8 image rows); each wave owns a
//     32(co) x 32(px = 2 image rows) accumulator tile of `v_mfma_f32_32x32x2_f32` (exact fp32,
//     16 accumulator registers, one chain).
//   * activations live in HBM in the Q4 layout [b][c/4][y][x][4]: a channel quad of a pixel is
//     16 contiguous bytes, an image row of a quad 256 B, so a wave-wide 1 KiB LDS-DMA piece is 4 image
//     rows of one quad -- contiguous in HBM *and* lane-linear in LDS (global_load_lds needs that).
//   * weights are pre-packed (pack_weights.hip) into the exact LDS image: per (8-channel group m, tap)
//     one 1 KiB piece [kq][co32][4].  Operand fragments are single ds_read_b128 reads, bank-conflict
//     free for both operands (16 consecutive 16-B slots per 16-lane group).
//   * K is streamed in chunks of 8*MC input channels through an NBUF-deep LDS ring filled by
//     global_load_lds_dwordx4 (no VGPR staging) with counted s_waitcnt vmcnt(N) + raw s_barrier, so
//     later chunks stay in flight while earlier ones are being multiplied.
//   * image-row halo: out-of-image rows are DMA'd from a zero page (per-lane source address), so
//     no LDS zero-fill pass; the x halo is handled by zeroing the B fragment of edge lanes.
#include "odehip_internal.h"

namespace odehip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ODEHIP_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define ODEHIP_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int KS, int MC>
struct ConvCfg {
  static constexpr int TAPS = KS * KS;
  static constexpr int HALO = KS / 2;
  static constexpr int NP = (KS == 1) ? 2 : 3;          // 1 KiB pieces (4 image rows) per channel quad
  static constexpr int W_BYTES = MC * TAPS * 1024;      // weights of one chunk (first in a stage)
  static constexpr int IN_BYTES = 2 * MC * NP * 1024;   // input rows of one chunk
  static constexpr int STAGE_BYTES = W_BYTES + IN_BYTES;
  static constexpr int PIECES = STAGE_BYTES / 1024;
  static constexpr int G = (PIECES + 3) / 4;            // global_load_lds per wave per chunk
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// waits until at most `younger` chunks (G loads each) issued after the awaited one are still in flight
template <int G, int MAXY>
__device__ __forceinline__ void wait_chunk(int younger) {
  static_assert(G * MAXY <= 63, "vmcnt is a 6-bit counter");
  if constexpr (MAXY >= 3) {
    if (younger >= 3) { wait_vmcnt<3 * G>(); return; }
  }
  if constexpr (MAXY >= 2) {
    if (younger == 2) { wait_vmcnt<2 * G>(); return; }
  }
  if constexpr (MAXY >= 1) {
    if (younger == 1) { wait_vmcnt<1 * G>(); return; }
  }
  wait_vmcnt<0>();
}

template <int KS, int MC, int NBUF>
__global__ __launch_bounds__(256, 1) void conv_q4_kernel(const ConvArgs a) {
  using C = ConvCfg<KS, MC>;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  // XCD-aware id: blocks p and p+8 share an XCD (round-robin dispatch, speed only); give each XCD a
  // contiguous range of logical ids so the 2*CT workgroups of one sample hit the same L2.
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);

  const int ct_count = a.qout >> 3;
  const int rh = bid & 1;
  const int ct = (bid >> 1) % ct_count;
  const int b = (bid >> 1) / ct_count;
  const int r0 = rh * 8;

  const int nchunk = a.qin / (2 * MC);
  const char* wp_tile = (const char*)a.w_packed + (size_t)ct * a.qin / 2 * C::TAPS * 1024;
  const char* s1 = (const char*)a.src1 + (size_t)b * a.q1 * kQuadBytes;
  const char* s2 = a.src2 ? (const char*)a.src2 + (size_t)b * (a.qin - a.q1) * kQuadBytes : nullptr;
  const char* zp = (const char*)a.zero_page + lane * 16;
  const int lrow = lane >> 4;          // row of this lane inside a 4-row piece
  const int lx16 = (lane & 15) * 16;   // byte offset of this lane's pixel inside a row

  auto issue = [&](int c, int s) {
    char* stage = smem + s * C::STAGE_BYTES;
#pragma unroll
    for (int g = 0; g < C::G; ++g) {
      int p = g * 4 + wave;
      if (p > C::PIECES - 1) p = C::PIECES - 1;  // surplus slots re-copy the last piece (same bytes)
      const char* src;
      if (p < MC * C::TAPS) {
        src = wp_tile + ((size_t)c * MC * C::TAPS + p) * 1024 + lane * 16;
      } else {
        const int ip = p - MC * C::TAPS;
        const int ql = ip / C::NP, j = ip - ql * C::NP;
        const int q = c * 2 * MC + ql;
        const int row = r0 - C::HALO + 4 * j + lrow;
        const char* plane = (q < a.q1) ? s1 + (size_t)q * kQuadBytes : s2 + (size_t)(q - a.q1) * kQuadBytes;
        src = (row >= 0 && row < kHW) ? plane + row * 256 + lx16 : zp;
      }
      __builtin_amdgcn_global_load_lds(ODEHIP_GLOBAL_PTR(src), ODEHIP_LDS_PTR(stage + p * 1024), 16, 0, 0);
    }
  };

  // ---- accumulators start at the bias (D row = co, D col = pixel)
  const int i32 = lane & 31;   // A row (co) / B col (pixel) of this lane
  const int kq = lane >> 5;    // which half of the 8-channel group this lane feeds
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
    acc[r] = a.bias ? a.bias[co] : 0.0f;
  }

  const int px = i32 & 15, pyl = i32 >> 4;
  const int a_off = kq * 512 + i32 * 16;
  const int b_off = kq * C::NP * 1024 + (wave * 2 + pyl + C::HALO) * 256 + px * 16;

  auto compute = [&](int s) {
    const char* wb = smem + s * C::STAGE_BYTES;
    const char* ib = wb + C::W_BYTES;
#pragma unroll
    for (int mm = 0; mm < MC; ++mm) {
#pragma unroll
      for (int tap = 0; tap < C::TAPS; ++tap) {
        const int dy = tap / KS - C::HALO, dx = tap % KS - C::HALO;
        const f32x4 wv = *(const f32x4*)(wb + (mm * C::TAPS + tap) * 1024 + a_off);
        f32x4 xv = *(const f32x4*)(ib + mm * 2 * C::NP * 1024 + b_off + dy * 256 + dx * 16);
        if (dx != 0) {
          const bool ok = (dx < 0) ? (px + dx >= 0) : (px + dx <= 15);
          xv.x = ok ? xv.x : 0.0f;
          xv.y = ok ? xv.y : 0.0f;
          xv.z = ok ? xv.z : 0.0f;
          xv.w = ok ? xv.w : 0.0f;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, xv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, xv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, xv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, xv.w, acc, 0, 0, 0);
      }
    }
  };

  // ---- K pipeline: NBUF-deep ring, counted waits, one raw barrier per chunk
  const int pre = (NBUF - 1 < nchunk) ? NBUF - 1 : nchunk;
  for (int c = 0; c < pre; ++c) issue(c, c);
  for (int c = 0; c < nchunk; ++c) {
    int issued = c + NBUF - 1;
    if (issued > nchunk) issued = nchunk;
    wait_chunk<C::G, NBUF - 2>(issued - (c + 1));
    __builtin_amdgcn_s_barrier();  // chunk c landed for every wave; every wave is done with chunk c-1
    if (c + NBUF - 1 < nchunk) issue(c + NBUF - 1, (c + NBUF - 1) % NBUF);
    compute(c % NBUF);
  }

  // ---- epilogue: lane (pixel i32, half kq) holds quads 2g+kq of this 32-channel tile
  const int P = (r0 + wave * 2) * 16 + i32;
  if (!a.combine) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
      if (a.relu) {
        v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
      }
      const size_t off = (((size_t)b * a.qout + ct * 8 + 2 * g + kq) * kPix + P) * 4;
      *(f32x4*)(a.dst + off) = v;
    }
  } else {
    const CombineArgs& m = a.cmb;
    const float h = m.h_ptr ? *m.h_ptr : 1.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int Q = ct * 8 + 2 * g + kq;
      const size_t off = (((size_t)b * a.qout + Q) * kPix + P) * 4;
      f32x4 kc = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
      kc *= m.k_scale;
      if (m.k_out) *(f32x4*)(m.k_out + off) = kc;
      if (m.y) {
        const f32x4 yv = *(const f32x4*)(m.y + off);
        f32x4 sa = kc * m.c1[m.n_prev];
        f32x4 sb = kc * m.c2[m.n_prev];
        for (int j = 0; j < m.n_prev; ++j) {
          const f32x4 kp = *(const f32x4*)(m.k_prev[j] + off);
          sa += kp * m.c1[j];
          sb += kp * m.c2[j];
        }
        if (m.out1) *(f32x4*)(m.out1 + off) = yv + sa * h;
        const f32x4 o2 = yv + sb * h;
        if (m.out2) *(f32x4*)(m.out2 + off) = o2;
        if (m.out2_nchw) {
          float* o = m.out2_nchw + ((size_t)b * a.qout * 4 + Q * 4) * kPix + P;
          o[0] = o2.x; o[kPix] = o2.y; o[2 * kPix] = o2.z; o[3 * kPix] = o2.w;
        }
      }
    }
  }
}

template <int KS, int MC, int NBUF>
static int launch_cfg(const ConvArgs& a, hipStream_t stream) {
  using C = ConvCfg<KS, MC>;
  const int nchunk = a.qin / (2 * MC);
  const int nbuf_alloc = (NBUF < nchunk) ? NBUF : nchunk;
  const size_t lds = (size_t)nbuf_alloc * C::STAGE_BYTES + 64;
  static bool attr_set = false;
  if (!attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)conv_q4_kernel<KS, MC, NBUF>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  ODEHIP_REQUIRE(lds <= 160 * 1024, "conv_q4: LDS request %zu exceeds 160 KiB", lds);
  const int grid = a.batch * (a.qout / 8) * 2;
  hipLaunchKernelGGL((conv_q4_kernel<KS, MC, NBUF>), dim3(grid), dim3(256), lds, stream, a);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

int launch_conv(const ConvArgs& a, int ks, hipStream_t stream) {
  ODEHIP_REQUIRE(a.batch > 0, "conv_q4: batch must be positive (got %d)", a.batch);
  ODEHIP_REQUIRE(a.qout > 0 && a.qout % 8 == 0, "conv_q4: cout must be a multiple of 32 (got %d)", a.qout * 4);
  ODEHIP_REQUIRE(a.src1 && a.w_packed && a.zero_page, "conv_q4: null pointer argument");
  ODEHIP_REQUIRE(a.q1 > 0 && a.q1 <= a.qin && (a.q1 == a.qin || a.src2), "conv_q4: bad input split");
  if (ks == 3) {
    ODEHIP_REQUIRE(a.qin % 4 == 0, "conv_q4: 3x3 needs cin %% 16 == 0 (got %d)", a.qin * 4);
    return launch_cfg<3, 2, 5>(a, stream);
  }
  if (ks == 5) {
    ODEHIP_REQUIRE(a.qin % 2 == 0, "conv_q4: 5x5 needs cin %% 8 == 0 (got %d)", a.qin * 4);
    return launch_cfg<5, 1, 4>(a, stream);
  }
  if (ks == 1) {
    ODEHIP_REQUIRE(a.qin % 4 == 0, "conv_q4: 1x1 needs cin %% 16 == 0 (got %d)", a.qin * 4);
    return launch_cfg<1, 2, 4>(a, stream);
  }
  set_error("conv_q4: unsupported kernel size %d (1, 3, 5 supported)", ks);
  return ODEHIP_EINVAL;
}

}  // namespace odehip
