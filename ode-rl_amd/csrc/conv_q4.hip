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
//     rows of one quad -- contiguous in HBM *and* lane-linear in LDS (LDS-DMA needs that).
//   * weights are pre-packed (layout.hip) into the exact LDS image: per (8-channel group m, tap)
//     one 1 KiB piece [kq][co32][4].  Operand fragments are single ds_read_b128 reads, bank-conflict
//     free for both operands (16 consecutive 16-B slots per 16-lane group).
//   * operands reach LDS by `buffer_load_dwordx4 ... lds` (no VGPR staging): per-lane offset in a VGPR
//     that never changes, piece offset in an SGPR, so a DMA costs two scalar instructions.  Image rows
//     outside [0,16) are given an out-of-range offset: the buffer range check writes zeros to LDS, which
//     is the conv's zero padding (measured: tools/experiments/buflds.hip).  The x halo is handled by
//     zeroing the B fragment of edge lanes.
//   * two kernels: `conv3x3_resident_kernel` (cin <= 64: the whole K extent, <= 120 KiB, is DMA'd up
//     front and multiplied behind two counted waits) and `conv_ring_kernel` (any cin, 1x1/3x3/5x5: K is
//     streamed in chunks through an NBUF-deep LDS ring with counted s_waitcnt vmcnt(N) + raw s_barrier).
#include <type_traits>

#include "conv_common.h"

namespace odehip {

__device__ __forceinline__ void mfma4(f32x16& acc, const f32x4& wv, const f32x4& xv) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, xv.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, xv.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, xv.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, xv.w, acc, 0, 0, 0);
}

// =================================================================================================
// Resident 3x3 kernel: cin = 16*NCHUNK <= 64.  LDS stage c = [18 KiB weights of channels 16c..16c+15 |
// 4 quads x 12 rows (3 pieces) of input], all NCHUNK stages DMA'd at kernel start (8 DMAs per wave
// per stage: 5 weight pieces (waves 2,3 re-copy piece 17 once) + the 3 pieces of quad `wave`).
// =================================================================================================
// The first six dwords (src, weights, cin/4, cout/4) are separate leading arguments so that the hardware preloads them
// into SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count): the first DMA does not wait for an s_load.
template <int NCHUNK, bool DBG>
__global__ __launch_bounds__(256, 1) void conv3x3_resident_kernel(const float* __restrict__ p_src, const float* __restrict__ p_w,
                                                                  int p_qin, int p_qout, const ConvArgs a) {
  constexpr int TAPS = 9, NP = 3, MC = 2;
  constexpr int W_BYTES = MC * TAPS * 1024, IN_BYTES = 2 * MC * NP * 1024, STAGE = W_BYTES + IN_BYTES;
  constexpr int G = 8;          // DMAs per wave per stage
  constexpr int NG = MC * TAPS; // fragment groups (4 MFMAs each) per stage
  constexpr int ZERO_OFF = NCHUNK * STAGE;  // 2 KiB of zeros: where edge lanes read their x halo
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Stamps st(a);
  if (DBG) st.take(0);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // grid = (2 * cout/32, batch): no division, no hidden-argument load on the way to the first DMA
  const int rh = blockIdx.x & 1;
  const int ct = blockIdx.x >> 1;
  const int b = blockIdx.y;
  const int r0 = rh * 8;
  const bool dma = !DBG || !(a.debug & 1), mfma = !DBG || !(a.debug & 2);

  (void)p_qout;
  const unsigned tile_w_bytes = (unsigned)NCHUNK * W_BYTES;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc((const char*)p_w + (size_t)ct * tile_w_bytes, tile_w_bytes);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc((const char*)p_src + (size_t)b * p_qin * kQuadBytes, (unsigned)p_qin * kQuadBytes);
  const int vw = lane * 16;
  int vx[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int row = r0 - 1 + 4 * j + (lane >> 4);
    vx[j] = (row >= 0 && row < kHW) ? row * 256 + (lane & 15) * 16 : kOobOffset;
  }
  // DMA number g (0..7) of stage c, for this wave: 5 weight pieces (waves 2,3 re-copy piece 17 once: same
  // bytes) and the 3 row pieces of channel quad `wave` of the stage
  auto issue_one = [&](int c, int g) {
    char* stage = smem + c * STAGE;
    if (g < 5) {
      int p = g * 4 + wave;
      p = p > MC * TAPS - 1 ? MC * TAPS - 1 : p;
      dma16(rw, stage + p * 1024, vw, (c * MC * TAPS + p) * 1024);
    } else {
      const int j = g - 5;
      dma16(rx, stage + W_BYTES + (wave * NP + j) * 1024, vx[j], (c * 2 * MC + wave) * kQuadBytes);
    }
  };
  if (dma) {
#pragma unroll
    for (int g = 0; g < G; ++g) issue_one(0, g);
  }
  if (a.skip && *a.skip) {  // adaptive solver finished while this launch was queued: drain the DMAs and leave
    wait_vmcnt<0>();
    return;
  }
  if (DBG) st.take(1);
  // the zero page of the x halo (read by edge lanes instead of the wrapped neighbour pixel)
  *(f32x4*)(smem + ZERO_OFF + (threadIdx.x >> 1) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

  const int i32 = lane & 31;   // A row (co) / B col (pixel) of this lane
  const int kq = lane >> 5;    // which half of an 8-channel group this lane feeds
  // two accumulator chains (even / odd channels of each quad)
  f32x16 acc0 = bias_init(a.bias, ct, kq), acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc1[r] = 0.0f;
  const int px = i32 & 15, pyl = i32 >> 4;
  const int w_lane = kq * 512 + i32 * 16;
  const int x_lane = W_BYTES + kq * NP * 1024 + (wave * 2 + pyl + 1) * 256 + px * 16;

  // Per-stage base addresses so that every fragment read is base + 16-bit immediate.  For the dx = -1 / +1 taps
  // the edge lanes (px = 0 / 15) are pointed at the zero page: no masking in the MFMA stream.  The x bases are
  // biased by -272 so that the tap offsets (dy+1)*256 + (dx+1)*16 are non-negative immediates.
  int wb[NCHUNK], xc[NCHUNK][MC], xl[NCHUNK][MC], xr[NCHUNK][MC];
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    wb[c] = c * STAGE + w_lane;
    asm volatile("" : "+v"(wb[c]));
#pragma unroll
    for (int mm = 0; mm < MC; ++mm) {
      xc[c][mm] = c * STAGE + x_lane + mm * 2 * NP * 1024 - 272;
      xl[c][mm] = px == 0 ? ZERO_OFF + 1024 - 272 : xc[c][mm];
      xr[c][mm] = px == 15 ? ZERO_OFF + 1024 - 272 : xc[c][mm];
      asm volatile("" : "+v"(xc[c][mm]), "+v"(xl[c][mm]), "+v"(xr[c][mm]));
    }
  }
  // fragments of flat group index fg = stage * NG + (mm, tap)
  auto load_frag = [&](int fg, f32x4& wv, f32x4& xv) {
    const int c = fg / NG, gi = fg % NG;
    const int mm = gi / TAPS, tap = gi % TAPS;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    wv = *(const f32x4*)(smem + wb[c] + (mm * TAPS + tap) * 1024);
    const int xb = dx < 0 ? xl[c][mm] : (dx > 0 ? xr[c][mm] : xc[c][mm]);
    xv = *(const f32x4*)(smem + xb + (dy + 1) * 256 + (dx + 1) * 16);
  };

  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();  // stage 0 and the zero page are visible to every wave
  if (DBG) st.take(2);
  if (mfma) {
    // One software pipeline over all NCHUNK*NG groups of 4 MFMAs; one scheduling region per group:
    //   MFMA | the 2 fragment reads of the NEXT group | MFMA | [one DMA of the next stage] | 2 MFMA
    // so the reads get ~200 cycles of MFMA cover.  The DMAs of stage c+1 are issued in the first G groups of
    // stage c and awaited (vmcnt(0) + barrier) two groups before stage c ends, so the pipeline never drains.
    f32x4 wv, xv, wn, xn;
    load_frag(0, wv, xv);
#pragma unroll
    for (int fg = 0; fg < NCHUNK * NG; ++fg) {
      const int c = fg / NG, gi = fg % NG;
      if (gi == NG - 2 && c + 1 < NCHUNK) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();  // stage c+1 landed for every wave
      }
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, xv.x, acc0, 0, 0, 0);
      if (fg + 1 < NCHUNK * NG) load_frag(fg + 1, wn, xn);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, xv.y, acc1, 0, 0, 0);
      const bool with_dma = c + 1 < NCHUNK && gi < G;
      if (with_dma && dma) issue_one(c + 1, gi);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, xv.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, xv.w, acc1, 0, 0, 0);
      wv = wn;
      xv = xn;
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (with_dma) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (DBG && gi == NG - 1 && st.on) {
        asm volatile("" ::"v"(acc0[0]), "v"(acc1[0]));
        st.take(3 + c);
      }
    }
  } else if (dma) {
#pragma unroll
    for (int c = 1; c < NCHUNK; ++c) {
#pragma unroll
      for (int g = 0; g < G; ++g) issue_one(c, g);
    }
    wait_vmcnt<0>();
  }
  f32x16 acc = acc0 + acc1;
  if (DBG && (a.debug & 4)) {
    if (acc[0] == 12345.678f) a.dst[0] = acc[1];  // keep the accumulators live
    return;
  }
  epilogue(a, acc, b, ct, (r0 + wave * 2) * 16 + i32, kq, wave);
  if (DBG) st.flush(a);
}

// =================================================================================================
// Ring kernel: any cin (multiple of 8*MC), KS in {1,3,5}, optional second source (torch.cat(x, h)).
// =================================================================================================
template <int KS, int MC>
struct RingCfg {
  static constexpr int TAPS = KS * KS;
  static constexpr int HALO = KS / 2;
  static constexpr int NP = (KS == 1) ? 2 : 3;          // 1 KiB pieces (4 image rows) per channel quad
  static constexpr int W_BYTES = MC * TAPS * 1024;      // weights of one chunk (first in a stage)
  static constexpr int IN_BYTES = 2 * MC * NP * 1024;   // input rows of one chunk
  static constexpr int STAGE_BYTES = W_BYTES + IN_BYTES;
  static constexpr int PIECES = STAGE_BYTES / 1024;
  static constexpr int G = (PIECES + 3) / 4;            // DMAs per wave per chunk
};

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

// MINB = 2: a two-stage ring (62 KiB for 5x5) so that TWO workgroups share a CU -- with one wave per SIMD nothing else covers a
// workgroup's ring fill, per-chunk barrier and epilogue; used when the grid has more workgroups than CUs anyway
template <int KS, int MC, int NBUF, int MINB = 1>
__global__ __launch_bounds__(256, MINB) void conv_ring_kernel(const ConvArgs a) {
  using C = RingCfg<KS, MC>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Stamps st(a);
  st.take(0);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = xcd_block_id();
  const int ct_count = a.qout >> 3;
  const int rh = bid & 1;
  const int ct = (bid >> 1) % ct_count;
  const int b = (bid >> 1) / ct_count;
  const int r0 = rh * 8;
  const int nchunk = a.qin / (2 * MC);
  const bool dma = !(a.debug & 1), mfma = !(a.debug & 2);

  const unsigned tile_w_bytes = (unsigned)(a.qin / 2) * C::TAPS * 1024;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc((const char*)a.w_packed + (size_t)ct * tile_w_bytes, tile_w_bytes);
  const __amdgpu_buffer_rsrc_t rx1 =
      make_rsrc((const char*)a.src1 + (size_t)b * a.q1 * kQuadBytes, (unsigned)a.q1 * kQuadBytes);
  const int q2n = a.qin - a.q1;
  const __amdgpu_buffer_rsrc_t rx2 = make_rsrc(
      a.src2 ? (const char*)a.src2 + (size_t)b * q2n * kQuadBytes : (const char*)a.src1, (unsigned)q2n * kQuadBytes);
  const int vw = lane * 16;
  int vx[C::NP];
#pragma unroll
  for (int j = 0; j < C::NP; ++j) {
    const int row = r0 - C::HALO + 4 * j + (lane >> 4);
    vx[j] = (row >= 0 && row < kHW) ? row * 256 + (lane & 15) * 16 : kOobOffset;
  }

  auto issue = [&](int c, int s) {
    char* stage = smem + s * C::STAGE_BYTES;
#pragma unroll
    for (int g = 0; g < C::G; ++g) {
      int p = g * 4 + wave;
      if (p > C::PIECES - 1) p = C::PIECES - 1;  // surplus slots re-copy the last piece (same bytes)
      if (p < MC * C::TAPS) {
        dma16(rw, stage + p * 1024, vw, (c * MC * C::TAPS + p) * 1024);
      } else {
        const int ip = p - MC * C::TAPS;
        const int ql = ip / C::NP, j = ip - ql * C::NP;
        const int q = c * 2 * MC + ql;
        int vo = vx[0];
        if (C::NP > 1 && j == 1) vo = vx[1];
        if (C::NP > 2 && j == 2) vo = vx[C::NP - 1];
        if (q < a.q1) dma16(rx1, stage + p * 1024, vo, q * kQuadBytes);
        else dma16(rx2, stage + p * 1024, vo, (q - a.q1) * kQuadBytes);
      }
    }
  };

  const int i32 = lane & 31, kq = lane >> 5;
  f32x16 acc = bias_init(a.bias, ct, kq);
  const int px = i32 & 15, pyl = i32 >> 4;
  const int a_off = kq * 512 + i32 * 16;
  const int b_off = kq * C::NP * 1024 + (wave * 2 + pyl + C::HALO) * 256 + px * 16;

  auto load_frag = [&](int s, int gi, f32x4& wv, f32x4& xv) {
    const char* wb = smem + s * C::STAGE_BYTES;
    const char* ib = wb + C::W_BYTES;
    const int mm = gi / C::TAPS, tap = gi % C::TAPS;
    const int dy = tap / KS - C::HALO, dx = tap % KS - C::HALO;
    wv = *(const f32x4*)(wb + (mm * C::TAPS + tap) * 1024 + a_off);
    xv = *(const f32x4*)(ib + mm * 2 * C::NP * 1024 + b_off + dy * 256 + dx * 16);
    if (dx != 0) {
      const bool ok = (dx < 0) ? (px + dx >= 0) : (px + dx <= 15);
      xv.x = ok ? xv.x : 0.0f;
      xv.y = ok ? xv.y : 0.0f;
      xv.z = ok ? xv.z : 0.0f;
      xv.w = ok ? xv.w : 0.0f;
    }
  };
  auto compute = [&](int s) {
    constexpr int NG = MC * C::TAPS;
    f32x4 wv, xv, wn, xn;
    load_frag(s, 0, wv, xv);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 1 < NG) load_frag(s, gi + 1, wn, xn);
      mfma4(acc, wv, xv);
      wv = wn;
      xv = xn;
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    }
  };

  // ---- K pipeline: NBUF-deep ring, counted waits, one raw barrier per chunk
  if (a.skip && *a.skip) return;  // adaptive solver finished while this launch was queued
  const int pre = (NBUF - 1 < nchunk) ? NBUF - 1 : nchunk;
  if (dma)
    for (int c = 0; c < pre; ++c) issue(c, c);
  st.take(1);
  for (int c = 0; c < nchunk; ++c) {
    int issued = c + NBUF - 1;
    if (issued > nchunk) issued = nchunk;
    wait_chunk<C::G, NBUF - 2>(issued - (c + 1));
    __builtin_amdgcn_s_barrier();  // chunk c landed for every wave; every wave is done with chunk c-1
    if (c == 0) st.take(2);
    if (dma && c + NBUF - 1 < nchunk) issue(c + NBUF - 1, (c + NBUF - 1) % NBUF);
    if (mfma) compute(c % NBUF);
  }
  if (st.on) {
    asm volatile("" ::"v"(acc[0]));
    st.take(6);
  }
  if (a.debug & 4) {
    if (acc[0] == 12345.678f) a.dst[0] = acc[1];
    return;
  }
  epilogue(a, acc, b, ct, (r0 + wave * 2) * 16 + i32, kq, wave);
  st.flush(a);
}

// ---------------------------------------------------------------------------------------------- host
template <typename K>
static int prepare_kernel(K kernel, size_t lds, bool* attr_set) {
  ODEHIP_REQUIRE(lds <= 160 * 1024, "conv_q4: LDS request %zu exceeds 160 KiB", lds);
  if (!*attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    *attr_set = true;
  }
  return ODEHIP_OK;
}

template <int NCHUNK>
static int launch_resident(const ConvArgs& a, hipStream_t stream) {
  static bool attr_set = false, attr_set_dbg = false;
  const size_t lds = (size_t)NCHUNK * 30 * 1024 + 2048;
  const dim3 grid((a.qout / 8) * 2, a.batch);
  int rc;
  if (a.debug) {
    if ((rc = prepare_kernel(conv3x3_resident_kernel<NCHUNK, true>, lds, &attr_set_dbg)) != ODEHIP_OK) return rc;
    hipLaunchKernelGGL((conv3x3_resident_kernel<NCHUNK, true>), grid, dim3(256), lds, stream, a.src1, a.w_packed, a.qin, a.qout, a);
  } else {
    if ((rc = prepare_kernel(conv3x3_resident_kernel<NCHUNK, false>, lds, &attr_set)) != ODEHIP_OK) return rc;
    hipLaunchKernelGGL((conv3x3_resident_kernel<NCHUNK, false>), grid, dim3(256), lds, stream, a.src1, a.w_packed, a.qin, a.qout, a);
  }
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

template <int KS, int MC, int NBUF, int MINB = 1>
static int launch_ring(const ConvArgs& a, hipStream_t stream) {
  using C = RingCfg<KS, MC>;
  static bool attr_set = false;
  const int nchunk = a.qin / (2 * MC);
  const int nbuf_alloc = (NBUF < nchunk) ? NBUF : nchunk;
  const size_t lds = (size_t)nbuf_alloc * C::STAGE_BYTES + 64;
  int rc = prepare_kernel(conv_ring_kernel<KS, MC, NBUF, MINB>, lds, &attr_set);
  if (rc != ODEHIP_OK) return rc;
  hipLaunchKernelGGL((conv_ring_kernel<KS, MC, NBUF, MINB>), dim3(a.batch * (a.qout / 8) * 2), dim3(256), lds, stream, a);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

int g_debug_flags = 0;
unsigned long long* g_debug_buf = nullptr;

thread_local ConvRecorder* g_conv_recorder = nullptr;

// an elementwise row (ConvArgs::combine == 4) outside a persistent walk: the same fma sequence per element (ew_quad)
__global__ __launch_bounds__(256) void ew_row_kernel(const ConvArgs a, long long n_quads) {
  if (a.skip && *a.skip) return;
  const float hs = a.cmb.h_ptr ? *a.cmb.h_ptr : 1.0f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_quads; i += (long long)gridDim.x * 256) ew_quad(a.cmb, (size_t)i * 4, hs);
}
int launch_ew_row(const ConvArgs& a, hipStream_t stream) {
  ODEHIP_REQUIRE(a.batch > 0 && a.qout > 0 && (a.cmb.out1 || a.cmb.out2) && a.cmb.n_prev >= 0 && a.cmb.n_prev <= ODEHIP_MAX_STAGES,
                 "elementwise row: bad arguments");
  const long long n_quads = (long long)a.batch * a.qout * kPix;
  hipLaunchKernelGGL(ew_row_kernel, dim3((unsigned)((n_quads + 255) / 256 < 2048 ? (n_quads + 255) / 256 : 2048)), dim3(256), 0, stream, a, n_quads);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

int launch_conv(const ConvArgs& a_in, int ks, hipStream_t stream) {
  if (g_conv_recorder) {
    ConvRecorder* r = g_conv_recorder;
    ODEHIP_REQUIRE(r->count < r->capacity, "conv recorder: more than %d layers", r->capacity);
    r->items[r->count++] = a_in;
    return ODEHIP_OK;
  }
  if (a_in.combine == 4) return launch_ew_row(a_in, stream);
  ODEHIP_REQUIRE(a_in.combine != 5, "a norm row only exists inside the adaptive persistent walk");
  ConvArgs a = a_in;
  a.debug = g_debug_flags;
  a.dbg = g_debug_buf;
  if (!a.dbg) a.debug &= ~8;
  ODEHIP_REQUIRE(a.batch > 0, "conv_q4: batch must be positive (got %d)", a.batch);
  ODEHIP_REQUIRE(a.qout > 0 && a.qout % 8 == 0, "conv_q4: cout must be a multiple of 32 (got %d)", a.qout * 4);
  ODEHIP_REQUIRE(a.src1 && a.w_packed, "conv_q4: null pointer argument");
  ODEHIP_REQUIRE(a.q1 > 0 && a.q1 <= a.qin && (a.q1 == a.qin || a.src2), "conv_q4: bad input split");
  ODEHIP_REQUIRE((size_t)a.qin * kQuadBytes < (size_t)kOobOffset, "conv_q4: cin too large");
  if (ks == 3) {
    ODEHIP_REQUIRE(a.qin % 4 == 0, "conv_q4: 3x3 needs cin %% 16 == 0 (got %d)", a.qin * 4);
    if (a.w_bf16) {
      const int rb = launch_bf16(a, stream);
      if (rb != 1) return rb;
    }
    if (a.w_wino && a.q1 == a.qin && !(g_debug_flags & 64)) {
      const int rw = launch_wino(a, stream);
      if (rw != 1) return rw;
    }
    if (a.q1 == a.qin && !(g_debug_flags & 16)) {
      switch (a.qin / 4) {
        case 1: return launch_resident<1>(a, stream);
        case 2: return launch_resident<2>(a, stream);
        case 3: return launch_resident<3>(a, stream);
        case 4: return launch_resident<4>(a, stream);
        default: break;
      }
    }
    return launch_ring<3, 2, 5>(a, stream);
  }
  if (ks == 5) {
    ODEHIP_REQUIRE(a.qin % 2 == 0, "conv_q4: 5x5 needs cin %% 8 == 0 (got %d)", a.qin * 4);
    if (a.w_bf16) {
      const int rb = launch_bf16_5x5(a, stream);
      if (rb != 1) return rb;
    }
    if (a.w_wino && !(g_debug_flags & 64)) {
      const int rw = launch_wino5(a, stream);
      if (rw != 1) return rw;
    }
    if (a.batch * (a.qout / 8) * 2 > 256 && !(g_debug_flags & 32)) return launch_ring<5, 1, 2, 2>(a, stream);
    return launch_ring<5, 1, 4>(a, stream);
  }
  if (ks == 1) {
    ODEHIP_REQUIRE(a.qin % 4 == 0, "conv_q4: 1x1 needs cin %% 16 == 0 (got %d)", a.qin * 4);
    return launch_ring<1, 2, 4>(a, stream);
  }
  set_error("conv_q4: unsupported kernel size %d (1, 3, 5 supported)", ks);
  return ODEHIP_EINVAL;
}

}  // namespace odehip
