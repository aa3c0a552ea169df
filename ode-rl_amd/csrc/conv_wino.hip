// conv_wino.hip -- 3x3 convolution on 16x16 maps by Winograd F(2x2, 3x3) on exact-fp32 MFMA (gfx950).
//
// Same contract as conv3x3_resident_kernel (conv_q4.hip) -- one nn.Conv2d(cin, cout, 3, 1, 1) of the dynamics f
// (/root/reference/helpers/utils.py:167-177) with the fused epilogues -- but 2.25x fewer multiplies:
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A     per 2x2 output tile, 4x4 input patch d, 3x3 filter g.
// The sum over ci for each of the 16 transform positions xi is a GEMM  M_xi[co][tile] = U_xi[co][ci] V_xi[ci][tile].
// The arithmetic is still fp32 throughout (reassociated, not reduced precision); rel-L2 vs direct fp32 conv ~1e-6.
//
// Workgroup = (sample, 32 output channels, 8 output rows = 4x8 tiles of 2x2), 512 threads, ROLE-SPECIALISED waves
// (two per SIMD: one of each role):
//   waves 0-3  CONSUMERS: wave w owns the (16 co x 16 tiles) block (w>>1, w&1) for ALL 16 xi -- 16 accumulators of
//              v_mfma_f32_16x16x4_f32 (64 AGPRs).  Per 16-channel chunk: 32 ds_read_b128 feed 64 MFMAs.  Because a
//              lane ends up holding M_xi[4 consecutive co][its tile] for every xi, the output transform A^T M A is
//              register-only and its result is exactly one Q4 channel quad per pixel: no LDS exchange.
//   waves 4-7  PRODUCERS: LDS-DMA of the next chunk's pre-transformed weights U (32 KiB) and of the raw input tile
//              two chunks ahead (4 quads x 10 rows x 18 cols, zero-PADDED BY THE DMA: per-lane source offsets, out-of-
//              image slots get an out-of-range offset and the buffer range check writes zeros), then the input
//              transform V = B^T d B of the next chunk (12 ds_read_b128 + 8 ds_write_b128 + float4 VALU per thread).
// One s_barrier per chunk hands U_c / V_c to the consumers and the freed buffers back to the producers.
// LDS: U[2] 64 KiB + raw[2] 32 KiB + V[2] 64 KiB = 160 KiB.
#include "conv_common.h"

namespace odehip {

constexpr int kWU = 32 * 1024;        // U chunk: 16 xi x [quad 4][co 32][4 ci]
constexpr int kWRaw = 16 * 1024;      // raw chunk: 4 quads x 4 KiB: 10 rows x 20 slots of 16 B (18 padded cols, de-interleaved)
constexpr int kWV = 32 * 1024;        // V chunk: 16 xi x [quad 4][tile 32][4 ci]
constexpr int kWinoLds = 2 * kWU + 2 * kWRaw + 2 * kWV;  // 160 KiB

// a - b on 4 floats as two v_pk_add_f32 with the negate modifier (hipcc emits four v_sub_f32 otherwise; every VALU
// cycle here is taken from the fp32 MFMAs of the consumer wave on the same SIMD)
__device__ __forceinline__ f32x4 pk_sub(f32x4 a, f32x4 b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 lo, hi;
  const f32x2 alo = {a.x, a.y}, ahi = {a.z, a.w}, blo = {b.x, b.y}, bhi = {b.z, b.w};
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(alo), "v"(blo));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(ahi), "v"(bhi));
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}

template <int NCHUNK, bool DBG>
__global__ __launch_bounds__(512, 1) void conv3x3_wino_kernel(const float* __restrict__ p_src, const float* __restrict__ p_u,
                                                              int p_qin, int p_qout, const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ub = smem;
  char* const Rb = smem + 2 * kWU;
  char* const Vb = smem + 2 * kWU + 2 * kWRaw;
  (void)p_qout;
  Stamps st(a, (DBG && (a.debug & 16)) ? 256 : (DBG ? 0 : -1));  // debug 16: stamps from a producer wave instead of a consumer
  st.take(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroups are dealt round-robin over the 8 XCDs in dispatch order: remap so that every XCD gets a CONTIGUOUS range of
  // logical ids, i.e. the four workgroups of a sample (two co tiles x two image halves, which read the same input) share an L2
  const int nwg = gridDim.x * gridDim.y;
  int lid = blockIdx.x + gridDim.x * blockIdx.y;
  if ((nwg & 7) == 0) lid = (lid & 7) * (nwg >> 3) + (lid >> 3);
  const int rh = lid & 1, ct = (lid >> 1) % (gridDim.x >> 1), b = (lid >> 1) / (gridDim.x >> 1);
  const int r0 = rh * 8;
  constexpr int nchunk = NCHUNK;
  const bool skip = a.skip && *a.skip;  // adaptive solver finished while this launch was queued (uniform)
  const bool dbg_noprod = DBG && (a.debug & 1), dbg_nomfma = DBG && (a.debug & 2), dbg_notr = DBG && (a.debug & 32),
             dbg_nodma = DBG && (a.debug & 128);  // diagnostic ablations (tools/conv_microbench.py)

  if (wave >= 4) {
    // =========================================== PRODUCERS ===========================================
    const int pw = wave - 4, ptid = tid - 256;
    const unsigned u_tile_bytes = (unsigned)nchunk * kWU;
    const __amdgpu_buffer_rsrc_t ru = make_rsrc((const char*)p_u + (size_t)ct * u_tile_bytes, u_tile_bytes);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc((const char*)p_src + (size_t)b * p_qin * kQuadBytes, (unsigned)p_qin * kQuadBytes);
    // Raw tile in LDS: row r (image row r0 - 1 + r) = 20 slots of 16 B: [even padded cols 0,2,..,16 | odd padded cols
    // 1,3,..,17 | 2 unused]; padded col pc = image col + 1.  A 4x4 patch reads cols 2tx + j: for a fixed j the 8 tiles of a
    // row hit CONSECUTIVE slots, and the 20-slot row stride puts the 4 tile rows of a ds_read_b128 lane group in
    // different bank quarters: conflict-free (a plain [row][18 cols] layout is 4-way conflicted on these stride-2 reads).
    int vr[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int s = 64 * p + lane;
      const int row = s / 20, w = s - row * 20;
      const int pc = w < 9 ? 2 * w : 2 * (w - 9) + 1;
      const int irow = r0 - 1 + row, col = pc - 1;
      vr[p] = (s < 200 && w < 18 && irow >= 0 && irow < kHW && col >= 0 && col < kHW) ? irow * 256 + col * 16 : kOobOffset;
    }
    const int vw = lane * 16;
    auto issue_u = [&](int c, int buf) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int p = pw * 8 + g;
        dma16(ru, Ub + buf * kWU + p * 1024, vw, (c * 32 + p) * 1024);
      }
    };
    auto issue_raw = [&](int c, int buf) {
#pragma unroll
      for (int p = 0; p < 4; ++p) dma16(rx, Rb + buf * kWRaw + (pw * 4 + p) * 1024, vr[p], (c * 4 + pw) * kQuadBytes);
    };
    // input transform task: (half th, quad tq, tile tt) -> V rows 2*th, 2*th+1 of B^T d B.  Each producer wave
    // transforms the quad it DMA'd itself (tq = pw), so raw data needs no cross-wave hand-off: its own vmcnt suffices.
    const int th = lane >> 5, tq = pw, tt = lane & 31;
    (void)ptid;
    const int tty = tt >> 3, ttx = tt & 7;
    // Patch rows p0..p3 (LDS rows 2ty..2ty+3).  B^T d:  T0 = p0 - p2, T1 = p1 + p2, T2 = p2 - p1, T3' = p3 - p1 (= -T3; the
    // sign is folded into the packed U rows of xi = 12..15).  Half 0 loads (p2, p0, p1), half 1 loads (p1, p2, p3): both
    // halves then compute  Ta = d1 - d0,  Tb = d2 + sgn*d0  with sgn = +1 / -1 -- no per-lane selects in the VALU stream.
    const int r_a = th == 0 ? 2 : 1, r_b = th == 0 ? 0 : 2, r_c = th == 0 ? 1 : 3;
    const int raw_base = tq * 4096 + (2 * tty * 20 + ttx) * 16;  // + (j&1)*9*16 + (j>>1)*16 per patch column j
    const int off_a = raw_base + r_a * 320, off_b = raw_base + r_b * 320, off_c = raw_base + r_c * 320;
    const float sgn = th == 0 ? 1.0f : -1.0f;
    const int v_off = tq * 512 + tt * 16;                                  // + xi * 2048
    auto transform = [&](int rbuf, int vbuf) {
      const char* r = Rb + rbuf * kWRaw;
      f32x4 d0[4], d1[4], d2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (DBG && (a.debug & 512)) {
          d0[j] = d1[j] = d2[j] = f32x4{1.f, 2.f, 3.f, (float)j};
        } else {
          const int cj = ((j & 1) * 9 + (j >> 1)) * 16;
          d0[j] = *(const f32x4*)(r + off_a + cj);
          d1[j] = *(const f32x4*)(r + off_b + cj);
          d2[j] = *(const f32x4*)(r + off_c + cj);
        }
      }
      f32x4 T[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        T[0][j] = pk_sub(d1[j], d0[j]);
        T[1][j] = d2[j] + d0[j] * sgn;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        char* v = Vb + vbuf * kWV + v_off + (2 * th + i) * 4 * 2048;
        if (DBG && (a.debug & 256)) {
          const f32x4 sum = (T[i][0] - T[i][2]) + (T[i][1] + T[i][2]) + (T[i][2] - T[i][1]) + (T[i][1] - T[i][3]);
          asm volatile("" ::"v"(sum.x), "v"(sum.y), "v"(sum.z), "v"(sum.w));
        } else {
          *(f32x4*)(v + 0 * 2048) = pk_sub(T[i][0], T[i][2]);
          *(f32x4*)(v + 1 * 2048) = T[i][1] + T[i][2];
          *(f32x4*)(v + 2 * 2048) = pk_sub(T[i][2], T[i][1]);
          *(f32x4*)(v + 3 * 2048) = pk_sub(T[i][1], T[i][3]);
        }
      }
    };

    // DMA issue order per wave: raw_0 (4) | U_0 (8) | raw_1 (4) | then per iteration c: U_{c+1} (8) | raw_{c+2} (4).
    // Counted waits (vmcnt counts this wave's DMAs in issue order) leave the younger ones in flight.
    if (!skip && !dbg_noprod) {
      issue_raw(0, 0);
      issue_u(0, 0);
      if (nchunk > 1) issue_raw(1, 1);
      if (nchunk > 1) wait_vmcnt<12>(); else wait_vmcnt<8>();  // raw_0 landed (U_0, raw_1 still in flight)
      transform(0, 0);
      if (nchunk > 1) wait_vmcnt<4>(); else wait_vmcnt<0>();   // U_0 landed
    }
    st.take(2);
#pragma unroll
    for (int c = 0; c < nchunk; ++c) {
      __builtin_amdgcn_s_barrier();  // [c] V_c and U_c ready for the consumers; they are done with chunk c-1
      if (!skip && !dbg_noprod && c + 1 < nchunk) {
        if (!dbg_nodma) issue_u(c + 1, (c + 1) & 1);      // U buffer last read by the MFMAs of chunk c-1
        if (c + 2 < nchunk) {
          if (!dbg_nodma) issue_raw(c + 2, c & 1);        // raw buffer consumed by this wave's transform of chunk c
          wait_vmcnt<12>();                               // raw_{c+1} landed
        } else {
          wait_vmcnt<8>();
        }
        if (!dbg_notr) transform((c + 1) & 1, (c + 1) & 1);  // V buffer last read by the MFMAs of chunk c-1
        if (c + 2 < nchunk) wait_vmcnt<4>(); else wait_vmcnt<0>();  // U_{c+1} landed
      }
      if (c < 4) st.take(3 + c);
    }
    st.flush(a);
    return;
  }

  // ============================================= CONSUMERS =============================================
  const int ch = wave >> 1, thh = wave & 1;
  const int i16 = lane & 15, kq = lane >> 4;
  f32x4 acc[16];
#pragma unroll
  for (int x = 0; x < 16; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int u_off = kq * 512 + (ch * 16 + i16) * 16;
  const int v_off = kq * 512 + (thh * 16 + i16) * 16;
  const int Q = ct * 8 + ch * 4 + kq;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bias4 = *(const f32x4*)(a.bias + Q * 4);  // loaded now, used after the last MFMA
  st.take(2);
#pragma unroll
  for (int c = 0; c < nchunk; ++c) {
    __builtin_amdgcn_s_barrier();  // [c]
    if (!skip && !dbg_nomfma) {
      const char* u = Ub + (c & 1) * kWU + u_off;
      const char* v = Vb + (c & 1) * kWV + v_off;
      // two xi at a time: consecutive MFMAs hit different accumulators (a dependent 16x16x4 issues 8 cycles late);
      // the fragments of the next pair are read while this pair's 8 MFMAs run
      f32x4 w0 = *(const f32x4*)(u), x0 = *(const f32x4*)(v), w1 = *(const f32x4*)(u + 2048), x1 = *(const f32x4*)(v + 2048);
      f32x4 w0n, x0n, w1n, x1n;
#pragma unroll
      for (int x = 0; x < 16; x += 2) {
        if (x + 2 < 16) {
          w0n = *(const f32x4*)(u + (x + 2) * 2048);
          x0n = *(const f32x4*)(v + (x + 2) * 2048);
          w1n = *(const f32x4*)(u + (x + 3) * 2048);
          x1n = *(const f32x4*)(v + (x + 3) * 2048);
        }
        acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, x0.x, acc[x], 0, 0, 0);
        acc[x + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, x1.x, acc[x + 1], 0, 0, 0);
        acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, x0.y, acc[x], 0, 0, 0);
        acc[x + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, x1.y, acc[x + 1], 0, 0, 0);
        acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, x0.z, acc[x], 0, 0, 0);
        acc[x + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, x1.z, acc[x + 1], 0, 0, 0);
        acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, x0.w, acc[x], 0, 0, 0);
        acc[x + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, x1.w, acc[x + 1], 0, 0, 0);
        w0 = w0n; x0 = x0n; w1 = w1n; x1 = x1n;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (st.on && c < 4) {
      asm volatile("" ::"v"(acc[15][0]));
      st.take(3 + c);
    }
  }
  if (skip) return;

  // ---- output transform in registers: lane (tile i16 of half thh, row group kq) holds M_xi[co quad][tile] for all xi
  f32x4 S[2][4];
#pragma unroll
  for (int cI = 0; cI < 4; ++cI) {
    S[0][cI] = acc[0 * 4 + cI] + acc[1 * 4 + cI] + acc[2 * 4 + cI];
    S[1][cI] = pk_sub(pk_sub(acc[1 * 4 + cI], acc[2 * 4 + cI]), acc[3 * 4 + cI]);
  }
  const int tile = thh * 16 + i16, oty = tile >> 3, otx = tile & 7;
  float esum = 0.0f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x4 y0 = S[i][0] + S[i][1] + S[i][2] + bias4;
    const f32x4 y1 = pk_sub(pk_sub(S[i][1] + bias4, S[i][2]), S[i][3]);
    const int P = (r0 + 2 * oty + i) * 16 + 2 * otx;
    emit_quad(a, b, Q, P, y0, esum);
    emit_quad(a, b, Q, P + 1, y1, esum);
  }
  finish_err(a, esum, wave);
  st.flush(a);
}

template <int NCHUNK>
static int launch_wino_n(const ConvArgs& a, hipStream_t stream) {
  static bool attr_set = false, attr_set_dbg = false;
  const dim3 grid((a.qout / 8) * 2, a.batch);
  if (a.debug) {
    if (!attr_set_dbg) {
      ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_wino_kernel<NCHUNK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set_dbg = true;
    }
    hipLaunchKernelGGL((conv3x3_wino_kernel<NCHUNK, true>), grid, dim3(512), kWinoLds, stream, a.src1, a.w_wino, a.qin, a.qout, a);
    ODEHIP_CHECK_HIP(hipGetLastError());
    return ODEHIP_OK;
  }
  if (!attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_wino_kernel<NCHUNK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_wino_kernel<NCHUNK, false>), grid, dim3(512), kWinoLds, stream, a.src1, a.w_wino, a.qin, a.qout, a);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

// returns 1 if the shape has no Winograd instantiation (the caller then runs the direct kernel)
int launch_wino(const ConvArgs& a, hipStream_t stream) {
  switch (a.qin / 4) {
    case 1: return launch_wino_n<1>(a, stream);
    case 2: return launch_wino_n<2>(a, stream);
    case 3: return launch_wino_n<3>(a, stream);
    case 4: return launch_wino_n<4>(a, stream);
    case 8: return launch_wino_n<8>(a, stream);
    default: return 1;
  }
}

}  // namespace odehip
