// conv_wino5.hip -- 5x5 convolution on 16x16 maps by Winograd F(2x2, 5x5) on exact-fp32 MFMA (gfx950): the two convolutions of
// the ConvGRU cell (/root/reference/modules/ConvGRUCell.py:40-50, :72-80; 128 -> 128 and 128 -> 64 channels in ODEConvGRU) and
// their input-gradient forms.  36 instead of 100 multiplies per 2x2 outputs:
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 6x6 input patch d, 5x5 filter g,
// interpolation points 0, +-1, +-2, inf (B^T is Lavin & Gray's F(4,3) matrix; G, A^T follow from the same points;
// tools/experiments/winograd_f25_error.py checks the identity in fp64 and prices the fp32 error: 2.6e-6 rel-L2 per layer
// against 3.4e-7 for the direct kernel).  The arithmetic is fp32 throughout.
//
// Workgroup = (sample, 32 output channels, 8 output rows = 4x8 tiles), 512 threads, role-specialised waves as in conv_wino.hip:
//   waves 0-3  CONSUMERS: wave w owns transform positions xi = 9w .. 9w+8 for the WHOLE (32 co x 32 tiles) tile: 36 accumulators
//              of v_mfma_f32_16x16x4_f32 (144 VGPRs).  Per 8-channel chunk 36 ds_read_b64 feed 72 MFMAs; splitting xi (not the
//              tile) across the waves halves the LDS bytes per MFMA -- the LDS, not the matrix core, is the first limit of this
//              kernel.  The output transform is linear, so every wave applies it to ITS nine positions and the four partial
//              2x2 outputs meet once per layer in LDS.
//   waves 4-7  PRODUCERS: LDS-DMA of the next chunk's transformed weights U (36 KiB) and of the raw input tile two chunks ahead
//              (2 quads x 12 rows x 20 columns, zero-padded by the DMA's range check), then V = B^T d B of the next chunk: four
//              threads per (tile, channel quad) -- (3 of the 6 V columns) x (2 of the 4 channels) -- 72 packed-fp32 VALU
//              instructions, 30 ds_read_b64 and 18 ds_write_b64 each.
// One s_barrier per chunk.  LDS: U[2] + V[2] 144 KiB + raw[2] 16 KiB = 160 KiB.
//
// SPLIT (round 4; small batches -- the reference trains at batch 4, configs.yaml:7): a launch is a serial chain over the K = cin / 8
// chunks (33 us for 128 input channels however small the batch: ~12 us fixed + 16 x 1.3 us) on (cout / 16) x batch workgroups -- 32 or 16
// of 256 CUs at batch 4.  With SPLIT, S workgroups share an output tile: each walks K / S chunks, writes its partial 2x2 outputs to a
// library-owned workspace and counts itself in (one counter per tile and consumer wave); the wave that arrives LAST adds the S
// partials in split order (a fixed order: deterministic, no float atomics), then bias and the usual epilogue.  The S workgroups have
// consecutive logical ids, i.e. one XCD, whose L2 is the coherence point of the hand-off (stores acknowledged, agent-scope counter,
// sc1 loads); every workgroup publishes the XCD it really runs on and a tile whose workgroups differ adds agent-scope fences.
#include <stddef.h>
#include <stdlib.h>

#include <mutex>

#include "conv_common.h"
#include "pack_elems.h"

namespace odehip {

constexpr int k5U = 36 * 1024;          // U chunk: 36 xi x [quad 2][co 32][4 ci]
constexpr int k5V = 36 * 1024;          // V chunk: 36 xi x [quad 2][tile 32][4 ci]
constexpr int k5RawQuad = 4096;        // one quad of the raw tile: rows r0-2 .. r0+9, 20 de-interleaved column slots of 16 B = 240 slots,
                                       // padded to the 256 slots four DMA instructions write
constexpr int k5Raw = 2 * k5RawQuad;
constexpr int k5Lds = 2 * (k5U + k5V) + 2 * k5Raw;  // 163,840 B = all of it
static_assert(k5Lds <= 160 * 1024, "LDS budget");

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 fma2(float c, f32x2 a, f32x2 b) { return f32x2{__builtin_fmaf(c, a.x, b.x), __builtin_fmaf(c, a.y, b.y)}; }

// one 6-point transform with B^T: out[i] = sum_k BT[i][k] in[k]
//   BT = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void bt6(const f32x2* in, f32x2* out) {
  out[0] = fma2(4.0f, in[0], fma2(-5.0f, in[2], in[4]));
  const f32x2 a = fma2(-4.0f, in[2], in[4]), b = fma2(-4.0f, in[1], in[3]);
  out[1] = a + b;
  out[2] = a - b;
  const f32x2 c = in[4] - in[2], s = in[3] - in[1];
  out[3] = fma2(2.0f, s, c);
  out[4] = fma2(-2.0f, s, c);
  out[5] = fma2(4.0f, in[1], fma2(-5.0f, in[3], in[5]));
}

// V = B^T d B for one (tile, channel pair), V columns 3 CH .. 3 CH + 2: first W = d B restricted to those columns (row by row; they
// only need patch columns 0..4 resp. 1..5), then B^T W column by column.  r: the lane's patch origin in the raw tile (rows of 20
// slots: even padded columns, then odd); v: the lane's slot of V[xi = 0] (xi = 6 i + j, 1 KiB apart).
template <int CH>
__device__ __forceinline__ void transform_half(const char* r, char* v) {
  f32x2 W[3][6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    f32x2 d[6];
#pragma unroll
    for (int l = CH; l < 5 + CH; ++l) d[l] = *(const f32x2*)(r + (k * 20 + ((l & 1) ? 10 : 0) + (l >> 1)) * 16);
    if (CH == 0) {
      W[0][k] = fma2(4.0f, d[0], fma2(-5.0f, d[2], d[4]));
      const f32x2 aa = fma2(-4.0f, d[2], d[4]), bb = fma2(-4.0f, d[1], d[3]);
      W[1][k] = aa + bb;
      W[2][k] = aa - bb;
    } else {
      const f32x2 cc = d[4] - d[2], ss = d[3] - d[1];
      W[0][k] = fma2(2.0f, ss, cc);
      W[1][k] = fma2(-2.0f, ss, cc);
      W[2][k] = fma2(4.0f, d[1], fma2(-5.0f, d[3], d[5]));
    }
  }
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    f32x2 o[6];
    bt6(W[jj], o);
#pragma unroll
    for (int i = 0; i < 6; ++i) *(f32x2*)(v + (6 * i + 3 * CH + jj) * 1024) = o[i];
  }
}

// A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 1]
__device__ __forceinline__ float at_coef(int row, int i) {
  if (row == 0) return i < 5 ? 1.0f : 0.0f;
  return i == 0 ? 0.0f : (i == 1 ? 1.0f : (i == 2 ? -1.0f : (i == 3 ? 2.0f : (i == 4 ? -2.0f : 1.0f))));
}

struct Split5 {
  int S;               // workgroups per output tile (1: no split)
  float* part;         // [tile][split][wave 4][e 4][lane 64] quads
  unsigned* count;     // [tile][wave 4]: splits that have delivered (reset by the last one)
  unsigned* xcc;       // [workgroup]: epoch << 4 | XCC_ID + 1
  unsigned epoch;
};

template <bool DBG, bool SPLIT>  // DBG: ablation bits in a.debug (tools/conv5_microbench.py): 256 no transform, 512 no DMA, 1024 no fragment reads, 2048 no MFMA
__global__ __launch_bounds__(512, 1) void conv5x5_wino_kernel(const ConvArgs a, const Split5 sp5) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // regions: [U0 | V0 | U1 | V1 | raw0 | raw1]: buffer b of U / V at b * (k5U + k5V) (+ k5U): the pair NOT used by the last chunk is
  // one contiguous 72 KiB block for the partial outputs
  char* const Rb = smem + 2 * (k5U + k5V);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x * gridDim.y;
  int lid = blockIdx.x + gridDim.x * blockIdx.y;
  if ((nwg & 7) == 0) lid = (lid & 7) * (nwg >> 3) + (lid >> 3);  // XCD-contiguous logical ids (conv_wino.hip)
  const int S = SPLIT ? sp5.S : 1;
  const int sp = SPLIT ? lid % S : 0;            // the S workgroups of an output tile have consecutive logical ids
  const int tile_id = SPLIT ? lid / S : lid;     // (b, ct, rh)
  const int ntile = (gridDim.x / S) >> 1;
  const int rh = tile_id & 1, ct = (tile_id >> 1) % ntile, b = (tile_id >> 1) / ntile;
  const int r0 = rh * 8;
  const int nchunk = (a.qin >> 1) / S;           // this workgroup's share of the chunks: [c0, c0 + nchunk)
  const int c0 = sp * nchunk;
  bool fence = false;
  if (SPLIT) {
    // which XCD do the tile's workgroups really run on?  (dispatch order puts consecutive logical ids on one; not a guarantee)
    const unsigned my_xcc = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) + 1u;
    const unsigned tag = sp5.epoch << 4;
    if (tid == 0) __hip_atomic_store(sp5.xcc + lid, tag | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave < 4) {   // the consumers have nothing to do until the first chunk is in LDS
      for (int p = 0; p < S; ++p) {
        unsigned v = 0;
        int n = 0;
        while (((v = __hip_atomic_load(sp5.xcc + tile_id * S + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & ~15u) != tag) {
          __builtin_amdgcn_s_sleep(2);
          if (++n > (1 << 22)) { v = 0; break; }   // a partner that never started (the grid did not fit): be safe, fence
        }
        fence |= ((v & 15u) != my_xcc);
      }
      fence = __builtin_amdgcn_readfirstlane(fence);
    }
  }

  if (wave >= 4) {
    // =========================================== PRODUCERS ===========================================
    const int pw = wave - 4;
    const int quad = pw & 1, colhalf = pw >> 1;  // this wave transforms quad `quad` of every chunk, V columns 3 colhalf .. +2
    const unsigned u_tile_bytes = (unsigned)(nchunk * S) * k5U;
    const __amdgpu_buffer_rsrc_t ru = make_rsrc((const char*)a.w_wino + (size_t)ct * u_tile_bytes, u_tile_bytes);
    const int q2 = a.qin - a.q1;
    const __amdgpu_buffer_rsrc_t rx1 = make_rsrc((const char*)a.src1 + (size_t)b * a.q1 * kQuadBytes, (unsigned)a.q1 * kQuadBytes);
    const __amdgpu_buffer_rsrc_t rx2 =
        q2 > 0 ? make_rsrc((const char*)a.src2 + (size_t)b * q2 * kQuadBytes, (unsigned)q2 * kQuadBytes) : rx1;
    // raw tile of one quad: row r (image row r0 - 2 + r), 20 slots of 16 B: [even padded columns 0,2,..,18 | odd 1,3,..,19], padded
    // column pc = image column + 2: for a fixed patch column the 8 tiles of a tile row read consecutive slots
    int vr[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int s = 64 * p + lane;
      const int row = s / 20, w = s - row * 20;
      const int pc = w < 10 ? 2 * w : 2 * (w - 10) + 1;
      const int irow = r0 - 2 + row, col = pc - 2;
      vr[p] = (s < 240 && irow >= 0 && irow < kHW && col >= 0 && col < kHW) ? irow * 256 + col * 16 : kOobOffset;
    }
    const int vw = lane * 16;
    auto issue_u = [&](int c, int buf) {
#pragma unroll
      for (int g = 0; g < 9; ++g) {
        const int p = pw * 9 + g;
        dma16(ru, smem + buf * (k5U + k5V) + p * 1024, vw, ((c0 + c) * 36 + p) * 1024);
      }
    };
    // BOTH waves that transform a quad load it (identical bytes to identical addresses): each then only has to wait for its own
    // DMAs, and the raw tile is a fifth of the chunk's traffic
    auto issue_raw = [&](int c, int buf) {
      const int q = 2 * (c0 + c) + quad;
      const bool second = q >= a.q1;
      const int soff = (second ? q - a.q1 : q) * kQuadBytes;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        char* const dst = Rb + buf * k5Raw + quad * k5RawQuad + p * 1024;
        if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, ODEHIP_LDS_PTR(dst), 16, vr[p], soff, 0, 0);
        else        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx1, ODEHIP_LDS_PTR(dst), 16, vr[p], soff, 0, 0);
      }
    };
    const int cp = lane & 1, tile = lane >> 1, ty = tile >> 3, tx = tile & 7;
    // patch element (k, l): row 2 ty + k, slot (l odd ? 10 : 0) + tx + l / 2
    const int raw_off = quad * k5RawQuad + (2 * ty * 20 + tx) * 16 + cp * 8;
    const int v_off = quad * 512 + tile * 16 + cp * 8;
    auto transform = [&](int rbuf, int vbuf) {
      const char* r = Rb + rbuf * k5Raw + raw_off;
      char* v = smem + vbuf * (k5U + k5V) + k5U + v_off;
      if (colhalf == 0) transform_half<0>(r, v); else transform_half<1>(r, v);  // wave-uniform
    };

    // DMA issue order per wave: raw_0 (4) | U_0 (9) | raw_1 (4) | then per iteration c: U_{c+1} (9) | raw_{c+2} (4)
    const bool no_tr = DBG && (a.debug & 256), no_dma = DBG && (a.debug & 512);
    if (!no_dma) {
      issue_raw(0, 0);
      issue_u(0, 0);
      if (nchunk > 1) issue_raw(1, 1);
    }
    if (nchunk > 1) wait_vmcnt<13>(); else wait_vmcnt<9>();  // raw_0 landed
    if (!no_tr) transform(0, 0);
    if (nchunk > 1) wait_vmcnt<4>(); else wait_vmcnt<0>();   // U_0 landed
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
      __builtin_amdgcn_s_barrier();  // [c] V_c and U_c ready for the consumers; they are done with chunk c-1
      if (c + 1 < nchunk) {
        if (!no_dma) issue_u(c + 1, (c + 1) & 1);      // U buffer last read by the MFMAs of chunk c-1
        if (c + 2 < nchunk) {
          if (!no_dma) issue_raw(c + 2, c & 1);        // raw buffer consumed by the transforms of chunk c (all of them before barrier [c])
          wait_vmcnt<13>();               // raw_{c+1} landed
        } else {
          wait_vmcnt<9>();
        }
        if (!no_tr) transform((c + 1) & 1, (c + 1) & 1);  // V buffer last read by the MFMAs of chunk c-1
        if (c + 2 < nchunk) wait_vmcnt<4>(); else wait_vmcnt<0>();  // U_{c+1} landed
      }
    }
    __builtin_amdgcn_s_barrier();  // [end] the consumers' partial outputs are in LDS (nothing for the producers to do)
    return;
  }

  // ============================================= CONSUMERS =============================================
  const int i16 = lane & 15, kq = lane >> 4;
  f32x4 acc[9][2][2];
#pragma unroll
  for (int n = 0; n < 9; ++n)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) acc[n][cb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frag_off = wave * 9 * 1024 + (kq >> 1) * 512 + (kq & 1) * 8 + i16 * 16;
#pragma unroll 1
  for (int c = 0; c < nchunk; ++c) {
    __builtin_amdgcn_s_barrier();  // [c]
    const char* u = smem + (c & 1) * (k5U + k5V) + frag_off;
    const char* v = u + k5U;
    if (DBG && (a.debug & 2048)) continue;
    // (measured and dropped, same microbenchmark: holding the last positions' second k-step back until after the next barrier to
    // cover the latency of the next chunk's first fragments: 74.7 vs 73.5 us per launch; forcing both k-steps of a position to issue
    // together so that fragments are consumed at the rate the LDS delivers them: 76.0 vs 73.5 us)
#pragma unroll
    for (int n = 0; n < 9; ++n) {
      f32x2 u0, u1, v0, v1;
      if (DBG && (a.debug & 1024)) {
        u0 = u1 = f32x2{1.0f, (float)n};
        v0 = v1 = f32x2{(float)c, 2.0f};
      } else {
        u0 = *(const f32x2*)(u + n * 1024); u1 = *(const f32x2*)(u + n * 1024 + 256);
        v0 = *(const f32x2*)(v + n * 1024); v1 = *(const f32x2*)(v + n * 1024 + 256);
      }
      acc[n][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.x, v0.x, acc[n][0][0], 0, 0, 0);
      acc[n][0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.x, v1.x, acc[n][0][1], 0, 0, 0);
      acc[n][1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.x, v0.x, acc[n][1][0], 0, 0, 0);
      acc[n][1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.x, v1.x, acc[n][1][1], 0, 0, 0);
      acc[n][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.y, v0.y, acc[n][0][0], 0, 0, 0);
      acc[n][0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.y, v1.y, acc[n][0][1], 0, 0, 0);
      acc[n][1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.y, v0.y, acc[n][1][0], 0, 0, 0);
      acc[n][1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.y, v1.y, acc[n][1][1], 0, 0, 0);
    }
  }
  // ---- this wave's share of Y = A^T M A: Y[ra][rb] += AT[ra][i] AT[rb][j] M[i][j] over its nine (i, j)
  f32x4 part[2][2][2][2];  // [cb][tb][ra][rb]
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int e = 0; e < 4; ++e) part[cb][tb][e >> 1][e & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < 9; ++n) {
    const int xi = wave * 9 + n, i = xi / 6, j = xi - 6 * i;  // wave-uniform
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float cf = at_coef(e >> 1, i) * at_coef(e & 1, j);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) part[cb][tb][e >> 1][e & 1] += acc[n][cb][tb] * cf;
    }
  }
  // the buffer pair the last chunk did not use: free since barrier [nchunk - 1]
  char* const px = smem + (nchunk & 1) * (k5U + k5V);
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *(f32x4*)(px + ((wave * 4 + cb * 2 + tb) * 4 + e) * 1024 + lane * 16) = part[cb][tb][e >> 1][e & 1];
  __builtin_amdgcn_s_barrier();  // [end]
  // wave w finishes block (cb, tb) = (w >> 1, w & 1): the four partials in wave order, bias, store
  const int cb = wave >> 1, tb = wave & 1;
  const int Q = ct * 8 + cb * 4 + kq;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bias4 = *(const f32x4*)(a.bias + Q * 4);
  const int tile = tb * 16 + i16, oty = tile >> 3, otx = tile & 7;
  f32x4 yv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f32x4 y = *(const f32x4*)(px + ((0 * 4 + wave) * 4 + e) * 1024 + lane * 16);
#pragma unroll
    for (int w = 1; w < 4; ++w) y += *(const f32x4*)(px + ((w * 4 + wave) * 4 + e) * 1024 + lane * 16);
    yv[e] = y;
  }
  if (SPLIT) {
    // deliver this split's partial, count in; the wave that completes the tile adds the S partials in split order
    float* const mine = sp5.part + (((size_t)tile_id * S + sp) * 4 + wave) * (4 * 64 * 4) + lane * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) *(f32x4*)(mine + e * 256) = yv[e];
    wait_vmcnt<0>();
    if (fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    unsigned before = 0;
    if (lane == 0) before = __hip_atomic_fetch_add(sp5.count + tile_id * 4 + wave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    before = __builtin_amdgcn_readfirstlane(before);
    if (before != (unsigned)(S - 1)) return;
    if (lane == 0) __hip_atomic_store(sp5.count + tile_id * 4 + wave, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const __amdgpu_buffer_rsrc_t rp = make_rsrc((const char*)(sp5.part + (size_t)tile_id * S * 4 * (4 * 64 * 4)), (unsigned)S * 4 * 4 * 64 * 16);
    f32x4 tot[4];
#pragma unroll 1
    for (int k = 0; k < S; ++k) {
      f32x4 pk[4];
      if (k == sp) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = yv[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)   // sc1: never from this CU's vector cache (the bytes were written by another CU a moment ago)
          pk[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, lane * 16, ((k * 4 + wave) * 4 + e) * 1024, 16));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) tot[e] = k == 0 ? pk[e] : tot[e] + pk[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) yv[e] = tot[e];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x4 y = yv[e] + bias4;
    const int P = (r0 + 2 * oty + (e >> 1)) * 16 + 2 * otx + (e & 1);
    float esum = 0.0f;
    emit_quad(a, b, Q, P, y, esum);  // plain / ReLU store, or the backward sweep's epilogues (conv_common.h)
  }
}

// library-owned scratch of the split launches (partials, counters, XCD words), grown on demand
static struct Split5State {
  float* part = nullptr;
  size_t part_floats = 0;
  unsigned* words = nullptr;   // [count: kTiles x 4 | xcc: kWgs]
  unsigned epoch = 0;
  int device = -1;             // the device the scratch lives on (one process drives one GPU: a call from another device is refused)
  hipStream_t last_stream = nullptr;   // the stream of the previous split launch (compared, never dereferenced)
  bool used = false;
} g_split5;
constexpr int kSplit5Tiles = 256, kSplit5Wgs = 256;

// returns 1 if the layer has no F(2x2,5x5) form here (the caller then runs the direct kernel)
int launch_wino5(const ConvArgs& a, hipStream_t stream) {
  if (a.combine == 1 || a.skip || (a.q1 & 1) || (a.qin & 1)) return 1;  // (no Runge-Kutta stage combine on a 5x5 layer)
  static bool attr_set = false;
  if (!attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)conv5x5_wino_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)conv5x5_wino_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)conv5x5_wino_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  Split5 sp5 = {1, nullptr, nullptr, nullptr, 0u};
  const int nwg1 = (a.qout / 8) * 2 * a.batch, nchunk = a.qin >> 1;
  // split the chunk chain while ALL workgroups still fit the chip at once (they wait for each other's XCD word: every one of them
  // must be resident) and a share is at least two chunks; ODEHIP_WINO5_SPLIT=0 switches it off (A/B)
  static const int split_max = [] { const char* e = getenv("ODEHIP_WINO5_SPLIT"); return e ? atoi(e) : 8; }();
  int S = 1;
  for (int k = 8; k >= 2; k >>= 1)
    if (k <= split_max && nwg1 * k <= kSplit5Wgs && nchunk % k == 0 && nchunk / k >= 2 && !(a.debug & (256 | 512 | 1024 | 2048))) { S = k; break; }
  if (S > 1) {   // a captured launch would replay with this launch's epoch and scratch: graphs get the unsplit kernel (no shared state)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) S = 1;
  }
  if (S > 1) {
    static std::mutex split_mu;   // the scratch, its epoch and its (re)allocation: one caller at a time (autograd runs backward passes
    std::lock_guard<std::mutex> lk(split_mu);   // on its own thread); launches that share it are ordered by the stream they go to
    Split5State& G = g_split5;
    const size_t need = (size_t)nwg1 * S * 4 * 4 * 64 * 4;
    int cur_dev = -1;
    ODEHIP_CHECK_HIP(hipGetDevice(&cur_dev));
    if (!G.words) {
      ODEHIP_CHECK_HIP(hipMalloc((void**)&G.words, (kSplit5Tiles * 4 + kSplit5Wgs) * sizeof(unsigned)));
      ODEHIP_CHECK_HIP(hipMemset(G.words, 0, (kSplit5Tiles * 4 + kSplit5Wgs) * sizeof(unsigned)));
      G.device = cur_dev;
    }
    ODEHIP_REQUIRE(cur_dev == G.device, "conv5x5: the split launches' scratch belongs to device %d, this call runs on device %d (one process drives one GPU)",
                   G.device, cur_dev);
    // the counters and partials are shared by consecutive launches: launches on ONE stream are ordered by it; a caller that changes
    // streams waits for the device first (rare: the whole library is driven from one stream)
    if (G.used && G.last_stream != stream) ODEHIP_CHECK_HIP(hipDeviceSynchronize());
    if (G.part_floats < need) {   // (rare, synchronous: a launch in flight may still use the old buffer)
      ODEHIP_CHECK_HIP(hipDeviceSynchronize());
      if (G.part) (void)hipFree(G.part);
      G.part = nullptr;
      G.part_floats = 0;
      ODEHIP_CHECK_HIP(hipMalloc((void**)&G.part, need * sizeof(float)));
      G.part_floats = need;
    }
    G.epoch = (G.epoch + 1) & 0x0fffffffu;
    if (G.epoch == 0) G.epoch = 1;
    sp5 = Split5{S, G.part, G.words, G.words + kSplit5Tiles * 4, G.epoch};
    const dim3 grid((a.qout / 8) * 2 * S, a.batch);
    hipLaunchKernelGGL((conv5x5_wino_kernel<false, true>), grid, dim3(512), k5Lds, stream, a, sp5);
    ODEHIP_CHECK_HIP(hipGetLastError());
    G.last_stream = stream;
    G.used = true;
    return ODEHIP_OK;
  }
  const dim3 grid((a.qout / 8) * 2, a.batch);
  if (a.debug & (256 | 512 | 1024 | 2048)) hipLaunchKernelGGL((conv5x5_wino_kernel<true, false>), grid, dim3(512), k5Lds, stream, a, sp5);
  else hipLaunchKernelGGL((conv5x5_wino_kernel<false, false>), grid, dim3(512), k5Lds, stream, a, sp5);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

// U[ct][chunk][xi 36][quad 2][co 32][4] = (G g G^T)[xi / 6][xi % 6] of filter (co = 32 ct + i, ci = 8 chunk + 4 quad + s);
// transpose_flip: the input-gradient convolution's filter g'[ci][co][ky][kx] = g[co][ci][4 - ky][4 - kx]
__global__ __launch_bounds__(256) void pack_winograd5_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin,
                                                             int transpose_flip, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < total) pack_winograd5_elem(w, out, cout, cin, transpose_flip, idx);  // pack_elems.h
}

}  // namespace odehip

using namespace odehip;

extern "C" size_t odehip_winograd5_weight_floats(int cout, int cin) { return (size_t)cout * cin * 36; }

extern "C" int odehip_pack_conv_weight_winograd5(const float* w_oihw, float* w_wino, int cout, int cin, int transpose_flip, void* stream) {
  ODEHIP_REQUIRE(w_oihw && w_wino, "pack_conv_weight_winograd5: null pointer");
  ODEHIP_REQUIRE(cout > 0 && cout % 32 == 0, "pack_conv_weight_winograd5: cout must be a multiple of 32 (got %d)", cout);
  ODEHIP_REQUIRE(cin > 0 && cin % 8 == 0, "pack_conv_weight_winograd5: cin must be a multiple of 8 (got %d)", cin);
  const int total = cout * cin * 36;
  hipLaunchKernelGGL(pack_winograd5_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, w_wino, cout, cin,
                     transpose_flip, total);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}
