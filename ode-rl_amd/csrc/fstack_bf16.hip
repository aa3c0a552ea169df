// fstack_bf16.hip -- a whole evaluation of the dynamics f (every conv3x3 + ReLU of create_convnet, helpers/utils.py:158-183,
// plus the Runge-Kutta stage combine of the last layer) in ONE launch, bf16 operands / fp32 accumulation (BASELINE.json
// configs[4]).  One workgroup = one sample: with bf16 on the matrix cores a 64-channel layer of one 16x16 map is 2.2 us of MFMA
// on one CU, so the five layers of f need no other workgroup -- and therefore no launch boundary and no trip through HBM between
// layers: the hidden activations live in LDS as bf16 (one [18][18][64] tile with a zero border, rewritten in place after a
// barrier), the weights of all layers stream through a 3-stage LDS ring, one kernel row (3 taps, 24 KiB) per stage, counted
// s_waitcnt + one raw barrier per row (four barriers per layer).
//   wave w (8 waves, two per SIMD) owns one 32-channel half (w & 1) of four image rows (4 (w >> 1) .. +3 = two 32-pixel MFMA
//   blocks): 8 MFMAs (v_mfma_f32_32x32x16_bf16) per tap, 360 per wave per 5-layer f.  The loop is LDS-read bound, not MFMA
//   bound (0.96 us of MFMA per layer), so the tile is chosen for operand reuse: a weight fragment feeds both pixel blocks, and
//   the activation fragments are read once per KERNEL ROW -- the dx = -1 / +1 taps are DPP row shifts of the centre fragment
//   (an MFMA B-operand row of 16 lanes is 16 pixels of an image row; the lane shifted in from outside reads 0 = the padding):
//   20 ds_read_b128 per wave per kernel row instead of 36.  Biases are staged in LDS once; the saved mask of a gradient chain
//   is prefetched a layer ahead and the hidden-layer stores stay in flight, both counted in the ring's vmcnt waits.
// The same kernel runs the input-gradient chain of the backward passes (transposed+flipped weights in execution order, the
// ReLU replaced by the saved mask, every layer's fp32 gradient stored for the weight-gradient kernels) and can store the
// hidden activations (fp32, for a later backward).  The last layer ends in the shared fused epilogue (conv_common.h).
// Used when every layer is 64 -> 64 (the ODEConvGRU dynamics) and a fused weight image is present; batches below 256 leave
// CUs idle -- at B = 64 one f evaluation still takes ~1/3 of five bf16 launches.
#include "conv_common.h"

namespace odehip {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

constexpr int kFS = 144;                   // bytes per pixel of an activation tile (64 ch bf16 + 16 pad)
constexpr int kFTile = 18 * 18 * kFS;      // 46,656 B
constexpr int kFUnit = 3 * 8192;           // one kernel row (3 taps) of one layer
constexpr int kFStages = 3;
constexpr int kFBias = ODEHIP_MAX_LAYERS * 64 * 4;  // every layer's bias, staged once (a global load per layer would sit in front of the ring's vmcnt waits)
constexpr int kFusedLds = kFTile + kFStages * kFUnit + kFBias;

// issued without the compiler's own s_waitcnt bookkeeping: the ring's counted waits cover it (see wait_younger)
__device__ __forceinline__ f32x4 gload_untracked(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int N>
__device__ __forceinline__ void wait_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// n (wave-uniform) = vector-memory operations this wave issued AFTER the ring unit it is about to read: 3 (the next unit's
// DMAs) + 8 per hidden-layer store set + 8 per prefetched mask set
__device__ __forceinline__ void wait_younger(int n) {
  if (n <= 3) wait_le<3>();
  else if (n <= 11) wait_le<11>();
  else wait_le<19>();
}

// DBG: diagnostic instantiation that honours the ablation flags in fa.last.debug (1 no weight DMA, 2 no MFMA and no operand
// reads, 64 MFMA on constant operands); the production instantiation has no such branches in its inner loop.
template <bool DBG>
__global__ __launch_bounds__(512, 1) void fstack_bf16_kernel(const FusedArgs fa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const act = smem;
  char* const ring = smem + kFTile;
  float* const bias_l = (float*)(smem + kFTile + kFStages * kFUnit);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x;
  if (fa.last.skip && *fa.last.skip) return;
  const int NL = fa.n_layers, U = NL * 3;
  const int dbg = DBG ? fa.last.debug : 0;

  const __amdgpu_buffer_rsrc_t rw = make_rsrc(fa.w_fused, (unsigned)(U * kFUnit));
  const int vw = lane * 16;
  auto issue = [&](int u, int stage) {  // three 1-KiB pieces per wave
    if (DBG && (dbg & 1)) return;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      dma16(rw, ring + stage * kFUnit + (wave * 3 + j) * 1024, vw, u * kFUnit + (wave * 3 + j) * 1024);
  };
  issue(0, 0);
  if (U > 1) issue(1, 1);

  for (int i = threadIdx.x; i < NL * 64; i += 512) bias_l[i] = fa.bias[i >> 6] ? fa.bias[i >> 6][i & 63] : 0.0f;
  // zero border of the tile (68 pixels x 144 B), then the input: fp32 quads -> bf16
  for (int i = threadIdx.x; i < 68 * 9; i += 512) {
    const int p = i / 9, c16 = i % 9;
    int row, col;
    if (p < 18) { row = 0; col = p; }
    else if (p < 36) { row = 17; col = p - 18; }
    else if (p < 52) { row = p - 36 + 1; col = 0; }
    else { row = p - 52 + 1; col = 17; }
    *(f32x4*)(act + (row * 18 + col) * kFS + c16 * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  {
    const f32x4* src = (const f32x4*)(fa.x + (size_t)b * 64 * kPix);
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = src[i * 512 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = i * 512 + (int)threadIdx.x, p = idx & 255, q = idx >> 8;
      *(u32x2*)(act + (((p >> 4) + 1) * 18 + (p & 15) + 1) * kFS + q * 8) = u32x2{pk_bf16(v[i].x, v[i].y), pk_bf16(v[i].z, v[i].w)};
    }
  }

  const int i32 = lane & 31, kq = lane >> 5;
  const int px = i32 & 15, pyl = i32 >> 4;
  const int mb = wave & 1, row0 = (wave >> 1) * 4 + pyl;          // 32-channel half; image row of this lane in pixel block 0
  const int P0 = row0 * 16 + px, P1 = P0 + 32;                    // pixel of this lane in the two pixel blocks (rows +0, +2)
  const char* const in = act + ((row0 + 1) * 18 + px + 1) * kFS + kq * 16;
  // The saved ReLU mask of a gradient chain is PREFETCHED at the start of its layer (32 VGPRs) and the hidden-layer stores are
  // left in flight: both are counted in the ring's vmcnt waits instead of draining the ring at every layer boundary.
  f32x4 mreg[8];
  auto load_mask = [&](int e) {
    const float* mk = fa.mask[e];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int Q = mb * 8 + 2 * (i & 3) + kq;
      mreg[i] = gload_untracked(mk + (((size_t)b * 16 + Q) * kPix + ((i >> 2) ? P1 : P0)) * 4);
    }
  };
  f32x16 acc0, acc1;
  for (int e = 0; e < NL; ++e) {
    const bool has_mask = e < NL - 1 && fa.mask[e];
    if (has_mask) load_mask(e);
    // operations younger than this layer's first two units besides the next unit's DMAs (layer 0 drains at its first unit)
    const int extra = e == 0 ? 0 : (fa.store[e - 1] ? 8 : 0) + (has_mask ? 8 : 0);
#pragma unroll
    for (int r = 0; r < 3; ++r) {  // unit = kernel row r of layer e; the ring stage is u mod 3 = r because units per layer = stages
      const int u = e * 3 + r;
      // unit u landed?  each wave has three DMAs per unit in flight, one unit issued beyond u (the last unit drains)
      if (u == 0 || u + 1 >= U) wait_le<0>(); else wait_younger(r < 2 ? 3 + extra : 3);
      // raw barrier (a __syncthreads() would drain vmcnt to 0 and serialise the ring): after lgkmcnt(0) this wave's LDS writes
      // (input staging, previous layer's epilogue) are complete; unit u is in LDS for every wave, every wave is done with u-1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (u + 2 < U) issue(u + 2, (r + 2) % 3);
      if (r == 0) {  // accumulators start from the bias: lane half kq holds channels 8g + 4kq .. +3 of its 32-channel half
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b0 = *(const f32x4*)(bias_l + e * 64 + mb * 32 + 8 * g + 4 * kq);
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc0[4 * g + j] = b0[j]; acc1[4 * g + j] = b0[j]; }
        }
      }
      if (DBG && (dbg & 2)) continue;
      // activations of image rows + (r - 1), read ONCE per kernel row: the side taps are lane shifts inside the 16-pixel rows
      // of the operand (DPP row_shr / row_shl; the lane shifted in from outside a row reads 0 = the zero padding)
      u32x4 xc[2][4];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) xc[nb][cb] = *(const u32x4*)(in + ((r - 1) * 18 + nb * 36) * kFS + cb * 32);
#pragma unroll
      for (int c = 0; c < 3; ++c) {  // tap (dy, dx) = (r - 1, c - 1)
        const char* wb = ring + r * kFUnit + c * 8192 + mb * 1024 + vw;
        bf16x8 w[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) w[cb] = *(const bf16x8*)(wb + cb * 2048);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            u32x4 xs = xc[nb][cb];
            if (c == 0) {
#pragma unroll
              for (int j = 0; j < 4; ++j) xs[j] = __builtin_amdgcn_update_dpp(0u, xc[nb][cb][j], 0x111, 0xf, 0xf, true);  // row_shr:1
            } else if (c == 2) {
#pragma unroll
              for (int j = 0; j < 4; ++j) xs[j] = __builtin_amdgcn_update_dpp(0u, xc[nb][cb][j], 0x101, 0xf, 0xf, true);  // row_shl:1
            }
            bf16x8 xv = __builtin_bit_cast(bf16x8, xs);
            if (DBG && (dbg & 64)) xv = w[0];
            if (nb == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(DBG && (dbg & 64) ? w[0] : w[cb], xv, acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(DBG && (dbg & 64) ? w[0] : w[cb], xv, acc1, 0, 0, 0);
          }
        }
      }
    }
    if (e == NL - 1) break;
    // ---- hidden layer: ReLU (or the saved mask), optional fp32 store, bf16 back into the SAME tile once every wave has read it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* st = fa.store[e];
    if (has_mask) {  // landed: the wait of kernel row 2 left only the next unit's DMAs in flight
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(mreg[i]));
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const f32x16& acc = nb ? acc1 : acc0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        const int Q = mb * 8 + 2 * g + kq;
        const size_t off = (((size_t)b * 16 + Q) * kPix + (nb ? P1 : P0)) * 4;
        if (has_mask) {
          const f32x4 m = mreg[nb * 4 + g];
          v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        } else {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (st) *(f32x4*)(st + off) = v;
        *(u32x2*)(act + ((row0 + 2 * nb + 1) * 18 + px + 1) * kFS + Q * 8) = u32x2{pk_bf16(v.x, v.y), pk_bf16(v.z, v.w)};
      }
    }
  }
  // ---- last layer: the shared fused epilogue (stage combine, error partials, reverse-sweep targets, ...)
  epilogue(fa.last, acc0, b, mb, P0, kq, wave, b * 16 + wave);
  epilogue(fa.last, acc1, b, mb, P1, kq, wave, b * 16 + 8 + wave);
}

// fused image of executed layer `e`: [tap][cb][mb][h][co32][8]
__global__ __launch_bounds__(256) void pack_fused_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int transpose_flip) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // 9 * 4 * 2 * 64 * 8 = 36864 elements
  if (idx >= 36864) return;
  int r = idx;
  const int j = r & 7; r >>= 3;
  const int co_l = r & 31; r >>= 5;
  const int h = r & 1; r >>= 1;
  const int mb = r & 1; r >>= 1;
  const int cb = r & 3; r >>= 2;
  const int tap = r;
  const int co = mb * 32 + co_l, ci = cb * 16 + h * 8 + j;
  const float v = transpose_flip ? w[((size_t)ci * 64 + co) * 9 + (8 - tap)] : w[((size_t)co * 64 + ci) * 9 + tap];
  out[idx] = (__bf16)v;
}

int launch_fstack_bf16(const FusedArgs& fa_in, int batch, hipStream_t stream) {
  FusedArgs fa = fa_in;
  fa.last.debug = g_debug_flags;
  static bool attr_set = false;
  if (!attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)fstack_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)fstack_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  if (fa.last.debug) hipLaunchKernelGGL(fstack_bf16_kernel<true>, dim3(batch), dim3(512), kFusedLds, stream, fa);
  else hipLaunchKernelGGL(fstack_bf16_kernel<false>, dim3(batch), dim3(512), kFusedLds, stream, fa);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

}  // namespace odehip

using namespace odehip;

extern "C" size_t odehip_fused_bf16_weight_bytes(int n_layers) { return (size_t)n_layers * 9 * 8192; }

extern "C" int odehip_pack_convstack_fused_bf16(const float* w_oihw, void* w_fused, int exec_index, int transpose_flip, void* stream) {
  ODEHIP_REQUIRE(w_oihw && w_fused && exec_index >= 0 && exec_index < ODEHIP_MAX_LAYERS, "pack_convstack_fused_bf16: bad argument");
  hipLaunchKernelGGL(pack_fused_bf16_kernel, dim3(144), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     (__bf16*)((char*)w_fused + (size_t)exec_index * 9 * 8192), transpose_flip);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}
