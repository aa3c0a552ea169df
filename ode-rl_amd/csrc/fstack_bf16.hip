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
#include <string.h>

#include "fused_bf16.h"

namespace odehip {

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
          v.x = relu_f(v.x); v.y = relu_f(v.y); v.z = relu_f(v.z); v.w = relu_f(v.w);
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

// ------------------------------------------------------------------------------------------------------------------------------
// ftraj_bf16_kernel<METHOD, SAVE> -- a WHOLE fixed-grid trajectory (every interval, every Runge-Kutta stage, every layer of
// every evaluation of f) in ONE launch, one workgroup per sample: between two evaluations nothing leaves the CU.  The solver
// state y and the stage derivatives k_1 .. k_{S-1} stay in REGISTERS (fp32, the accumulator layout: lane = (pixel, channel-quad
// pair)), the next stage's input y + h*sum c_j k_j is formed there and written straight into the LDS activation tile as bf16 --
// the per-evaluation launch above pays for that hand-over with a launch boundary, an fp32 round trip of x through HBM and a
// staging pass (about 10 of its 21 us).  The weight ring simply keeps cycling through the 3*NL units of f; the result frames
// (fp32 NCHW) and, with SAVE, every stage input and hidden activation (bf16 "Q4h": what the backward sweep's masks and the
// weight-gradient kernel consume) are stored asynchronously -- counted in the ring's vmcnt waits, never drained.  Arithmetic and
// its order are those of the per-evaluation path (same bf16 roundings, same stage-combine expressions as
// conv_common.h::epilogue), so the trajectories are bit-identical.
struct TrajArgs {
  const float* z0_nchw;                  // (B,64,16,16)
  float* out_nchw;                       // (T,B,64,16,16); frame 0 is written by the caller
  const float* hdev;                     // [n_steps] step sizes (fp32, as they meet the state)
  const void* w_fused;
  const float* bias[ODEHIP_MAX_LAYERS];
  int n_layers, n_steps, batch;
  float k_scale;                         // -1: negated dynamics (decreasing time grid)
  // SAVE: base of the stage inputs / hidden activations, slot strides in bytes (evaluation e = n*S + s; layer l)
  char* save_x;                          // [e] : Q4h of x_s (s = 0: y_n)
  char* save_h;                          // [e][l] : Q4h of the ReLU output of conv l < NL-1
  unsigned long long stride_x, stride_h_eval, stride_h_layer;
};

// the stage programs of fixed_grid.hip (torchdiffeq _impl/fixed_grid.py; rk4 = the 3/8 rule), compile-time constants here: per
// stage, how many earlier k enter its combine (n_prev) and the weights of k_1 .. k_{n_prev} and (last) of the stage's own k
template <int METHOD> struct StageProgram;
template <> struct StageProgram<ODEHIP_EULER> {
  static constexpr int S = 1;
  static constexpr int n_prev[4] = {0, 0, 0, 0};
  static constexpr float c[4][4] = {{1.0f, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
};
template <> struct StageProgram<ODEHIP_MIDPOINT> {
  static constexpr int S = 2;
  static constexpr int n_prev[4] = {0, 0, 0, 0};   // the midpoint rule's result does not use k1
  static constexpr float c[4][4] = {{0.5f, 0, 0, 0}, {1.0f, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
};
template <> struct StageProgram<ODEHIP_RK4> {
  static constexpr int S = 4;
  static constexpr int n_prev[4] = {0, 1, 2, 3};
  static constexpr float c[4][4] = {{1.0f / 3.0f, 0, 0, 0}, {-(1.0f / 3.0f), 1.0f, 0, 0}, {1.0f, -1.0f, 1.0f, 0}, {0.125f, 0.375f, 0.375f, 0.125f}};
};

template <int METHOD, bool SAVE>
__global__ __launch_bounds__(512, 1) void ftraj_bf16_kernel(const TrajArgs ta) {
  typedef StageProgram<METHOD> Prog;
  constexpr int S = Prog::S;
  constexpr int NK = Prog::n_prev[S - 1];   // stage derivatives that must be kept (rk4: 3, midpoint / euler: 0)
  // TWO activation tiles (round 3): a layer reads one and writes its output into the other, so no wave has to wait for the slowest
  // reader before it rewrites -- the in-place rewrite of the single-tile kernels needs a barrier of its own per layer and
  // serialises the epilogue behind it.  Both fit next to the weight ring because a tile here has NO column borders: the dx = -1 /
  // +1 taps are lane shifts of the centre fragment whose shifted-in lane is zero, columns 0 and 17 of the [18][18] tile were never
  // read.  [18 rows][16 columns][64 ch + pad]: 41,472 B each.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem + 2 * kTTile;
  float* const bias_l = (float*)(smem + 2 * kTTile + kFStages * kFUnit);
  int cur = 0;   // the tile the next layer reads (wave-uniform)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x;
  const int NL = ta.n_layers, UE = NL * 3;
  const long long U = (long long)ta.n_steps * S * UE;   // ring units of the whole trajectory

  const __amdgpu_buffer_rsrc_t rw = make_rsrc(ta.w_fused, (unsigned)(UE * kFUnit));
  const int vw = lane * 16;
  auto issue = [&](int ue, int stage) {  // unit `ue` of f (0 .. UE-1) into ring stage `stage`: three 1-KiB pieces per wave
#pragma unroll
    for (int j = 0; j < 3; ++j)
      dma16(rw, ring + stage * kFUnit + (wave * 3 + j) * 1024, vw, ue * kFUnit + (wave * 3 + j) * 1024);
  };
  issue(0, 0);
  issue(1 % UE, 1);

  for (int i = threadIdx.x; i < NL * 64; i += 512) bias_l[i] = ta.bias[i >> 6] ? ta.bias[i >> 6][i & 63] : 0.0f;
  for (int i = threadIdx.x; i < 64 * 9; i += 512) {  // zero row borders (rows 0 and 17) of both tiles
    const int p = i / 9, c16 = i % 9;
    const int tile = p >> 5, row = (p & 16) ? 17 : 0, col = p & 15;
    *(f32x4*)(smem + tile * kTTile + (row * 16 + col) * kFS + c16 * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int i32 = lane & 31, kq = lane >> 5;
  const int px = i32 & 15, pyl = i32 >> 4;
  const int mb = wave & 1, row0 = (wave >> 1) * 4 + pyl;
  const int P0 = row0 * 16 + px, P1 = P0 + 32;
  const int in_off = ((row0 + 1) * 16 + px) * kFS + kq * 16;

  // solver state: y[nb][4g + j] = channel 32 mb + 8 g + 4 kq + j of pixel (nb ? P1 : P0) -- the accumulator layout
  f32x16 y[2], k[NK > 0 ? NK : 1][2];
  // channels of quad Q = 8 mb + 2 g + kq as bf16: into the activation tile and, if `dst` (wave-uniform: this sample's Q4h tensor),
  // into HBM -- one lane offset, the quads 4 KiB apart
  const unsigned q4h_off = (unsigned)(((mb * 8 + kq) * kPix + P0) * 8);
  // written into the tile the CURRENT layer does not read (cur ^ 1); the caller flips `cur` once both pixel blocks are out
  auto emit = [&](const f32x16& v, int nb, char* dst) {
    char* const wt = smem + (cur ^ 1) * kTTile;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int Q = mb * 8 + 2 * g + kq;
      const u32x2 pk = {pk_bf16(v[4 * g], v[4 * g + 1]), pk_bf16(v[4 * g + 2], v[4 * g + 3])};
      *(u32x2*)(wt + ((row0 + 2 * nb + 1) * 16 + px) * kFS + Q * 8) = pk;
      if (SAVE && dst) *(u32x2*)(dst + (q4h_off + (unsigned)(g * 2 * kPix * 8 + nb * 32 * 8))) = pk;
    }
  };
  char* const save_x = SAVE ? wave_uniform(ta.save_x + (size_t)b * kQ4hSample) : nullptr;
  char* const save_h = SAVE ? wave_uniform(ta.save_h + (size_t)b * kQ4hSample) : nullptr;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float* src = ta.z0_nchw + ((size_t)b * 64 + mb * 32 + 8 * g + 4 * kq) * kPix + (nb ? P1 : P0);
#pragma unroll
      for (int j = 0; j < 4; ++j) y[nb][4 * g + j] = src[(size_t)j * kPix];
    }
    emit(y[nb], nb, save_x);
  }
  cur ^= 1;

  long long u = 0;
  int frame_pending = 0;   // the previous stage stored a result frame: 32 more stores younger than the first two units' DMAs (wave-uniform)
  f32x16 acc0, acc1;
  for (int n = 0; n < ta.n_steps; ++n) {
    const float h = ta.hdev[n];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const size_t ev = (size_t)n * S + s;
      for (int e = 0; e < NL; ++e) {
        const char* const in = smem + cur * kTTile + in_off;
#pragma unroll
        for (int r = 0; r < 3; ++r, ++u) {
          // unit u landed?  Younger than its DMAs: the next unit's three DMAs and -- for the first two units of a layer -- what was
          // stored after the previous layer: its saved activation (8 stores per lane) or, after a stage, the saved stage input
          // (8) and, after an interval's last stage, the result frame (32)
          // (one two-way branch with immediates: a chain of alternative waits in this loop costs the register allocator its footing)
          if (u == 0 || u + 1 >= U) wait_le<0>();
          else if (e == 0 && r < 2 && frame_pending) wait_le<3 + 32 + (SAVE ? 8 : 0)>();
          else if (r < 2) wait_le<3 + (SAVE ? 8 : 0)>();
          else wait_le<3>();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (u + 2 < U) issue((int)((u + 2) % UE), (r + 2) % 3);
          if (r == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 b0 = *(const f32x4*)(bias_l + e * 64 + mb * 32 + 8 * g + 4 * kq);
#pragma unroll
              for (int j = 0; j < 4; ++j) { acc0[4 * g + j] = b0[j]; acc1[4 * g + j] = b0[j]; }
            }
          }
          u32x4 xc[2][4];
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) xc[nb][cb] = *(const u32x4*)(in + ((r - 1) * 16 + nb * 32) * kFS + cb * 32);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
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
                  for (int j = 0; j < 4; ++j) xs[j] = __builtin_amdgcn_update_dpp(0u, xc[nb][cb][j], 0x111, 0xf, 0xf, true);
                } else if (c == 2) {
#pragma unroll
                  for (int j = 0; j < 4; ++j) xs[j] = __builtin_amdgcn_update_dpp(0u, xc[nb][cb][j], 0x101, 0xf, 0xf, true);
                }
                const bf16x8 xv = __builtin_bit_cast(bf16x8, xs);
                if (nb == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[cb], xv, acc0, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[cb], xv, acc1, 0, 0, 0);
              }
            }
          }
        }
        if (e == 0) frame_pending = 0;
        // the output goes to the OTHER tile: no wave has to be waited for (hidden layer: ReLU; last layer: the next stage input /
        // new state, below); the first barrier of the next layer makes it visible
        if (e < NL - 1) {
          char* const dst = SAVE ? wave_uniform(save_h + ev * ta.stride_h_eval + (size_t)e * ta.stride_h_layer) : nullptr;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            f32x16 v = nb ? acc1 : acc0;
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = relu_f(v[i]);
            emit(v, nb, dst);
          }
          cur ^= 1;
        }
      }
      // ---- stage combine (conv_common.h::epilogue, combine == 1): kc = k_scale * f(x_s);  out = y + h * (c[n_prev] kc + sum_j c[j] k_j)
      const bool final_stage = s == S - 1;
      // the next evaluation's input: x_{s+1} of this interval, or y_{n+1} (= x_1 of the next interval; not needed after the last)
      char* const dst = SAVE ? wave_uniform(save_x + (ev + 1) * ta.stride_x) : nullptr;
      const bool last_eval = final_stage && n + 1 == ta.n_steps;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const f32x16& acc = nb ? acc1 : acc0;
        f32x16 o;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 kc = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
          kc *= ta.k_scale;
          f32x4 sa = kc * Prog::c[s][Prog::n_prev[s]];
#pragma unroll
          for (int j = 0; j < NK; ++j)
            if (j < Prog::n_prev[s]) {
              const f32x4 kp = {k[j][nb][4 * g], k[j][nb][4 * g + 1], k[j][nb][4 * g + 2], k[j][nb][4 * g + 3]};
              sa += kp * Prog::c[s][j];
            }
          const f32x4 yv = {y[nb][4 * g], y[nb][4 * g + 1], y[nb][4 * g + 2], y[nb][4 * g + 3]};
          const f32x4 ov = yv + sa * h;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[4 * g + j] = ov[j];
          if (s < NK) {
#pragma unroll
            for (int j = 0; j < 4; ++j) k[s < NK ? s : 0][nb][4 * g + j] = kc[j];
          }
        }
        if (final_stage) {
          y[nb] = o;
          float* fr = ta.out_nchw + ((size_t)(n + 1) * ta.batch + b) * 64 * kPix + (nb ? P1 : P0);
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) fr[(size_t)(mb * 32 + 8 * g + 4 * kq + j) * kPix] = o[4 * g + j];
        }
        emit(o, nb, last_eval ? nullptr : dst);
      }
      cur ^= 1;
      if (final_stage) frame_pending = 1;
    }
  }
}

static void traj_args(TrajArgs& ta, const odehip_convstack* f, const float* z0_nchw, float* out_nchw, const float* hdev, int n_times,
                      int batch, int negate) {
  memset(&ta, 0, sizeof(ta));
  ta.z0_nchw = z0_nchw; ta.out_nchw = out_nchw; ta.hdev = hdev; ta.w_fused = f->w_fused;
  for (int l = 0; l < f->n_convs; ++l) ta.bias[l] = f->bias[l];
  ta.n_layers = f->n_convs; ta.n_steps = n_times - 1; ta.batch = batch;
  ta.k_scale = negate ? -1.0f : 1.0f;
}

template <int METHOD, bool SAVE>
static int launch_ftraj(const TrajArgs& ta, int batch, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)ftraj_bf16_kernel<METHOD, SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((ftraj_bf16_kernel<METHOD, SAVE>), dim3(batch), dim3(512), kTrajLds, stream, ta);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

int launch_ftraj_bf16(const odehip_convstack* f, int method, const float* z0_nchw, float* out_nchw, const float* hdev, int n_times,
                      int batch, int negate, hipStream_t stream) {
  TrajArgs ta;
  traj_args(ta, f, z0_nchw, out_nchw, hdev, n_times, batch, negate);
  if (method == ODEHIP_EULER) return launch_ftraj<ODEHIP_EULER, false>(ta, batch, stream);
  if (method == ODEHIP_MIDPOINT) return launch_ftraj<ODEHIP_MIDPOINT, false>(ta, batch, stream);
  if (method == ODEHIP_RK4) return launch_ftraj<ODEHIP_RK4, false>(ta, batch, stream);
  set_error("ftraj_bf16: unknown method %d", method);
  return ODEHIP_EINVAL;
}

// rk4 training forward: also saves every stage input (slot e = n*4 + s of save_x) and hidden activation (save_h) as bf16 Q4h
int launch_ftraj_bf16_saving(const odehip_convstack* f, const float* z0_nchw, float* out_nchw, const float* hdev, int n_times, int batch,
                             void* save_x, size_t stride_x, void* save_h, size_t stride_h_eval, size_t stride_h_layer,
                             hipStream_t stream) {
  TrajArgs ta;
  traj_args(ta, f, z0_nchw, out_nchw, hdev, n_times, batch, 0);
  ta.save_x = (char*)save_x; ta.save_h = (char*)save_h;
  ta.stride_x = stride_x; ta.stride_h_eval = stride_h_eval; ta.stride_h_layer = stride_h_layer;
  return launch_ftraj<ODEHIP_RK4, true>(ta, batch, stream);
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
