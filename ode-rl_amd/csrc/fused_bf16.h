// fused_bf16.h -- constants and device helpers shared by the bf16 whole-stack / whole-trajectory kernels (fstack_bf16.hip,
// btraj_bf16.hip): LDS layout (one [18][18][64] bf16 activation tile, a 3-stage weight ring of one kernel row per stage, the
// staged biases), bf16 packing, untracked global loads and counted vmcnt waits.
#pragma once
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
// the whole-trajectory forward kernels: two tiles without column borders (fstack_bf16.hip, ftraj_bf16_kernel)
constexpr int kTTile = 18 * 16 * kFS;      // 41,472 B
constexpr int kTrajLds = 2 * kTTile + kFStages * kFUnit + kFBias;
static_assert(kTrajLds <= 160 * 1024, "two activation tiles, the weight ring and the biases must fit the CU's LDS");

// issued without the compiler's own s_waitcnt bookkeeping: the ring's counted waits cover it (see wait_younger)
__device__ __forceinline__ f32x4 gload_untracked(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int N>
__device__ __forceinline__ void wait_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// a pointer the compiler must treat as wave-uniform (SGPR base + 32-bit lane offset addressing instead of a 64-bit address per lane)
template <typename T>
__device__ __forceinline__ T* wave_uniform(T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}

// largest supported immediate <= n: waiting for FEWER outstanding operations than allowed is always safe
__device__ __forceinline__ void wait_at_most(int n) {
  if (n >= 51) wait_le<51>();
  else if (n >= 43) wait_le<43>();
  else if (n >= 35) wait_le<35>();
  else if (n >= 27) wait_le<27>();
  else if (n >= 19) wait_le<19>();
  else if (n >= 11) wait_le<11>();
  else if (n >= 3) wait_le<3>();
  else wait_le<0>();
}

// Saved tensors of the bf16 training path ("Q4h"): [B][16 quads][256 px] x 4 bf16 -- 8 bytes per (quad, pixel), 32 KiB per sample
// of a 64-channel map: exactly what the weight-gradient kernel multiplies and what the ReLU masks need (sign only).
constexpr int kQ4hSample = 16 * kPix * 8;

}  // namespace odehip
