// layout.hip -- boundary layout conversion and weight packing (HBM-bound helpers, gfx950).
//
//   NCHW (the reference's tensors, modules/DiffEqSolver.py:24-52)  <->  Q4 [b][c/4][pixel][4]
//   OIHW conv weight (nn.Conv2d, helpers/utils.py:167-177)          ->  MFMA-ordered LDS image
#include <string.h>

#include "odehip_internal.h"
#include "pack_elems.h"

namespace odehip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one thread = one (b, quad, pixel): reads 4 channel planes (coalesced over pixels), writes 16 B.
__global__ __launch_bounds__(256) void nchw_to_q4_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         int total /* b*quads*256 */, int quads) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int p = idx & 255;
  const int bq = idx >> 8;  // b*quads + q
  const float* s = src + (size_t)bq * 4 * kPix + p;
  f32x4 v = {s[0], s[kPix], s[2 * kPix], s[3 * kPix]};
  *(f32x4*)(dst + (size_t)idx * 4) = v;
  (void)quads;
}

// Prologue of a fixed-grid trajectory in ONE launch: y0 NCHW -> Q4 and a verbatim NCHW copy of it (solution[0] = y0), the step
// sizes (carried in the kernel arguments) into device memory, and the persistent launch's flag area zeroed -- instead of a
// device-to-device memcpy, the layout kernel, an upload kernel and a memset (4 launches, ~20 us of a 1.4 ms trajectory).
struct ProloguePack {
  float h[64];
};
__global__ __launch_bounds__(256) void traj_prologue_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            float* __restrict__ copy, int total, ProloguePack hp, int n_h,
                                                            float* __restrict__ hdev, unsigned* __restrict__ zero_words, int n_zero) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && (int)threadIdx.x < n_h) hdev[threadIdx.x] = hp.h[threadIdx.x];
  for (int i = idx; i < n_zero; i += gridDim.x * 256) zero_words[i] = 0u;
  if (idx >= total) return;
  const int p = idx & 255;
  const int bq = idx >> 8;
  const size_t o = (size_t)bq * 4 * kPix + p;
  const f32x4 v = {src[o], src[o + kPix], src[o + 2 * kPix], src[o + 3 * kPix]};
  *(f32x4*)(dst + (size_t)idx * 4) = v;
  copy[o] = v.x; copy[o + kPix] = v.y; copy[o + 2 * kPix] = v.z; copy[o + 3 * kPix] = v.w;
}

// n_h <= 64 step sizes (more: the caller uploads them itself and passes n_h = 0); zero_words may be null
int traj_prologue(const float* src, float* dst_q4, float* copy_nchw, int batch, int channels, const float* h_host, int n_h, float* hdev,
                  unsigned* zero_words, int n_zero, hipStream_t stream) {
  ODEHIP_REQUIRE(src && dst_q4 && copy_nchw && batch > 0 && channels > 0 && channels % 4 == 0 && n_h >= 0 && n_h <= 64,
                 "traj_prologue: bad arguments");
  const int total = batch * (channels / 4) * kPix;
  ProloguePack hp;
  for (int i = 0; i < 64; ++i) hp.h[i] = i < n_h ? h_host[i] : 0.0f;
  hipLaunchKernelGGL(traj_prologue_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, src, dst_q4, copy_nchw, total, hp, n_h, hdev,
                     zero_words, zero_words ? n_zero : 0);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

__global__ __launch_bounds__(256) void q4_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int p = idx & 255;
  const int bq = idx >> 8;
  const f32x4 v = *(const f32x4*)(src + (size_t)idx * 4);
  float* d = dst + (size_t)bq * 4 * kPix + p;
  d[0] = v.x; d[kPix] = v.y; d[2 * kPix] = v.z; d[3 * kPix] = v.w;
}

// packed[ct][m][tap][kq][i][s] = W[co = ct*32+i][ci = 8m+4kq+s][tap]      (transpose_flip == 0)
//                              = W[co' = ci][ci' = co][taps-1-tap]        (dgrad weights)
__device__ __forceinline__ void pack_weight_elem(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int taps,
                                                 int transpose_flip, int idx) {
  int r = idx;
  const int s = r & 3; r >>= 2;
  const int i = r & 31; r >>= 5;
  const int kq = r & 1; r >>= 1;
  const int tap = r % taps; r /= taps;
  const int mcount = cin / 8;
  const int m = r % mcount;
  const int ct = r / mcount;
  const int co = ct * 32 + i, ci = 8 * m + 4 * kq + s;
  float v;
  if (!transpose_flip) {
    v = w[((size_t)co * cin + ci) * taps + tap];
  } else {
    // source tensor is (cin_of_this_conv = original cout ... ) : original layout W[o][i'][tap] with
    // o = ci (this conv's input channel), i' = co (this conv's output channel)
    v = w[((size_t)ci * cout + co) * taps + (taps - 1 - tap)];
  }
  out[idx] = v;
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                          int cout, int cin, int taps, int transpose_flip,
                                                          int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < total) pack_weight_elem(w, out, cout, cin, taps, transpose_flip, idx);
}

// Winograd F(2x2,3x3) weights U = G g G^T in the LDS image of conv_wino.hip:
//   out[ct][c][xi][quad][i][s] = U_xi[co = ct*32+i][ci = 16c + 4quad + s],  xi = 4r + col
__device__ __forceinline__ void pack_winograd_elem(const float* __restrict__ w, float* __restrict__ out, int cout, int cin,
                                                   int transpose_flip, int idx) {
  int r = idx;
  const int s = r & 3; r >>= 2;
  const int i = r & 31; r >>= 5;
  const int quad = r & 3; r >>= 2;
  const int xi = r & 15; r >>= 4;
  const int nc = cin / 16;
  const int c = r % nc;
  const int ct = r / nc;
  const int co = ct * 32 + i, ci = 16 * c + 4 * quad + s;
  float g[3][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
      g[ky][kx] = transpose_flip ? w[((size_t)ci * cout + co) * 9 + (2 - ky) * 3 + (2 - kx)] : w[((size_t)co * cin + ci) * 9 + ky * 3 + kx];
  const int ur = xi >> 2, uc = xi & 3;
  // row ur of G g (a 3-vector), then its product with column uc of G^T
  float t[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const float g0 = g[0][kx], g1 = g[1][kx], g2 = g[2][kx];
    t[kx] = ur == 0 ? g0 : (ur == 1 ? 0.5f * (g0 + g1 + g2) : (ur == 2 ? 0.5f * (g0 - g1 + g2) : g2));
  }
  const float uv = uc == 0 ? t[0] : (uc == 1 ? 0.5f * (t[0] + t[1] + t[2]) : (uc == 2 ? 0.5f * (t[0] - t[1] + t[2]) : t[2]));
  out[idx] = ur == 3 ? -uv : uv;  // conv_wino.hip computes row 3 of B^T d with the opposite sign
}

__global__ __launch_bounds__(256) void pack_winograd_kernel(const float* __restrict__ w, float* __restrict__ out, int cout,
                                                            int cin, int transpose_flip, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < total) pack_winograd_elem(w, out, cout, cin, transpose_flip, idx);
}

// every pack of a conv stack in ONE launch (a training step repacks ~50 weight tensors after the optimizer's update: as single
// launches of 5 us each they were 0.3 ms of a 12 ms step)
struct PackJobs {
  odehip_pack_job job[ODEHIP_MAX_PACK_JOBS];
  int start[ODEHIP_MAX_PACK_JOBS + 1];  // first element index of job j in the launch's index space (multiples of 256)
  int n;
};

__global__ __launch_bounds__(256) void pack_many_kernel(const PackJobs p) {
  const int blk = blockIdx.x * 256;
  int j = 0;
  while (j + 1 < p.n && blk >= p.start[j + 1]) ++j;  // block-uniform: job boundaries are multiples of the block size
  const odehip_pack_job& q = p.job[j];
  const int idx = blk - p.start[j] + threadIdx.x;
  const int taps = q.ks * q.ks;
  const int total = q.cout * q.cin * (q.kind == 1 ? 16 : (q.kind == 2 ? 36 : taps));
  if (idx >= total) return;
  if (q.kind == 1) pack_winograd_elem(q.w, q.out, q.cout, q.cin, q.transpose_flip, idx);
  else if (q.kind == 2) pack_winograd5_elem(q.w, q.out, q.cout, q.cin, q.transpose_flip, idx);
  else pack_weight_elem(q.w, q.out, q.cout, q.cin, taps, q.transpose_flip, idx);
}

}  // namespace odehip

using namespace odehip;

extern "C" size_t odehip_winograd_weight_floats(int cout, int cin) { return (size_t)cout * cin * 16; }

extern "C" int odehip_pack_conv_weight_winograd(const float* w_oihw, float* w_wino, int cout, int cin, int transpose_flip,
                                                void* stream) {
  ODEHIP_REQUIRE(w_oihw && w_wino, "pack_conv_weight_winograd: null pointer");
  ODEHIP_REQUIRE(cout > 0 && cout % 32 == 0, "pack_conv_weight_winograd: cout must be a multiple of 32 (got %d)", cout);
  ODEHIP_REQUIRE(cin > 0 && cin % 16 == 0, "pack_conv_weight_winograd: cin must be a multiple of 16 (got %d)", cin);
  const int total = cout * cin * 16;
  hipLaunchKernelGGL(pack_winograd_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, w_wino, cout, cin,
                     transpose_flip, total);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

extern "C" size_t odehip_packed_weight_floats(int cout, int cin, int ks) {
  return (size_t)cout * cin * ks * ks;
}

extern "C" int odehip_pack_conv_weight(const float* w_oihw, float* w_packed, int cout, int cin, int ks,
                                       int transpose_flip, void* stream) {
  ODEHIP_REQUIRE(w_oihw && w_packed, "pack_conv_weight: null pointer");
  ODEHIP_REQUIRE(cout > 0 && cout % 32 == 0, "pack_conv_weight: cout must be a multiple of 32 (got %d)", cout);
  ODEHIP_REQUIRE(cin > 0 && cin % 8 == 0, "pack_conv_weight: cin must be a multiple of 8 (got %d)", cin);
  ODEHIP_REQUIRE(ks == 1 || ks == 3 || ks == 5, "pack_conv_weight: kernel size %d unsupported", ks);
  const int total = cout * cin * ks * ks;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     w_packed, cout, cin, ks * ks, transpose_flip, total);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

extern "C" int odehip_pack_conv_weights(const odehip_pack_job* jobs, int n_jobs, void* stream) {
  ODEHIP_REQUIRE(jobs && n_jobs > 0 && n_jobs <= ODEHIP_MAX_PACK_JOBS, "pack_conv_weights: 1..%d jobs (got %d)", ODEHIP_MAX_PACK_JOBS, n_jobs);
  PackJobs p;
  memset(&p, 0, sizeof(p));
  p.n = n_jobs;
  long long at = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const odehip_pack_job& q = jobs[j];
    ODEHIP_REQUIRE(q.w && q.out, "pack_conv_weights: job %d: null pointer", j);
    ODEHIP_REQUIRE(q.kind >= 0 && q.kind <= 2, "pack_conv_weights: job %d: kind %d (0: odehip_pack_conv_weight, 1: ..._winograd, 2: ..._winograd5)", j, q.kind);
    ODEHIP_REQUIRE(q.cout > 0 && q.cout % 32 == 0, "pack_conv_weights: job %d: cout must be a multiple of 32 (got %d)", j, q.cout);
    if (q.kind == 1) {
      ODEHIP_REQUIRE(q.cin > 0 && q.cin % 16 == 0 && q.ks == 3, "pack_conv_weights: job %d: Winograd packs need 3x3 and cin %% 16 == 0", j);
    } else {
      ODEHIP_REQUIRE(q.cin > 0 && q.cin % 8 == 0, "pack_conv_weights: job %d: cin must be a multiple of 8 (got %d)", j, q.cin);
      ODEHIP_REQUIRE(q.kind == 2 ? q.ks == 5 : (q.ks == 1 || q.ks == 3 || q.ks == 5), "pack_conv_weights: job %d: kernel size %d unsupported", j, q.ks);
    }
    p.job[j] = q;
    p.start[j] = (int)at;
    const long long total = (long long)q.cout * q.cin * (q.kind == 1 ? 16 : (q.kind == 2 ? 36 : q.ks * q.ks));
    at += (total + 255) / 256 * 256;
    ODEHIP_REQUIRE(at < (1LL << 31), "pack_conv_weights: too many elements in one call");
  }
  p.start[n_jobs] = (int)at;
  hipLaunchKernelGGL(pack_many_kernel, dim3((unsigned)(at / 256)), dim3(256), 0, (hipStream_t)stream, p);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

extern "C" int odehip_nchw_to_q4(const float* src, float* dst, int batch, int channels, void* stream) {
  ODEHIP_REQUIRE(src && dst, "nchw_to_q4: null pointer");
  ODEHIP_REQUIRE(batch > 0 && channels > 0 && channels % 4 == 0, "nchw_to_q4: bad shape (%d, %d)", batch, channels);
  const int total = batch * (channels / 4) * kPix;
  hipLaunchKernelGGL(nchw_to_q4_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, total,
                     channels / 4);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

extern "C" int odehip_q4_to_nchw(const float* src, float* dst, int batch, int channels, void* stream) {
  ODEHIP_REQUIRE(src && dst, "q4_to_nchw: null pointer");
  ODEHIP_REQUIRE(batch > 0 && channels > 0 && channels % 4 == 0, "q4_to_nchw: bad shape (%d, %d)", batch, channels);
  const int total = batch * (channels / 4) * kPix;
  hipLaunchKernelGGL(q4_to_nchw_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, total);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}
