// upsample.hip -- nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) of VidODE's flow decoder
// (/root/reference/models/VidODE.py:34: one in front of each of the decoder's two 3x3 convolutions, applied once per predicted frame:
// get_flowmaps :143-158), forward and backward.
//
// Why it exists (round 4, tools/vidode_bench.py + rocprofv3): torch's own kernel for this op takes 4.24 ms per call on this stack
// (upsample_bilinear2d_out_frame, NCHW fp32, 64 x 256 x 16 x 16 -> 32 x 32: 84 MB of traffic at 20 GB/s) -- 85 of the 102 ms of a VidODE
// forward at the config's per-GPU batch of 64, while the library convolutions next to it run at 143 TFLOP/s algorithmic.  The op is
// pure data movement: every output pixel is a fixed-weight blend of <= 4 input pixels, so the bound is HBM: (1 + 4) x input bytes.
//
// ATen's arithmetic (UpSample.h, area_pixel_compute_source_index with align_corners = false, scale = 1 / 2):
//   src = max(0, 0.5 (dst + 0.5) - 0.5);  i0 = (int) src;  i1 = i0 + (i0 < n - 1);  l1 = src - i0;  l0 = 1 - l1
//   out = h0 (w0 in[y0][x0] + w1 in[y0][x1]) + h1 (w0 in[y1][x0] + w1 in[y1][x1])
// in fp32, in this order of operations.  One thread writes four consecutive output pixels of a row (a 16-byte store; the two input
// rows' <= 4 pixels each come from cache: the input is read from HBM once).
// Backward = the transpose as a GATHER (no atomics: bitwise reproducible): an input pixel collects from the <= 4 x 4 output pixels
// whose stencils contain it, with the weights the forward would have used.
#include "odehip_internal.h"

namespace odehip {

typedef float f32x4u __attribute__((ext_vector_type(4)));

struct Src {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Src src_of(int dst, int n) {
#pragma clang fp contract(off)
  float s = 0.5f * ((float)dst + 0.5f) - 0.5f;
  s = s < 0.0f ? 0.0f : s;
  Src r;
  r.i0 = (int)s;
  r.i1 = r.i0 + (r.i0 < n - 1 ? 1 : 0);
  r.l1 = s - (float)r.i0;
  r.l0 = 1.0f - r.l1;
  return r;
}

// planes = N * C images of (H, W) -> (2H, 2W); W % 2 == 0 (so 2W % 4 == 0)
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ in, float* __restrict__ out, long long planes, int H, int W) {
#pragma clang fp contract(off)
  const int W2 = 2 * W, H2 = 2 * H, q_per_row = W2 / 4;
  const long long total = planes * (long long)H2 * q_per_row;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int q = (int)(i % q_per_row);
    const long long r = i / q_per_row;
    const int oy = (int)(r % H2);
    const long long pl = r / H2;
    const Src sy = src_of(oy, H);
    const float* r0 = in + (pl * H + sy.i0) * (long long)W;
    const float* r1 = in + (pl * H + sy.i1) * (long long)W;
    f32x4u o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const Src sx = src_of(4 * q + k, W);
      const float top = sx.l0 * r0[sx.i0] + sx.l1 * r0[sx.i1];
      const float bot = sx.l0 * r1[sx.i0] + sx.l1 * r1[sx.i1];
      o[k] = sy.l0 * top + sy.l1 * bot;
    }
    *(f32x4u*)(out + (pl * H2 + oy) * (long long)W2 + 4 * q) = o;
  }
}

// grad_in[y][x] = sum over the output pixels (oy, ox) whose stencil contains (y, x) of weight(oy -> y) * weight(ox -> x) * g[oy][ox];
// candidates oy in [2y - 1, 2y + 2], ox likewise (a border pixel is hit twice by the clamped stencils: both hits are added)
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ g, float* __restrict__ gin, long long planes, int H, int W) {
#pragma clang fp contract(off)
  const int W2 = 2 * W, H2 = 2 * H;
  const long long total = planes * (long long)H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const long long r = i / W;
    const int y = (int)(r % H);
    const long long pl = r / H;
    // weights of the output rows 2y-1 .. 2y+2 on input row y: {1/4, 3/4, 3/4, 1/4}; at the borders the clamped stencils fold onto the
    // edge row (y = 0: rows 0, 1, 2 -> {1, 3/4, 1/4}; y = H-1: rows 2H-3 .. 2H-1 -> {1/4, 3/4, 1}) -- exactly what src_of() yields (the
    // weights are exact binary fractions), without its float -> int conversions per tap
    float wy[4] = {0.25f, 0.75f, 0.75f, 0.25f}, wx[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    if (y == 0) { wy[0] = 0.0f; wy[1] = 1.0f; }
    if (y == H - 1) { wy[2] = 1.0f; wy[3] = 0.0f; }
    if (x == 0) { wx[0] = 0.0f; wx[1] = 1.0f; }
    if (x == W - 1) { wx[2] = 1.0f; wx[3] = 0.0f; }
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int oy = 2 * y - 1 + a;
      if (oy < 0 || oy >= H2) continue;
      const float* row = g + (pl * H2 + oy) * (long long)W2;
      float s = 0.0f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ox = 2 * x - 1 + b;
        if (ox >= 0 && ox < W2) s += wx[b] * row[ox];
      }
      acc += wy[a] * s;
    }
    gin[i] = acc;
  }
}

}  // namespace odehip

using namespace odehip;

extern "C" int odehip_upsample2x_bilinear(const float* in, float* out, long long planes, int height, int width, void* stream) {
  ODEHIP_REQUIRE(in && out, "upsample2x_bilinear: null pointer");
  ODEHIP_REQUIRE(planes > 0 && height > 0 && width > 0 && width % 2 == 0, "upsample2x_bilinear: bad shape (planes %lld, %d x %d; the width must be even)",
                 planes, height, width);
  const long long total = planes * 2LL * height * (width / 2);
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, in, out, planes, height, width);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

extern "C" int odehip_upsample2x_bilinear_backward(const float* grad_out, float* grad_in, long long planes, int height, int width, void* stream) {
  ODEHIP_REQUIRE(grad_out && grad_in, "upsample2x_bilinear_backward: null pointer");
  ODEHIP_REQUIRE(planes > 0 && height > 0 && width > 0, "upsample2x_bilinear_backward: bad shape (planes %lld, %d x %d)", planes, height, width);
  const long long total = planes * (long long)height * width;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, grad_out, grad_in, planes, height,
                     width);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}
