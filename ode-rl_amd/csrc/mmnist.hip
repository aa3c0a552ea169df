// mmnist.hip -- Moving-MNIST-shaped frames rendered on the device: the reference's on-the-fly generator
// /root/reference/dataloader.py:47-79 (get_random_trajectory: unit-square random walk with reflecting walls, step 0.1, scaled to
// the 36-pixel canvas and truncated) and :81-103 (generate_moving_mnist: np.maximum compositing of 28x28 digits on a 64x64
// canvas), followed by the (x / 255.0) - 0.5 normalisation of __getitem__ (:217-218) -- so that end-to-end benchmarks and the
// training step need no dataset, cv2 or host->device frame copies.  The host draws the random initial state (position, heading,
// digit id) and hands it over as a few doubles per digit; the walk itself is sequential in time but tiny, so every workgroup
// (one frame) replays it up to its own frame in float64 with the reference's operation order (no FMA contraction: positions are
// bit-identical to the numpy loop).  HBM-bound: 16 KiB written per frame, glyphs and LUT stay in cache.
#include "odehip_internal.h"

namespace odehip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kImg = 64, kDigit = 28, kMaxDigits = 8;

struct MmnistArgs {
  const double* init;           // [B][D][4] = x, y, v_x, v_y in the unit square (dataloader.py:50-54)
  const int* ids;               // [B][D] glyph index
  const unsigned char* glyphs;  // [G][28][28]
  const float* lut;             // [256]: (float32(v) / 255) - 0.5 in float32, as numpy evaluates it on the reference's frames
  float* out_in;                // (B, t_in, 1, 64, 64)
  float* out_pred;              // (B, t_out, 1, 64, 64)
  int n_digits, t_in, t_out;
};

__global__ __launch_bounds__(256) void mmnist_render_kernel(const MmnistArgs a) {
#pragma clang fp contract(off)
  const int t = blockIdx.x, b = blockIdx.y;
  int top[kMaxDigits], left[kMaxDigits], id[kMaxDigits];
  const double canvas = (double)(kImg - kDigit), step = 0.1;
  for (int d = 0; d < a.n_digits; ++d) {
    const double* s = a.init + ((size_t)b * a.n_digits + d) * 4;
    double x = s[0], y = s[1], vx = s[2], vy = s[3];
    for (int i = 0; i <= t; ++i) {  // dataloader.py:58-76, same order of updates and tests
      y += vy * step;
      x += vx * step;
      if (x <= 0) { x = 0; vx = -vx; }
      if (x >= 1.0) { x = 1.0; vx = -vx; }
      if (y <= 0) { y = 0; vy = -vy; }
      if (y >= 1.0) { y = 1.0; vy = -vy; }
    }
    top[d] = (int)(canvas * y);   // .astype(np.int32): truncation (:79-80)
    left[d] = (int)(canvas * x);
    id[d] = a.ids[(size_t)b * a.n_digits + d];
  }
  float* dst = t < a.t_in ? a.out_in + ((size_t)b * a.t_in + t) * (kImg * kImg)
                          : a.out_pred + ((size_t)b * a.t_out + (t - a.t_in)) * (kImg * kImg);
  for (int p = threadIdx.x * 4; p < kImg * kImg; p += 256 * 4) {
    const int r = p >> 6, c0 = p & 63;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + j;
      int v = 0;
      for (int d = 0; d < a.n_digits; ++d) {
        const int gr = r - top[d], gc = c - left[d];
        if (gr >= 0 && gr < kDigit && gc >= 0 && gc < kDigit) {
          const int g = a.glyphs[((size_t)id[d] * kDigit + gr) * kDigit + gc];
          v = g > v ? g : v;  // np.maximum(canvas, digit) (:100)
        }
      }
      o[j] = a.lut[v];
    }
    *(f32x4*)(dst + p) = f32x4{o[0], o[1], o[2], o[3]};
  }
}

}  // namespace odehip

using namespace odehip;

extern "C" int odehip_mmnist_render(const double* init, const int* digit_ids, const unsigned char* glyphs, int n_glyphs,
                                    const float* lut256, int batch, int n_digits, int t_in, int t_out, float* out_in,
                                    float* out_pred, void* stream) {
  ODEHIP_REQUIRE(init && digit_ids && glyphs && lut256, "mmnist_render: null pointer argument");
  ODEHIP_REQUIRE(batch > 0 && t_in >= 0 && t_out >= 0 && t_in + t_out > 0, "mmnist_render: bad batch / frame counts");
  ODEHIP_REQUIRE(n_digits >= 1 && n_digits <= kMaxDigits, "mmnist_render: 1..%d digits per frame (got %d)", kMaxDigits, n_digits);
  ODEHIP_REQUIRE(n_glyphs >= 1, "mmnist_render: no glyphs");
  ODEHIP_REQUIRE((t_in == 0 || out_in) && (t_out == 0 || out_pred), "mmnist_render: null output");
  MmnistArgs a;
  a.init = init; a.ids = digit_ids; a.glyphs = glyphs; a.lut = lut256; a.out_in = out_in; a.out_pred = out_pred;
  a.n_digits = n_digits; a.t_in = t_in; a.t_out = t_out;
  hipLaunchKernelGGL(mmnist_render_kernel, dim3(t_in + t_out, batch), dim3(256), 0, (hipStream_t)stream, a);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}
