// adam.hip -- one-launch Adam step over every parameter tensor of the model (the optimizer step of
// train_test.py:24,205 `optim.Adam(model.parameters(), lr=opt.lr)`).  HBM-bound elementwise work: 4 reads + 3 writes of 4 B per
// parameter; the ODEConvGRU model has 1.04 M parameters in 40 tensors, so the point is ONE launch instead of a launch per
// tensor and op.  Arithmetic in torch.optim.Adam's order (amsgrad off):
//   g += wd*p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g*g;  p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
#include <math.h>
#include <string.h>

#include "odehip_internal.h"

namespace odehip {

constexpr int kAdamChunk = 24;  // tensors per launch (pointers travel as kernel arguments)
struct AdamTable {
  float* p[kAdamChunk];
  const float* g[kAdamChunk];
  float* m[kAdamChunk];
  float* v[kAdamChunk];
  long long n[kAdamChunk];
};

__global__ __launch_bounds__(256) void adam_kernel(AdamTable t, float lr_over_bc1, float inv_sqrt_bc2, float b1, float b2, float eps,
                                                   float wd) {
  const int k = blockIdx.y;
  float* __restrict__ p = t.p[k];
  const float* __restrict__ g = t.g[k];
  float* __restrict__ m = t.m[k];
  float* __restrict__ v = t.v[k];
  const long long n = t.n[k];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float pv = p[i];
    const float gv = g[i] + wd * pv;
    const float mv = b1 * m[i] + (1.0f - b1) * gv;
    const float vv = b2 * v[i] + (1.0f - b2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    p[i] = pv - lr_over_bc1 * (mv / (sqrtf(vv) * inv_sqrt_bc2 + eps));
  }
}

}  // namespace odehip

using namespace odehip;

extern "C" int odehip_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                const long long* numel, int n_tensors, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int step, void* stream) {
  ODEHIP_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel && n_tensors >= 0, "adam_step: null pointer");
  ODEHIP_REQUIRE(step >= 1, "adam_step: step counts from 1 (got %d)", step);
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float lr_over_bc1 = (float)((double)lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  for (int o = 0; o < n_tensors; o += kAdamChunk) {
    AdamTable t;
    memset(&t, 0, sizeof(t));
    const int m = n_tensors - o < kAdamChunk ? n_tensors - o : kAdamChunk;
    long long nmax = 0;
    for (int i = 0; i < m; ++i) {
      ODEHIP_REQUIRE(params[o + i] && grads[o + i] && exp_avg[o + i] && exp_avg_sq[o + i] && numel[o + i] >= 0,
                     "adam_step: tensor %d has a null pointer", o + i);
      t.p[i] = params[o + i];
      t.g[i] = grads[o + i];
      t.m[i] = exp_avg[o + i];
      t.v[i] = exp_avg_sq[o + i];
      t.n[i] = numel[o + i];
      nmax = numel[o + i] > nmax ? numel[o + i] : nmax;
    }
    int gx = (int)((nmax + 255) / 256);
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    hipLaunchKernelGGL(adam_kernel, dim3(gx, m), dim3(256), 0, (hipStream_t)stream, t, lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps,
                       weight_decay);
  }
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}
