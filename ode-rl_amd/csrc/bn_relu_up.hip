// bn_relu_up.hip -- BatchNorm2d -> ReLU (-> bilinear x2 upsampling) of VidODE's flow decoder as ONE pass over the convolution's output
// (/root/reference/models/VidODE.py:34-36: [Upsample, Conv2d, BatchNorm2d, ReLU] x 2 + Conv2d, applied once per predicted frame by
// get_flowmaps :143-158; in train() mode the statistics are those of the call, i.e. of one frame's batch).
//
// Unfused (round 4 profile, B = 64, `profiles/r04_vidode_train_kernel_stats.txt`): MIOpenBatchNormFwdTrainSpatial 221 us + a ReLU launch +
// the upsampling per level and frame, each a full read and write of a 33 / 67 MB tensor; backward: MIOpenBatchNormBwdSpatial 116 us + ReLU
// backward + upsampling backward.  Here:
//   forward   bn_stats_kernel (one read: per-channel sum and sum of squares in fp64, fixed-order two-level reduction) -> bn_finalize_kernel
//             (mean, biased variance, inverse std, scale / shift; running statistics updated as nn.BatchNorm2d does: momentum, unbiased
//             variance) -> bn_relu_up_kernel (read x once, write relu(x * scale + shift), upsampled x2 with ATen's bilinear arithmetic
//             when the next layer is the decoder's Upsample);
//   backward  bn_bwd_reduce_kernel: g_pre = [upsampling transposed as a gather](grad_out) * (pre-activation > 0), written once, with the
//             per-channel sums of g_pre and g_pre * xhat -> finalize -> bn_bwd_apply_kernel: dx = scale * (g_pre - mean(g_pre) - xhat *
//             mean(g_pre * xhat)) (train) or scale * g_pre (eval).
// All HBM-bound; every reduction has a fixed order (bitwise reproducible).  NCHW fp32 as torch holds the tensors.
#include "odehip_internal.h"

namespace odehip {

constexpr int kBnSplit = 64;   // partial sums per channel

struct UpSrc {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ UpSrc up_src_of(int dst, int n) {   // = src_of of upsample.hip (ATen, align_corners = false, scale 1/2)
#pragma clang fp contract(off)
  float src = 0.5f * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  UpSrc r;
  r.i0 = (int)src;
  r.i1 = r.i0 + (r.i0 < n - 1 ? 1 : 0);
  r.l1 = src - (float)r.i0;
  r.l0 = 1.0f - r.l1;
  return r;
}

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// grid (C, kBnSplit): block (c, s) sums the images n = s, s + kBnSplit, ... of channel c
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int N, int C, int HW, double* __restrict__ part) {
  __shared__ double sh[256];
  const int c = blockIdx.x, s = blockIdx.y;
  double a = 0.0, b = 0.0;
  for (int n = s; n < N; n += kBnSplit) {
    const float* p = x + ((size_t)n * C + c) * HW;
    for (int i = threadIdx.x * 4; i < HW; i += 256 * 4) {   // HW % 4 == 0 (checked by the caller)
      const float4 v = *(const float4*)(p + i);
      a += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      b += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  }
  a = block_sum_d(a, sh);
  b = block_sum_d(b, sh);
  if (threadIdx.x == 0) {
    part[((size_t)c * kBnSplit + s) * 2] = a;
    part[((size_t)c * kBnSplit + s) * 2 + 1] = b;
  }
}

// one thread per channel: statistics of the call, the affine map of the normalisation, nn.BatchNorm2d's running statistics
__global__ void bn_finalize_kernel(const double* __restrict__ part, int C, double count, float eps, float momentum, const float* __restrict__ weight,
                                   const float* __restrict__ bias, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < kBnSplit; ++s) {
    a += part[((size_t)c * kBnSplit + s) * 2];
    b += part[((size_t)c * kBnSplit + s) * 2 + 1];
  }
  const double mean = a / count;
  double var = b / count - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  mean_out[c] = (float)mean;
  invstd_out[c] = invstd;
  const float g = weight ? weight[c] : 1.0f, be = bias ? bias[c] : 0.0f;
  scale[c] = g * invstd;
  shift[c] = be - (float)mean * g * invstd;
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// eval mode: the affine map from the running statistics
__global__ void bn_eval_affine_kernel(int C, float eps, const float* __restrict__ weight, const float* __restrict__ bias,
                                      const float* __restrict__ running_mean, const float* __restrict__ running_var, float* __restrict__ mean_out,
                                      float* __restrict__ invstd_out, float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(running_var[c] + eps);
  const float g = weight ? weight[c] : 1.0f, be = bias ? bias[c] : 0.0f;
  mean_out[c] = running_mean[c];
  invstd_out[c] = invstd;
  scale[c] = g * invstd;
  shift[c] = be - running_mean[c] * g * invstd;
}

__device__ __forceinline__ float bn_relu(float v, float a, float b) {
  const float y = __builtin_fmaf(v, a, b);
  return y < 0.0f ? 0.0f : y;   // (NaN passes through, as torch.relu)
}

// out = relu(x * scale[c] + shift[c]); UP: upsampled x2 (one thread = four consecutive output pixels of a row)
template <bool UP>
__global__ __launch_bounds__(256) void bn_relu_up_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                         float* __restrict__ out, long long planes, int C, int H, int W) {
#pragma clang fp contract(off)
  if (!UP) {
    const long long total = planes * (long long)H * W / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
      const int c = (int)((i * 4 / ((long long)H * W)) % C);
      const float a = scale[c], b = shift[c];
      const float4 v = ((const float4*)x)[i];
      ((float4*)out)[i] = float4{bn_relu(v.x, a, b), bn_relu(v.y, a, b), bn_relu(v.z, a, b), bn_relu(v.w, a, b)};
    }
    return;
  }
  const int W2 = 2 * W, H2 = 2 * H, q_per_row = W2 / 4;
  const long long total = planes * (long long)H2 * q_per_row;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int q = (int)(i % q_per_row);
    const long long r = i / q_per_row;
    const int oy = (int)(r % H2);
    const long long pl = r / H2;
    const int c = (int)(pl % C);
    const float a = scale[c], b = shift[c];
    const UpSrc sy = up_src_of(oy, H);
    const float* r0 = x + (pl * H + sy.i0) * (long long)W;
    const float* r1 = x + (pl * H + sy.i1) * (long long)W;
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const UpSrc sx = up_src_of(4 * q + k, W);
      const float top = sx.l0 * bn_relu(r0[sx.i0], a, b) + sx.l1 * bn_relu(r0[sx.i1], a, b);
      const float bot = sx.l0 * bn_relu(r1[sx.i0], a, b) + sx.l1 * bn_relu(r1[sx.i1], a, b);
      o[k] = sy.l0 * top + sy.l1 * bot;
    }
    *(float4*)(out + (pl * H2 + oy) * (long long)W2 + 4 * q) = float4{o[0], o[1], o[2], o[3]};
  }
}

// gradient w.r.t. the ReLU's output at input pixel (y, x) of plane pl: the upsampling's transpose as a gather (upsample.hip), or g itself
template <bool UP>
__device__ __forceinline__ float grad_at(const float* __restrict__ g, long long pl, int y, int x, int H, int W) {
#pragma clang fp contract(off)
  if (!UP) return g[(pl * H + y) * (long long)W + x];
  const int W2 = 2 * W, H2 = 2 * H;
  // weights of the output rows 2y-1 .. 2y+2 on input row y: {1/4, 3/4, 3/4, 1/4}; at the borders the clamped stencils fold onto the edge
  // row (y = 0: rows 0, 1, 2 -> {1, 3/4, 1/4}; y = H-1: rows 2H-3 .. 2H-1 -> {1/4, 3/4, 1}) -- exactly what src_of() yields (the weights
  // are exact binary fractions), without its float -> int conversions per tap
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
  return acc;
}

// grid (C, kBnSplit): g_pre = grad * (pre-activation > 0) written once; partial sums of g_pre and g_pre * xhat per channel
template <bool UP>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* __restrict__ g_pre, int N, int C, int H, int W,
                                                            double* __restrict__ part) {
  __shared__ double sh[256];
  const int c = blockIdx.x, s = blockIdx.y, HW = H * W;
  const float a = scale[c], b = shift[c], mu = mean[c], is = invstd[c];
  double s1 = 0.0, s2 = 0.0;
  for (int n = s; n < N; n += kBnSplit) {
    const long long pl = (long long)n * C + c;
    for (int i = threadIdx.x; i < HW; i += 256) {
      const int y = i / W, xx = i - y * W;
      const float xv = x[pl * HW + i];
      const float pre = __builtin_fmaf(xv, a, b);
      const float gv = pre > 0.0f ? grad_at<UP>(g, pl, y, xx, H, W) : 0.0f;
      g_pre[pl * HW + i] = gv;
      s1 += (double)gv;
      s2 += (double)gv * (double)((xv - mu) * is);
    }
  }
  s1 = block_sum_d(s1, sh);
  s2 = block_sum_d(s2, sh);
  if (threadIdx.x == 0) {
    part[((size_t)c * kBnSplit + s) * 2] = s1;
    part[((size_t)c * kBnSplit + s) * 2 + 1] = s2;
  }
}

// one thread per channel: d beta = sum g_pre, d gamma = sum g_pre * xhat (+ the means the apply pass needs)
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ part, int C, double count, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ m1, float* __restrict__ m2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < kBnSplit; ++s) {
    a += part[((size_t)c * kBnSplit + s) * 2];
    b += part[((size_t)c * kBnSplit + s) * 2 + 1];
  }
  dbeta[c] = (float)a;
  dgamma[c] = (float)b;
  m1[c] = (float)(a / count);
  m2[c] = (float)(b / count);
}

// dx = scale * (g_pre - m1 - xhat * m2) (train) | scale * g_pre (eval: m1 = m2 = null)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g_pre, const float* __restrict__ x, const float* __restrict__ scale,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ m1, const float* __restrict__ m2, float* __restrict__ dx,
                                                           long long planes, int C, int HW) {
  const long long total = planes * (long long)HW / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)((i * 4 / HW) % C);
    const float a = scale[c];
    const float4 gv = ((const float4*)g_pre)[i];
    float4 o;
    if (m1) {
      const float mu = mean[c], is = invstd[c], a1 = m1[c], a2 = m2[c];
      const float4 xv = ((const float4*)x)[i];
      o.x = a * (gv.x - a1 - (xv.x - mu) * is * a2);
      o.y = a * (gv.y - a1 - (xv.y - mu) * is * a2);
      o.z = a * (gv.z - a1 - (xv.z - mu) * is * a2);
      o.w = a * (gv.w - a1 - (xv.w - mu) * is * a2);
    } else {
      o = float4{a * gv.x, a * gv.y, a * gv.z, a * gv.w};
    }
    ((float4*)dx)[i] = o;
  }
}

static unsigned grid_for(long long work) {
  const long long blocks = (work + 255) / 256;
  return (unsigned)(blocks < 65536 ? (blocks < 1 ? 1 : blocks) : 65536);
}

}  // namespace odehip

using namespace odehip;

extern "C" size_t odehip_bn_workspace_bytes(int channels) { return (size_t)channels * kBnSplit * 2 * sizeof(double); }

extern "C" int odehip_bn_relu_up2x_forward(const float* x, int batch, int channels, int height, int width, const float* weight, const float* bias,
                                           float* running_mean, float* running_var, int training, float momentum, float eps, int upsample,
                                           float* out, float* mean_out, float* invstd_out, float* scale_out, float* shift_out, void* workspace,
                                           size_t workspace_bytes, void* stream_) {
  ODEHIP_REQUIRE(x && out && mean_out && invstd_out && scale_out && shift_out, "bn_relu_up2x_forward: null pointer");
  ODEHIP_REQUIRE(batch > 0 && channels > 0 && height > 0 && width > 0 && width % 4 == 0,
                 "bn_relu_up2x_forward: bad shape (%d, %d, %d, %d); the width must be a multiple of 4", batch, channels, height, width);
  ODEHIP_REQUIRE(training || (running_mean && running_var), "bn_relu_up2x_forward: eval mode needs the running statistics");
  hipStream_t stream = (hipStream_t)stream_;
  const int HW = height * width;
  if (training) {
    ODEHIP_REQUIRE(workspace && workspace_bytes >= odehip_bn_workspace_bytes(channels), "bn_relu_up2x_forward: workspace too small");
    hipLaunchKernelGGL(bn_stats_kernel, dim3(channels, kBnSplit), dim3(256), 0, stream, x, batch, channels, HW, (double*)workspace);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((channels + 63) / 64), dim3(64), 0, stream, (const double*)workspace, channels,
                       (double)batch * HW, eps, momentum, weight, bias, running_mean, running_var, mean_out, invstd_out, scale_out, shift_out);
  } else {
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((channels + 63) / 64), dim3(64), 0, stream, channels, eps, weight, bias, running_mean, running_var,
                       mean_out, invstd_out, scale_out, shift_out);
  }
  const long long planes = (long long)batch * channels;
  if (upsample) {
    hipLaunchKernelGGL((bn_relu_up_kernel<true>), dim3(grid_for(planes * 2LL * height * (width / 2))), dim3(256), 0, stream, x, scale_out, shift_out, out,
                       planes, channels, height, width);
  } else {
    hipLaunchKernelGGL((bn_relu_up_kernel<false>), dim3(grid_for(planes * (long long)HW / 4)), dim3(256), 0, stream, x, scale_out, shift_out, out, planes,
                       channels, height, width);
  }
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

extern "C" int odehip_bn_relu_up2x_backward(const float* grad_out, const float* x, int batch, int channels, int height, int width,
                                            const float* mean, const float* invstd, const float* scale, const float* shift, int training,
                                            int upsample, float* grad_x, float* grad_weight, float* grad_bias, float* g_pre, void* workspace,
                                            size_t workspace_bytes, void* stream_) {
  ODEHIP_REQUIRE(grad_out && x && mean && invstd && scale && shift && grad_x && grad_weight && grad_bias && g_pre, "bn_relu_up2x_backward: null pointer");
  ODEHIP_REQUIRE(batch > 0 && channels > 0 && height > 0 && width > 0 && width % 4 == 0,
                 "bn_relu_up2x_backward: bad shape (%d, %d, %d, %d); the width must be a multiple of 4", batch, channels, height, width);
  ODEHIP_REQUIRE(workspace && workspace_bytes >= odehip_bn_workspace_bytes(channels) + 2 * (size_t)channels * sizeof(float),
                 "bn_relu_up2x_backward: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  const int HW = height * width;
  double* part = (double*)workspace;
  float* m1 = (float*)((char*)workspace + odehip_bn_workspace_bytes(channels));
  float* m2 = m1 + channels;
  if (upsample)
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<true>), dim3(channels, kBnSplit), dim3(256), 0, stream, grad_out, x, scale, shift, mean, invstd, g_pre, batch,
                       channels, height, width, part);
  else
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<false>), dim3(channels, kBnSplit), dim3(256), 0, stream, grad_out, x, scale, shift, mean, invstd, g_pre, batch,
                       channels, height, width, part);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((channels + 63) / 64), dim3(64), 0, stream, (const double*)part, channels, (double)batch * HW, grad_weight,
                     grad_bias, m1, m2);
  const long long planes = (long long)batch * channels;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(planes * (long long)HW / 4)), dim3(256), 0, stream, g_pre, x, scale, mean, invstd,
                     training ? m1 : (const float*)nullptr, training ? m2 : (const float*)nullptr, grad_x, planes, channels, HW);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}
