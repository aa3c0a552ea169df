// warp.hip -- the VidODE flow/mask/warp decoder's tail: /root/reference/models/VidODE.py:119-140
// (get_warped_images :160-186 -- the image is warped by the predicted flow of step t, and the RESULT is what step t+1 warps:
// a chain that is sequential in T -- followed by the mask compositing of :137-138).  The reference runs, per predicted frame,
// a division, a permute, an add, F.grid_sample(bilinear, padding_mode="border", align_corners=False -- the default since
// torch 1.3), an unsqueeze/clone and, after the loop, a cat, a sigmoid and the compositing arithmetic: ~10 launches and 6 HBM
// round trips of the frame per step.  Here ONE launch walks the whole chain: a workgroup owns a sample, the current image lives
// in LDS (64x64 fp32 = 16 KiB per channel, double-buffered), every step reads its flow / mask logit / intermediate frame once
// and writes warped frame, mask and composited prediction once.  HBM-bound: (c + 3) floats read and (2c + 1) written per pixel
// and step -- 36 KiB per frame at c = 1, against the >200 KiB the op-by-op sequence moves.
//
// grid_sample arithmetic follows ATen's grid_sampler_2d (bilinear / border / align_corners=False):
//   ix = ((gx + 1) * W - 1) / 2, clamped to [0, W-1];  corners nw = floor, weights (x1 - ix)(y1 - iy) ...; out-of-range corners
//   (only x1 = W or y1 = H after the clamp, with weight 0) contribute nothing.
// The backward kernel is the transpose: gradients w.r.t. flow (through the bilinear weights, zero where the coordinate was
// clamped), mask logit, intermediate frame and the start image; the scatter into the previous image's gradient uses LDS float
// atomics (summation order within a step is not fixed: results are reproducible to rounding, not bitwise -- as torch's own
// grid_sample backward on a GPU).
#include "odehip_internal.h"

namespace odehip {

constexpr int kWarpThreads = 256;

struct WarpArgs {
  const float* po;      // pred_outputs (B, T, c + 3, H, W): [0:2] flow (x, y) in pixels, [2:2+c] intermediate frame, [2+c] mask logit
  const float* start;   // (B, c, H, W): the last observed frame
  const float* grid_x;  // [W] = torch.linspace(-1, 1, W) as the caller's torch computes it
  const float* grid_y;  // [H]
  float* pred_x;        // (B, T, c, H, W)
  float* warped;        // (B, T, c, H, W)
  float* masks;         // (B, T, 1, H, W) = sigmoid(logit)
  int T, c, H, W;
};

struct Bilinear {
  int x0, y0;
  float ix, iy;         // clamped source coordinates
  bool cx, cy;          // coordinate was clamped (its gradient is 0)
};

__device__ __forceinline__ Bilinear source_of(float flow_x, float flow_y, float gx0, float gy0, int H, int W) {
#pragma clang fp contract(off)
  Bilinear s;
  const float fx = flow_x / ((W - 1.0f) / 2.0f);   // VidODE.py:177
  const float fy = flow_y / ((H - 1.0f) / 2.0f);
  const float gx = gx0 + fx, gy = gy0 + fy;        // :179
  float ix = ((gx + 1.0f) * W - 1.0f) / 2.0f;      // grid_sampler_unnormalize, align_corners = False
  float iy = ((gy + 1.0f) * H - 1.0f) / 2.0f;
  s.cx = !(ix > 0.0f) || !(ix < (float)(W - 1));   // clip_coordinates_set_grad: gradient 0 at and beyond the border
  s.cy = !(iy > 0.0f) || !(iy < (float)(H - 1));
  ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
  iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
  s.ix = ix;
  s.iy = iy;
  s.x0 = (int)floorf(ix);
  s.y0 = (int)floorf(iy);
  return s;
}

__global__ __launch_bounds__(kWarpThreads) void warp_composite_kernel(const WarpArgs a) {
#pragma clang fp contract(off)
  extern __shared__ float lds[];
  const int b = blockIdx.x, HW = a.H * a.W, c = a.c;
  float* cur = lds;
  float* nxt = lds + (size_t)c * HW;
  for (int i = threadIdx.x; i < c * HW; i += kWarpThreads) cur[i] = a.start[(size_t)b * c * HW + i];
  __syncthreads();
  for (int t = 0; t < a.T; ++t) {
    const float* po = a.po + ((size_t)b * a.T + t) * (c + 3) * HW;
    float* px = a.pred_x + ((size_t)b * a.T + t) * c * HW;
    float* wp = a.warped + ((size_t)b * a.T + t) * c * HW;
    float* mk = a.masks + ((size_t)b * a.T + t) * HW;
    for (int p = threadIdx.x; p < HW; p += kWarpThreads) {
      const int y = p / a.W, x = p - y * a.W;
      const Bilinear s = source_of(po[p], po[HW + p], a.grid_x[x], a.grid_y[y], a.H, a.W);
      const int x1 = s.x0 + 1, y1 = s.y0 + 1;
      const float wnw = ((float)x1 - s.ix) * ((float)y1 - s.iy), wne = (s.ix - (float)s.x0) * ((float)y1 - s.iy);
      const float wsw = ((float)x1 - s.ix) * (s.iy - (float)s.y0), wse = (s.ix - (float)s.x0) * (s.iy - (float)s.y0);
      const bool inx1 = x1 < a.W, iny1 = y1 < a.H;
      const float m = 1.0f / (1.0f + expf(-po[(size_t)(2 + c) * HW + p]));
      mk[p] = m;
      for (int ch = 0; ch < c; ++ch) {
        const float* im = cur + (size_t)ch * HW;
        float v = im[s.y0 * a.W + s.x0] * wnw;
        if (inx1) v += im[s.y0 * a.W + x1] * wne;
        if (iny1) v += im[y1 * a.W + s.x0] * wsw;
        if (inx1 && iny1) v += im[y1 * a.W + x1] * wse;
        nxt[(size_t)ch * HW + p] = v;
        wp[(size_t)ch * HW + p] = v;
        const float inter = po[(size_t)(2 + ch) * HW + p];
        px[(size_t)ch * HW + p] = m * v + (1.0f - m) * inter;   // VidODE.py:137
      }
    }
    __syncthreads();
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
}

struct WarpBwdArgs {
  const float* po;
  const float* start;
  const float* warped;   // saved by the forward
  const float* grid_x;
  const float* grid_y;
  const float* g_pred_x; // (B, T, c, H, W)
  const float* g_warped; // (B, T, c, H, W) or null
  const float* g_masks;  // (B, T, 1, H, W) or null
  float* g_po;           // (B, T, c + 3, H, W)
  float* g_start;        // (B, c, H, W) or null
  int T, c, H, W;
};

__global__ __launch_bounds__(kWarpThreads) void warp_composite_bwd_kernel(const WarpBwdArgs a) {
#pragma clang fp contract(off)
  extern __shared__ float lds[];
  const int b = blockIdx.x, HW = a.H * a.W, c = a.c;
  float* gcur = lds;                       // gradient w.r.t. W_t that later steps have produced
  float* gprev = lds + (size_t)c * HW;     // gradient w.r.t. W_{t-1}, scattered by this step
  float* src = lds + (size_t)2 * c * HW;   // W_{t-1} (the image step t sampled from)
  for (int i = threadIdx.x; i < c * HW; i += kWarpThreads) gcur[i] = 0.0f;
  for (int t = a.T - 1; t >= 0; --t) {
    const float* sp = t == 0 ? a.start + (size_t)b * c * HW : a.warped + ((size_t)b * a.T + (t - 1)) * c * HW;
    for (int i = threadIdx.x; i < c * HW; i += kWarpThreads) {
      src[i] = sp[i];
      gprev[i] = 0.0f;
    }
    __syncthreads();
    const size_t frame = (size_t)b * a.T + t;
    const float* po = a.po + frame * (c + 3) * HW;
    float* gpo = a.g_po + frame * (c + 3) * HW;
    for (int p = threadIdx.x; p < HW; p += kWarpThreads) {
      const int y = p / a.W, x = p - y * a.W;
      const Bilinear s = source_of(po[p], po[HW + p], a.grid_x[x], a.grid_y[y], a.H, a.W);
      const int x1 = s.x0 + 1, y1 = s.y0 + 1;
      const float ax = (float)x1 - s.ix, bx = s.ix - (float)s.x0, ay = (float)y1 - s.iy, by = s.iy - (float)s.y0;
      const bool inx1 = x1 < a.W, iny1 = y1 < a.H;
      const float m = 1.0f / (1.0f + expf(-po[(size_t)(2 + c) * HW + p]));
      float gm = a.g_masks ? a.g_masks[frame * HW + p] : 0.0f, gix = 0.0f, giy = 0.0f;
      for (int ch = 0; ch < c; ++ch) {
        const size_t o = (size_t)ch * HW + p;
        const float gp = a.g_pred_x[frame * c * HW + o];
        const float wv = a.warped[frame * c * HW + o];
        const float inter = po[(size_t)(2 + ch) * HW + p];
        gm += gp * (wv - inter);
        gpo[(size_t)(2 + ch) * HW + p] = gp * (1.0f - m);
        const float g = gcur[o] + gp * m + (a.g_warped ? a.g_warped[frame * c * HW + o] : 0.0f);   // total gradient w.r.t. W_t[p]
        const float* im = src + (size_t)ch * HW;
        float* gi = gprev + (size_t)ch * HW;
        const float vnw = im[s.y0 * a.W + s.x0];
        const float vne = inx1 ? im[s.y0 * a.W + x1] : 0.0f;
        const float vsw = iny1 ? im[y1 * a.W + s.x0] : 0.0f;
        const float vse = (inx1 && iny1) ? im[y1 * a.W + x1] : 0.0f;
        atomicAdd(gi + s.y0 * a.W + s.x0, g * ax * ay);
        if (inx1) atomicAdd(gi + s.y0 * a.W + x1, g * bx * ay);
        if (iny1) atomicAdd(gi + y1 * a.W + s.x0, g * ax * by);
        if (inx1 && iny1) atomicAdd(gi + y1 * a.W + x1, g * bx * by);
        gix += g * ((vne - vnw) * ay + (vse - vsw) * by);
        giy += g * ((vsw - vnw) * ax + (vse - vne) * bx);
      }
      gpo[(size_t)(2 + c) * HW + p] = gm * m * (1.0f - m);
      // d ix / d flow_x = (W / 2) / ((W - 1) / 2), zero where the coordinate was clamped
      gpo[p] = s.cx ? 0.0f : gix * (0.5f * (float)a.W) / ((a.W - 1.0f) / 2.0f);
      gpo[HW + p] = s.cy ? 0.0f : giy * (0.5f * (float)a.H) / ((a.H - 1.0f) / 2.0f);
    }
    __syncthreads();
    float* tmp = gcur; gcur = gprev; gprev = tmp;
  }
  if (a.g_start)
    for (int i = threadIdx.x; i < c * HW; i += kWarpThreads) a.g_start[(size_t)b * c * HW + i] = gcur[i];
}

static int check_warp(const char* who, int batch, int n_times, int channels, int H, int W, size_t lds_images) {
  ODEHIP_REQUIRE(batch > 0 && n_times > 0, "%s: batch and n_times must be positive", who);
  ODEHIP_REQUIRE(channels >= 1 && channels <= 4, "%s: 1..4 image channels (got %d)", who, channels);
  ODEHIP_REQUIRE(H >= 2 && W >= 2 && (size_t)H * W * channels * lds_images * 4 <= 160 * 1024,
                 "%s: a %dx%dx%d image does not fit in LDS", who, channels, H, W);
  return ODEHIP_OK;
}

}  // namespace odehip

using namespace odehip;

extern "C" int odehip_warp_composite(const float* pred_outputs, const float* start_image, const float* grid_x, const float* grid_y,
                                     int batch, int n_times, int channels, int height, int width, float* pred_x, float* warped,
                                     float* masks, void* stream) {
  ODEHIP_REQUIRE(pred_outputs && start_image && grid_x && grid_y && pred_x && warped && masks, "warp_composite: null pointer argument");
  int rc = check_warp("warp_composite", batch, n_times, channels, height, width, 2);
  if (rc != ODEHIP_OK) return rc;
  WarpArgs a;
  a.po = pred_outputs; a.start = start_image; a.grid_x = grid_x; a.grid_y = grid_y; a.pred_x = pred_x; a.warped = warped; a.masks = masks;
  a.T = n_times; a.c = channels; a.H = height; a.W = width;
  const size_t lds = (size_t)2 * channels * height * width * 4;
  static size_t lds_set = 0;
  if (lds > 64 * 1024 && lds > lds_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)warp_composite_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    lds_set = 160 * 1024;
  }
  hipLaunchKernelGGL(warp_composite_kernel, dim3(batch), dim3(kWarpThreads), lds, (hipStream_t)stream, a);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

extern "C" int odehip_warp_composite_backward(const float* pred_outputs, const float* start_image, const float* warped,
                                              const float* grid_x, const float* grid_y, const float* grad_pred_x,
                                              const float* grad_warped, const float* grad_masks, int batch, int n_times, int channels,
                                              int height, int width, float* grad_pred_outputs, float* grad_start_image, void* stream) {
  ODEHIP_REQUIRE(pred_outputs && start_image && warped && grid_x && grid_y && grad_pred_x && grad_pred_outputs,
                 "warp_composite_backward: null pointer argument");
  int rc = check_warp("warp_composite_backward", batch, n_times, channels, height, width, 3);
  if (rc != ODEHIP_OK) return rc;
  WarpBwdArgs a;
  a.po = pred_outputs; a.start = start_image; a.warped = warped; a.grid_x = grid_x; a.grid_y = grid_y; a.g_pred_x = grad_pred_x;
  a.g_warped = grad_warped; a.g_masks = grad_masks; a.g_po = grad_pred_outputs; a.g_start = grad_start_image;
  a.T = n_times; a.c = channels; a.H = height; a.W = width;
  const size_t lds = (size_t)3 * channels * height * width * 4;
  static size_t lds_set = 0;
  if (lds > 64 * 1024 && lds > lds_set) {
    ODEHIP_CHECK_HIP(hipFuncSetAttribute((const void*)warp_composite_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    lds_set = 160 * 1024;
  }
  hipLaunchKernelGGL(warp_composite_bwd_kernel, dim3(batch), dim3(kWarpThreads), lds, (hipStream_t)stream, a);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}
