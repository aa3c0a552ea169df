// Internal declarations shared by the HIP translation units of libodecgru_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/odecgru_hip.h"

namespace odehip {

constexpr int kHW = 16;           // latent maps are 16x16 (models/ODEConvGRU.py:18-20)
constexpr int kPix = kHW * kHW;   // 256 pixels
constexpr int kQuadBytes = kPix * 16;  // one channel-quad plane of the Q4 layout: 256 px * 4 ch * 4 B

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define ODEHIP_CHECK_HIP(expr)                                   \
  do {                                                           \
    hipError_t e__ = (expr);                                     \
    if (e__ != hipSuccess) return ::odehip::hip_fail(e__, #expr); \
  } while (0)

#define ODEHIP_REQUIRE(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      ::odehip::set_error(__VA_ARGS__);      \
      return ODEHIP_EINVAL;                  \
    }                                        \
  } while (0)

// ---- stage-combine epilogue of the LAST conv of f (Runge-Kutta bookkeeping fused in) -------
// k_cur = conv output.  Optional outputs, all Q4 unless stated:
//   k_out             <- k_cur
//   out1              <- y + h * ( sum_{j<n_prev} c1[j]*k_prev[j] + c1[n_prev]*k_cur )
//   out2 (+out2_nchw) <- y + h * ( sum c2 ... )            (used for the step result y_{n+1})
// h = *h_ptr (device scalar; the step size of this interval), so one captured graph serves any t.
struct CombineArgs {
  const float* y;
  const float* k_prev[ODEHIP_MAX_STAGES];
  float* k_out;
  float* out1;
  float* out2;
  float* out2_nchw;
  const float* h_ptr;
  int n_prev;
  float c1[ODEHIP_MAX_STAGES + 1];
  float c2[ODEHIP_MAX_STAGES + 1];
  float k_scale;  // k_cur is multiplied by this first (-1 for backwards=True / reversed time)
  // adaptive error norm (dopri5, last stage): err = h * (sum_j ce[j]*k_prev[j] + ce[n_prev]*k_cur),
  // tol = atol + rtol*max(|y|, |err_y1|); every wave writes sum((err/tol)^2) of its 32x32 tile to
  // err_partials[4*workgroup + wave] (no atomics: the controller adds them in a fixed order)
  const float* err_y1;
  float* err_partials;
  float ce[ODEHIP_MAX_STAGES + 1];
  float rtol, atol;
};

// ---- epilogues of the input-gradient (dgrad) convolutions of the backward sweep
//   combine == 2: dst = scale * acc * (mask_src > 0)            (ReLU backward fused; scale = sc_c + sc_h*h)
//   combine == 3: gx = acc; up to 4 targets  out_t = (a_c+a_h*h)*srcA_t + (b_c+b_h*h)*srcB_t + (g_c+g_h*h)*gx
//                 (the reverse Runge-Kutta bookkeeping: gy += gx, gk_j += c*h*gx, next interval's seed, ...)
struct BwdTarget {
  float* out;
  const float* srcA;
  const float* srcB;
  float a_c, a_h, b_c, b_h, g_c, g_h;
};
struct BwdArgs {
  const float* mask_src;
  float sc_c, sc_h;
  int n_targets;
  BwdTarget tgt[4];
  const float* h_ptr;
};

struct ConvArgs {
  const float* src1;
  const float* src2;
  const float* w_packed;
  const float* w_wino;   // Winograd-transformed weights (pack_winograd) or null
  const void* w_bf16;    // bf16 A-operand image (pack_conv_weight_bf16) or null: non-null selects bf16 compute for 3x3 layers
  const float* bias;
  float* dst;
  int q1;      // channel quads in src1
  int qin;     // total input quads
  int qout;    // output quads (cout / 4)
  int batch;
  int relu;
  int combine; // 0: plain store (+relu); 1: CombineArgs epilogue; 2/3: BwdArgs epilogues
  int debug;   // diagnostic ablation bits (tools/conv_microbench.py): 1 skip DMA, 2 skip MFMA, 4 skip epilogue
  const int* skip;          // if non-null and *skip != 0 the kernel does nothing (adaptive solver already done)
  unsigned long long* dbg;  // debug & 8: per-workgroup stamps (8 x u64 per workgroup)
  CombineArgs cmb;
  BwdArgs bwd;
};

int launch_conv(const ConvArgs& a, int ks, hipStream_t stream);
int launch_wino(const ConvArgs& a, hipStream_t stream);
int launch_bf16(const ConvArgs& a, hipStream_t stream);
extern int g_debug_flags;
extern unsigned long long* g_debug_buf;

}  // namespace odehip
