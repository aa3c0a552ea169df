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
  // Summation order of the stage combine.  0 (fixed-grid drivers): s = c[n]*k_cur, then += c[j]*k_prev[j].  1 (the adaptive
  // solver's drivers): explicit fmas over k_prev[0..n-1] FIRST, k_cur last, out = fma(s, h, y) -- the part that does not depend
  // on the conv output can then be formed while the matrix cores still work on the layer (the persistent walk's adaptive kernel
  // holds three partial sums instead of up to six earlier stages), and every kernel that evaluates the expression rounds it the
  // same way (no contraction left to the compiler), so the walk stays bit-identical to one launch per layer.
  int order;
  const float* c_dev;   // elementwise rows only: if non-null, c1[j] is read from this DEVICE array instead (coefficients a device-side
                        // controller computed, e.g. the dense-output weights of an accepted step)
};
// ---- elementwise rows (ConvArgs::combine == 4): no convolution.  Uses the CombineArgs fields:
//   out1 = (y ? y : 0) + sum_{j < n_prev} (c1[j] * hs) * k_prev[j]        hs = *h_ptr (1 if null)
//   out2 = (y ? y : 0) + sum_{j < n_prev} (c2[j] * hs) * k_prev[j]        (only if out2 != null)
// evaluated as explicit fmas in j order.  In a persistent walk a lane only touches the elements the conv epilogues give it
// (channel quad Q of its 2x2 output tile), i.e. data it wrote itself: no wait, no barrier; the row then announces itself like a layer.

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
  int combine; // 0: plain store (+relu); 1: CombineArgs epilogue; 2/3: BwdArgs epilogues; 4: elementwise row (no conv; see above);
               // 5: norm row (adaptive walk only): err_partials <- per-wave sums of ((k_prev[0] - k_prev[1]) / (atol + |y| rtol))^2
  int debug;   // diagnostic ablation bits (tools/conv_microbench.py): 1 skip DMA, 2 skip MFMA, 4 skip epilogue
  int h_by_value;  // persistent tables of fixed-grid drivers: cmb.atol holds the step size itself (read instead of *h_ptr)
  int dep_back;    // adaptive walk: 1 = this row does NOT depend on the row right in front of it (two independent chains woven into
                   // one table, e.g. the input-gradient chain of a stage and the forward chain of the next): its producers only wait
                   // for the row before that, i.e. they load and transform its input while the consumers still multiply the other
                   // chain's row -- the partner hand-off leaves the critical path
  const int* skip;          // if non-null and *skip != 0 the kernel does nothing (adaptive solver already done)
  unsigned long long* dbg;  // debug & 8: per-workgroup stamps (8 x u64 per workgroup)
  CombineArgs cmb;
  BwdArgs bwd;
};

int launch_conv(const ConvArgs& a, int ks, hipStream_t stream);
// While non-null (set by a driver on its own thread), launch_conv() appends its argument here INSTEAD of launching: the driver
// collects the layer sequence of a whole trajectory and hands it to the persistent kernel (conv_wino.hip).
struct ConvRecorder {
  ConvArgs* items;
  int count, capacity;
};
extern thread_local ConvRecorder* g_conv_recorder;
int launch_wino_persist_small(const ConvArgs* items, int n_layers, int batch, unsigned* done, unsigned* xcc_of, unsigned epoch,
                              unsigned* host_err_dev, int grid, hipStream_t stream);
int launch_wino_persist(const ConvArgs* table_dev, int n_layers, int batch, unsigned* done, unsigned* xcc_of, unsigned* host_err_dev,
                        float* out_nchw, int grid, hipStream_t stream, bool wide = false,  // wide: the table has 128-channel layers
                        bool adaptive = false, const int* n_layers_ptr = nullptr,           // adaptive: wino_persist_d_kernel (order-1
                        const unsigned long long* reloc = nullptr);                         // combines, elementwise rows, h on the device;
                                                                                            // {row0, rows} and relocation bases on the device)
int launch_ew_row(const ConvArgs& a, hipStream_t stream);
// forward tables of a 64-channel stack at batch <= 16: sixteen workgroups per sample (conv_wino.hip, wino_persist16_kernel)
int launch_wino_persist16(const ConvArgs* table_dev, int n_layers, int batch, unsigned* done, unsigned* xcc_of, unsigned* host_err_dev,
                          float* out_nchw, hipStream_t stream, const int* n_layers_ptr = nullptr, const unsigned long long* reloc = nullptr);
// error-norm / norm-row partials per sample that a walk of `batch` samples writes: 64 on the sixteen-workgroup walk (batch <= 16, walk
// available and not switched off), else the per-layer kernels' 16 (64-channel stacks).  Decided BEFORE the rows run: a controller must
// know how many partials to add.
int persist_partials_per_sample(int batch);   // an elementwise row (combine == 4) as an ordinary launch
int launch_wino(const ConvArgs& a, hipStream_t stream);
int launch_wino5(const ConvArgs& a, hipStream_t stream);  // 5x5 layers with a.w_wino (conv_wino5.hip); 1 = no such form, run the direct kernel
int launch_bf16(const ConvArgs& a, hipStream_t stream);
int launch_bf16_5x5(const ConvArgs& a, hipStream_t stream);

// whole-stack bf16 launch (fstack_bf16.hip)
struct FusedArgs {
  const float* x;                          // stage input (B,64) Q4 fp32
  const void* w_fused;                     // [layer][tap][cb 4][mb 2][lane 64][8] bf16, execution order
  const float* bias[ODEHIP_MAX_LAYERS];    // per executed layer (null: none)
  float* store[ODEHIP_MAX_LAYERS];         // fp32 Q4 output of executed layer e < n-1 (null: not stored)
  const float* mask[ODEHIP_MAX_LAYERS];    // e < n-1: null -> ReLU, else output *= (mask > 0)
  int n_layers;
  ConvArgs last;                           // epilogue of the last executed layer (qout = 16)
};
int launch_fstack_bf16(const FusedArgs& fa, int batch, hipStream_t stream);
// whole fixed-grid trajectory of a fused-bf16 64-channel stack in one launch (forward only, nothing saved)
int launch_ftraj_bf16(const odehip_convstack* f, int method, const float* z0_nchw, float* out_nchw, const float* hdev, int n_times,
                      int batch, int negate, hipStream_t stream);
extern int g_debug_flags;
extern unsigned long long* g_debug_buf;


// ---- shared host-side helpers (api.hip, wgrad.hip) ---------------------------------------------------------------------------
static inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }

int check_stack(const odehip_convstack* f);
// every layer 64 -> 64, 3x3: the stacks the adaptive persistent walk takes (their adaptive drivers write order-1 stage combines)
inline bool all_64(const odehip_convstack* f) {
  if (f->ks != 3) return false;
  for (int l = 0; l <= f->n_convs; ++l)
    if (f->channels[l] != 64) return false;
  return true;
}
// prologue of a fixed-grid trajectory in one launch: NCHW -> Q4 + verbatim copy (solution[0] = y0), n_h <= 64 step sizes into hdev,
// n_zero words zeroed (the persistent launch's flag area; may be null)
int traj_prologue(const float* src, float* dst_q4, float* copy_nchw, int batch, int channels, const float* h_host, int n_h, float* hdev,
                  unsigned* zero_words, int n_zero, hipStream_t stream);
int max_hidden(const odehip_convstack* f);
int upload_floats(float* dst, const float* src, int n, hipStream_t stream);  // scalars travel as kernel arguments (async)
// f(x) with the stage combine fused into the last conv; `hidden` (n_convs-1 buffers) keeps the ReLU outputs for a backward pass
int enqueue_f(const odehip_convstack* f, const float* x_q4, int batch, float* ping, float* pong, const CombineArgs* cmb,
              float* plain_dst, const int* skip, hipStream_t stream);
int enqueue_f_saving(const odehip_convstack* f, const float* x_q4, int batch, float* const* hidden, float* ping, float* pong,
                     const CombineArgs* cmb, float* plain_dst, const int* skip, hipStream_t stream);

// Input-gradient chain of one evaluation of f: gp[n_convs-1] holds the gradient w.r.t. f's output; for l = n_convs-1 .. 1 the
// gradient w.r.t. conv l's input is masked with hidden[l-1] (ReLU) and written to gp[l-1]; conv 0 ends in `last` (combine 3
// with last.bwd, or combine 1 with last.cmb; its src/weights/shape fields are filled here).  One fused bf16 launch when
// f_dgrad carries a fused image, else one launch per layer.
int enqueue_dgrad_chain(const odehip_convstack* f, const odehip_convstack* f_dgrad, int batch, float* const* gp,
                        const float* const* hidden, const ConvArgs& last, hipStream_t stream);

// one (gradient, activation, weight) triple of the batched weight-gradient kernels (wgrad.hip)
struct WgradPair {
  const float* g;  // gradient w.r.t. the layer's output (Q4)
  const float* a;  // the layer's input (Q4)
  float scale;     // weight of this evaluation in the sum (1 for discretise-then-optimise; dt*b_s for the adjoint)
  float pad_[3];
};
// bf16 = true: operands rounded to bf16 (fp32 accumulation), for stacks running in bf16 compute mode
// floats per workgroup slab of the weight-gradient kernels: the Winograd-domain tile [16 positions][64][64] + 64 bias sums is the
// largest; every slab region holds batch * esplit of them PLUS ONE (the fixed-order sum before the final G^T . G)
constexpr int kWgradSlabFloats = 16 * 64 * 64 + 64;
// Workgroups per sample of a batched weight-gradient launch (each walks every esplit-th evaluation): 4 at the batches the chip is
// full anyway; small batches split the evaluations further so that the launch still covers the CUs (B = 4: 16 workgroups walked
// nine evaluations each -- 215 us per layer, the time of a B = 64 launch).  At most 256 slabs, as at B = 64.
inline int wgrad_esplit(int batch, int n_eval) {
  int e = batch > 0 ? 256 / batch : 4;
  if (e > n_eval) e = n_eval;
  return e < 4 ? 4 : e;
}
inline int wgrad_esplit_max(int batch) { return batch >= 64 ? 4 : (256 / batch < 4 ? 4 : 256 / batch); }
int launch_wgrad_wino(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, float* db, int cout,
                      int cin, hipStream_t stream);  // wgrad_wino.hip; 1 = switched off
int launch_wgrad(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, float* db, int cout,
                 int cin, hipStream_t stream, bool bf16 = false);
int launch_wgrad_q4h(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, int accumulate,
                     hipStream_t stream);
int launch_ftraj_bf16_saving(const odehip_convstack* f, const float* z0_nchw, float* out_nchw, const float* hdev, int n_times, int batch,
                             void* save_x, size_t stride_x, void* save_h, size_t stride_h_eval, size_t stride_h_layer,
                             hipStream_t stream);
// reverse sweep over the intervals n_hi-1 .. n_lo (n_hi == n_times-1: starts from grad_out[T-1], else from state_g / state_seed as the
// previous segment left them; n_lo == 0: writes grad_z0, else leaves its state in state_g / state_seed); bias_part: [B][NL][64]
int launch_btraj_bf16_rk4(const odehip_convstack* f_dgrad, const float* grad_out_nchw, float* grad_z0_nchw, const float* hdev, int n_times,
                          int batch, int n_lo, int n_hi, float* state_g, float* state_seed, const void* save_h, size_t stride_h_eval,
                          size_t stride_h_layer, void* save_g, size_t stride_g_eval, size_t stride_g_layer, float* bias_part,
                          hipStream_t stream);
// grad_b[l][ch] = sum over segments and samples of bias_part[seg][b][l][ch], in order
int launch_bias_reduce(const float* bias_part, int n_parts, int n_layers, float* const* grad_b, hipStream_t stream);
int launch_wgrad_tile(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, float* db, int ks,
                      int cin_total, int co0, int ci0, int g_quads, int g_quad0, int a_quads, int a_quad0, bool write_bias,
                      hipStream_t stream);

// a 64 x 64 tile of a 5x5 weight gradient in the Winograd F(2x2,5x5) domain (wgrad_wino5.hip); 1 = switched off
// sum[i] = sum over k < n_slabs of slabs[k * stride + i], i < n_vals, in a fixed order (16 interleaved partial sums of 4 chains each,
// 16-byte loads: bitwise reproducible).  stride and n_vals multiples of 4 floats, slabs and sum 16-byte aligned.  (wgrad.hip)
void launch_slab_sum4(const float* slabs, int n_slabs, int stride, int n_vals, float* sum, hipStream_t stream);
int launch_wgrad_wino5(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, float* db, int cin_total,
                       int co0, int ci0, int g_quads, int g_quad0, int a_quads, int a_quad0, bool write_bias, hipStream_t stream);
int launch_wgrad_tile_bf16_5x5(const WgradPair* table_dev, int n_eval, int batch, int esplit, float* slabs, float* dw, float* db,
                               int cin_total, int co0, int ci0, int g_quads, int g_quad0, int a_quads, int a_quad0, bool write_bias,
                               hipStream_t stream);

// Dormand-Prince 5(4) tableau as torchdiffeq 0.2.1 holds it (_impl/dopri5.py): beta rows, c_sol (= last beta row, padded),
// c_error = c_sol - 4th-order weights, c_mid (dense-output midpoint weights)
namespace dp5 {
inline constexpr double kBeta[6][6] = {
    {1.0 / 5, 0, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
    {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84},
};
inline constexpr double kCSol[7] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84, 0};
inline constexpr double kCErr[7] = {35.0 / 384 - 1951.0 / 21600, 0, 500.0 / 1113 - 22642.0 / 50085, 125.0 / 192 - 451.0 / 720,
                                    -2187.0 / 6784 + 12231.0 / 42400, 11.0 / 84 - 649.0 / 6300, -1.0 / 60};
inline constexpr double kCMid[7] = {6025192743.0 / 30085553152.0 / 2, 0, 51252292925.0 / 65400821598.0 / 2,
                                    -2691868925.0 / 45128329728.0 / 2, 187940372067.0 / 1594534317056.0 / 2,
                                    -1776094331.0 / 19743644256.0 / 2, 11237099.0 / 235043384.0 / 2};
// weight of stage s in the dense output at x: value(x) = y0 + h * sum_s dense_weight(s, x) * k_s   (the quartic of _interp_fit)
inline double dense_weight(int s, double x) {
  const double d1 = s == 0 ? 1.0 : 0.0, d7 = s == 6 ? 1.0 : 0.0, b = kCSol[s], m = kCMid[s];
  const double A4 = 2.0 * (d7 - d1) - 8.0 * b + 16.0 * m;
  const double B3 = 5.0 * d1 - 3.0 * d7 + 14.0 * b - 32.0 * m;
  const double C2 = d7 - 4.0 * d1 - 5.0 * b + 16.0 * m;
  return x * d1 + x * x * C2 + x * x * x * B3 + x * x * x * x * A4;
}
}  // namespace dp5

}  // namespace odehip
