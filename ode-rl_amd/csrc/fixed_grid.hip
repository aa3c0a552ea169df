// fixed_grid.hip -- euler / midpoint / rk4(3/8) trajectories (torchdiffeq FixedGridODESolver, one step per output
// interval; call sites /root/reference/modules/DiffEqSolver.py:37,45-46) and their BACKWARD pass.
//
// Backward = what the reference gets from `loss.backward()` (train_test.py:204): reverse-mode differentiation
// through every op of the discrete solver ("discretise-then-optimise"; the reference imports `odeint`, not
// `odeint_adjoint`: modules/DiffEqSolver.py:9).  Here it is an explicit reverse sweep:
//   * the forward pass (save_for_backward) keeps every layer input A[n][s][l] of every evaluation of f in the
//     workspace (HBM is 288 GB; B=64,T=10 needs 0.7 GB) -- nothing is recomputed;
//   * per evaluation, the input-gradient chain is the SAME MFMA conv kernel run on transposed+flipped weights
//     (odehip_pack_conv_weight(transpose_flip=1)) with the ReLU mask fused in the epilogue, and the reverse
//     Runge-Kutta bookkeeping (gy += gx, gk_j += c h gx, seed of the next interval) fused into the epilogue of
//     the chain's last conv -- no standalone elementwise kernels;
//   * all weight gradients are ONE launch per layer over all evaluations (wgrad.hip).
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "odehip_internal.h"
#include "persist.h"

namespace odehip {


typedef float f32x4 __attribute__((ext_vector_type(4)));

static int n_stages(int method) { return method == ODEHIP_RK4 ? 4 : (method == ODEHIP_MIDPOINT ? 2 : 1); }
constexpr int kEsplit = 4;

// Everything lives in the caller's workspace; this is the one place that knows where.
struct FixedLayout {
  int T, B, C, S, NH, save;  // NH = hidden activations per evaluation of f (n_convs - 1)
  size_t st;                 // bytes of one (B,C,16,16) tensor
  size_t hid;                // bytes of one hidden activation (B,max_hidden,16,16)
  size_t off_h, off_ping, off_pong, off_xs, off_k, off_y, off_xin, off_hid, off_gp, off_go, off_gy, off_g2, off_tab, off_slab, off_psync, total;
  FixedLayout(const odehip_convstack* f, int batch, int n_times, int method, int save_) {
    T = n_times; B = batch; C = f->channels[0]; S = n_stages(method); NH = f->n_convs - 1; save = save_;
    st = al256((size_t)B * C * kPix * 4);
    int cmax = 32;
    for (int i = 0; i <= f->n_convs; ++i) cmax = f->channels[i] > cmax ? f->channels[i] : cmax;
    hid = al256((size_t)B * cmax * kPix * 4);   // one slot fits any activation / gradient of the stack
    size_t o = 0;
    auto take = [&](size_t b) { size_t r = o; o += al256(b); return r; };
    off_h = take((size_t)T * 4);
    off_ping = take(hid);
    off_pong = take(hid);
    off_xs = take(st);
    off_k = take(3 * st);
    off_y = take((size_t)T * st);
    off_xin = off_hid = off_gp = off_go = off_gy = off_g2 = off_tab = off_slab = 0;
    off_psync = take(persist_sync_bytes(B));  // persistent kernel: flag line per sample + xcc_of[grid] + abort word
    if (save) {
      const size_t ne = (size_t)(T - 1) * S;
      off_xin = take(ne * st);             // stage inputs (slot s = 0 unused: it is y[n])
      off_hid = take(ne * NH * hid);       // ReLU outputs of every evaluation
      off_gp = take(ne * (NH + 1) * hid);  // gradients w.r.t. every conv output (filled by the backward sweep)
      off_go = take((size_t)T * st);       // grad_out in Q4
      off_gy = take(st);
      off_g2 = take(2 * st);
      off_tab = take(ne * sizeof(WgradPair) * ODEHIP_MAX_LAYERS);   // one table per layer, uploaded together
      off_slab = take(((size_t)B * wgrad_esplit_max(B) + 1) * kWgradSlabFloats * 4);
    }
    total = o;
  }
  float* p(void* ws, size_t off) const { return (float*)((char*)ws + off); }
  float* y(void* ws, int n) const { return p(ws, off_y + (size_t)n * st); }
  float* xin(void* ws, int n, int s) const { return s == 0 ? y(ws, n) : p(ws, off_xin + ((size_t)n * S + s) * st); }
  float* hidden(void* ws, int n, int s, int l) const { return p(ws, off_hid + (((size_t)n * S + s) * NH + l) * hid); }
  float* gp(void* ws, int n, int s, int l) const { return p(ws, off_gp + (((size_t)n * S + s) * (NH + 1) + l) * hid); }
  float* go(void* ws, int j) const { return p(ws, off_go + (size_t)j * st); }
};

// out = (c_c + c_h*h) * in
__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ out, const float* __restrict__ in, float c_c, float c_h,
                                                    const float* h_ptr, long long n4) {
  const float c = c_c + c_h * (h_ptr ? *h_ptr : 0.0f);
  for (long long i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
    ((f32x4*)out)[i] = ((const f32x4*)in)[i] * c;
}

struct PtrPack {
  unsigned long long v[32];
};
__global__ void fill_u64_kernel(unsigned long long* dst, PtrPack p, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = p.v[threadIdx.x];
}

// table[e] = {g0 + e*gs, a0 + e*as, scale 1}: the evaluations of one layer are evenly spaced in the workspace, so the table is
// built by one tiny launch instead of being shipped 32 values at a time through kernel arguments
__global__ void wgrad_table_kernel(WgradPair* table, int n_eval, const char* g0, unsigned long long gs, const char* a0,
                                   unsigned long long as) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_eval) return;
  WgradPair p;
  p.g = (const float*)(g0 + (size_t)e * gs);
  p.a = (const float*)(a0 + (size_t)e * as);
  p.scale = 1.0f;
  p.pad_[0] = p.pad_[1] = p.pad_[2] = 0.0f;
  table[e] = p;
}

static int check_common(const odehip_convstack* f, int method, const double* t_host, int n_times, int batch, const char* who) {
  int rc = check_stack(f);
  if (rc != ODEHIP_OK) return rc;
  ODEHIP_REQUIRE(method == ODEHIP_EULER || method == ODEHIP_MIDPOINT || method == ODEHIP_RK4,
                 "%s: method %d is not a fixed-grid method", who, method);
  ODEHIP_REQUIRE(t_host, "%s: null t", who);
  ODEHIP_REQUIRE(n_times >= 1 && n_times <= 4096 && batch > 0, "%s: bad sizes (n_times %d, batch %d)", who, n_times, batch);
  ODEHIP_REQUIRE(f->channels[0] == f->channels[f->n_convs], "%s: f must map C -> C channels (%d -> %d)", who, f->channels[0],
                 f->channels[f->n_convs]);
  for (int i = 1; i < n_times; ++i)
    ODEHIP_REQUIRE(t_host[i] > t_host[i - 1], "%s: t must be strictly increasing (t[%d]=%g, t[%d]=%g)", who, i - 1, t_host[i - 1],
                   i, t_host[i]);
  return ODEHIP_OK;
}

// dW_l, db_l = sum over evaluations e = (n, s) of scale_e * wgrad(GP[n][s][l], A[n][s][l]); one launch per layer.
// A[n][s][0] is the stage input: y[n] (forward sweep) or y[n+1] (adjoint, integrating backwards) for s = 0.
static int wgrad_all_layers(const odehip_convstack* f, const FixedLayout& L, void* ws, int n_times, int batch, bool adjoint,
                            const float* eval_scale /* host, (T-1)*S entries or null = 1 */, float* const* grad_w,
                            float* const* grad_b, hipStream_t stream) {
  const int S = L.S, NL = f->n_convs;
  const int n_eval = (n_times - 1) * S;
  ODEHIP_REQUIRE(n_eval <= 32 * 64, "odeint backward: too many evaluations (%d)", n_eval);
  WgradPair* table = (WgradPair*)L.p(ws, L.off_tab);
  float* slabs = L.p(ws, L.off_slab);
  // the NL tables (g, a, scale per evaluation) travel in ONE asynchronous staged upload (25 kernel-argument uploads per training step before)
  std::vector<WgradPair> host((size_t)NL * n_eval);
  for (int l = 0; l < NL; ++l)
    for (int e = 0; e < n_eval; ++e) {
      const int n = e / S, s = e % S;
      WgradPair& p = host[(size_t)l * n_eval + e];
      p.g = L.gp(ws, n, s, l);
      p.a = l > 0 ? L.hidden(ws, n, s, l - 1) : (s > 0 ? L.xin(ws, n, s) : L.y(ws, adjoint ? n + 1 : n));
      p.scale = eval_scale ? eval_scale[e] : 1.0f;
      p.pad_[0] = p.pad_[1] = p.pad_[2] = 0.0f;
    }
  int rc = staged_upload(table, host.data(), host.size() * sizeof(WgradPair), stream);
  if (rc != ODEHIP_OK) return rc;
  const int esplit = f->w_bf16[0] ? kEsplit : wgrad_esplit(batch, n_eval);
  for (int l = 0; l < NL && !(g_debug_flags & 32); ++l) {
    rc = launch_wgrad(table + (size_t)l * n_eval, n_eval, batch, esplit, slabs, grad_w[l], grad_b[l], f->channels[l + 1], f->channels[l], stream,
                      f->w_bf16[l] != nullptr);
    if (rc != ODEHIP_OK) return rc;
  }
  return ODEHIP_OK;
}

}  // namespace odehip

using namespace odehip;

extern "C" size_t odehip_odeint_workspace_bytes(const odehip_convstack* f, int batch, int n_times, int method,
                                                int save_for_backward) {
  if (!f || batch <= 0 || n_times <= 0 || f->n_convs < 1) return 0;
  return FixedLayout(f, batch, n_times, method, save_for_backward).total;
}

// the saved tensors of a forward pass are bf16 "Q4h" written by the whole-trajectory launch (format 1) when: bf16 fused stack,
// every layer 64 -> 64, rk4, and the persistent switch is on
static bool bf16_trajectory_ok(const odehip_convstack* f, int method) {
  if (!f->w_fused || f->ks != 3 || g_debug_flags || !persist_switch_on()) return false;
  for (int l = 0; l <= f->n_convs; ++l)
    if (f->channels[l] != 64) return false;
  return method == ODEHIP_EULER || method == ODEHIP_MIDPOINT || method == ODEHIP_RK4;
}
constexpr int kMaxWgradEvals = 32 * 64;   // evaluations one weight-gradient table holds

extern "C" int odehip_odeint_fixed(const odehip_convstack* f, int method, const float* z0_nchw, const double* t_host,
                                   int n_times, int batch, float* out_nchw, int save_for_backward, int negate,
                                   void* workspace, size_t workspace_bytes, int* saved_format_out, void* stream_) {
  if (saved_format_out) *saved_format_out = 0;
  int rc = check_common(f, method, t_host, n_times, batch, "odeint_fixed");
  ODEHIP_REQUIRE(!(negate && save_for_backward), "odeint_fixed: backward through negated dynamics is not supported");
  if (rc != ODEHIP_OK) return rc;
  ODEHIP_REQUIRE(z0_nchw && out_nchw && workspace, "odeint_fixed: null pointer");
  ODEHIP_REQUIRE(saved_format_out || !save_for_backward, "odeint_fixed: saved_format_out is required with save_for_backward");
  const FixedLayout L(f, batch, n_times, method, save_for_backward);
  ODEHIP_REQUIRE(workspace_bytes >= L.total, "odeint_fixed: workspace too small (%zu < %zu)", workspace_bytes, L.total);
  hipStream_t stream = (hipStream_t)stream_;
  void* ws = workspace;
  const size_t st_b = (size_t)batch * L.C * kPix * 4, st_f = st_b / 4;
  float* hdev = L.p(ws, L.off_h);
  float* ping = L.p(ws, L.off_ping);
  float* pong = L.p(ws, L.off_pong);
  float* k[3] = {L.p(ws, L.off_k), L.p(ws, L.off_k + L.st), L.p(ws, L.off_k + 2 * L.st)};

  // step sizes: dt = t1 - t0 in float64, rounded to fp32 when it meets the state (torchdiffeq semantics)
  float hbuf[4096];
  for (int i = 0; i + 1 < n_times; ++i) hbuf[i] = (float)(t_host[i + 1] - t_host[i]);
  // ONE prologue launch: solution[0] = y0, y0 in the kernels' layout, the step sizes on the device, the persistent launch's flags zeroed
  const bool h_in_prologue = n_times - 1 <= 64;
  unsigned* psync = (unsigned*)L.p(ws, L.off_psync);
  rc = traj_prologue(z0_nchw, L.y(ws, 0), out_nchw, batch, L.C, hbuf, h_in_prologue ? n_times - 1 : 0, hdev, psync,
                     (int)(persist_sync_bytes(batch) / 4), stream);
  if (rc != ODEHIP_OK) return rc;
  if (n_times == 1) return ODEHIP_OK;
  if (!h_in_prologue) {
    rc = upload_floats(hdev, hbuf, n_times - 1, stream);
    if (rc != ODEHIP_OK) return rc;
  }

  // bf16 compute, 64-channel stack: the whole trajectory as ONE launch with one workgroup per sample -- state and stage derivatives
  // in registers, activations in LDS (fstack_bf16.hip: ftraj_bf16_kernel).  A training forward (rk4) also saves every stage input
  // and hidden activation as bf16 for the one-launch reverse sweep (btraj_bf16.hip): saved format 1.
  if (bf16_trajectory_ok(f, method) && (!save_for_backward || (method == ODEHIP_RK4 && (n_times - 1) * 4 <= kMaxWgradEvals))) {
    if (save_for_backward) {
      rc = launch_ftraj_bf16_saving(f, z0_nchw, out_nchw, hdev, n_times, batch, L.p(ws, L.off_xin), L.st, L.p(ws, L.off_hid),
                                    (size_t)L.NH * L.hid, L.hid, stream);
      if (rc == ODEHIP_OK) *saved_format_out = 1;   // non-null: checked with the arguments, before anything was enqueued
    } else {
      rc = launch_ftraj_bf16(f, method, z0_nchw, out_nchw, hdev, n_times, batch, negate, stream);
    }
    if (rc == ODEHIP_OK) persist_count_launch();
    return rc;
  }

  // One launch for the whole trajectory when the dynamics are the 64-channel fp32 stack: the loop below then only RECORDS its
  // layers (with save_for_backward only the destinations of the hidden layers differ).
  PersistScope persist;
  if ((rc = persist.begin(f, nullptr, (n_times - 1) * L.S * f->n_convs)) != ODEHIP_OK) return rc;
  float* hidv[ODEHIP_MAX_LAYERS];
  auto enqueue_steps = [&]() -> int {
  for (int n = 0; n + 1 < n_times; ++n) {
    const float* y = L.y(ws, n);
    float* ynew = L.y(ws, n + 1);
    float* ynew_nchw = out_nchw + (size_t)(n + 1) * st_f;
    // stage input / hidden-activation buffers of stage s (distinct per evaluation when saving)
    auto xin = [&](int s) { return save_for_backward ? L.xin(ws, n, s) : L.p(ws, L.off_xs); };
    auto run = [&](int s, const float* x, const CombineArgs& c) {
      float* const* hp = nullptr;
      if (save_for_backward) {
        for (int l = 0; l < L.NH; ++l) hidv[l] = L.hidden(ws, n, s, l);
        hp = hidv;
      }
      return enqueue_f_saving(f, x, batch, hp, ping, pong, &c, nullptr, nullptr, stream);
    };
    CombineArgs c;
    memset(&c, 0, sizeof(c));
    c.y = y;
    c.h_ptr = hdev + n;
    c.k_scale = negate ? -1.0f : 1.0f;
    if (method == ODEHIP_EULER) {  // y1 = y + h*f(y)
      c.c2[0] = 1.0f;
      c.out2 = ynew;
      c.out2_nchw = ynew_nchw;
      if ((rc = run(0, y, c)) != ODEHIP_OK) return rc;
    } else if (method == ODEHIP_MIDPOINT) {  // x = y + h/2*k1 ; y1 = y + h*f(x)
      c.c1[0] = 0.5f;
      c.out1 = xin(1);
      if ((rc = run(0, y, c)) != ODEHIP_OK) return rc;
      c.c1[0] = 0.0f;
      c.out1 = nullptr;
      c.c2[0] = 1.0f;
      c.out2 = ynew;
      c.out2_nchw = ynew_nchw;
      if ((rc = run(1, xin(1), c)) != ODEHIP_OK) return rc;
    } else {  // 3/8 rule (torchdiffeq rk4_alt_step_func)
      const float third = 1.0f / 3.0f;
      c.k_out = k[0];  // k1 = f(y); x2 = y + h*(k1/3)
      c.c1[0] = third;
      c.out1 = xin(1);
      if ((rc = run(0, y, c)) != ODEHIP_OK) return rc;
      c.n_prev = 1;  // k2 = f(x2); x3 = y + h*(k2 - k1/3)
      c.k_prev[0] = k[0];
      c.k_out = k[1];
      c.c1[0] = -third;
      c.c1[1] = 1.0f;
      c.out1 = xin(2);
      if ((rc = run(1, xin(1), c)) != ODEHIP_OK) return rc;
      c.n_prev = 2;  // k3 = f(x3); x4 = y + h*(k1 - k2 + k3)
      c.k_prev[1] = k[1];
      c.k_out = k[2];
      c.c1[0] = 1.0f;
      c.c1[1] = -1.0f;
      c.c1[2] = 1.0f;
      c.out1 = xin(3);
      if ((rc = run(2, xin(2), c)) != ODEHIP_OK) return rc;
      c.n_prev = 3;  // k4 = f(x4); y1 = y + h*(k1 + 3(k2+k3) + k4)/8
      c.k_prev[2] = k[2];
      c.k_out = nullptr;
      c.out1 = nullptr;
      c.c2[0] = 0.125f;
      c.c2[1] = 0.375f;
      c.c2[2] = 0.375f;
      c.c2[3] = 0.125f;
      c.out2 = ynew;
      c.out2_nchw = ynew_nchw;
      if ((rc = run(3, xin(3), c)) != ODEHIP_OK) return rc;
    }
  }
  return ODEHIP_OK;
  };
  rc = enqueue_steps();
  int rc2 = persist.finish(hbuf, hdev, out_nchw, batch, psync, f->ks, stream, /*sync_is_zero=*/true);
  if (rc == ODEHIP_OK && rc2 == ODEHIP_OK) {
    float* const regions[1] = {out_nchw};
    const size_t floats[1] = {(size_t)n_times * st_f};
    rc2 = persist.guard(regions, floats, 1, stream);
  }
  return rc != ODEHIP_OK ? rc : rc2;
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of odehip_odeint_fixed(save_for_backward = 1) on the SAME workspace.
//   f_dgrad: the stack of input-gradient convs, f_dgrad->w_packed[l] = pack(W_l, transpose_flip = 1), in the
//            forward layer order; bias pointers unused.
//   grad_out (T,B,C,16,16) NCHW -> grad_z0 (B,C,16,16) NCHW, grad_w[l] (OIHW), grad_b[l].
// behind the last kernel of a backward pass whose sweep ran as a persistent launch: NaN-fill every gradient if that launch gave up
static int guard_gradients(PersistScope& persist, const odehip_convstack* f, int batch, float* grad_z0, float* const* grad_w,
                           float* const* grad_b, hipStream_t stream) {
  float* regions[1 + 2 * ODEHIP_MAX_LAYERS];
  size_t floats[1 + 2 * ODEHIP_MAX_LAYERS];
  int n = 0;
  regions[n] = grad_z0;
  floats[n++] = (size_t)batch * f->channels[0] * kPix;
  for (int l = 0; l < f->n_convs; ++l) {
    regions[n] = grad_w ? grad_w[l] : nullptr;
    floats[n++] = (size_t)f->channels[l + 1] * f->channels[l] * f->ks * f->ks;
    regions[n] = grad_b ? grad_b[l] : nullptr;
    floats[n++] = (size_t)f->channels[l + 1];
  }
  return persist.guard(regions, floats, n, stream);
}

// ---------------------------------------------------------------------------------------------------------------
extern "C" int odehip_odeint_fixed_backward(const odehip_convstack* f, const odehip_convstack* f_dgrad, int method,
                                            const double* t_host, int n_times, int batch, const float* grad_out_nchw,
                                            float* grad_z0_nchw, float* const* grad_w, float* const* grad_b, int saved_format,
                                            void* workspace, size_t workspace_bytes, void* stream_) {
  int rc = check_common(f, method, t_host, n_times, batch, "odeint_fixed_backward");
  ODEHIP_REQUIRE(saved_format == 0 || saved_format == 1, "odeint_fixed_backward: unknown saved format %d", saved_format);
  if (rc != ODEHIP_OK) return rc;
  ODEHIP_REQUIRE(f_dgrad && grad_out_nchw && grad_z0_nchw && grad_w && grad_b && workspace, "odeint_fixed_backward: null pointer");
  for (int l = 0; l <= f->n_convs; ++l)
    ODEHIP_REQUIRE(f->channels[l] % 64 == 0, "odeint_fixed_backward: channel counts must be multiples of 64 (channels[%d] = %d)", l,
                   f->channels[l]);
  ODEHIP_REQUIRE(f->ks == 3, "odeint_fixed_backward: 3x3 dynamics only");
  for (int l = 0; l < f->n_convs; ++l)
    ODEHIP_REQUIRE(f_dgrad->w_packed[l] && grad_w[l] && grad_b[l], "odeint_fixed_backward: layer %d has null pointers", l);
  const FixedLayout L(f, batch, n_times, method, 1);
  ODEHIP_REQUIRE(workspace_bytes >= L.total, "odeint_fixed_backward: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  void* ws = workspace;
  const int NH = L.NH, S = L.S, NL = f->n_convs;
  const size_t st_b = (size_t)batch * L.C * kPix * 4;
  const long long n4 = (long long)(st_b / 16);
  float* hdev = L.p(ws, L.off_h);
  float* gy = L.p(ws, L.off_gy);
  float* gbuf[2] = {L.p(ws, L.off_g2), L.p(ws, L.off_g2 + L.st)};

  if (saved_format == 1 && n_times > 1) {
    // ---- the forward was the whole-trajectory bf16 launch: ONE launch for the reverse sweep (gradient state in registers, every
    // conv-output gradient stored as bf16 Q4h), then one weight-gradient launch per layer on the bf16 operands
    ODEHIP_REQUIRE(method == ODEHIP_RK4 && f->w_fused && f_dgrad->w_fused, "odeint_fixed_backward: saved format 1 belongs to the fused bf16 rk4 path");
    // The sweep occupies one CU per sample -- half the chip at 128 samples per GPU -- and the weight gradients only need what the
    // sweep has already written: the sweep is cut into up to four launches (its state crosses the cut in two state-sized tensors)
    // and the weight gradients of a finished segment run on a library-owned SIDE STREAM, on the idle CUs, while the next segment
    // is swept.  The concurrent weight-gradient launches are sized to the CUs the sweep leaves free (one workgroup per sample at
    // 128 samples) so that they cannot take the CUs the sweep's next launch needs; the last segment's run on the whole chip.  dW accumulates over the segments in order.
    // The side stream and its events are library state of ONE device (one process per GPU is the deployment model): created once
    // under a lock, bound to the device that was current then, and every later call must come from that device -- a call from
    // another one is refused instead of launching the weight gradients on the wrong device's stream.  Calls are serialised by the
    // caller (one host thread per process drives the library: SURVEY.md section 8b "Threading"); the lock only makes the lazy
    // creation itself safe.
    constexpr int kMaxSeg = 4;   // measured at B=128, T=40: 4 segments 9.3 ms, 6: 9.4, 8: 9.6, uncut 10.2
    static std::mutex side_mu;
    static hipStream_t side = nullptr;
    static int side_device = -1;
    static hipEvent_t ev_seg[kMaxSeg] = {}, ev_done = nullptr;
    {
      std::lock_guard<std::mutex> lk(side_mu);
      int cur_dev = -1;
      ODEHIP_CHECK_HIP(hipGetDevice(&cur_dev));
      if (!side) {
        int lo_pri = 0, hi_pri = 0;
        ODEHIP_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
        hipStream_t s_new = nullptr;
        ODEHIP_CHECK_HIP(hipStreamCreateWithPriority(&s_new, hipStreamNonBlocking, lo_pri));   // the sweep's launches go first
        for (int i = 0; i < kMaxSeg; ++i) ODEHIP_CHECK_HIP(hipEventCreateWithFlags(&ev_seg[i], hipEventDisableTiming));
        ODEHIP_CHECK_HIP(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
        side_device = cur_dev;
        side = s_new;
      }
      ODEHIP_REQUIRE(cur_dev == side_device, "odeint_fixed_backward: the library's side stream belongs to device %d, this call runs on device %d "
                                               "(one process drives one GPU)", side_device, cur_dev);
    }
    const int n_steps = n_times - 1;
    const int n_eval = n_steps * S;
    ODEHIP_REQUIRE(n_eval <= kMaxWgradEvals, "odeint backward: too many evaluations (%d)", n_eval);
    static const bool overlap_on = [] { const char* e = getenv("ODEHIP_BF16_OVERLAP"); return !(e && e[0] == '0'); }();
    static const int seg_env = [] { const char* e = getenv("ODEHIP_BF16_SEGMENTS"); return e ? atoi(e) : 0; }();
    int cus = 256;
    {
      int dev = 0;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    }
    const int free_cus = cus - batch;                         // the sweep holds one CU per sample
    const int esplit_c = free_cus / batch < kEsplit ? free_cus / batch : kEsplit;   // concurrent weight-gradient workgroups per sample
    int n_seg = overlap_on && esplit_c >= 1 ? n_steps / 8 : 1;   // at least eight intervals per segment; no idle CUs: no cut
    if (seg_env > 0) n_seg = seg_env;
    n_seg = n_seg < 1 ? 1 : (n_seg > kMaxSeg ? kMaxSeg : n_seg);
    if (n_seg > n_steps) n_seg = n_steps;
    float* bias_part = L.p(ws, L.off_g2);                      // [segment][B][NL][64] fp32: fits the two state-sized scratch tensors
    float* state_g = L.go(ws, 0);                              // the per-launch path's Q4 copy of grad_out is not needed here:
    float* state_seed = L.go(ws, 1);                           // two of its T tensors carry the sweep's state across a cut
    WgradPair* table = (WgradPair*)L.p(ws, L.off_tab);
    float* slabs = L.p(ws, L.off_slab);
    int hi = n_steps;
    for (int k = 0; k < n_seg; ++k) {
      const int lo = (int)((long long)n_steps * (n_seg - 1 - k) / n_seg);
      rc = launch_btraj_bf16_rk4(f_dgrad, grad_out_nchw, grad_z0_nchw, hdev, n_times, batch, lo, hi, state_g, state_seed,
                                 L.p(ws, L.off_hid), (size_t)NH * L.hid, L.hid, L.p(ws, L.off_gp), (size_t)(NH + 1) * L.hid, L.hid,
                                 bias_part + (size_t)k * batch * NL * 64, stream);
      if (rc != ODEHIP_OK) return rc;
      const bool concurrent = n_seg > 1 && k + 1 < n_seg;     // another segment is swept while these weight gradients run
      hipStream_t ws_stream = n_seg > 1 ? side : stream;
      if (n_seg > 1) {
        ODEHIP_CHECK_HIP(hipEventRecord(ev_seg[k], stream));
        ODEHIP_CHECK_HIP(hipStreamWaitEvent(side, ev_seg[k], 0));
      }
      const int e_lo = lo * S, n_e = (hi - lo) * S;
      for (int l = 0; l < NL; ++l) {
        const char* g0 = (const char*)L.p(ws, L.off_gp) + ((size_t)e_lo * (NH + 1) + l) * L.hid;
        const char* a0 = l > 0 ? (const char*)L.p(ws, L.off_hid) + ((size_t)e_lo * NH + (l - 1)) * L.hid
                               : (const char*)L.p(ws, L.off_xin) + (size_t)e_lo * L.st;
        hipLaunchKernelGGL(wgrad_table_kernel, dim3((n_e + 255) / 256), dim3(256), 0, ws_stream, table, n_e, g0, (size_t)(NH + 1) * L.hid,
                           a0, l > 0 ? (size_t)NH * L.hid : L.st);
        rc = launch_wgrad_q4h(table, n_e, batch, concurrent ? (esplit_c >= 1 ? esplit_c : 1) : kEsplit, slabs, grad_w[l], /*accumulate=*/k > 0, ws_stream);
        if (rc != ODEHIP_OK) return rc;
      }
      hi = lo;
    }
    if (n_seg > 1) {   // the caller's stream continues only when the side stream is done with the workspace and the gradients
      ODEHIP_CHECK_HIP(hipEventRecord(ev_done, side));
      ODEHIP_CHECK_HIP(hipStreamWaitEvent(stream, ev_done, 0));
    }
    return launch_bias_reduce(bias_part, n_seg * batch, NL, grad_b, stream);
  }
  ODEHIP_REQUIRE(saved_format == 0 || n_times == 1, "odeint_fixed_backward: bad saved format");
  rc = odehip_nchw_to_q4(grad_out_nchw, L.go(ws, 0), n_times * batch, L.C, stream);
  if (rc != ODEHIP_OK) return rc;
  if (n_times == 1) {
    ODEHIP_CHECK_HIP(hipMemcpyAsync(grad_z0_nchw, grad_out_nchw, st_b, hipMemcpyDeviceToDevice, stream));
    for (int l = 0; l < NL; ++l) {
      ODEHIP_CHECK_HIP(hipMemsetAsync(grad_w[l], 0, (size_t)f->channels[l + 1] * f->channels[l] * 9 * 4, stream));
      ODEHIP_CHECK_HIP(hipMemsetAsync(grad_b[l], 0, (size_t)f->channels[l + 1] * 4, stream));
    }
    return ODEHIP_OK;
  }

  // dgrad chain of evaluation (n, s): GP[n][s][NH] (gradient w.r.t. k) -> ... -> gx, consumed by `targets`
  auto chain = [&](int n, int s, const BwdArgs& last) -> int {
    float* gpv[ODEHIP_MAX_LAYERS];
    const float* hv[ODEHIP_MAX_LAYERS];
    for (int l = 0; l < NL; ++l) gpv[l] = L.gp(ws, n, s, l);
    for (int l = 0; l + 1 < NL; ++l) hv[l] = L.hidden(ws, n, s, l);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.combine = 3;
    a.bwd = last;
    return enqueue_dgrad_chain(f, f_dgrad, batch, gpv, hv, a, stream);
  };
  auto tgt = [](float* out, const float* sa, float a_c, float a_h, const float* sb, float b_c, float b_h, float g_c, float g_h) {
    BwdTarget t;
    t.out = out; t.srcA = sa; t.srcB = sb;
    t.a_c = a_c; t.a_h = a_h; t.b_c = b_c; t.b_h = b_h; t.g_c = g_c; t.g_h = g_h;
    return t;
  };

  float hbuf[4096];
  for (int i = 0; i + 1 < n_times; ++i) hbuf[i] = (float)(t_host[i + 1] - t_host[i]);  // as uploaded by the forward pass
  const int last_s = S - 1;
  const float wlast = method == ODEHIP_RK4 ? 0.125f : 1.0f;  // weight of the last stage's k in the step
  // seed: gradient w.r.t. the last stage's k of the last interval = wlast * h * grad_out[T-1]
  hipLaunchKernelGGL(scale_kernel, dim3(1024), dim3(256), 0, stream, L.gp(ws, n_times - 2, last_s, NH), L.go(ws, n_times - 1), 0.0f,
                     wlast, hdev + (n_times - 2), n4);
  const float* g = L.go(ws, n_times - 1);  // total gradient w.r.t. y[n+1]
  // the reverse sweep is nothing but conv launches (all bookkeeping lives in their epilogues): one persistent launch
  PersistScope persist;
  if ((rc = persist.begin(f, f_dgrad, (n_times - 1) * S * NL)) != ODEHIP_OK) return rc;
  auto sweep = [&]() -> int {
  for (int n = n_times - 2; n >= 0; --n) {
    float* gnext = gbuf[n & 1];
    // epilogue of the FIRST stage's chain: closes the interval and seeds the next one (which uses h[n-1])
    auto close_interval = [&](const float* gy_src) {
      BwdArgs w;
      memset(&w, 0, sizeof(w));
      if (n > 0) {
        w.h_ptr = hdev + (n - 1);
        w.n_targets = 2;
        w.tgt[0] = tgt(gnext, gy_src, 1.f, 0.f, L.go(ws, n), 1.f, 0.f, 1.f, 0.f);
        w.tgt[1] = tgt(L.gp(ws, n - 1, last_s, NH), gy_src, 0.f, wlast, L.go(ws, n), 0.f, wlast, 0.f, wlast);
      } else {
        w.n_targets = 1;
        w.tgt[0] = tgt(gnext, gy_src, 1.f, 0.f, L.go(ws, 0), 1.f, 0.f, 1.f, 0.f);
      }
      return w;
    };
    BwdArgs w;
    if (method == ODEHIP_EULER) {
      // y1 = y + h k1:  gk1 = h g (already seeded), gy = g
      w = close_interval(g);
      if ((rc = chain(n, 0, w)) != ODEHIP_OK) return rc;
    } else if (method == ODEHIP_MIDPOINT) {
      // x = y + h/2 k1, y1 = y + h k2:  gk2 = h g (seeded); gx2 -> gy = g + gx2, gk1 = h/2 gx2
      memset(&w, 0, sizeof(w));
      w.h_ptr = hdev + n;
      w.n_targets = 2;
      w.tgt[0] = tgt(gy, g, 1.f, 0.f, nullptr, 0.f, 0.f, 1.f, 0.f);
      w.tgt[1] = tgt(L.gp(ws, n, 0, NH), nullptr, 0.f, 0.f, nullptr, 0.f, 0.f, 0.f, 0.5f);
      if ((rc = chain(n, 1, w)) != ODEHIP_OK) return rc;
      w = close_interval(gy);
      if ((rc = chain(n, 0, w)) != ODEHIP_OK) return rc;
    } else {
      const float third = 1.0f / 3.0f;
      float* gk1 = L.gp(ws, n, 0, NH);
      float* gk2 = L.gp(ws, n, 1, NH);
      float* gk3 = L.gp(ws, n, 2, NH);
      // stage 4 (gk4 = h/8 g seeded): gx4 -> gy = g + gx4; gk3 = 3h/8 g + h gx4; gk2 = 3h/8 g - h gx4; gk1 = h/8 g + h gx4
      memset(&w, 0, sizeof(w));
      w.h_ptr = hdev + n;
      w.n_targets = 4;
      w.tgt[0] = tgt(gy, g, 1.f, 0.f, nullptr, 0.f, 0.f, 1.f, 0.f);
      w.tgt[1] = tgt(gk3, g, 0.f, 0.375f, nullptr, 0.f, 0.f, 0.f, 1.f);
      w.tgt[2] = tgt(gk2, g, 0.f, 0.375f, nullptr, 0.f, 0.f, 0.f, -1.f);
      w.tgt[3] = tgt(gk1, g, 0.f, 0.125f, nullptr, 0.f, 0.f, 0.f, 1.f);
      if ((rc = chain(n, 3, w)) != ODEHIP_OK) return rc;
      // stage 3: gx3 -> gy += gx3; gk2 += h gx3; gk1 -= h/3 gx3
      w.n_targets = 3;
      w.tgt[0] = tgt(gy, gy, 1.f, 0.f, nullptr, 0.f, 0.f, 1.f, 0.f);
      w.tgt[1] = tgt(gk2, gk2, 1.f, 0.f, nullptr, 0.f, 0.f, 0.f, 1.f);
      w.tgt[2] = tgt(gk1, gk1, 1.f, 0.f, nullptr, 0.f, 0.f, 0.f, -third);
      if ((rc = chain(n, 2, w)) != ODEHIP_OK) return rc;
      // stage 2: gx2 -> gy += gx2; gk1 += h/3 gx2
      w.n_targets = 2;
      w.tgt[0] = tgt(gy, gy, 1.f, 0.f, nullptr, 0.f, 0.f, 1.f, 0.f);
      w.tgt[1] = tgt(gk1, gk1, 1.f, 0.f, nullptr, 0.f, 0.f, 0.f, third);
      if ((rc = chain(n, 1, w)) != ODEHIP_OK) return rc;
      // stage 1: gx1 -> g(y[n]) = gy + gx1 + grad_out[n]
      w = close_interval(gy);
      if ((rc = chain(n, 0, w)) != ODEHIP_OK) return rc;
    }
    g = gnext;
  }
  return ODEHIP_OK;
  };
  rc = sweep();
  const int rc2 = persist.finish(hbuf, hdev, nullptr, batch, (unsigned*)L.p(ws, L.off_psync), f->ks, stream);
  if (rc != ODEHIP_OK || rc2 != ODEHIP_OK) return rc != ODEHIP_OK ? rc : rc2;
  rc = odehip_q4_to_nchw(g, grad_z0_nchw, batch, L.C, stream);
  if (rc != ODEHIP_OK) return rc;

  // ---- weight / bias gradients: one launch per layer over all (T-1)*S evaluations
  rc = wgrad_all_layers(f, L, ws, n_times, batch, /*adjoint=*/false, nullptr, grad_w, grad_b, stream);
  if (rc != ODEHIP_OK) return rc;
  return guard_gradients(persist, f, batch, grad_z0_nchw, grad_w, grad_b, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Adjoint backward (torchdiffeq `odeint_adjoint` semantics, _impl/adjoint.py; NEW capability -- the reference itself
// never uses it: modules/DiffEqSolver.py:9).  For i = T-1 .. 1 the augmented state (y, a_y, a_theta) is integrated
// from t[i] back to t[i-1] with ONE step of the same fixed-grid method on the negated dynamics (torchdiffeq flips a
// decreasing time grid), y is then reset to the stored forward value y[i-1] and a_y += grad_out[i-1].
//   d a_y / dt     = -J_f(y)^T a_y          -> the dgrad chain (same MFMA conv kernel, transposed+flipped weights)
//   d a_theta / dt = -(df/dtheta)^T a_y     -> wgrad over every stage of every interval, weighted dt*b_s, one launch
// The stages of y are recomputed from y[i] (nothing saved by the forward pass except the trajectory itself).
// Workspace: odehip_odeint_workspace_bytes(..., save_for_backward = 1).
// ---------------------------------------------------------------------------------------------------------------
extern "C" int odehip_odeint_adjoint_backward(const odehip_convstack* f, const odehip_convstack* f_dgrad, int method,
                                              const double* t_host, int n_times, int batch, const float* y_traj_nchw,
                                              const float* grad_out_nchw, float* grad_z0_nchw, float* const* grad_w,
                                              float* const* grad_b, void* workspace, size_t workspace_bytes, void* stream_) {
  int rc = check_common(f, method, t_host, n_times, batch, "odeint_adjoint_backward");
  if (rc != ODEHIP_OK) return rc;
  ODEHIP_REQUIRE(f_dgrad && y_traj_nchw && grad_out_nchw && grad_z0_nchw && grad_w && grad_b && workspace,
                 "odeint_adjoint_backward: null pointer");
  for (int l = 0; l <= f->n_convs; ++l)
    ODEHIP_REQUIRE(f->channels[l] % 64 == 0, "odeint_adjoint_backward: channel counts must be multiples of 64 (channels[%d] = %d)", l,
                   f->channels[l]);
  ODEHIP_REQUIRE(f->ks == 3, "odeint_adjoint_backward: 3x3 dynamics only");
  const FixedLayout L(f, batch, n_times, method, 1);
  ODEHIP_REQUIRE(workspace_bytes >= L.total, "odeint_adjoint_backward: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  void* ws = workspace;
  const int NH = L.NH, S = L.S, NL = f->n_convs;
  const size_t st_b = (size_t)batch * L.C * kPix * 4;
  float* hdev = L.p(ws, L.off_h);
  float* ping = L.p(ws, L.off_ping);
  float* pong = L.p(ws, L.off_pong);
  float* k[3] = {L.p(ws, L.off_k), L.p(ws, L.off_k + L.st), L.p(ws, L.off_k + 2 * L.st)};
  float* q3 = L.p(ws, L.off_gy);             // partial sums of the a_y stages (see below)
  float* q4 = L.p(ws, L.off_g2);
  float* rr = L.p(ws, L.off_g2 + L.st);

  rc = odehip_nchw_to_q4(y_traj_nchw, L.y(ws, 0), n_times * batch, L.C, stream);
  if (rc != ODEHIP_OK) return rc;
  rc = odehip_nchw_to_q4(grad_out_nchw, L.go(ws, 0), n_times * batch, L.C, stream);
  if (rc != ODEHIP_OK) return rc;
  if (n_times == 1) {
    ODEHIP_CHECK_HIP(hipMemcpyAsync(grad_z0_nchw, grad_out_nchw, st_b, hipMemcpyDeviceToDevice, stream));
    for (int l = 0; l < NL; ++l) {
      ODEHIP_CHECK_HIP(hipMemsetAsync(grad_w[l], 0, (size_t)f->channels[l + 1] * f->channels[l] * 9 * 4, stream));
      ODEHIP_CHECK_HIP(hipMemsetAsync(grad_b[l], 0, (size_t)f->channels[l + 1] * 4, stream));
    }
    return ODEHIP_OK;
  }
  float hbuf[4096];
  for (int i = 0; i + 1 < n_times; ++i) hbuf[i] = (float)(t_host[i + 1] - t_host[i]);  // dt of the flipped grid, > 0
  rc = upload_floats(hdev, hbuf, n_times - 1, stream);
  if (rc != ODEHIP_OK) return rc;

  float* hidv[ODEHIP_MAX_LAYERS];
  auto run_f = [&](int n, int s, const float* x, const CombineArgs& c) {  // f at stage s, activations kept
    for (int l = 0; l < NH; ++l) hidv[l] = L.hidden(ws, n, s, l);
    return enqueue_f_saving(f, x, batch, hidv, ping, pong, &c, nullptr, nullptr, stream);
  };
  auto chain = [&](int n, int s, const BwdArgs& last) -> int {  // K^a = J_f(Y_s)^T A_s, A_s = GP[n][s][NH]
    float* gpv[ODEHIP_MAX_LAYERS];
    const float* hv[ODEHIP_MAX_LAYERS];
    for (int l = 0; l < NL; ++l) gpv[l] = L.gp(ws, n, s, l);
    for (int l = 0; l + 1 < NL; ++l) hv[l] = L.hidden(ws, n, s, l);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.combine = 3;
    a.bwd = last;
    return enqueue_dgrad_chain(f, f_dgrad, batch, gpv, hv, a, stream);
  };
  auto tgt = [](float* out, const float* sa, float a_c, const float* sb, float b_c, float g_h) {
    BwdTarget t;
    t.out = out; t.srcA = sa; t.srcB = sb;
    t.a_c = a_c; t.a_h = 0.f; t.b_c = b_c; t.b_h = 0.f; t.g_c = 0.f; t.g_h = g_h;
    return t;
  };

  // seed: a_y = grad_out[T-1] is the stage-1 adjoint of the last interval
  ODEHIP_CHECK_HIP(hipMemcpyAsync(L.gp(ws, n_times - 2, 0, NH), L.go(ws, n_times - 1), st_b, hipMemcpyDeviceToDevice, stream));
  float scales[4096];
  ODEHIP_REQUIRE((n_times - 1) * S <= 4096, "odeint_adjoint_backward: too many evaluations");
  float* a_final = L.p(ws, L.off_xs);
  // every interval is conv launches only (recomputed stages + input-gradient chains, all bookkeeping in their epilogues)
  PersistScope persist;
  if ((rc = persist.begin(f, f_dgrad, (n_times - 1) * S * NL * 2)) != ODEHIP_OK) return rc;
  auto sweep = [&]() -> int {
  for (int n = n_times - 2; n >= 0; --n) {
    const float* y = L.y(ws, n + 1);                       // integrate from t[n+1] back to t[n]
    const float* a = L.gp(ws, n, 0, NH);                   // a_y at t[n+1]
    float* a_next = n > 0 ? L.gp(ws, n - 1, 0, NH) : a_final;  // a_y at t[n] (+ grad_out[n]) seeds the next interval
    CombineArgs c;
    memset(&c, 0, sizeof(c));
    c.y = y;
    c.h_ptr = hdev + n;
    c.k_scale = -1.0f;                                     // negated dynamics
    BwdArgs w;
    memset(&w, 0, sizeof(w));
    w.h_ptr = hdev + n;
    const float dt = hbuf[n];
    if (method == ODEHIP_EULER) {
      scales[n * S + 0] = dt;
      if ((rc = run_f(n, 0, y, c)) != ODEHIP_OK) return rc;
      w.n_targets = 1;
      w.tgt[0] = tgt(a_next, a, 1.f, L.go(ws, n), 1.f, 1.f);            // a + dt K1 + grad_out[n]
      if ((rc = chain(n, 0, w)) != ODEHIP_OK) return rc;
    } else if (method == ODEHIP_MIDPOINT) {
      scales[n * S + 0] = 0.0f;                                         // b = (0, 1)
      scales[n * S + 1] = dt;
      c.c1[0] = 0.5f;
      c.out1 = L.xin(ws, n, 1);                                         // Y2 = y + dt/2 k1'
      if ((rc = run_f(n, 0, y, c)) != ODEHIP_OK) return rc;
      w.n_targets = 1;
      w.tgt[0] = tgt(L.gp(ws, n, 1, NH), a, 1.f, nullptr, 0.f, 0.5f);   // A2 = a + dt/2 K1
      if ((rc = chain(n, 0, w)) != ODEHIP_OK) return rc;
      c.c1[0] = 0.f;
      c.out1 = nullptr;
      if ((rc = run_f(n, 1, L.xin(ws, n, 1), c)) != ODEHIP_OK) return rc;
      w.tgt[0] = tgt(a_next, a, 1.f, L.go(ws, n), 1.f, 1.f);            // a + dt K2 + grad_out[n]
      if ((rc = chain(n, 1, w)) != ODEHIP_OK) return rc;
    } else {
      const float third = 1.0f / 3.0f;
      scales[n * S + 0] = dt * 0.125f;
      scales[n * S + 1] = dt * 0.375f;
      scales[n * S + 2] = dt * 0.375f;
      scales[n * S + 3] = dt * 0.125f;
      // stage 1: k1' = -f(y); Y2 = y + dt k1'/3.   K1: A2 = a + dt/3 K1; Q3 = a - dt/3 K1; Q4 = a + dt K1; R = a + dt/8 K1
      c.k_out = k[0];
      c.c1[0] = third;
      c.out1 = L.xin(ws, n, 1);
      if ((rc = run_f(n, 0, y, c)) != ODEHIP_OK) return rc;
      w.n_targets = 4;
      w.tgt[0] = tgt(L.gp(ws, n, 1, NH), a, 1.f, nullptr, 0.f, third);
      w.tgt[1] = tgt(q3, a, 1.f, nullptr, 0.f, -third);
      w.tgt[2] = tgt(q4, a, 1.f, nullptr, 0.f, 1.f);
      w.tgt[3] = tgt(rr, a, 1.f, nullptr, 0.f, 0.125f);
      if ((rc = chain(n, 0, w)) != ODEHIP_OK) return rc;
      // stage 2: Y3 = y + dt (k2' - k1'/3).   K2: A3 = Q3 + dt K2; Q4 -= dt K2; R += 3dt/8 K2
      c.n_prev = 1;
      c.k_prev[0] = k[0];
      c.k_out = k[1];
      c.c1[0] = -third;
      c.c1[1] = 1.f;
      c.out1 = L.xin(ws, n, 2);
      if ((rc = run_f(n, 1, L.xin(ws, n, 1), c)) != ODEHIP_OK) return rc;
      w.n_targets = 3;
      w.tgt[0] = tgt(L.gp(ws, n, 2, NH), q3, 1.f, nullptr, 0.f, 1.f);
      w.tgt[1] = tgt(q4, q4, 1.f, nullptr, 0.f, -1.f);
      w.tgt[2] = tgt(rr, rr, 1.f, nullptr, 0.f, 0.375f);
      if ((rc = chain(n, 1, w)) != ODEHIP_OK) return rc;
      // stage 3: Y4 = y + dt (k1' - k2' + k3').   K3: A4 = Q4 + dt K3; R += 3dt/8 K3
      c.n_prev = 2;
      c.k_prev[1] = k[1];
      c.k_out = nullptr;
      c.c1[0] = 1.f;
      c.c1[1] = -1.f;
      c.c1[2] = 1.f;
      c.out1 = L.xin(ws, n, 3);
      if ((rc = run_f(n, 2, L.xin(ws, n, 2), c)) != ODEHIP_OK) return rc;
      w.n_targets = 2;
      w.tgt[0] = tgt(L.gp(ws, n, 3, NH), q4, 1.f, nullptr, 0.f, 1.f);
      w.tgt[1] = tgt(rr, rr, 1.f, nullptr, 0.f, 0.375f);
      if ((rc = chain(n, 2, w)) != ODEHIP_OK) return rc;
      // stage 4: only the activations at Y4 are needed (y is reset to the stored y[n]).   K4: a_next = R + dt/8 K4 + grad_out[n]
      memset(&c, 0, sizeof(c));
      c.k_scale = -1.0f;
      c.k_out = k[2];  // k4' itself is unused; the epilogue needs some destination
      if ((rc = run_f(n, 3, L.xin(ws, n, 3), c)) != ODEHIP_OK) return rc;
      w.n_targets = 1;
      w.tgt[0] = tgt(a_next, rr, 1.f, L.go(ws, n), 1.f, 0.125f);
      if ((rc = chain(n, 3, w)) != ODEHIP_OK) return rc;
    }
  }
  return ODEHIP_OK;
  };
  rc = sweep();
  const int rc2 = persist.finish(hbuf, hdev, nullptr, batch, (unsigned*)L.p(ws, L.off_psync), f->ks, stream);
  if (rc != ODEHIP_OK || rc2 != ODEHIP_OK) return rc != ODEHIP_OK ? rc : rc2;
  rc = odehip_q4_to_nchw(a_final, grad_z0_nchw, batch, L.C, stream);
  if (rc != ODEHIP_OK) return rc;
  rc = wgrad_all_layers(f, L, ws, n_times, batch, /*adjoint=*/true, scales, grad_w, grad_b, stream);
  if (rc != ODEHIP_OK) return rc;
  return guard_gradients(persist, f, batch, grad_z0_nchw, grad_w, grad_b, stream);
}
