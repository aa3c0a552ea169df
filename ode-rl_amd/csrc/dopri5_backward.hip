// dopri5_backward.hip -- `loss.backward()` through the adaptive solver (the reference's default training path:
// configs.yaml:79 decode_diff_method 'dopri5', modules/DiffEqSolver.py:9 plain `odeint`, train_test.py:204).
//
// Autograd through torchdiffeq differentiates the arithmetic of the ACCEPTED steps (rejected attempts never reach the
// output; `_optimal_step_size` runs under torch.no_grad(), so step sizes are constants of the graph).  This file
// computes exactly that gradient by an explicit reverse sweep:
//   1. the accepted steps (t0_n, dt_n) logged by the forward controller (dopri5.hip) are re-integrated from z0 with the
//      ReLU outputs of every stage kept (no controller, no host synchronisation);
//   2. steps are walked backwards.  With value(x) = y0 + h*sum_s W_s(x) k_s for the dense output at x (W_s: the quartic
//      of _interp_fit as a linear form in the stages) and Y_s = y0 + h*sum_j beta_sj k_j, k_s = f(Y_s), y1 = Y_7:
//        seed_s = sum_i h W_s(x_i) G_i + h*sum_{s'>s} beta_{s's} gY_{s'}   (+ the gradient of the next step's k1 for s = 7)
//        gY_s   = J_f(Y_s)^T seed_s  (+ gy1 for s = 7)         -- dgrad chain on the MFMA conv kernels, ReLU mask fused
//        gy0    = sum_i G_i + sum_{s>=2} gY_s ;  gk_1 goes to the previous step's k7 (FSAL) or, at n = 0, through f(z0);
//   3. one wgrad launch per layer sums the weight gradients of all 6N+1 stage evaluations (wgrad.hip).
// Not differentiated: the dependence of the FIRST step size on y0/theta through _select_initial_step (a term of the
// size of the local error; absent altogether with options={'first_step': dt}).
#include <math.h>
#include <string.h>

#include <vector>

#include "odehip_internal.h"
#include "persist.h"
#include "dopri5_layout.h"

namespace odehip {

typedef float f32x4 __attribute__((ext_vector_type(4)));



}  // namespace odehip

using namespace odehip;

extern "C" size_t odehip_dopri5_backward_workspace_bytes(const odehip_convstack* f, int batch, int n_times, int n_steps) {
  if (!f || batch <= 0 || n_times <= 0 || n_steps < 0 || f->n_convs < 1) return 0;
  return BwdLayout(f, batch, n_times, n_steps).total;
}

// layout_steps: the number of slots the workspace was laid out for (n_steps, or the saving forward's max_accept); saved: the
// slots already hold the stage inputs and hidden activations of the accepted steps (odehip_odeint_dopri5_saving): no re-integration
static int dopri5_backward(const odehip_convstack* f, const odehip_convstack* f_dgrad, const double* t_host, int n_times, int batch,
                           const double* accepted_host, int n_steps, const float* z0_nchw, const float* grad_out_nchw,
                           float* grad_z0_nchw, float* const* grad_w, float* const* grad_b, void* workspace, size_t workspace_bytes,
                           void* stream_, int layout_steps, bool saved);

extern "C" int odehip_odeint_dopri5_backward(const odehip_convstack* f, const odehip_convstack* f_dgrad, const double* t_host,
                                             int n_times, int batch, const double* accepted_host, int n_steps,
                                             const float* z0_nchw, const float* grad_out_nchw, float* grad_z0_nchw,
                                             float* const* grad_w, float* const* grad_b, void* workspace, size_t workspace_bytes,
                                             void* stream_) {
  ODEHIP_REQUIRE(z0_nchw, "odeint_dopri5_backward: null pointer");
  return dopri5_backward(f, f_dgrad, t_host, n_times, batch, accepted_host, n_steps, z0_nchw, grad_out_nchw, grad_z0_nchw, grad_w, grad_b,
                         workspace, workspace_bytes, stream_, n_steps, false);
}

extern "C" int odehip_odeint_dopri5_backward_saved(const odehip_convstack* f, const odehip_convstack* f_dgrad, const double* t_host,
                                                   int n_times, int batch, const double* accepted_host, int n_steps,
                                                   const float* grad_out_nchw, float* grad_z0_nchw, float* const* grad_w,
                                                   float* const* grad_b, int max_accept, void* saved_workspace,
                                                   size_t saved_workspace_bytes, void* stream_) {
  ODEHIP_REQUIRE(f && saved_workspace && max_accept > 0 && n_steps >= 1 && n_steps <= max_accept,
                 "odeint_dopri5_backward_saved: needs the saving forward's workspace and 1 <= n_steps <= max_accept");
  ODEHIP_REQUIRE(saved_workspace_bytes >= odehip_dopri5_saving_workspace_bytes(f, batch, n_times, max_accept),
                 "odeint_dopri5_backward_saved: workspace smaller than the saving forward's");
  char* bws = (char*)saved_workspace + al256(odehip_dopri5_workspace_bytes(f, batch, n_times));   // the backward half (dopri5.hip)
  return dopri5_backward(f, f_dgrad, t_host, n_times, batch, accepted_host, n_steps, nullptr, grad_out_nchw, grad_z0_nchw, grad_w, grad_b,
                         bws, saved_workspace_bytes - (size_t)(bws - (char*)saved_workspace), stream_, max_accept, true);
}

static int dopri5_backward(const odehip_convstack* f, const odehip_convstack* f_dgrad, const double* t_host, int n_times, int batch,
                           const double* accepted_host, int n_steps, const float* z0_nchw, const float* grad_out_nchw,
                           float* grad_z0_nchw, float* const* grad_w, float* const* grad_b, void* workspace, size_t workspace_bytes,
                           void* stream_, int layout_steps, bool saved) {
  int rc = check_stack(f);
  if (rc != ODEHIP_OK) return rc;
  ODEHIP_REQUIRE(f_dgrad && t_host && (z0_nchw || saved) && grad_out_nchw && grad_z0_nchw && grad_w && grad_b && workspace,
                 "odeint_dopri5_backward: null pointer");
  ODEHIP_REQUIRE(n_times >= 1 && batch > 0 && n_steps >= 0, "odeint_dopri5_backward: bad sizes");
  ODEHIP_REQUIRE(n_steps == 0 || accepted_host, "odeint_dopri5_backward: null step log");
  ODEHIP_REQUIRE(f->ks == 3 && f->channels[0] == f->channels[f->n_convs], "odeint_dopri5_backward: 3x3 C -> C dynamics only");
  for (int l = 0; l <= f->n_convs; ++l)
    ODEHIP_REQUIRE(f->channels[l] % 64 == 0, "odeint_dopri5_backward: channel counts must be multiples of 64");
  for (int i = 1; i < n_times; ++i)
    ODEHIP_REQUIRE(t_host[i] > t_host[i - 1], "odeint_dopri5_backward: t must be strictly increasing");
  ODEHIP_REQUIRE(6 * n_steps + 1 <= 32 * 64, "odeint_dopri5_backward: too many accepted steps (%d)", n_steps);
  const BwdLayout L(f, batch, n_times, layout_steps);
  ODEHIP_REQUIRE(workspace_bytes >= L.total, "odeint_dopri5_backward: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  void* ws = workspace;
  const int NH = L.NH, NL = f->n_convs, N = n_steps;
  const size_t st_b = (size_t)batch * L.C * kPix * 4;

  // the log must tile [t[0], >= t[T-1]] without gaps, as the controller produced it
  std::vector<int> j_lo(N + 1), j_hi(N + 1);
  {
    int j = 1;
    double t1 = t_host[0];
    for (int n = 0; n < N; ++n) {
      const double t0 = accepted_host[2 * n], dt = accepted_host[2 * n + 1];
      ODEHIP_REQUIRE(t0 == t1 && dt > 0, "odeint_dopri5_backward: step log is not contiguous at step %d", n);
      t1 = t0 + dt;
      j_lo[n] = j;
      while (j < n_times && t_host[j] <= t1) ++j;
      j_hi[n] = j;
    }
    ODEHIP_REQUIRE(j == n_times, "odeint_dopri5_backward: the step log does not reach t[%d]", n_times - 1);
  }

  float* hdev = L.p(ws, L.off_h);
  float* ping = L.p(ws, L.off_ping);
  float* pong = L.p(ws, L.off_pong);
  float* k[7];
  float* gY[7];
  for (int i = 0; i < 7; ++i) {
    k[i] = L.p(ws, L.off_k + (size_t)i * L.st);
    gY[i] = L.p(ws, L.off_gY + (size_t)i * L.st);
  }
  auto goq = [&](int n) { return L.p(ws, L.off_go + (size_t)n * L.st); };
  rc = odehip_nchw_to_q4(grad_out_nchw, goq(0), n_times * batch, L.C, stream);
  if (rc != ODEHIP_OK) return rc;
  if (N == 0) {  // a single output time: out[0] = z0
    ODEHIP_REQUIRE(n_times == 1 && !saved, "odeint_dopri5_backward: empty step log");
    ODEHIP_CHECK_HIP(hipMemcpyAsync(grad_z0_nchw, grad_out_nchw, st_b, hipMemcpyDeviceToDevice, stream));
    for (int l = 0; l < NL; ++l) {
      ODEHIP_CHECK_HIP(hipMemsetAsync(grad_w[l], 0, (size_t)f->channels[l + 1] * f->channels[l] * 9 * 4, stream));
      ODEHIP_CHECK_HIP(hipMemsetAsync(grad_b[l], 0, (size_t)f->channels[l + 1] * 4, stream));
    }
    return ODEHIP_OK;
  }
  {
    std::vector<float> hs(N);
    for (int n = 0; n < N; ++n) hs[n] = (float)accepted_host[2 * n + 1];
    if ((rc = upload_floats(hdev, hs.data(), N, stream)) != ODEHIP_OK) return rc;
  }

  // Everything below is a sequence of ROWS -- conv layers with fused epilogues and elementwise rows (ConvArgs::combine == 4) --
  // recorded by one PersistScope: ONE launch of the adaptive persistent walk runs the re-integration and the whole reverse sweep
  // (the table follows the accepted steps, so it is uploaded asynchronously rather than cached); when the walk is unavailable the
  // same rows are replayed as ordinary launches, with bit-identical results.
  const bool order1 = all_64(f);   // order-1 stage combines are what the adaptive walk takes (other stacks: one launch per layer)
  ODEHIP_CHECK_HIP(hipMemsetAsync(L.p(ws, L.off_gy), 0, st_b, stream));   // (before the rows: everything recorded runs at finish())
  if (!saved) {
    rc = odehip_nchw_to_q4(z0_nchw, L.xin(ws, 0, 0), batch, L.C, stream);
    if (rc != ODEHIP_OK) return rc;
  }
  int n_out_max = 1;
  for (int n = 0; n < N; ++n) n_out_max = j_hi[n] - j_lo[n] > n_out_max ? j_hi[n] - j_lo[n] : n_out_max;
  const int ew_per_seed = (n_out_max + 6 + ODEHIP_MAX_STAGES) / ODEHIP_MAX_STAGES + 1;
  PersistScope persist;
  if ((rc = persist.begin(f, f_dgrad, 2 * (6 * N + 1) * NL + (8 * N + 2) * ew_per_seed)) != ODEHIP_OK) return rc;
  persist.set_volatile_table(true);

  float* hidv[ODEHIP_MAX_LAYERS];
  auto run_f = [&](int n, int s, const float* x, const CombineArgs& c) {
    for (int l = 0; l < NH; ++l) hidv[l] = L.hidden(ws, n, s, l);
    return enqueue_f_saving(f, x, batch, hidv, ping, pong, &c, nullptr, nullptr, stream);
  };
  // out = sum_j c[j] * src[j] as elementwise rows of at most ODEHIP_MAX_STAGES sources (later rows accumulate); zero if empty
  auto axpy = [&](float* out, std::vector<const float*>& src, std::vector<float>& c) -> int {
    size_t o = 0;
    bool acc = false;
    do {
      ConvArgs a;
      memset(&a, 0, sizeof(a));
      a.combine = 4;
      a.qout = L.C / 4;
      a.batch = batch;
      a.cmb.order = 1;
      a.cmb.y = acc ? out : nullptr;
      a.cmb.out1 = out;
      const size_t m = src.size() - o < (size_t)ODEHIP_MAX_STAGES ? src.size() - o : (size_t)ODEHIP_MAX_STAGES;
      a.cmb.n_prev = (int)m;
      for (size_t j = 0; j < m; ++j) {
        a.cmb.k_prev[j] = src[o + j];
        a.cmb.c1[j] = c[o + j];
      }
      int r = launch_conv(a, f->ks, stream);
      if (r != ODEHIP_OK) return r;
      o += m;
      acc = true;
    } while (o < src.size());
    return ODEHIP_OK;
  };

  // ---- 1. re-integrate the accepted steps, keeping activations ------------------------------------------------------------
  // The stage-2 input of a step, Y_2 = y0 + h*beta21*k1, rides in the epilogue of the evaluation that produces k1: f(z0) for the
  // first step, the previous step's stage 7 (FSAL) for the others.
  CombineArgs c;
  memset(&c, 0, sizeof(c));
  c.k_scale = 1.0f;
  c.order = order1;
  c.k_out = k[0];
  c.y = L.xin(ws, 0, 0);
  c.h_ptr = hdev;
  c.c1[0] = (float)dp5::kBeta[0][0];
  c.out1 = L.xin(ws, 0, 1);
  if (!saved && (rc = run_f(0, 0, L.xin(ws, 0, 0), c)) != ODEHIP_OK) return rc;  // k1 of the first step (+ its Y_2)
  for (int n = 0; n < N && !saved; ++n) {
    const float* y0 = n == 0 ? L.xin(ws, 0, 0) : L.xin(ws, n - 1, 6);
    for (int s = 2; s <= 7; ++s) {
      memset(&c, 0, sizeof(c));
      c.k_scale = 1.0f;
      c.order = order1;
      c.k_out = k[s - 1];
      if (s <= 6) {
        c.y = y0;
        c.h_ptr = hdev + n;
        c.n_prev = s - 1;
        for (int j = 0; j < s - 1; ++j) c.k_prev[j] = k[j];
        for (int j = 0; j < s; ++j) c.c1[j] = (float)dp5::kBeta[s - 1][j];
        c.out1 = L.xin(ws, n, s);  // Y_{s+1}; s = 6: y1
      } else if (n + 1 < N) {      // k7 = k1 of the next step: its Y_2 = y1 + h_{n+1}*beta21*k7
        c.y = L.xin(ws, n, 6);
        c.h_ptr = hdev + n + 1;
        c.c1[0] = (float)dp5::kBeta[0][0];
        c.out1 = L.xin(ws, n + 1, 1);
      }
      if ((rc = run_f(n, s - 1, L.xin(ws, n, s - 1), c)) != ODEHIP_OK) return rc;
    }
    float* t = k[0]; k[0] = k[6]; k[6] = t;  // FSAL
  }

  // ---- 2. reverse sweep ------------------------------------------------------------------------------------------------------
  auto chain = [&](int n, int s, const BwdArgs& last) -> int {  // J_f(Y)^T seed, seed = gp[n][s][NH]
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
  float* gy = L.p(ws, L.off_gy);                 // gradient w.r.t. the state at the end of the step being processed
  float* gy_new = L.p(ws, L.off_gy + L.st);
  float* gk1 = L.p(ws, L.off_gk1);               // gradient w.r.t. k1 of the step after the one being processed (= its k7)
  float* gk1_new = L.p(ws, L.off_gk1 + L.st);
  bool have_gk1 = false;
  std::vector<const float*> src;
  std::vector<float> cf;
  for (int n = N - 1; n >= 0; --n) {
    const double t0 = accepted_host[2 * n], t1 = t0 + accepted_host[2 * n + 1];
    const float h = (float)accepted_host[2 * n + 1];
    const int nout = j_hi[n] - j_lo[n];
    std::vector<float> W((size_t)nout * 7);
    for (int i = 0; i < nout; ++i) {
      const float x = (float)((t_host[j_lo[n] + i] - t0) / (t1 - t0));
      for (int s = 0; s < 7; ++s) {
        W[(size_t)i * 7 + s] = h * (float)dp5::dense_weight(s, (double)x);
      }
    }
    for (int s = 7; s >= 1; --s) {  // stage s lives at index s-1
      src.clear();
      cf.clear();
      for (int i = 0; i < nout; ++i)
        if (W[(size_t)i * 7 + s - 1] != 0.0f) {
          src.push_back(goq(j_lo[n] + i));
          cf.push_back(W[(size_t)i * 7 + s - 1]);
        }
      for (int sp = s + 1; sp <= 7; ++sp) {  // Y_{sp} = y0 + h*sum_j beta[sp-2][j] k_{j+1}
        const float b = (float)dp5::kBeta[sp - 2][s - 1];
        if (b != 0.0f) {
          src.push_back(gY[sp - 1]);
          cf.push_back(h * b);
        }
      }
      if (s == 7 && have_gk1) {
        src.push_back(gk1);
        cf.push_back(1.0f);
      }
      if (s >= 2) {
        if ((rc = axpy(L.gp(ws, n, s - 1, NH), src, cf)) != ODEHIP_OK) return rc;
        BwdArgs w;
        memset(&w, 0, sizeof(w));
        w.n_targets = 1;
        w.tgt[0].out = gY[s - 1];
        w.tgt[0].g_c = 1.0f;
        if (s == 7) {  // Y_7 is also the step's result y1
          w.tgt[0].srcA = gy;
          w.tgt[0].a_c = 1.0f;
        }
        if ((rc = chain(n, s - 1, w)) != ODEHIP_OK) return rc;
      } else if (n > 0) {
        if ((rc = axpy(gk1_new, src, cf)) != ODEHIP_OK) return rc;  // k1 of this step is k7 of the previous one
      } else {
        if ((rc = axpy(L.gp(ws, 0, 0, NH), src, cf)) != ODEHIP_OK) return rc;  // k1 = f(z0)
      }
    }
    // gradient w.r.t. y0 of this step
    src.clear();
    cf.clear();
    for (int i = 0; i < nout; ++i) {
      src.push_back(goq(j_lo[n] + i));
      cf.push_back(1.0f);
    }
    for (int s = 2; s <= 7; ++s) {
      src.push_back(gY[s - 1]);
      cf.push_back(1.0f);
    }
    if ((rc = axpy(gy_new, src, cf)) != ODEHIP_OK) return rc;
    float* t = gy; gy = gy_new; gy_new = t;
    t = gk1; gk1 = gk1_new; gk1_new = t;
    have_gk1 = true;
  }
  {  // through k1 = f(z0), plus out[0] = z0
    BwdArgs w;
    memset(&w, 0, sizeof(w));
    w.n_targets = 1;
    w.tgt[0].out = gy_new;
    w.tgt[0].g_c = 1.0f;
    w.tgt[0].srcA = gy;
    w.tgt[0].a_c = 1.0f;
    w.tgt[0].srcB = goq(0);
    w.tgt[0].b_c = 1.0f;
    if ((rc = chain(0, 0, w)) != ODEHIP_OK) return rc;
  }
  if ((rc = persist.finish(nullptr, nullptr, nullptr, batch, (unsigned*)L.p(ws, L.off_psync), f->ks, stream)) != ODEHIP_OK) return rc;
  rc = odehip_q4_to_nchw(gy_new, grad_z0_nchw, batch, L.C, stream);
  if (rc != ODEHIP_OK) return rc;

  // ---- 3. weight / bias gradients: one launch per layer over all stage evaluations; the NL tables travel in ONE staged upload
  const int n_eval = 6 * N + 1;
  WgradPair* table = (WgradPair*)L.p(ws, L.off_tab);
  float* slabs = L.p(ws, L.off_slab);
  std::vector<WgradPair> host((size_t)n_eval * NL);
  for (int l = 0; l < NL; ++l) {
    int e = 0;
    for (int n = 0; n < N; ++n)
      for (int s = (n == 0 ? 0 : 1); s < 7; ++s, ++e) {
        WgradPair& hp = host[(size_t)l * n_eval + e];
        hp.g = L.gp(ws, n, s, l);
        hp.a = l == 0 ? L.xin(ws, n, s) : L.hidden(ws, n, s, l - 1);
        hp.scale = 1.0f;
        hp.pad_[0] = hp.pad_[1] = hp.pad_[2] = 0.0f;
      }
  }
  if ((rc = staged_upload(table, host.data(), host.size() * sizeof(WgradPair), stream)) != ODEHIP_OK) return rc;
  for (int l = 0; l < NL; ++l) {
    rc = launch_wgrad(table + (size_t)l * n_eval, n_eval, batch, f->w_bf16[l] ? 4 : wgrad_esplit(batch, n_eval), slabs, grad_w[l], grad_b[l], f->channels[l + 1], f->channels[l], stream,
                      f->w_bf16[l] != nullptr);
    if (rc != ODEHIP_OK) return rc;
  }
  // a give-up of the walk must not hand plausible numbers to the caller: NaN-fill what this call returns
  float* regions[2 * ODEHIP_MAX_LAYERS + 1];
  size_t floats[2 * ODEHIP_MAX_LAYERS + 1];
  int nr = 0;
  regions[nr] = grad_z0_nchw; floats[nr++] = st_b / 4;
  for (int l = 0; l < NL; ++l) {
    regions[nr] = grad_w[l]; floats[nr++] = (size_t)f->channels[l + 1] * f->channels[l] * 9;
    regions[nr] = grad_b[l]; floats[nr++] = (size_t)f->channels[l + 1];
  }
  return persist.guard(regions, floats, nr, stream);
}
