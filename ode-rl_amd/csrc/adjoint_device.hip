// adjoint_device.hip -- the adaptive (dopri5) adjoint of adjoint_dopri5.hip with the step control ON THE DEVICE and every
// evaluation on the adaptive persistent walk (BASELINE.json configs[2]; torchdiffeq `odeint_adjoint(..., method="dopri5",
// adjoint_options={"norm": "seminorm"})`, _impl/adjoint.py).  Same arithmetic and the same decisions as the host loop there:
// per interval t[n+1] -> t[n] a fresh solve of the augmented state (y, a) on the flipped time axis -- k1, the initial-step search
// (_select_initial_step: d0, d1 -> h0; Euler point; d2 -> dt), attempted steps with the error ratio max(rms_y, rms_a), accept /
// reject and step-size update of _adaptive_step / _optimal_step_size, dense output at t[n] -- but the host never reads a verdict:
//
//   * ONE table (cached by content: it is the same for every call on a workspace) holds four row programs -- P3: dense output of an
//     interval's last step (+ grad_out[n]); P0: k1 of a fresh interval; P1: the Euler point of the initial-step search; P2: an attempted
//     step (two elementwise rows for the stage-2 inputs + six augmented evaluations = 62 rows) -- written with RELOCATABLE pointers
//     (class << 56 | offset, conv_wino.hip rel()) wherever a buffer depends on a decision: the slot an accepted step keeps for the
//     final weight-gradient launch, the state pointers (y, a) that move with FSAL, the k1 / k7 buffers that swap;
//   * a TICK = adj_control_kernel (one workgroup: digests the sums of the previous tick -- norms or error partials in fixed order --
//     takes the decision in torchdiffeq's types, appends the accepted step's stage evaluations to the per-layer weight-gradient
//     tables, writes {first row, rows} of the next program and the relocation bases, zeroes the walk's flag area) -> the walk
//     (the scaled sums of squares of the initial-step search are NORM ROWS at the end of P0 / P1: per-wave partials, no launch);
//   * the host only bounds its run-ahead by polling a pinned mailbox (one tick ahead) and, after `done`, launches the weight
//     gradients (one launch per layer over every recorded stage evaluation, as before) with the entry count it reads there.
// A training step at B=64, T=10 is 27 ticks instead of ~720 per-layer launches and 27 stream synchronisations.
#include <math.h>
#include <string.h>
#include <time.h>

#include <vector>

#include "adjoint_layout.h"
#include "odehip_internal.h"
#include "persist.h"

namespace odehip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { PH_INIT = 0, PH_P0 = 1, PH_P1 = 2, PH_P2 = 3, PH_FINAL = 4, PH_IDLE = 5 };
// relocation classes
enum { RC_SLOT = 1, RC_Y = 2, RC_A = 3, RC_KYA = 4, RC_KYB = 5, RC_KAA = 6, RC_KAB = 7, RC_ANEXT = 8, RC_GO = 9, RC_APREV = 10 };
constexpr int kAdjMaxTimes = 256;

struct AdjCtl {  // device-resident controller state
  // ---- constants of the call
  float rtol, atol;
  int n_times, n_part, max_slots, n_layers, nh, wtab_cap;
  double n_elems;
  unsigned long long slots_base, slot_bytes, st_bytes, hid_bytes;
  unsigned long long yq_base, goq_base, a2_base, ky_base, ka_base, wtab_base;
  double csol[7], cmid[7];
  int rows_p3p0[2], rows_p0[2], rows_p3[2], rows_p1[2], rows_p2[2];
  // ---- solver state
  int phase;   // program of the walk of THIS tick (what the next controller call digests)
  int n;       // interval t[n+1] -> t[n]
  int slot, parity;
  double t_cur, t_end, dt;
  float h, h0, d1;
  int k1_slot, k1_stage;
  unsigned long long k1_x0;
  int nfe, n_accept, n_reject, n_entries, done, status, ticks;
  float w[8];                    // weights of the stages in the last accepted step's contribution (device coefficients of P3)
  int rows[2];                   // {first row, rows} of this tick's walk
  unsigned long long reloc[16];  // bases of the relocation classes
  double t[kAdjMaxTimes];
};
static_assert(sizeof(AdjCtl) <= 4096, "the controller state has 4096 bytes of workspace");

struct AdjMailbox {  // pinned host memory
  volatile int ticks, done, status, nfe, n_accept, n_reject, n_entries;
  int pad_[9];
};

// Fixed-order sums of up to four arrays of n floats at once: wave w adds array w (lane i takes elements i, i + 64, ...; then a
// butterfly over the lanes) -- one pass instead of four block reductions behind each other.  Valid in every thread afterwards.
__device__ void adj_sums4(const float* base, int stride, const int* which, int count, int n, float* sh, float* out) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (w < count) {
    const float* v = base + (size_t)which[w] * stride;
    // four independent chains: a lane's 64 strided loads (n = 64 partials per sample x 64 samples) were one dependent chain of L2
    // round trips on the critical path of every tick
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int i = lane;
    for (; i + 192 < n; i += 256) {
      s0 += v[i];
      s1 += v[i + 64];
      s2 += v[i + 128];
      s3 += v[i + 192];
    }
    for (; i < n; i += 64) s0 += v[i];
    float s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) sh[w] = s;
  }
  __syncthreads();
  for (int j = 0; j < count; ++j) out[j] = sh[j];
  __syncthreads();
}

__device__ double adj_dense_weight(const AdjCtl* st, int s, double x) {  // dp5::dense_weight
  const double d1 = s == 0 ? 1.0 : 0.0, d7 = s == 6 ? 1.0 : 0.0, b = st->csol[s], m = st->cmid[s];
  const double A4 = 2.0 * (d7 - d1) - 8.0 * b + 16.0 * m;
  const double B3 = 5.0 * d1 - 3.0 * d7 + 14.0 * b - 32.0 * m;
  const double C2 = d7 - 4.0 * d1 - 5.0 * b + 16.0 * m;
  return x * d1 + x * x * C2 + x * x * x * B3 + x * x * x * x * A4;
}

__device__ void adj_decide(AdjCtl* st, float s0, float s1, float s2, float s3, AdjMailbox* mb);

__global__ __launch_bounds__(256) void adj_control_kernel(AdjCtl* st_global, const float* parts, int part_stride, unsigned* psync,
                                                          int psync_words, AdjMailbox* mb) {
  __shared__ float sh[256];
  // the state is worked on in LDS: thread 0's decision logic touches ~100 of its fields one after the other -- from global memory that
  // was 11 us per tick (a dependent round trip each), from LDS it is the block sums that set the time
  __shared__ AdjCtl st_lds;
  static_assert(sizeof(AdjCtl) % 16 == 0, "copied as 16-byte words");
  for (int i = threadIdx.x; i < (int)(sizeof(AdjCtl) / 16); i += 256) ((f32x4*)&st_lds)[i] = ((const f32x4*)st_global)[i];
  for (int i = threadIdx.x; i < psync_words; i += 256) psync[i] = 0u;  // the flag area of this tick's walk
  __syncthreads();
  AdjCtl* const st = &st_lds;
  const int phase = st->phase;
  float sv[4] = {0.f, 0.f, 0.f, 0.f};
  if (!st->done) {   // (uniform) the sums the walk of the previous tick left: per-wave partials, added in a fixed order
    const int np = st->n_part;
    const int w0[4] = {0, 1, 2, 3}, w2[4] = {4, 5, 0, 0};
    if (phase == PH_P0) adj_sums4(parts, part_stride, w0, 4, np, sh, sv);
    else if (phase == PH_P1) adj_sums4(parts, part_stride, w0, 2, np, sh, sv);
    else if (phase == PH_P2) adj_sums4(parts, part_stride, w2, 2, np, sh, sv);
  }
  if (threadIdx.x == 0) adj_decide(st, sv[0], sv[1], sv[2], sv[3], mb);
  __syncthreads();
  constexpr int kLive = (int)(offsetof(AdjCtl, t) / 16);   // everything in front of the (constant) time grid goes back
  static_assert(offsetof(AdjCtl, t) % 16 == 0, "copied as 16-byte words");
  for (int i = threadIdx.x; i < kLive; i += 256) ((f32x4*)st_global)[i] = ((const f32x4*)&st_lds)[i];
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store((int*)&mb->ticks, st->ticks, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the decision logic of a tick (one thread, state in LDS)
__device__ void adj_decide(AdjCtl* st, float s0, float s1, float s2, float s3, AdjMailbox* mb) {
  const int phase = st->phase;
  const double N = st->n_elems;
  auto rms = [&](float s) { return sqrtf((float)((double)s / N)); };
  const unsigned long long ST = st->st_bytes;
  auto set_k = [&]() {  // the k1 / k7 roles of the stage-derivative buffers 0 and 6
    const int a = st->parity ? 6 : 0, b = st->parity ? 0 : 6;
    st->reloc[RC_KYA] = st->ky_base + (unsigned long long)a * ST;
    st->reloc[RC_KYB] = st->ky_base + (unsigned long long)b * ST;
    st->reloc[RC_KAA] = st->ka_base + (unsigned long long)a * ST;
    st->reloc[RC_KAB] = st->ka_base + (unsigned long long)b * ST;
  };
  auto slot_addr = [&](int slot) { return st->slots_base + (unsigned long long)slot * st->slot_bytes; };
  auto start_interval = [&](int n, unsigned long long a_cur, bool with_p3) {  // k1 of a fresh solve over t[n+1] -> t[n]
    st->n = n;
    st->reloc[RC_Y] = st->yq_base + (unsigned long long)(n + 1) * ST;
    st->reloc[RC_A] = a_cur;
    st->reloc[RC_SLOT] = slot_addr(st->slot);
    st->k1_slot = st->slot;
    st->k1_stage = 0;
    st->k1_x0 = st->reloc[RC_Y];
    set_k();
    st->phase = PH_P0;
    st->rows[0] = with_p3 ? st->rows_p3p0[0] : st->rows_p0[0];
    st->rows[1] = with_p3 ? st->rows_p3p0[1] : st->rows_p0[1];
  };
  auto start_attempt = [&]() {
    if (st->slot >= st->max_slots) { st->status = ODEHIP_EINVAL; return false; }      // more than max_accept accepted steps
    if (!(st->t_cur + st->dt > st->t_cur)) { st->status = ODEHIP_ENOTCONV; return false; }   // "underflow in dt"
    st->h = (float)st->dt;
    st->reloc[RC_SLOT] = slot_addr(st->slot);
    st->phase = PH_P2;
    st->rows[0] = st->rows_p2[0];
    st->rows[1] = st->rows_p2[1];
    return true;
  };
  bool ok = true;
  if (st->done) {
    st->phase = PH_IDLE;
    st->rows[0] = st->rows[1] = 0;
  } else if (phase == PH_INIT) {
    st->slot = 0;
    st->parity = 0;
    if (st->slot >= st->max_slots) { st->status = ODEHIP_EINVAL; ok = false; }
    else start_interval(st->n_times - 2, st->goq_base + (unsigned long long)(st->n_times - 1) * ST, false);
  } else if (phase == PH_P0) {  // _select_initial_step, first half
    const float d0 = fmaxf(rms(s0), rms(s1)), d1 = fmaxf(rms(s2), rms(s3));
    st->h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
    st->d1 = d1;
    st->nfe += 1;
    st->phase = PH_P1;
    st->rows[0] = st->rows_p1[0];
    st->rows[1] = st->rows_p1[1];
  } else if (phase == PH_P1) {  // second half: d2 -> dt; first attempt
    const float h0 = st->h0, d1 = st->d1;
    const float d2 = fmaxf(rms(s0), rms(s1)) / h0;
    float h1;
    if (d1 <= 1e-15f && d2 <= 1e-15f) h1 = fmaxf(1e-6f, h0 * 1e-3f);
    else h1 = powf(0.01f / fmaxf(d1, d2), 1.0f / 5.0f);
    st->nfe += 1;
    st->dt = (double)fminf(100.0f * h0, h1);
    st->t_cur = -st->t[st->n + 1];
    st->t_end = -st->t[st->n];
    ok = start_attempt();
  } else if (phase == PH_P2) {  // _adaptive_step's scalar part
    const float ratio = fmaxf(rms(s0), rms(s1));
    st->nfe += 6;
    if (!(ratio == ratio)) {
      st->status = ODEHIP_ENAN;
      ok = false;
    } else {
      const bool accept = ratio <= 1.0f;
      const double dt = st->dt;
      double dtn;
      if (ratio == 0.0f) {
        dtn = dt * 10.0;
      } else {
        const double dfactor = ratio < 1.0f ? 1.0 : 0.2;
        dtn = dt * fmin(10.0, fmax(0.9 / pow((double)ratio, 0.2), dfactor));
      }
      if (!accept) {
        st->n_reject += 1;
        st->dt = dtn;
        ok = start_attempt();
      } else {
        st->n_accept += 1;
        const float h = st->h;
        const double t_new = st->t_cur + dt;
        const bool final_step = t_new >= st->t_end;
        float w[7];
        if (!final_step) {
          for (int s = 0; s < 7; ++s) w[s] = (float)st->csol[s] * h;
        } else {
          const float x = (float)((st->t_end - st->t_cur) / (t_new - st->t_cur));
          for (int s = 0; s < 7; ++s) w[s] = h * (float)adj_dense_weight(st, s, (double)x);
        }
        // the accepted step's stage evaluations join the final weight-gradient sum (a_theta is linear in the stages)
        const unsigned long long sb = slot_addr(st->slot);
        WgradPair* wt = (WgradPair*)st->wtab_base;
        const int NH = st->nh, NL = st->n_layers;
        for (int s = 0; s < 7; ++s) {
          st->w[s] = w[s];
          if (w[s] == 0.0f) continue;
          if (st->n_entries >= st->wtab_cap) { st->status = ODEHIP_EINVAL; ok = false; break; }
          const unsigned long long eb = s == 0 ? slot_addr(st->k1_slot) : sb;
          const int es = s == 0 ? st->k1_stage : s;
          const unsigned long long x0 = s == 0 ? st->k1_x0 : sb + (unsigned long long)s * ST;
          for (int l = 0; l < NL; ++l) {
            WgradPair& p = wt[(size_t)l * st->wtab_cap + st->n_entries];
            p.g = (const float*)(eb + 7 * ST + 7 * (unsigned long long)NH * st->hid_bytes +
                                 ((unsigned long long)es * (NH + 1) + l) * st->hid_bytes);
            p.a = l == 0 ? (const float*)x0 : (const float*)(eb + 7 * ST + ((unsigned long long)es * NH + (l - 1)) * st->hid_bytes);
            p.scale = w[s];
            p.pad_[0] = p.pad_[1] = p.pad_[2] = 0.0f;
          }
          st->n_entries += 1;
        }
        if (ok && final_step) {  // a(t[n]) = dense output + grad_out[n]: program P3 in front of the next interval's k1
          const int n = st->n;
          const unsigned long long a_next = st->a2_base + (unsigned long long)(n & 1) * ST;
          st->reloc[RC_APREV] = st->reloc[RC_A];
          st->reloc[RC_ANEXT] = a_next;
          st->reloc[RC_GO] = st->goq_base + (unsigned long long)n * ST;
          set_k();   // (unchanged: P3 reads the seven stage derivatives of the step just accepted)
          st->slot += 1;
          if (n == 0) {
            st->phase = PH_FINAL;
            st->rows[0] = st->rows_p3[0];
            st->rows[1] = st->rows_p3[1];
            st->done = 1;
          } else if (st->slot >= st->max_slots) {
            st->status = ODEHIP_EINVAL;
            ok = false;
          } else {
            start_interval(n - 1, a_next, true);
          }
        } else if (ok) {  // FSAL: (y, a, k1) <- (y1, a1, k7)
          st->reloc[RC_Y] = sb + 6 * ST;
          st->reloc[RC_A] = sb + 7 * ST + 7 * (unsigned long long)NH * st->hid_bytes + ((unsigned long long)6 * (NH + 1) + NH) * st->hid_bytes;
          st->parity ^= 1;
          set_k();
          st->k1_slot = st->slot;
          st->k1_stage = 6;
          st->k1_x0 = sb + 6 * ST;
          st->t_cur = t_new;
          st->dt = dtn;
          st->slot += 1;
          ok = start_attempt();
        }
      }
    }
  }
  if (!ok) {
    st->done = 1;
    st->phase = PH_IDLE;
    st->rows[0] = st->rows[1] = 0;
  }
  st->ticks += 1;
  mb->nfe = st->nfe;
  mb->n_accept = st->n_accept;
  mb->n_reject = st->n_reject;
  mb->n_entries = st->n_entries;
  mb->status = st->status;
  mb->done = st->done;
}

__global__ void adj_init_kernel(AdjCtl* st, AdjCtl init) { *st = init; }

static AdjMailbox* g_adj_mailbox = nullptr;

static double adj_now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int adjoint_dopri5_device(const odehip_convstack* f, const odehip_convstack* f_dgrad, const double* t_host, int n_times, int batch, float rtol,
                          float atol, const float* y_traj_nchw, const float* grad_out_nchw, float* grad_z0_nchw, float* const* grad_w,
                          float* const* grad_b, int max_accept, int* stats_host, void* workspace, size_t workspace_bytes, hipStream_t stream,
                          int* ran) {
  *ran = 0;
  // read per call (not cached): the parity tests run the host-driven loop and this one in ONE process on the same inputs
  const char* const env_e = getenv("ODEHIP_ADJOINT_DEVICE");
  const bool env_off = env_e && env_e[0] == '0';
  if (env_off || !all_64(f) || n_times < 2 || n_times > kAdjMaxTimes || g_debug_flags) return ODEHIP_OK;
  const AdjLayout L(f, batch, n_times, max_accept);
  ODEHIP_REQUIRE(workspace_bytes >= L.total, "odeint_adjoint_dopri5_backward: workspace too small");
  void* ws = workspace;
  const int NH = L.NH, NL = f->n_convs;
  const size_t st_b = (size_t)batch * L.C * kPix * 4;
  int rc;

  const int n_pps = persist_partials_per_sample(batch);   // (before the scope: it takes the library's lock)
  PersistScope persist;
  if ((rc = persist.begin(f, f_dgrad, 2 + 15 + 14 + 62)) != ODEHIP_OK) return rc;
  if (!persist.recording()) return ODEHIP_OK;   // no persistent walk on this device / switched off: the host loop takes the call

  if (!g_adj_mailbox) ODEHIP_CHECK_HIP(hipHostMalloc((void**)&g_adj_mailbox, sizeof(AdjMailbox), hipHostMallocCoherent));
  memset((void*)g_adj_mailbox, 0, sizeof(AdjMailbox));

  AdjCtl* state = (AdjCtl*)L.p(ws, L.off_state);
  unsigned* psync = (unsigned*)L.p(ws, L.off_psync);
  float* ping = L.p(ws, L.off_ping);
  float* pong = L.p(ws, L.off_pong);
  float* ky[7];
  float* ka[7];
  for (int i = 0; i < 7; ++i) {
    ky[i] = L.p(ws, L.off_ky + (size_t)i * L.st);
    ka[i] = L.p(ws, L.off_ka + (size_t)i * L.st);
  }
  const int part_stride = L.part_stride();

  rc = odehip_nchw_to_q4(y_traj_nchw, L.p(ws, L.off_y), n_times * batch, L.C, stream);
  if (rc != ODEHIP_OK) return rc;
  rc = odehip_nchw_to_q4(grad_out_nchw, L.p(ws, L.off_go), n_times * batch, L.C, stream);
  if (rc != ODEHIP_OK) return rc;

  // ---- the table: four row programs with relocatable pointers ----------------------------------------------------------------
  auto tag = [](int cls, size_t off) { return (float*)(((unsigned long long)cls << 56) | (unsigned long long)off); };
  float* const Ycur = tag(RC_Y, 0);
  float* const Acur = tag(RC_A, 0);
  float* const kyA = tag(RC_KYA, 0);
  float* const kyB = tag(RC_KYB, 0);
  float* const kaA = tag(RC_KAA, 0);
  float* const kaB = tag(RC_KAB, 0);
  auto s_xin = [&](int s) { return tag(RC_SLOT, adj_off_xin(L, s)); };
  auto s_hid = [&](int s, int l) { return tag(RC_SLOT, adj_off_hidden(L, s, l)); };
  auto s_gp = [&](int s, int l) { return tag(RC_SLOT, adj_off_gp(L, s, l)); };
  int n_rows = 0;
  auto ew = [&](float* out, const float* y, int n, const float* const* k, const float* c, const float* h_ptr, const float* c_dev) -> int {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.combine = 4;
    a.qout = L.C / 4;
    a.batch = batch;
    a.cmb.order = 1;
    a.cmb.y = y;
    a.cmb.out1 = out;
    a.cmb.n_prev = n;
    for (int j = 0; j < n; ++j) {
      a.cmb.k_prev[j] = k[j];
      a.cmb.c1[j] = c ? c[j] : 0.0f;
    }
    a.cmb.h_ptr = h_ptr;
    a.cmb.c_dev = c_dev;
    ++n_rows;
    return launch_conv(a, f->ks, stream);
  };
  // augmented dynamics at (Y, A = gp(slot, s, NH)) of stage s of the current slot: K^y = -f(Y) through `cy`, K^a = J^T A through `ca`
  auto eval_aug = [&](int s, const float* Y, const CombineArgs& cy, const CombineArgs& ca) -> int {
    float* hidv[ODEHIP_MAX_LAYERS];
    for (int l = 0; l < NH; ++l) hidv[l] = s_hid(s, l);
    int r = enqueue_f_saving(f, Y, batch, hidv, ping, pong, &cy, nullptr, nullptr, stream);
    if (r != ODEHIP_OK) return r;
    float* gpv[ODEHIP_MAX_LAYERS];
    for (int l = 0; l < NL; ++l) gpv[l] = s_gp(s, l);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.combine = 1;
    a.cmb = ca;
    n_rows += 2 * NL;
    return enqueue_dgrad_chain(f, f_dgrad, batch, gpv, hidv, a, stream);
  };
  // The two halves of an augmented evaluation as ROWS handed back to the caller (captured with a private recorder) instead of
  // appended: program P2 weaves the input-gradient chain of stage s with the forward chain of stage s + 1 (they are independent).
  ConvArgs cap_buf[2 * ODEHIP_MAX_LAYERS];
  auto capture = [&](std::vector<ConvArgs>& out, auto&& fn) -> int {
    ConvRecorder tmp = {cap_buf, 0, 2 * ODEHIP_MAX_LAYERS};
    ConvRecorder* const outer = g_conv_recorder;
    g_conv_recorder = &tmp;
    const int r = fn();
    g_conv_recorder = outer;
    out.assign(cap_buf, cap_buf + tmp.count);
    return r;
  };
  auto rows_f = [&](int s, const float* Y, const CombineArgs& cy_, std::vector<ConvArgs>& out) -> int {
    return capture(out, [&]() -> int {
      float* hidv[ODEHIP_MAX_LAYERS];
      for (int l = 0; l < NH; ++l) hidv[l] = s_hid(s, l);
      return enqueue_f_saving(f, Y, batch, hidv, ping, pong, &cy_, nullptr, nullptr, stream);
    });
  };
  auto rows_d = [&](int s, const CombineArgs& ca_, std::vector<ConvArgs>& out) -> int {
    return capture(out, [&]() -> int {
      float* hidv[ODEHIP_MAX_LAYERS];
      for (int l = 0; l < NH; ++l) hidv[l] = s_hid(s, l);
      float* gpv[ODEHIP_MAX_LAYERS];
      for (int l = 0; l < NL; ++l) gpv[l] = s_gp(s, l);
      ConvArgs a;
      memset(&a, 0, sizeof(a));
      a.combine = 1;
      a.cmb = ca_;
      return enqueue_dgrad_chain(f, f_dgrad, batch, gpv, hidv, a, stream);
    });
  };
  auto emit = [&](ConvArgs a, int dep_back) -> int {
    a.dep_back = dep_back;
    ++n_rows;
    return launch_conv(a, f->ks, stream);
  };
  // norm row: part(j) <- per-wave sums of ((a - b) / (atol + |y| rtol))^2
  auto norm_row = [&](int j, const float* a_, const float* b_, const float* y_) -> int {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.combine = 5;
    a.qout = L.C / 4;
    a.batch = batch;
    a.cmb.order = 1;
    a.cmb.y = y_;
    a.cmb.k_prev[0] = a_;
    a.cmb.k_prev[1] = b_;
    a.cmb.n_prev = b_ ? 2 : 1;
    a.cmb.err_partials = L.part(ws, j);
    a.cmb.rtol = rtol;
    a.cmb.atol = atol;
    ++n_rows;
    return launch_conv(a, f->ks, stream);
  };
  AdjCtl init;
  memset(&init, 0, sizeof(init));
  CombineArgs cy, ca;
  const float one = 1.0f;
  // -- P3: a_next = a_prev + sum_s w_s * K^a_s (+ grad_out[n])
  init.rows_p3[0] = init.rows_p3p0[0] = n_rows;
  {
    const float* kk[7] = {kaA, ka[1], ka[2], ka[3], ka[4], ka[5], kaB};
    if ((rc = ew(tag(RC_ANEXT, 0), tag(RC_APREV, 0), 7, kk, nullptr, nullptr, (const float*)((char*)state + offsetof(AdjCtl, w)))) != ODEHIP_OK) return rc;
    const float* gg[1] = {tag(RC_GO, 0)};
    if ((rc = ew(tag(RC_ANEXT, 0), tag(RC_ANEXT, 0), 1, gg, &one, nullptr, nullptr)) != ODEHIP_OK) return rc;
  }
  init.rows_p3[1] = n_rows - init.rows_p3[0];
  // -- P0: k1 = augmented dynamics at the interval start (stage 0 of the slot)
  init.rows_p0[0] = n_rows;
  if ((rc = ew(s_gp(0, NH), Acur, 0, nullptr, nullptr, nullptr, nullptr)) != ODEHIP_OK) return rc;
  memset(&cy, 0, sizeof(cy));
  memset(&ca, 0, sizeof(ca));
  cy.order = ca.order = 1;
  cy.k_scale = -1.0f;
  cy.k_out = kyA;
  ca.k_scale = 1.0f;
  ca.k_out = kaA;
  if ((rc = eval_aug(0, Ycur, cy, ca)) != ODEHIP_OK) return rc;
  // _select_initial_step: d0 over (y, a), d1 over (k1_y, k1_a)
  if ((rc = norm_row(0, Ycur, nullptr, Ycur)) != ODEHIP_OK || (rc = norm_row(1, Acur, nullptr, Acur)) != ODEHIP_OK ||
      (rc = norm_row(2, kyA, nullptr, Ycur)) != ODEHIP_OK || (rc = norm_row(3, kaA, nullptr, Acur)) != ODEHIP_OK)
    return rc;
  init.rows_p0[1] = n_rows - init.rows_p0[0];
  init.rows_p3p0[1] = n_rows - init.rows_p3p0[0];
  // -- P1: the Euler point y + h0 * k1 (stage 1 of the slot)
  init.rows_p1[0] = n_rows;
  {
    const float* h0p = (const float*)((char*)state + offsetof(AdjCtl, h0));
    const float* k0y[1] = {kyA};
    const float* k0a[1] = {kaA};
    if ((rc = ew(s_xin(1), Ycur, 1, k0y, &one, h0p, nullptr)) != ODEHIP_OK) return rc;
    if ((rc = ew(s_gp(1, NH), Acur, 1, k0a, &one, h0p, nullptr)) != ODEHIP_OK) return rc;
    cy.k_out = ky[1];
    ca.k_out = ka[1];
    if ((rc = eval_aug(1, s_xin(1), cy, ca)) != ODEHIP_OK) return rc;
    // d2 over (k(Euler point) - k1)
    if ((rc = norm_row(0, ky[1], kyA, Ycur)) != ODEHIP_OK || (rc = norm_row(1, ka[1], kaA, Acur)) != ODEHIP_OK) return rc;
  }
  init.rows_p1[1] = n_rows - init.rows_p1[0];
  // -- P2: an attempted step
  init.rows_p2[0] = n_rows;
  {
    const float* hp = (const float*)((char*)state + offsetof(AdjCtl, h));
    const float b21 = (float)dp5::kBeta[0][0];
    const float* k0y[1] = {kyA};
    const float* k0a[1] = {kaA};
    if ((rc = ew(s_xin(1), Ycur, 1, k0y, &b21, hp, nullptr)) != ODEHIP_OK) return rc;
    if ((rc = ew(s_gp(1, NH), Acur, 1, k0a, &b21, hp, nullptr)) != ODEHIP_OK) return rc;
    std::vector<ConvArgs> F[7], D[7];   // rows of evaluation e = s - 1 (stage s lives at index s-1 of the slot)
    for (int s = 2; s <= 7; ++s) {
      memset(&cy, 0, sizeof(cy));
      memset(&ca, 0, sizeof(ca));
      cy.order = ca.order = 1;
      cy.k_scale = -1.0f;
      ca.k_scale = 1.0f;
      cy.y = Ycur;
      ca.y = Acur;
      cy.h_ptr = ca.h_ptr = hp;
      cy.n_prev = ca.n_prev = s - 1;
      for (int j = 0; j < s - 1; ++j) {
        cy.k_prev[j] = j == 0 ? kyA : ky[j];
        ca.k_prev[j] = j == 0 ? kaA : ka[j];
      }
      cy.k_out = s == 7 ? kyB : ky[s - 1];
      ca.k_out = s == 7 ? kaB : ka[s - 1];
      if (s <= 6) {
        for (int j = 0; j < s; ++j) cy.c1[j] = ca.c1[j] = (float)dp5::kBeta[s - 1][j];
        cy.out1 = s_xin(s);       // Y_{s+1}  (s = 6: y1)
        ca.out1 = s_gp(s, NH);    // A_{s+1}  (s = 6: a1)
      } else {
        for (int j = 0; j < 7; ++j) cy.ce[j] = ca.ce[j] = (float)dp5::kCErr[j];
        cy.err_y1 = s_xin(6);
        ca.err_y1 = s_gp(6, NH);
        cy.err_partials = L.part(ws, 4);
        ca.err_partials = L.part(ws, 5);
        cy.rtol = ca.rtol = rtol;
        cy.atol = ca.atol = atol;
      }
      if ((rc = rows_f(s - 1, s_xin(s - 1), cy, F[s - 1])) != ODEHIP_OK) return rc;
      if ((rc = rows_d(s - 1, ca, D[s - 1])) != ODEHIP_OK) return rc;
      ODEHIP_REQUIRE((int)F[s - 1].size() == NL && (int)D[s - 1].size() == NL, "adjoint: unexpected row count of an evaluation");
    }
    // Order: F1 | (F2, D1) woven | (F3, D2) | ... | (F6, D5) | D6.  Y_{s+1} needs only the forward chains up to stage s and A_{s+1}
    // only the input-gradient chains, whose masks come from the SAME stage's forward chain: D_e and F_{e+1} are independent, so
    // in a woven pair every row's predecessor in its own chain lies two rows back (dep_back = 1: its input is loaded and transformed
    // while the consumers still multiply the other chain's row).
    static const bool weave = [] { const char* e = getenv("ODEHIP_ADJOINT_WEAVE"); return !(e && e[0] == '0'); }();
    if (weave) {
      for (int j = 0; j < NL; ++j)
        if ((rc = emit(F[1][j], 0)) != ODEHIP_OK) return rc;
      for (int e = 1; e <= 5; ++e)
        for (int j = 0; j < NL; ++j) {
          if ((rc = emit(F[e + 1][j], (e == 1 && j == 0) ? 0 : 1)) != ODEHIP_OK) return rc;   // F2's first row follows F1's last
          if ((rc = emit(D[e][j], 1)) != ODEHIP_OK) return rc;
        }
      for (int j = 0; j < NL; ++j)
        if ((rc = emit(D[6][j], 0)) != ODEHIP_OK) return rc;
    } else {
      for (int e = 1; e <= 6; ++e) {
        for (int j = 0; j < NL; ++j)
          if ((rc = emit(F[e][j], 0)) != ODEHIP_OK) return rc;
        for (int j = 0; j < NL; ++j)
          if ((rc = emit(D[e][j], 0)) != ODEHIP_OK) return rc;
      }
    }
  }
  init.rows_p2[1] = n_rows - init.rows_p2[0];

  // ---- controller state ---------------------------------------------------------------------------------------------------------
  init.rtol = rtol;
  init.atol = atol;
  init.n_times = n_times;
  init.n_part = batch * n_pps;   // partials per array: 16 per sample on the 4-workgroup walk, 64 on the sixteen-workgroup one
  init.max_slots = L.max_slots;
  init.n_layers = NL;
  init.nh = NH;
  init.wtab_cap = L.max_slots * 7;
  init.n_elems = (double)(st_b / 4);
  init.slots_base = (unsigned long long)(uintptr_t)L.p(ws, L.off_slots);
  init.slot_bytes = L.slot_bytes;
  init.st_bytes = L.st;
  init.hid_bytes = L.hid;
  init.yq_base = (unsigned long long)(uintptr_t)L.p(ws, L.off_y);
  init.goq_base = (unsigned long long)(uintptr_t)L.p(ws, L.off_go);
  init.a2_base = (unsigned long long)(uintptr_t)L.p(ws, L.off_a2);
  init.ky_base = (unsigned long long)(uintptr_t)ky[0];
  init.ka_base = (unsigned long long)(uintptr_t)ka[0];
  init.wtab_base = (unsigned long long)(uintptr_t)L.p(ws, L.off_wtab);
  for (int s = 0; s < 7; ++s) {
    init.csol[s] = dp5::kCSol[s];
    init.cmid[s] = dp5::kCMid[s];
  }
  for (int i = 0; i < n_times; ++i) init.t[i] = t_host[i];
  init.phase = PH_INIT;
  if ((rc = staged_upload(state, &init, sizeof(init), stream)) != ODEHIP_OK) return rc;

  persist.set_device_steering((const int*)((char*)state + offsetof(AdjCtl, rows)), (const unsigned long long*)((char*)state + offsetof(AdjCtl, reloc)));
  const int psync_words = (int)(persist_sync_bytes(batch) / 4);

  // ---- ticks: the host stays at most one tick ahead of the device ------------------------------------------------------------
  int enq = 0, seen = 0;
  double t_progress = adj_now_s();
  for (;;) {
    hipLaunchKernelGGL(adj_control_kernel, dim3(1), dim3(256), 0, stream, state, L.part(ws, 0), part_stride, psync, psync_words, g_adj_mailbox);
    ODEHIP_CHECK_HIP(hipGetLastError());
    if (enq == 0) rc = persist.finish(nullptr, nullptr, nullptr, batch, psync, f->ks, stream, /*sync_is_zero=*/true);
    else rc = persist.relaunch(batch, psync, stream, /*sync_is_zero=*/true);
    if (rc != ODEHIP_OK) return rc;
    ++enq;
    while (g_adj_mailbox->ticks + 1 < enq) {
      if (g_adj_mailbox->ticks != seen) {
        seen = g_adj_mailbox->ticks;
        t_progress = adj_now_s();
      }
      if (adj_now_s() - t_progress > 120.0) {
        set_error("odeint_adjoint_dopri5_backward: no progress from the device for 120 s (%d of %d ticks)", g_adj_mailbox->ticks, enq);
        return ODEHIP_EHIP;
      }
    }
    if (g_adj_mailbox->done) break;
  }
  if (g_adj_mailbox->status != 0) {
    const int s = g_adj_mailbox->status;
    ODEHIP_CHECK_HIP(hipStreamSynchronize(stream));   // (error path: nothing of this call may still write the mailbox afterwards)
    if (s == ODEHIP_ENAN) set_error("odeint_adjoint_dopri5_backward: non-finite error ratio");
    else if (s == ODEHIP_ENOTCONV) set_error("odeint_adjoint_dopri5_backward: underflow in dt");
    else set_error("odeint_adjoint_dopri5_backward: more than max_accept = %d accepted steps", max_accept);
    return s;
  }
  // a(t[0]) is the result of the last P3 program: a2[0]
  rc = odehip_q4_to_nchw(L.p(ws, L.off_a2), grad_z0_nchw, batch, L.C, stream);
  if (rc != ODEHIP_OK) return rc;
  // ---- ONE weight-gradient launch per layer over every recorded stage evaluation (tables written by the controller)
  const int n_eval = g_adj_mailbox->n_entries;
  ODEHIP_REQUIRE(n_eval >= 1, "odeint_adjoint_dopri5_backward: the controller recorded no stage evaluation");
  float* slabs = L.p(ws, L.off_slab);
  const WgradPair* wt = (const WgradPair*)L.p(ws, L.off_wtab);
  for (int l = 0; l < NL; ++l) {
    rc = launch_wgrad(wt + (size_t)l * init.wtab_cap, n_eval, batch, wgrad_esplit(batch, n_eval), slabs, grad_w[l], grad_b[l], f->channels[l + 1], f->channels[l], stream,
                      f->w_bf16[l] != nullptr);
    if (rc != ODEHIP_OK) return rc;
  }
  if (stats_host) {
    stats_host[0] = g_adj_mailbox->nfe;
    stats_host[1] = g_adj_mailbox->n_accept;
    stats_host[2] = g_adj_mailbox->n_reject;
  }
  float* regions[2 * ODEHIP_MAX_LAYERS + 1];
  size_t floats[2 * ODEHIP_MAX_LAYERS + 1];
  int nr = 0;
  regions[nr] = grad_z0_nchw; floats[nr++] = st_b / 4;
  for (int l = 0; l < NL; ++l) {
    regions[nr] = grad_w[l]; floats[nr++] = (size_t)f->channels[l + 1] * f->channels[l] * 9;
    regions[nr] = grad_b[l]; floats[nr++] = (size_t)f->channels[l + 1];
  }
  if ((rc = persist.guard(regions, floats, nr, stream)) != ODEHIP_OK) return rc;
  // A tick may still be queued behind `done` (the host runs one ahead): its controller writes the mailbox once more.  Wait for it
  // (the work enqueued above keeps the device busy meanwhile) so that the next call's freshly cleared mailbox cannot be overwritten
  // by this call's last controller.
  t_progress = adj_now_s();
  while (g_adj_mailbox->ticks < enq) {
    if (adj_now_s() - t_progress > 120.0) {
      set_error("odeint_adjoint_dopri5_backward: the device did not drain (%d of %d ticks)", g_adj_mailbox->ticks, enq);
      return ODEHIP_EHIP;
    }
  }
  *ran = 1;
  return ODEHIP_OK;
}

}  // namespace odehip
