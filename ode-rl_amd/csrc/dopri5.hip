// dopri5.hip -- adaptive Dormand-Prince 5(4) with a DEVICE-SIDE step controller (gfx950).
//
// Restates torchdiffeq 0.2.1's Dopri5Solver (_impl/rk_common.py: _adaptive_step, _runge_kutta_step,
// _compute_error_ratio, _optimal_step_size; _impl/misc.py: _select_initial_step; _impl/interp.py) as called by the
// reference at /root/reference/modules/DiffEqSolver.py:37,45-46 (default method, configs.yaml:79), with the same
// decisions in the same floating-point types (float64 time, fp32 state and error ratio), but without torchdiffeq's
// >= 3 host synchronisations per attempted step: accept/reject, the next step size, which output times fall inside
// the accepted step and the "done" test all live in a small state block in device memory written by a one-workgroup
// controller kernel.  The host only bounds its run-ahead by polling a pinned mailbox that the controller updates.
//
// One attempted step = 6 evaluations of f (30 conv launches; stage combines and the error-norm partial sums are
// fused into the last conv of each f) + controller + finish (dense output for the output times inside the step,
// FSAL hand-over y <- y1, k1 <- k7, and the first stage input of the next attempt).
#include <math.h>
#include <string.h>
#include <time.h>

#include "odehip_internal.h"
#include "persist.h"
#include "dopri5_layout.h"

namespace odehip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Dormand-Prince-Shampine tableau (torchdiffeq/_impl/dopri5.py)

struct DopriState {
  double t0, t1, dt;   // accepted interval [t0, t1]; size of the attempt in flight
  float h;             // (float)dt: what meets the fp32 state (torchdiffeq casts dt to y.dtype)
  float h_used;        // step size of the attempt that was just judged (for the dense output)
  float h0;            // initial-step heuristic scratch
  float ratio;
  int accept, done, status;
  int j_next, j_lo, j_hi;
  int n_accept, n_reject, nfe, n_steps;
  int n_ctrl;          // controller launches (attempts enqueued by the host, real or skipped)
  float rtol, atol;
  int n_times, n_partials;
  long long n_elems;
  int max_steps;
  // SAVING mode (the forward of a training step keeps the stage inputs and hidden activations of every ACCEPTED step for the
  // backward pass: no re-integration).  An attempt writes them into slot `slot_next` of the backward workspace through
  // relocatable pointers (class 1, conv_wino.hip rel()); a rejected attempt's slot is simply reused, an accepted one moves on.
  int save_max;        // slots available (0: not saving)
  int slot_used;       // slot of the attempt that was just judged
  int slot_next;       // slot of the next attempt
  int save_ok;         // cleared when more steps were accepted than there are slots (the caller then re-integrates)
  unsigned long long slot_base, slot_bytes;
  unsigned long long* reloc;   // [16] relocation bases read by the walk; entry 1 = address of slot_next
};

constexpr int kMailboxBytes = 65536;
constexpr int kLogCap = (kMailboxBytes - 64) / 16;
struct Mailbox {  // pinned host memory, written by the controller with system-scope stores
  volatile int steps_done, done, status, n_accept, n_reject, nfe, save_ok;
  volatile int truncated;   // asynchronous solve: the attempts enqueued by start() did not finish it (seal_kernel)
  int pad_[8];
  volatile double log[kLogCap][2];  // (t0, dt) of every accepted step, in order (what a backward pass re-integrates)
};

// ---- deterministic block reduction of `n` floats (fixed order), result valid in thread 0
__device__ float block_sum(const float* v, int n, float* sh) {
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  return sh[0];
}

// partial sums of ((a - b) / (atol + |y|*rtol))^2 per workgroup (b may be null)
__global__ __launch_bounds__(256) void scaled_sumsq_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ y, float atol, float rtol,
                                                           long long n4, float* __restrict__ partials) {
  __shared__ float sh[256];
  float s = 0.0f;
  for (long long i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 av = ((const f32x4*)a)[i], yv = ((const f32x4*)y)[i];
    f32x4 d = av;
    if (b) d -= ((const f32x4*)b)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float r = d[k] / (atol + fabsf(yv[k]) * rtol);
      s += r * r;
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

// out = y + h * sum_j c[j] * k[j]   (h from device memory)
struct LinComb {
  const float* y;
  const float* k[ODEHIP_MAX_STAGES];
  float c[ODEHIP_MAX_STAGES];
  int n;
  const float* h_ptr;
  float* out;
  const int* skip;
};
__global__ __launch_bounds__(256) void lincomb_kernel(LinComb a, long long n4) {
  if (a.skip && *a.skip) return;
  const float h = *a.h_ptr;
  for (long long i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 s = ((const f32x4*)a.k[0])[i] * a.c[0];
    for (int j = 1; j < a.n; ++j) s += ((const f32x4*)a.k[j])[i] * a.c[j];
    ((f32x4*)a.out)[i] = ((const f32x4*)a.y)[i] + s * h;
  }
}

// _select_initial_step, first half: d0, d1 -> h0
__global__ __launch_bounds__(256) void init1_kernel(DopriState* st, const float* p0, const float* p1, int np) {
  __shared__ float sh[256];
  const float s0 = block_sum(p0, np, sh);
  __syncthreads();
  const float s1 = block_sum(p1, np, sh);
  if (threadIdx.x == 0) {
    const float n = (float)st->n_elems;
    const float d0 = sqrtf(s0 / n), d1 = sqrtf(s1 / n);
    st->h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
    st->ratio = d1;  // parked for init2
  }
}

// second half: d2 -> dt; arms the first attempt
__global__ __launch_bounds__(256) void init2_kernel(DopriState* st, const float* p2, int np, const double* t_out) {
  __shared__ float sh[256];
  const float s2 = block_sum(p2, np, sh);
  if (threadIdx.x == 0) {
    const float n = (float)st->n_elems;
    const float h0 = st->h0, d1 = st->ratio;
    const float d2 = sqrtf(s2 / n) / h0;
    float h1;
    if (d1 <= 1e-15f && d2 <= 1e-15f) h1 = fmaxf(1e-6f, h0 * 1e-3f);
    else h1 = powf(0.01f / fmaxf(d1, d2), 1.0f / 5.0f);
    const double dt = (double)fminf(100.0f * h0, h1);
    st->t0 = st->t1 = t_out[0];
    st->dt = dt;
    st->h = (float)dt;
    st->nfe = 2;
    st->j_next = 1;
    st->done = st->n_times <= 1;
  }
}

// _adaptive_step's scalar part: error ratio, accept, output range, next dt (torchdiffeq _optimal_step_size)
__global__ __launch_bounds__(256) void controller_kernel(DopriState* st, const float* partials, const double* t_out,
                                                          Mailbox* mb, unsigned* psync, int psync_words) {
  __shared__ float sh[256];
  // the flag area of the NEXT attempt's persistent walk (this attempt's walk has finished: stream order) -- saves a memset launch
  for (int i = threadIdx.x; i < psync_words; i += 256) psync[i] = 0u;
  if (st->done) {  // an attempt the host enqueued after the solve finished: only acknowledge it
    if (threadIdx.x == 0) {
      st->accept = 0;
      st->j_lo = st->j_hi = st->j_next;
      st->n_ctrl += 1;
      __hip_atomic_store((int*)&mb->steps_done, st->n_ctrl, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  const float s = block_sum(partials, st->n_partials, sh);
  if (threadIdx.x != 0) return;
  const float ratio = sqrtf(s / (float)st->n_elems);
  const bool finite = ratio == ratio && ratio < INFINITY;
  const bool accept = ratio <= 1.0f;
  st->ratio = ratio;
  st->accept = accept;
  st->h_used = st->h;
  st->nfe += 6;
  st->n_steps += 1;
  const double dt = st->dt;
  int status = 0;
  if (!finite) status = ODEHIP_ENAN;  // torchdiffeq asserts isfinite(y) at the next step; a NaN ratio never accepts
  if (accept) {
    const double t1n = st->t1 + dt;
    if (st->n_accept < kLogCap) {
      mb->log[st->n_accept][0] = st->t1;
      mb->log[st->n_accept][1] = dt;
    }
    st->t0 = st->t1;
    st->t1 = t1n;
    st->n_accept += 1;
    int j = st->j_next;
    st->j_lo = j;
    while (j < st->n_times && t_out[j] <= t1n) ++j;  // outputs inside (t0, t1]: `while next_t > t1` is false
    st->j_hi = j;
    st->j_next = j;
    // torchdiffeq counts max_num_steps per `_advance(next_t)` call (a local of it): once this step has covered output j_lo the
    // call returns, and the calls for the further outputs it covers take no step at all -- the next attempt starts a fresh count
    if (j > st->j_lo) st->n_steps = 0;
  } else {
    st->n_reject += 1;
    st->j_lo = st->j_hi = st->j_next;
  }
  // dt_next (float64, order 5, safety 0.9, ifactor 10, dfactor 0.2 -- 1 after an accepted step)
  double dtn;
  if (ratio == 0.0f) {
    dtn = dt * 10.0;
  } else {
    const double dfactor = ratio < 1.0f ? 1.0 : 0.2;
    const double fac = fmin(10.0, fmax(0.9 / pow((double)ratio, 0.2), dfactor));
    dtn = dt * fac;
  }
  st->dt = dtn;
  st->h = (float)dtn;
  const bool done = st->j_next >= st->n_times;
  if (!done) {
    if (!(st->t1 + dtn > st->t1)) status = ODEHIP_ENOTCONV;       // "underflow in dt"
    if (st->n_steps >= st->max_steps) status = ODEHIP_ENOTCONV;    // max_num_steps: attempts spent on the output time in progress
  }
  if (status) st->status = status;
  st->done = done || status != 0;
  if (st->save_max > 0) {
    st->slot_used = st->slot_next;
    if (accept) {
      int nx = st->n_accept;
      if (nx >= st->save_max) {   // out of slots: keep going (the last slot is overwritten), the caller falls back to re-integration
        if (!st->done) st->save_ok = 0;
        nx = st->save_max - 1;
      }
      st->slot_next = nx;
    }
    st->reloc[1] = st->slot_base + (unsigned long long)st->slot_next * st->slot_bytes;
  }
  mb->save_ok = st->save_ok;
  mb->n_accept = st->n_accept;
  mb->n_reject = st->n_reject;
  mb->nfe = st->nfe;
  mb->status = st->status;
  mb->done = st->done;
  st->n_ctrl += 1;
  __hip_atomic_store((int*)&mb->steps_done, st->n_ctrl, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Elementwise tail of an attempt.  On accept: dense output for the output times inside the step (quartic of
// _interp_fit/_interp_evaluate), then y <- y1, k1 <- k7 (FSAL).  Always: x2 = y + h_next * beta21 * k1.
struct FinishArgs {
  const DopriState* st;
  const double* t_out;
  float* y;          // current state (Q4), updated on accept
  const float* y1;   // candidate state
  float* k[7];       // k[0] = k1 (updated on accept from k[6])
  float* x2;         // stage-2 input of the next attempt
  float* out_nchw;   // (T,B,C,16,16)
  int channels;
  long long state_floats;
  long long off_y1, off_x2;   // saving mode: byte offsets of y1 (stage input 6) and of the stage-2 input inside a slot
};
__global__ __launch_bounds__(256) void finish_kernel(FinishArgs a, long long n4, float cm0, float cm2, float cm3, float cm4,
                                                     float cm5, float cm6) {
  const DopriState* st = a.st;
  const int accept = st->accept, j_lo = st->j_lo, j_hi = st->j_hi;
  if (st->done && !(accept && j_hi > j_lo)) return;  // nothing left to write
  if (st->save_max > 0) {   // y1 of the attempt just judged lies in ITS slot; the next attempt's stage-2 input goes to the next slot
    a.y1 = (const float*)(st->slot_base + (unsigned long long)st->slot_used * st->slot_bytes + (unsigned long long)a.off_y1);
    a.x2 = (float*)(st->slot_base + (unsigned long long)st->slot_next * st->slot_bytes + (unsigned long long)a.off_x2);
  }
  const float hu = st->h_used, hn = st->h;
  const double t0 = st->t0, t1 = st->t1;
  const bool done = st->done;
  for (long long i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 y0 = ((const f32x4*)a.y)[i];
    f32x4 k1 = ((const f32x4*)a.k[0])[i];
    if (accept) {
      const f32x4 y1 = ((const f32x4*)a.y1)[i];
      const f32x4 k7 = ((const f32x4*)a.k[6])[i];
      if (j_hi > j_lo) {
        const f32x4 k3 = ((const f32x4*)a.k[2])[i], k4 = ((const f32x4*)a.k[3])[i], k5 = ((const f32x4*)a.k[4])[i],
                    k6 = ((const f32x4*)a.k[5])[i];
        const f32x4 ymid = y0 + (k1 * cm0 + k3 * cm2 + k4 * cm3 + k5 * cm4 + k6 * cm5 + k7 * cm6) * hu;
        const f32x4 ca = (k7 - k1) * (2.0f * hu) - (y1 + y0) * 8.0f + ymid * 16.0f;
        const f32x4 cb = (k1 * 5.0f - k7 * 3.0f) * hu + y0 * 18.0f + y1 * 14.0f - ymid * 32.0f;
        const f32x4 cc = (k7 - k1 * 4.0f) * hu - y0 * 11.0f - y1 * 5.0f + ymid * 16.0f;
        const f32x4 cd = k1 * hu;
        // element i*4.. of the Q4 state -> NCHW: i = (b*Q + q)*256 + p
        const long long p = i & 255, bq = i >> 8;
        for (int j = j_lo; j < j_hi; ++j) {
          const float x = (float)((a.t_out[j] - t0) / (t1 - t0));
          f32x4 tot = y0 + cd * x;
          float xp = x * x;
          tot += cc * xp;
          xp *= x;
          tot += cb * xp;
          xp *= x;
          tot += ca * xp;
          float* o = a.out_nchw + (size_t)j * a.state_floats + (size_t)bq * 4 * kPix + p;
          o[0] = tot.x; o[kPix] = tot.y; o[2 * kPix] = tot.z; o[3 * kPix] = tot.w;
        }
      }
      y0 = y1;
      k1 = k7;
      if (!done) {
        ((f32x4*)a.y)[i] = y0;
        ((f32x4*)a.k[0])[i] = k1;
      }
    }
    if (!done) ((f32x4*)a.x2)[i] = y0 + k1 * (hn * 0.2f);  // beta21 = 1/5
  }
}

// The SEAL of an asynchronous solve (odehip_odeint_dopri5_start): enqueued behind the last attempt start() enqueues.  The caller has
// been handed `out` and goes on enqueueing its consumers (decoder, loss, the backward pass) behind this kernel, so nothing enqueued
// LATER can still complete the trajectory for them.  If the solve is not done here, the frames it has not reached are filled with NaN
// -- the consumers then compute NaN instead of reading uninitialised memory -- and the mailbox says so: collect() fails with
// ODEHIP_ETRUNC instead of carrying on behind the consumers' backs.
__global__ __launch_bounds__(256) void seal_kernel(const DopriState* st, Mailbox* mb, float* out_nchw, long long state_floats) {
  if (st->done) return;
  const int j0 = st->j_next, nt = st->n_times;
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store((int*)&mb->truncated, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  const long long n = (long long)(nt - j0) * state_floats;
  float* o = out_nchw + (long long)j0 * state_floats;
  const float nan = __builtin_nanf("");
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) o[i] = nan;
}

struct DoublePack {
  double v[32];
};
__global__ void fill_doubles_kernel(double* dst, DoublePack p, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = p.v[threadIdx.x];
}
// options={'first_step': dt} of torchdiffeq: skip the heuristic (f0 is still evaluated: nfe starts at 1)
__global__ void arm_first_step_kernel(DopriState* st, const double* t_out, double dt) {
  st->t0 = st->t1 = t_out[0];
  st->dt = dt;
  st->h = (float)dt;
  st->nfe = 1;
  st->j_next = 1;
  st->done = st->n_times <= 1;
}
__global__ void init_state_kernel(DopriState* st, float rtol, float atol, int n_times, int n_partials, long long n_elems,
                                  int max_steps, int save_max, unsigned long long slot_base, unsigned long long slot_bytes,
                                  unsigned long long* reloc) {
  DopriState z;
  memset(&z, 0, sizeof(z));
  z.save_max = save_max;
  z.save_ok = save_max > 0;
  z.slot_base = slot_base;
  z.slot_bytes = slot_bytes;
  z.reloc = reloc;
  if (reloc) reloc[1] = slot_base;
  z.rtol = rtol;
  z.atol = atol;
  z.n_times = n_times;
  z.n_partials = n_partials;
  z.n_elems = n_elems;
  z.max_steps = max_steps;
  *st = z;
}


// out[j] = fixed-order sum of partial array j (one workgroup)
struct SumSet {
  const float* p[4];
  int n[4];
  int count;
};
__global__ __launch_bounds__(256) void sum_partials_kernel(SumSet ss, float* __restrict__ out, const int* skip) {
  __shared__ float sh[256];
  if (skip && *skip) return;
  for (int j = 0; j < ss.count; ++j) {
    const float s = block_sum(ss.p[j], ss.n[j], sh);
    if (threadIdx.x == 0) out[j] = s;
    __syncthreads();
  }
}

static double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

// ---- one dopri5 solve in flight.  The synchronous entry points build it on their stack and run it to completion; the asynchronous
// pair (odehip_odeint_dopri5_start / _collect) parks it in a slot between the two calls, so that the host can go on enqueueing the
// work behind the solver (decoder, loss, the whole backward pass) while the device is still integrating.
struct D5Ctx {
  bool in_use = false;
  odehip_convstack f;          // by value: the caller's descriptor need not outlive the start call
  int n_times = 0, batch = 0, n_conv_partials = 0;
  float rtol = 0, atol = 0, ksc = 1;
  hipStream_t stream = nullptr;
  DopriState* state = nullptr;
  double* t_dev = nullptr;
  float *part0 = nullptr, *ping = nullptr, *pong = nullptr, *xs = nullptr, *y = nullptr, *y1 = nullptr, *k[7] = {};
  unsigned* psync = nullptr;
  unsigned long long* reloc = nullptr;
  bool saving = false, global_norm = false, partials64 = false;
  size_t bl_st = 0, bl_hid = 0;   // BwdLayout::st / hid / NH of the saving slots
  int bl_nh = 0;
  FinishArgs fa;
  long long n4 = 0;
  Mailbox* mb = nullptr;
  int enq = 0, seen = 0;
  double t_progress = 0;
};
constexpr int kD5Slots = 4;
static D5Ctx g_d5[kD5Slots];
static Mailbox* g_d5_mailbox[kD5Slots + 1] = {};   // [kD5Slots]: the synchronous calls'

static int d5_mailbox(int slot, Mailbox** out) {
  if (!g_d5_mailbox[slot]) {
    // 64 KiB of pinned, coherent host memory for the controller's progress word and its log of accepted steps
    ODEHIP_CHECK_HIP(hipHostMalloc((void**)&g_d5_mailbox[slot], kMailboxBytes, hipHostMallocCoherent));
  }
  memset((void*)g_d5_mailbox[slot], 0, sizeof(Mailbox));
  *out = g_d5_mailbox[slot];
  return ODEHIP_OK;
}

// exact-global step control under batch sharding: every sum of squares is summed over the ranks before it is used
static odehip_allreduce_fn g_reduce_cb = nullptr;
static void* g_reduce_user = nullptr;
static int g_reduce_world = 1;
static float* g_reduce_buf = nullptr;
static int d5_reduce_sums(hipStream_t stream, int count, const float* const* arrays, const int* lens, const int* skip_flag) {
  SumSet ss;
  memset(&ss, 0, sizeof(ss));
  ss.count = count;
  for (int j = 0; j < count; ++j) {
    ss.p[j] = arrays[j];
    ss.n[j] = lens[j];
  }
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, stream, ss, g_reduce_buf, skip_flag);
  ODEHIP_CHECK_HIP(hipGetLastError());
  const int rcb = g_reduce_cb(g_reduce_buf, count, (void*)stream, g_reduce_user);
  ODEHIP_REQUIRE(rcb == 0, "odeint_dopri5: the all-reduce callback failed (%d)", rcb);
  return ODEHIP_OK;
}

// enqueue ONE attempted step: the six evaluations (one persistent launch: the same table for every attempt -- the step size is read
// through h_ptr, the queued-behind-`done` case through the skip word), the controller, the elementwise tail
static int d5_attempt(D5Ctx& x) {
  const odehip_convstack* f = &x.f;
  int rc;
  constexpr unsigned long long kTag1 = 1ull << 56;
  auto s_xin = [&](int e) { return (float*)(kTag1 | (unsigned long long)((size_t)e * x.bl_st)); };
  auto s_hid = [&](int e, int l) { return (float*)(kTag1 | (unsigned long long)(7 * x.bl_st + ((size_t)e * x.bl_nh + l) * x.bl_hid)); };
  const int* skip = &x.state->done;
  PersistScope persist;
  if ((rc = persist.begin(f, nullptr, 6 * f->n_convs)) != ODEHIP_OK) return rc;
  CombineArgs c;
  for (int s = 2; s <= 7; ++s) {  // k_s = f(x_s); fused: x_{s+1} = y + h*sum beta_{s+1,j} k_j   (s = 7: error norm)
    memset(&c, 0, sizeof(c));
    c.k_scale = x.ksc;
    c.order = all_64(f);   // 64-channel stacks: the adaptive walk (stage sums formed ahead of the matrix work), same bits per layer
    c.y = x.y;
    c.h_ptr = &x.state->h;
    c.n_prev = s - 1;
    for (int j = 0; j < s - 1; ++j) c.k_prev[j] = x.k[j];
    c.k_out = x.k[s - 1];
    if (s <= 6) {
      for (int j = 0; j < s; ++j) c.c1[j] = (float)dp5::kBeta[s - 1][j];
      c.out1 = x.saving ? s_xin(s) : (s < 6 ? x.xs : x.y1);  // x7 = y1 (c_sol equals the last beta row)
    } else {
      for (int j = 0; j < 7; ++j) c.ce[j] = (float)dp5::kCErr[j];
      c.err_y1 = x.saving ? s_xin(6) : x.y1;
      c.err_partials = x.part0;
      c.rtol = x.rtol;
      c.atol = x.atol;
    }
    if (x.saving) {   // evaluation (slot, s - 1): input and hidden activations live in the slot the device picks
      ODEHIP_REQUIRE(persist.recording(), "odeint_dopri5: the persistent walk became unavailable during a saving forward");
      float* hid_s[ODEHIP_MAX_LAYERS];
      for (int l = 0; l < x.bl_nh; ++l) hid_s[l] = s_hid(s - 1, l);
      rc = enqueue_f_saving(f, s_xin(s - 1), x.batch, hid_s, x.ping, x.pong, &c, nullptr, skip, x.stream);
    } else {
      rc = enqueue_f(f, s < 7 ? x.xs : x.y1, x.batch, x.ping, x.pong, &c, nullptr, skip, x.stream);
    }
    if (rc != ODEHIP_OK) return rc;
  }
  if (x.saving) persist.set_device_steering(nullptr, x.reloc);
  if ((rc = persist.finish(nullptr, nullptr, nullptr, x.batch, x.psync, f->ks, x.stream, /*sync_is_zero=*/true)) != ODEHIP_OK) return rc;
  if (x.partials64 && !persist.launched()) {
    // the controller sums 64 error-norm partials per sample (the sixteen-workgroup walk's); a refused launch has just been replayed
    // as per-layer launches, which write 16: the norm of this attempt is not the solver's
    set_error("odeint_dopri5: the persistent walk became unavailable during the solve (its error-norm partials are laid out for it)");
    return ODEHIP_EHIP;
  }
  if (x.global_norm) {
    const float* arr[1] = {x.part0};
    const int lens[1] = {x.n_conv_partials};
    if ((rc = d5_reduce_sums(x.stream, 1, arr, lens, nullptr)) != ODEHIP_OK) return rc;
  }
  hipLaunchKernelGGL(controller_kernel, dim3(1), dim3(256), 0, x.stream, x.state, x.global_norm ? g_reduce_buf : x.part0, x.t_dev, x.mb, x.psync,
                     (int)(persist_sync_bytes(x.batch) / 4));
  hipLaunchKernelGGL(finish_kernel, dim3(1024), dim3(256), 0, x.stream, x.fa, x.n4, (float)dp5::kCMid[0], (float)dp5::kCMid[2],
                     (float)dp5::kCMid[3], (float)dp5::kCMid[4], (float)dp5::kCMid[5], (float)dp5::kCMid[6]);
  ODEHIP_CHECK_HIP(hipGetLastError());
  ++x.enq;
  return ODEHIP_OK;
}

// attempts until the controller reports done (the host at most RUN_AHEAD attempts ahead of the device: one attempt queued behind the
// running one keeps the GPU busy; exact-global mode enqueues a collective per attempt, so every rank must enqueue the same number:
// no run-ahead there), then wait for the last enqueued controller so that the mailbox is final
static int d5_run_to_done(D5Ctx& x) {
  const int RUN_AHEAD = x.global_norm ? 0 : 1;
  int rc;
  x.t_progress = now_s();
  auto stalled = [&](const char* what) {
    if (x.mb->steps_done != x.seen) {   // the limit is on time WITHOUT progress, not on the whole integration
      x.seen = x.mb->steps_done;
      x.t_progress = now_s();
    }
    if (now_s() - x.t_progress > 120.0) {
      set_error("odeint_dopri5: %s (steps done %d of %d enqueued)", what, x.mb->steps_done, x.enq);
      return true;
    }
    return false;
  };
  for (;;) {
    while (!x.mb->done && x.mb->steps_done + RUN_AHEAD < x.enq)
      if (stalled("no progress from the device for 120 s")) return ODEHIP_EHIP;
    if (x.mb->done) break;
    if ((rc = d5_attempt(x)) != ODEHIP_OK) return rc;
  }
  // attempts enqueued after `done` do nothing (skip flag)
  while (x.mb->steps_done < x.enq)
    if (stalled("device did not drain")) return ODEHIP_EHIP;
  return ODEHIP_OK;
}

static int d5_collect(D5Ctx& x, int* stats_host, double* accepted_host, int accepted_cap, int* saved_out) {
  // the device has finished every attempt of this call: a persistent launch that gave up a wait is known NOW
  if (const unsigned code = persist_error(true)) {
    set_error("odeint_dopri5: a persistent launch gave up waiting for a partner workgroup (code %u); the trajectory is invalid.  "
              "Persistent launches are now disabled for this process", code);
    return ODEHIP_EHIP;
  }
  Mailbox* mb = x.mb;
  if (saved_out) *saved_out = x.saving && mb->save_ok && mb->status == 0;
  if (stats_host) {
    stats_host[0] = mb->nfe;
    stats_host[1] = mb->n_accept;
    stats_host[2] = mb->n_reject;
    stats_host[3] = x.enq;
  }
  if (accepted_host) {  // (t0, dt) pairs; the caller sees from stats_host[1] > accepted_cap that the log is incomplete
    int n = mb->n_accept < kLogCap ? mb->n_accept : kLogCap;
    if (n > accepted_cap) n = accepted_cap;
    for (int i = 0; i < n; ++i) {
      accepted_host[2 * i] = mb->log[i][0];
      accepted_host[2 * i + 1] = mb->log[i][1];
    }
  }
  if (mb->status == ODEHIP_ENAN) {
    set_error("odeint_dopri5: non-finite error ratio (non-finite values in state `y`)");
    return ODEHIP_ENAN;
  }
  if (mb->status == ODEHIP_ENOTCONV) {
    set_error("odeint_dopri5: underflow in dt or max_num_steps exceeded");
    return ODEHIP_ENOTCONV;
  }
  return ODEHIP_OK;
}


}  // namespace odehip

using namespace odehip;


extern "C" int odehip_set_norm_allreduce(odehip_allreduce_fn cb, void* user, int world_size, float* scratch_dev) {
  if (!cb) {
    g_reduce_cb = nullptr;
    g_reduce_user = nullptr;
    g_reduce_world = 1;
    g_reduce_buf = nullptr;
    return ODEHIP_OK;
  }
  ODEHIP_REQUIRE(world_size >= 1 && scratch_dev, "set_norm_allreduce: world_size must be >= 1 and scratch_dev non-null");
  g_reduce_cb = cb;
  g_reduce_user = user;
  g_reduce_world = world_size;
  g_reduce_buf = scratch_dev;
  return ODEHIP_OK;
}

// Workspace: [state | t_out[T] | partials x3 | ping | pong | xs | y | y1 | k1..k7]
extern "C" size_t odehip_dopri5_workspace_bytes(const odehip_convstack* f, int batch, int n_times) {
  if (!f || batch <= 0 || n_times <= 0) return 0;
  const size_t st = al256((size_t)batch * f->channels[0] * kPix * 4);
  const size_t hid = al256((size_t)batch * max_hidden(f) * kPix * 4);
  const size_t np = (size_t)batch * (f->channels[0] / 32) * 2 * 4;
  return al256(sizeof(DopriState)) + al256((size_t)n_times * 8) + 3 * al256((np > 1024 ? np : 1024) * 4) + 2 * hid + 10 * st +
         al256(persist_sync_bytes(batch)) + al256(16 * 8);
}

// SAVING forward: the plain forward's workspace followed by the backward pass's (dopri5_layout.h) with max_accept slots
extern "C" size_t odehip_dopri5_saving_workspace_bytes(const odehip_convstack* f, int batch, int n_times, int max_accept) {
  if (!f || batch <= 0 || n_times <= 0 || max_accept <= 0 || f->n_convs < 1) return 0;
  return odehip_dopri5_workspace_bytes(f, batch, n_times) + BwdLayout(f, batch, n_times, max_accept).total;
}

static int dopri5_forward(const odehip_convstack* f, const float* z0_nchw, const double* t_host, int n_times, int batch, float rtol,
                          float atol, double first_step, int max_steps, int negate, float* out_nchw, int* stats_host,
                          double* accepted_host, int accepted_cap, void* workspace, size_t workspace_bytes, void* stream_,
                          int save_max_accept, int* saved_out, int async_attempts = 0, int* token_out = nullptr);

extern "C" int odehip_odeint_dopri5(const odehip_convstack* f, const float* z0_nchw, const double* t_host, int n_times,
                                    int batch, float rtol, float atol, double first_step, int max_steps, int negate,
                                    float* out_nchw, int* stats_host, double* accepted_host, int accepted_cap, void* workspace,
                                    size_t workspace_bytes, void* stream_) {
  return dopri5_forward(f, z0_nchw, t_host, n_times, batch, rtol, atol, first_step, max_steps, negate, out_nchw, stats_host, accepted_host,
                        accepted_cap, workspace, workspace_bytes, stream_, 0, nullptr);
}

extern "C" int odehip_odeint_dopri5_saving(const odehip_convstack* f, const float* z0_nchw, const double* t_host, int n_times,
                                           int batch, float rtol, float atol, double first_step, int max_steps, float* out_nchw,
                                           int* stats_host, double* accepted_host, int accepted_cap, int max_accept, int* saved_out,
                                           void* workspace, size_t workspace_bytes, void* stream_) {
  ODEHIP_REQUIRE(saved_out && max_accept > 0, "odeint_dopri5_saving: saved_out is required and max_accept must be positive");
  *saved_out = 0;
  ODEHIP_REQUIRE(f && workspace_bytes >= odehip_dopri5_saving_workspace_bytes(f, batch, n_times, max_accept),
                 "odeint_dopri5_saving: workspace too small");
  return dopri5_forward(f, z0_nchw, t_host, n_times, batch, rtol, atol, first_step, max_steps, 0, out_nchw, stats_host, accepted_host,
                        accepted_cap, workspace, workspace_bytes, stream_, max_accept, saved_out);
}

static int dopri5_forward(const odehip_convstack* f, const float* z0_nchw, const double* t_host, int n_times, int batch, float rtol,
                          float atol, double first_step, int max_steps, int negate, float* out_nchw, int* stats_host,
                          double* accepted_host, int accepted_cap, void* workspace, size_t workspace_bytes, void* stream_,
                          int save_max_accept, int* saved_out, int async_attempts, int* token_out) {
  int rc = check_stack(f);
  if (rc != ODEHIP_OK) return rc;
  ODEHIP_REQUIRE(z0_nchw && t_host && out_nchw && workspace, "odeint_dopri5: null pointer");
  ODEHIP_REQUIRE(n_times >= 1 && batch > 0, "odeint_dopri5: bad sizes (n_times %d, batch %d)", n_times, batch);
  ODEHIP_REQUIRE(f->channels[0] == f->channels[f->n_convs], "odeint_dopri5: f must map C -> C channels");
  ODEHIP_REQUIRE(rtol > 0 && atol >= 0, "odeint_dopri5: rtol must be > 0 and atol >= 0");
  for (int i = 1; i < n_times; ++i)
    ODEHIP_REQUIRE(t_host[i] > t_host[i - 1], "odeint_dopri5: t must be strictly increasing (t[%d]=%g, t[%d]=%g)", i - 1,
                   t_host[i - 1], i, t_host[i]);
  ODEHIP_REQUIRE(workspace_bytes >= odehip_dopri5_workspace_bytes(f, batch, n_times), "odeint_dopri5: workspace too small");
  if (max_steps <= 0) max_steps = 1 << 30;
  hipStream_t stream = (hipStream_t)stream_;
  const int C = f->channels[0];
  const size_t st_b = (size_t)batch * C * kPix * 4, st_f = st_b / 4;
  const long long n4 = (long long)st_f / 4;
  // error-norm partials the attempts' last layers write: 4 per workgroup of the per-layer grid -- or, on the sixteen-workgroup walk
  // (batch <= 16, 64-channel stack, walk available), 64 per sample
  bool walk_ok = false;
  if (all_64(f)) {
    PersistScope probe;
    if ((rc = probe.begin(f, nullptr, 1)) != ODEHIP_OK) return rc;
    walk_ok = probe.recording();
  }
  const int pps = walk_ok ? persist_partials_per_sample(batch) : 16;
  const int n_conv_partials = pps > 16 ? batch * pps : batch * (C / 32) * 2 * 4;
  const int red_grid = 256;

  // the context of this solve: on the stack for a synchronous call, in a free slot for an asynchronous one
  D5Ctx stack_ctx;
  int slot = kD5Slots;
  if (async_attempts > 0) {
    ODEHIP_REQUIRE(token_out && !g_reduce_cb, "odeint_dopri5_start: needs token_out; not available under exact-global step control");
    slot = -1;
    for (int i = 0; i < kD5Slots; ++i)
      if (!g_d5[i].in_use) { slot = i; break; }
    ODEHIP_REQUIRE(slot >= 0, "odeint_dopri5_start: %d solves are already pending (collect one first)", kD5Slots);
  }
  D5Ctx& cx = slot < kD5Slots ? g_d5[slot] : stack_ctx;
  cx = D5Ctx();
  if ((rc = d5_mailbox(slot, &cx.mb)) != ODEHIP_OK) return rc;

  char* base = (char*)workspace;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base + off; off += al256(bytes); return p; };
  DopriState* state = (DopriState*)take(sizeof(DopriState));
  double* t_dev = (double*)take((size_t)n_times * 8);
  const size_t pbytes = (size_t)(n_conv_partials > 1024 ? n_conv_partials : 1024) * 4;
  float* part0 = (float*)take(pbytes);
  float* part1 = (float*)take(pbytes);
  float* part2 = (float*)take(pbytes);
  float* ping = (float*)take((size_t)batch * max_hidden(f) * kPix * 4);
  float* pong = (float*)take((size_t)batch * max_hidden(f) * kPix * 4);
  float* xs = (float*)take(st_b);
  float* y = (float*)take(st_b);
  float* y1 = (float*)take(st_b);
  float* k[7];
  for (int i = 0; i < 7; ++i) k[i] = (float*)take(st_b);
  unsigned* psync = (unsigned*)take(persist_sync_bytes(batch));
  unsigned long long* reloc = (unsigned long long*)take(16 * 8);

  // ---- SAVING mode: every attempt writes its stage inputs and hidden activations into a slot of the backward workspace that
  // follows this one; the slot is chosen on the device (class-1 relocatable pointers), so the attempt's table is still the same
  // for every attempt.  Needs the adaptive walk: without it (or for other stacks) nothing is saved and the caller re-integrates.
  bool saving = false;
  char* bws = base + al256(odehip_dopri5_workspace_bytes(f, batch, n_times));
  const BwdLayout BL(f, batch, n_times, save_max_accept > 0 ? save_max_accept : 1);
  if (save_max_accept > 0 && all_64(f) && !g_reduce_cb && n_times > 1) saving = walk_ok;

  // exact-global mode: the kernels below read ONE already all-reduced scalar instead of the local partial arrays
  const bool global_norm = g_reduce_cb != nullptr;
  auto reduce_sums = [&](int count, const float* const* arrays, const int* lens, const int* skip_flag) -> int {
    return d5_reduce_sums(stream, count, arrays, lens, skip_flag);
  };
  hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(1), 0, stream, state, rtol, atol, n_times,
                     global_norm ? 1 : n_conv_partials, (long long)st_f * (global_norm ? g_reduce_world : 1), max_steps,
                     saving ? save_max_accept : 0, (unsigned long long)(uintptr_t)(bws + BL.off_slots), (unsigned long long)BL.slot_bytes,
                     saving ? reloc : (unsigned long long*)nullptr);
  for (int o = 0; o < n_times; o += 32) {
    DoublePack p;
    const int m = n_times - o < 32 ? n_times - o : 32;
    for (int i = 0; i < m; ++i) p.v[i] = t_host[o + i];
    hipLaunchKernelGGL(fill_doubles_kernel, dim3(1), dim3(32), 0, stream, t_dev + o, p, m);
  }
  ODEHIP_CHECK_HIP(hipGetLastError());
  ODEHIP_CHECK_HIP(hipMemcpyAsync(out_nchw, z0_nchw, st_b, hipMemcpyDeviceToDevice, stream));  // solution[0] = y0
  rc = odehip_nchw_to_q4(z0_nchw, y, batch, C, stream);
  if (rc != ODEHIP_OK) return rc;
  if (saving) ODEHIP_CHECK_HIP(hipMemcpyAsync(BL.xin(bws, 0, 0), y, st_b, hipMemcpyDeviceToDevice, stream));   // stage input (0, 0) = z0
  if (n_times == 1) {
    if (async_attempts > 0) {   // nothing to integrate: collect() only has to hand back empty stats
      cx.n_times = 1;
      cx.in_use = true;
      *token_out = slot;
      return ODEHIP_OK;
    }
    if (stats_host) stats_host[0] = stats_host[1] = stats_host[2] = stats_host[3] = 0;
    return ODEHIP_OK;
  }

  // ---- _select_initial_step: f0, d0, d1 -> h0; f(y0 + h0 f0); d2 -> dt
  CombineArgs c;
  memset(&c, 0, sizeof(c));
  const float ksc = negate ? -1.0f : 1.0f;
  c.k_scale = ksc;
  c.k_out = k[0];
  if (saving) {   // k1 of the first step = evaluation (slot 0, stage 0): its hidden activations are kept
    float* hid0[ODEHIP_MAX_LAYERS];
    for (int l = 0; l < BL.NH; ++l) hid0[l] = BL.hidden(bws, 0, 0, l);
    rc = enqueue_f_saving(f, y, batch, hid0, ping, pong, &c, nullptr, nullptr, stream);
  } else {
    rc = enqueue_f(f, y, batch, ping, pong, &c, nullptr, nullptr, stream);
  }
  if (rc != ODEHIP_OK) return rc;
  LinComb lc;
  memset(&lc, 0, sizeof(lc));
  lc.y = y;
  lc.k[0] = k[0];
  lc.n = 1;
  lc.out = xs;
  if (first_step > 0.0) {
    hipLaunchKernelGGL(arm_first_step_kernel, dim3(1), dim3(1), 0, stream, state, t_dev, first_step);
  } else {
    hipLaunchKernelGGL(scaled_sumsq_kernel, dim3(red_grid), dim3(256), 0, stream, y, (const float*)nullptr, y, atol, rtol, n4, part0);
    hipLaunchKernelGGL(scaled_sumsq_kernel, dim3(red_grid), dim3(256), 0, stream, k[0], (const float*)nullptr, y, atol, rtol, n4, part1);
    if (global_norm) {
      const float* arr[2] = {part0, part1};
      const int lens[2] = {red_grid, red_grid};
      if ((rc = reduce_sums(2, arr, lens, nullptr)) != ODEHIP_OK) return rc;
      hipLaunchKernelGGL(init1_kernel, dim3(1), dim3(256), 0, stream, state, g_reduce_buf, g_reduce_buf + 1, 1);
    } else {
      hipLaunchKernelGGL(init1_kernel, dim3(1), dim3(256), 0, stream, state, part0, part1, red_grid);
    }
    lc.c[0] = 1.0f;
    lc.h_ptr = &state->h0;
    hipLaunchKernelGGL(lincomb_kernel, dim3(1024), dim3(256), 0, stream, lc, n4);
    c.k_out = k[1];
    rc = enqueue_f(f, xs, batch, ping, pong, &c, nullptr, nullptr, stream);
    if (rc != ODEHIP_OK) return rc;
    hipLaunchKernelGGL(scaled_sumsq_kernel, dim3(red_grid), dim3(256), 0, stream, k[1], k[0], y, atol, rtol, n4, part2);
    if (global_norm) {
      const float* arr[1] = {part2};
      const int lens[1] = {red_grid};
      if ((rc = reduce_sums(1, arr, lens, nullptr)) != ODEHIP_OK) return rc;
      hipLaunchKernelGGL(init2_kernel, dim3(1), dim3(256), 0, stream, state, g_reduce_buf, 1, t_dev);
    } else {
      hipLaunchKernelGGL(init2_kernel, dim3(1), dim3(256), 0, stream, state, part2, red_grid, t_dev);
    }
  }
  lc.c[0] = (float)dp5::kBeta[0][0];
  lc.h_ptr = &state->h;
  if (saving) lc.out = BL.xin(bws, 0, 1);
  hipLaunchKernelGGL(lincomb_kernel, dim3(1024), dim3(256), 0, stream, lc, n4);  // x2 of the first attempt
  ODEHIP_CHECK_HIP(hipGetLastError());

  FinishArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.st = state;
  fa.t_out = t_dev;
  fa.y = y;
  fa.y1 = y1;
  for (int i = 0; i < 7; ++i) fa.k[i] = k[i];
  fa.x2 = xs;
  fa.out_nchw = out_nchw;
  fa.channels = C;
  fa.state_floats = (long long)st_f;
  fa.off_y1 = (long long)(6 * BL.st);
  fa.off_x2 = (long long)(1 * BL.st);

  ODEHIP_CHECK_HIP(hipMemsetAsync(psync, 0, persist_sync_bytes(batch), stream));   // first attempt's flag area; the controller zeroes it for the next
  // ---- attempted steps
  cx.f = *f;
  cx.n_times = n_times; cx.batch = batch; cx.n_conv_partials = n_conv_partials;
  cx.rtol = rtol; cx.atol = atol; cx.ksc = ksc;
  cx.stream = stream;
  cx.state = state; cx.t_dev = t_dev; cx.part0 = part0; cx.ping = ping; cx.pong = pong; cx.xs = xs; cx.y = y; cx.y1 = y1;
  for (int i = 0; i < 7; ++i) cx.k[i] = k[i];
  cx.psync = psync; cx.reloc = reloc;
  cx.saving = saving; cx.global_norm = global_norm;
  cx.partials64 = n_conv_partials != batch * (C / 32) * 2 * 4;   // (the walks' 64 or 32 partials per sample)
  cx.bl_st = BL.st; cx.bl_hid = BL.hid; cx.bl_nh = BL.NH;
  cx.fa = fa;
  cx.n4 = n4;
  if (async_attempts > 0) {
    // ASYNCHRONOUS start: a fixed number of attempts is enqueued (those behind `done` return at once) and the call returns without
    // waiting for the device; odehip_odeint_dopri5_collect() reads the outcome later and carries on should the solve need more
    for (int i = 0; i < async_attempts; ++i)
      if ((rc = d5_attempt(cx)) != ODEHIP_OK) return rc;
    hipLaunchKernelGGL(seal_kernel, dim3(1024), dim3(256), 0, stream, state, cx.mb, out_nchw, (long long)st_f);
    ODEHIP_CHECK_HIP(hipGetLastError());
    cx.in_use = true;
    *token_out = slot;
    return ODEHIP_OK;
  }
  if ((rc = d5_run_to_done(cx)) != ODEHIP_OK) return rc;
  return d5_collect(cx, stats_host, accepted_host, accepted_cap, saved_out);
}

// ---- the asynchronous pair (ABI 8; sealed since ABI 10).  start = odehip_odeint_dopri5_saving's arguments (max_accept = 0: nothing
// is kept) + how many attempted steps to enqueue before returning; nothing is waited for and no outcome is reported.  Behind the
// last attempt start() enqueues the SEAL: the caller's consumers of `out` are enqueued behind start(), so attempts enqueued any later
// could not complete the trajectory for them -- a solve that is not done at the seal gets its unreached frames NaN-filled and
// collect() reports ODEHIP_ETRUNC.  collect(token) waits for the device (normally long done) and returns what the synchronous call
// returns (status code, stats, accepted-step log, saved flag).  Everything the start call was given (workspace, out, z0) must stay
// untouched until collect; at most four solves may be pending.
extern "C" int odehip_odeint_dopri5_start(const odehip_convstack* f, const float* z0_nchw, const double* t_host, int n_times, int batch,
                                          float rtol, float atol, double first_step, int max_steps, float* out_nchw, int max_accept,
                                          int attempts, int* token_out, void* workspace, size_t workspace_bytes, void* stream_) {
  ODEHIP_REQUIRE(token_out && attempts > 0, "odeint_dopri5_start: token_out is required and attempts must be positive");
  *token_out = -1;
  ODEHIP_REQUIRE(f && workspace_bytes >= (max_accept > 0 ? odehip_dopri5_saving_workspace_bytes(f, batch, n_times, max_accept)
                                                           : odehip_dopri5_workspace_bytes(f, batch, n_times)),
                 "odeint_dopri5_start: workspace too small");
  int saved_dummy = 0;
  return dopri5_forward(f, z0_nchw, t_host, n_times, batch, rtol, atol, first_step, max_steps, 0, out_nchw, nullptr, nullptr, 0, workspace,
                        workspace_bytes, stream_, max_accept, &saved_dummy, attempts, token_out);
}

extern "C" int odehip_odeint_dopri5_collect(int token, int* stats_host, double* accepted_host, int accepted_cap, int* saved_out) {
  ODEHIP_REQUIRE(token >= 0 && token < kD5Slots && g_d5[token].in_use, "odeint_dopri5_collect: no pending solve with token %d", token);
  D5Ctx& cx = g_d5[token];
  cx.in_use = false;   // whatever happens below, the slot is free again
  if (saved_out) *saved_out = 0;
  if (cx.n_times == 1) {
    if (stats_host) stats_host[0] = stats_host[1] = stats_host[2] = stats_host[3] = 0;
    return ODEHIP_OK;
  }
  // the attempts start() enqueued and the seal behind them are all this solve ever gets: wait for them, enqueue nothing
  cx.t_progress = now_s();
  while (cx.mb->steps_done < cx.enq) {
    if (cx.mb->steps_done != cx.seen) {
      cx.seen = cx.mb->steps_done;
      cx.t_progress = now_s();
    }
    if (now_s() - cx.t_progress > 120.0) {
      set_error("odeint_dopri5_collect: no progress from the device for 120 s (steps done %d of %d enqueued)", cx.mb->steps_done, cx.enq);
      return ODEHIP_EHIP;
    }
  }
  if (!cx.mb->done) {
    // the seal runs right behind the last controller: its verdict is at most a kernel away
    ODEHIP_CHECK_HIP(hipStreamSynchronize(cx.stream));
    if (stats_host) {
      stats_host[0] = cx.mb->nfe;
      stats_host[1] = cx.mb->n_accept;
      stats_host[2] = cx.mb->n_reject;
      stats_host[3] = cx.enq;
    }
    set_error("odeint_dopri5_collect: the solve was not finished by the %d attempted steps start() enqueued (%d accepted, %d rejected "
              "so far); the frames it had not reached were NaN-filled, because the work enqueued behind start() has already consumed "
              "`out`.  Start again with more attempts, or use the synchronous call", cx.enq, cx.mb->n_accept, cx.mb->n_reject);
    return ODEHIP_ETRUNC;
  }
  return d5_collect(cx, stats_host, accepted_host, accepted_cap, saved_out);
}
