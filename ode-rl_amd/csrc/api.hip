// api.hip -- host side of the C ABI (include/odecgru_hip.h): argument checks, workspace carving and the
// launch sequences of f(y) and of the fixed-grid solvers.  Enqueue-only on the caller's stream.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "odehip_internal.h"

namespace odehip {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return ODEHIP_EHIP;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base((char*)p) {}
  float* take(size_t bytes) {
    float* r = (float*)(base + off);
    off += align_up(bytes, 256);
    return r;
  }
};

int check_stack(const odehip_convstack* f) {
  ODEHIP_REQUIRE(f, "convstack: null descriptor");
  ODEHIP_REQUIRE(f->n_convs >= 1 && f->n_convs <= ODEHIP_MAX_LAYERS, "convstack: n_convs %d out of range", f->n_convs);
  ODEHIP_REQUIRE(f->ks == 3 || f->ks == 1 || f->ks == 5, "convstack: kernel size %d unsupported", f->ks);
  ODEHIP_REQUIRE(!f->final_tanh, "convstack: final_act=True (Tanh head) is not supported by the HIP path");
  for (int i = 0; i <= f->n_convs; ++i)
    ODEHIP_REQUIRE(f->channels[i] > 0 && f->channels[i] % 32 == 0,
                   "convstack: channels[%d] = %d must be a positive multiple of 32", i, f->channels[i]);
  for (int i = 0; i < f->n_convs; ++i)
    ODEHIP_REQUIRE(f->w_packed[i] && f->bias[i], "convstack: layer %d has null weights/bias", i);
  return ODEHIP_OK;
}

int max_hidden(const odehip_convstack* f) {
  int m = 32;
  for (int i = 1; i < f->n_convs; ++i) m = f->channels[i] > m ? f->channels[i] : m;
  return m;
}

// Scalars travel to the device as kernel arguments (no pageable-host memcpy on the stream).
struct FloatPack {
  float v[64];
};
__global__ void fill_floats_kernel(float* dst, FloatPack p, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = p.v[threadIdx.x];
}
static int upload_floats(float* dst, const float* src, int n, hipStream_t stream) {
  for (int o = 0; o < n; o += 64) {
    FloatPack p;
    const int m = (n - o < 64) ? n - o : 64;
    for (int i = 0; i < m; ++i) p.v[i] = src[o + i];
    hipLaunchKernelGGL(fill_floats_kernel, dim3(1), dim3(64), 0, stream, dst + o, p, m);
    ODEHIP_CHECK_HIP(hipGetLastError());
  }
  return ODEHIP_OK;
}

static inline size_t state_bytes(int batch, int channels) { return (size_t)batch * channels * kPix * sizeof(float); }

// Enqueue f(x) with the stage-combine fused into the last conv.  ping/pong hold hidden activations.
int enqueue_f(const odehip_convstack* f, const float* x_q4, int batch, float* ping, float* pong, const CombineArgs* cmb,
              float* plain_dst, const int* skip, hipStream_t stream) {
  const float* cur = x_q4;
  for (int l = 0; l < f->n_convs; ++l) {
    const bool last = (l == f->n_convs - 1);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.src1 = cur;
    a.src2 = nullptr;
    a.q1 = a.qin = f->channels[l] / 4;
    a.qout = f->channels[l + 1] / 4;
    a.w_packed = f->w_packed[l];
    a.bias = f->bias[l];
    a.batch = batch;
    a.skip = skip;
    if (!last) {
      a.relu = 1;
      a.dst = (l & 1) ? pong : ping;
      cur = a.dst;
    } else if (cmb) {
      a.combine = 1;
      a.cmb = *cmb;
    } else {
      a.relu = 0;
      a.dst = plain_dst;
    }
    int rc = launch_conv(a, f->ks, stream);
    if (rc != ODEHIP_OK) return rc;
  }
  return ODEHIP_OK;
}

}  // namespace odehip

using namespace odehip;

extern "C" const char* odehip_last_error(void) { return g_err; }
extern "C" int odehip_version(void) { return 1; }
extern "C" void odehip_set_debug_flags(int flags) { g_debug_flags = flags; }
extern "C" void odehip_set_debug_buffer(void* p) { g_debug_buf = (unsigned long long*)p; }

extern "C" int odehip_conv_q4(const odehip_conv_desc* d, void* stream) {
  ODEHIP_REQUIRE(d, "conv_q4: null descriptor");
  ODEHIP_REQUIRE(d->cin1 > 0 && d->cin1 % 4 == 0 && d->cin % 4 == 0 && d->cin1 <= d->cin, "conv_q4: bad cin split %d/%d",
                 d->cin1, d->cin);
  ODEHIP_REQUIRE(d->cout > 0 && d->cout % 32 == 0, "conv_q4: cout must be a multiple of 32 (got %d)", d->cout);
  ODEHIP_REQUIRE(d->dst, "conv_q4: null dst");
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.src1 = d->src1;
  a.src2 = d->src2;
  a.q1 = d->cin1 / 4;
  a.qin = d->cin / 4;
  a.qout = d->cout / 4;
  a.w_packed = d->w_packed;
  a.bias = d->bias;
  a.dst = d->dst;
  a.batch = d->batch;
  a.relu = d->relu;
  return launch_conv(a, d->ks, (hipStream_t)stream);
}

// diagnostic: n back-to-back launches from C (tools/conv_microbench.py), to take Python out of the loop
extern "C" int odehip_debug_repeat_conv(const odehip_conv_desc* d, int n, void* stream) {
  for (int i = 0; i < n; ++i) {
    int rc = odehip_conv_q4(d, stream);
    if (rc != ODEHIP_OK) return rc;
  }
  return ODEHIP_OK;
}

// workspace of f(y): [x_q4 | ping | pong | out_q4]
extern "C" size_t odehip_convstack_workspace_bytes(const odehip_convstack* f, int batch) {
  if (!f || batch <= 0) return 0;
  const size_t hid = align_up(state_bytes(batch, max_hidden(f)), 256);
  return align_up(state_bytes(batch, f->channels[0]), 256) + 2 * hid +
         align_up(state_bytes(batch, f->channels[f->n_convs]), 256);
}

extern "C" int odehip_convstack_forward(const odehip_convstack* f, const float* y_nchw, float* out_nchw, int batch,
                                        int negate, void* workspace, size_t workspace_bytes, void* stream_) {
  int rc = check_stack(f);
  if (rc != ODEHIP_OK) return rc;
  ODEHIP_REQUIRE(y_nchw && out_nchw && workspace, "convstack_forward: null pointer");
  ODEHIP_REQUIRE(batch > 0, "convstack_forward: batch must be positive");
  ODEHIP_REQUIRE(workspace_bytes >= odehip_convstack_workspace_bytes(f, batch), "convstack_forward: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  Carver ws(workspace);
  float* x = ws.take(state_bytes(batch, f->channels[0]));
  float* ping = ws.take(state_bytes(batch, max_hidden(f)));
  float* pong = ws.take(state_bytes(batch, max_hidden(f)));
  float* out = ws.take(state_bytes(batch, f->channels[f->n_convs]));
  rc = odehip_nchw_to_q4(y_nchw, x, batch, f->channels[0], stream);
  if (rc != ODEHIP_OK) return rc;
  CombineArgs cmb;
  memset(&cmb, 0, sizeof(cmb));
  cmb.k_out = out;
  cmb.k_scale = negate ? -1.0f : 1.0f;
  rc = enqueue_f(f, x, batch, ping, pong, &cmb, nullptr, nullptr, stream);
  if (rc != ODEHIP_OK) return rc;
  return odehip_q4_to_nchw(out, out_nchw, batch, f->channels[f->n_convs], stream);
}

// ---------------------------------------------------------------------------------------------
// Fixed-grid odeint.  Workspace: [h[n_times-1] | ping | pong | x_stage | k0..k2 | y_q4[n_times]]
// ---------------------------------------------------------------------------------------------
static int n_k_buffers(int method) { return method == ODEHIP_RK4 ? 3 : (method == ODEHIP_MIDPOINT ? 1 : 0); }

extern "C" size_t odehip_odeint_workspace_bytes(const odehip_convstack* f, int batch, int n_times, int method,
                                                int save_for_backward) {
  if (!f || batch <= 0 || n_times <= 0) return 0;
  (void)save_for_backward;
  const size_t st = align_up(state_bytes(batch, f->channels[0]), 256);
  const size_t hid = align_up(state_bytes(batch, max_hidden(f)), 256);
  return align_up((size_t)n_times * sizeof(float), 256) + 2 * hid + st + (size_t)n_k_buffers(method) * st +
         (size_t)n_times * st;
}

extern "C" int odehip_odeint_fixed(const odehip_convstack* f, int method, const float* z0_nchw, const double* t_host,
                                   int n_times, int batch, float* out_nchw, int save_for_backward, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
  int rc = check_stack(f);
  if (rc != ODEHIP_OK) return rc;
  ODEHIP_REQUIRE(method == ODEHIP_EULER || method == ODEHIP_MIDPOINT || method == ODEHIP_RK4,
                 "odeint_fixed: method %d is not a fixed-grid method", method);
  ODEHIP_REQUIRE(z0_nchw && t_host && out_nchw && workspace, "odeint_fixed: null pointer");
  ODEHIP_REQUIRE(n_times >= 1 && batch > 0, "odeint_fixed: bad sizes (n_times %d, batch %d)", n_times, batch);
  ODEHIP_REQUIRE(f->channels[0] == f->channels[f->n_convs], "odeint_fixed: f must map C -> C channels (%d -> %d)",
                 f->channels[0], f->channels[f->n_convs]);
  ODEHIP_REQUIRE(!save_for_backward, "odeint_fixed: save_for_backward is not implemented yet");
  for (int i = 1; i < n_times; ++i)
    ODEHIP_REQUIRE(t_host[i] > t_host[i - 1], "odeint_fixed: t must be strictly increasing (t[%d]=%g, t[%d]=%g)", i - 1,
                   t_host[i - 1], i, t_host[i]);
  ODEHIP_REQUIRE(workspace_bytes >= odehip_odeint_workspace_bytes(f, batch, n_times, method, save_for_backward),
                 "odeint_fixed: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  const int C = f->channels[0];
  const size_t st = state_bytes(batch, C);
  const size_t st_f = st / sizeof(float);

  Carver ws(workspace);
  float* hdev = ws.take((size_t)n_times * sizeof(float));
  float* ping = ws.take(state_bytes(batch, max_hidden(f)));
  float* pong = ws.take(state_bytes(batch, max_hidden(f)));
  float* xs = ws.take(st);
  float* k[3] = {nullptr, nullptr, nullptr};
  for (int i = 0; i < n_k_buffers(method); ++i) k[i] = ws.take(st);
  float* yq = ws.take((size_t)n_times * st);

  // out[0] = z0 (torchdiffeq: solution[0] = y0)
  ODEHIP_CHECK_HIP(hipMemcpyAsync(out_nchw, z0_nchw, st, hipMemcpyDeviceToDevice, stream));
  rc = odehip_nchw_to_q4(z0_nchw, yq, batch, C, stream);
  if (rc != ODEHIP_OK) return rc;
  if (n_times == 1) return ODEHIP_OK;

  // step sizes: dt = t1 - t0 in float64, rounded to fp32 when it meets the state (torchdiffeq semantics)
  float hbuf[4096];
  ODEHIP_REQUIRE(n_times <= 4096, "odeint_fixed: at most 4096 time points per call (got %d)", n_times);
  for (int i = 0; i + 1 < n_times; ++i) hbuf[i] = (float)(t_host[i + 1] - t_host[i]);
  rc = upload_floats(hdev, hbuf, n_times - 1, stream);
  if (rc != ODEHIP_OK) return rc;

  for (int n = 0; n + 1 < n_times; ++n) {
    const float* y = yq + (size_t)n * st_f;
    float* ynew = yq + (size_t)(n + 1) * st_f;
    float* ynew_nchw = out_nchw + (size_t)(n + 1) * st_f;
    CombineArgs c;
    memset(&c, 0, sizeof(c));
    c.y = y;
    c.h_ptr = hdev + n;
    c.k_scale = 1.0f;
    if (method == ODEHIP_EULER) {
      // y1 = y + h*f(y)
      c.n_prev = 0;
      c.c2[0] = 1.0f;
      c.out2 = ynew;
      c.out2_nchw = ynew_nchw;
      rc = enqueue_f(f, y, batch, ping, pong, &c, nullptr, nullptr, stream);
      if (rc != ODEHIP_OK) return rc;
    } else if (method == ODEHIP_MIDPOINT) {
      // x = y + h/2*k1 ; y1 = y + h*f(x)
      c.n_prev = 0;
      c.c1[0] = 0.5f;
      c.out1 = xs;
      rc = enqueue_f(f, y, batch, ping, pong, &c, nullptr, nullptr, stream);
      if (rc != ODEHIP_OK) return rc;
      c.c1[0] = 0.0f;
      c.out1 = nullptr;
      c.c2[0] = 1.0f;
      c.out2 = ynew;
      c.out2_nchw = ynew_nchw;
      rc = enqueue_f(f, xs, batch, ping, pong, &c, nullptr, nullptr, stream);
      if (rc != ODEHIP_OK) return rc;
    } else {
      // 3/8 rule (torchdiffeq rk4_alt_step_func)
      const float third = 1.0f / 3.0f;
      // stage 1: k1 = f(y); x2 = y + h*(k1/3)
      c.n_prev = 0;
      c.k_out = k[0];
      c.c1[0] = third;
      c.out1 = xs;
      rc = enqueue_f(f, y, batch, ping, pong, &c, nullptr, nullptr, stream);
      if (rc != ODEHIP_OK) return rc;
      // stage 2: k2 = f(x2); x3 = y + h*(k2 - k1/3)
      c.n_prev = 1;
      c.k_prev[0] = k[0];
      c.k_out = k[1];
      c.c1[0] = -third;
      c.c1[1] = 1.0f;
      rc = enqueue_f(f, xs, batch, ping, pong, &c, nullptr, nullptr, stream);
      if (rc != ODEHIP_OK) return rc;
      // stage 3: k3 = f(x3); x4 = y + h*(k1 - k2 + k3)
      c.n_prev = 2;
      c.k_prev[1] = k[1];
      c.k_out = k[2];
      c.c1[0] = 1.0f;
      c.c1[1] = -1.0f;
      c.c1[2] = 1.0f;
      rc = enqueue_f(f, xs, batch, ping, pong, &c, nullptr, nullptr, stream);
      if (rc != ODEHIP_OK) return rc;
      // stage 4: k4 = f(x4); y1 = y + h*(k1 + 3(k2+k3) + k4)/8
      c.n_prev = 3;
      c.k_prev[2] = k[2];
      c.k_out = nullptr;
      c.out1 = nullptr;
      c.c2[0] = 0.125f;
      c.c2[1] = 0.375f;
      c.c2[2] = 0.375f;
      c.c2[3] = 0.125f;
      c.out2 = ynew;
      c.out2_nchw = ynew_nchw;
      rc = enqueue_f(f, xs, batch, ping, pong, &c, nullptr, nullptr, stream);
      if (rc != ODEHIP_OK) return rc;
    }
  }
  return ODEHIP_OK;
}
