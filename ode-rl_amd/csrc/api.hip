// api.hip -- host side of the C ABI (include/odecgru_hip.h): argument checks, workspace carving and the
// launch sequences of f(y) and of the fixed-grid solvers.  Enqueue-only on the caller's stream.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "odehip_internal.h"
#include "persist.h"

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
  if (f->w_fused) {
    ODEHIP_REQUIRE(f->ks == 3, "convstack: the fused bf16 image needs 3x3 layers");
    for (int i = 0; i <= f->n_convs; ++i)
      ODEHIP_REQUIRE(f->channels[i] == 64, "convstack: the fused bf16 image needs 64-channel layers (channels[%d] = %d)", i, f->channels[i]);
  }
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
int upload_floats(float* dst, const float* src, int n, hipStream_t stream) {
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
// `hidden`, if given, holds n_convs-1 distinct buffers for the ReLU outputs (save_for_backward); else ping/pong.

int enqueue_f(const odehip_convstack* f, const float* x_q4, int batch, float* ping, float* pong, const CombineArgs* cmb,
              float* plain_dst, const int* skip, hipStream_t stream) {
  return enqueue_f_saving(f, x_q4, batch, nullptr, ping, pong, cmb, plain_dst, skip, stream);
}

int enqueue_f_saving(const odehip_convstack* f, const float* x_q4, int batch, float* const* hidden, float* ping, float* pong,
                     const CombineArgs* cmb, float* plain_dst, const int* skip, hipStream_t stream) {
  if (f->w_fused) {  // bf16, every layer 64 -> 64: one launch for the whole stack, one workgroup per sample
    FusedArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.x = x_q4;
    fa.w_fused = f->w_fused;
    fa.n_layers = f->n_convs;
    for (int l = 0; l < f->n_convs; ++l) {
      fa.bias[l] = f->bias[l];
      if (l < f->n_convs - 1) fa.store[l] = hidden ? hidden[l] : nullptr;
    }
    fa.last.qout = 16;
    fa.last.batch = batch;
    fa.last.skip = skip;
    if (cmb) {
      fa.last.combine = 1;
      fa.last.cmb = *cmb;
    } else {
      fa.last.dst = plain_dst;
    }
    return launch_fstack_bf16(fa, batch, stream);
  }
  // the layers of one evaluation as ONE persistent launch (64-channel fp32 stacks of at most 5 layers; inside a driver that
  // records a longer sequence this scope stays inactive and the layers go to that recorder)
  PersistScope one_eval;
  int rcp = one_eval.begin(f, nullptr, f->n_convs, /*small=*/true);
  if (rcp != ODEHIP_OK) return rcp;
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
    a.w_wino = f->w_wino[l];
    a.w_bf16 = f->w_bf16[l];
    a.bias = f->bias[l];
    a.batch = batch;
    a.skip = skip;
    if (!last) {
      a.relu = 1;
      a.dst = hidden ? hidden[l] : ((l & 1) ? pong : ping);
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
  return one_eval.finish(nullptr, nullptr, nullptr, batch, nullptr, f->ks, stream);
}

int enqueue_dgrad_chain(const odehip_convstack* f, const odehip_convstack* fd, int batch, float* const* gp,
                        const float* const* hidden, const ConvArgs& last, hipStream_t stream) {
  const int NL = f->n_convs;
  if (fd->w_fused) {  // bf16, 64-channel stack: the whole chain in one launch (layers in execution order NL-1 .. 0)
    FusedArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.x = gp[NL - 1];
    fa.w_fused = fd->w_fused;
    fa.n_layers = NL;
    for (int e = 0; e < NL - 1; ++e) {
      const int l = NL - 1 - e;
      fa.store[e] = gp[l - 1];
      fa.mask[e] = hidden[l - 1];
    }
    fa.last = last;
    fa.last.qout = 16;
    fa.last.batch = batch;
    return launch_fstack_bf16(fa, batch, stream);
  }
  PersistScope one_chain;  // as in enqueue_f_saving: the chain as one persistent launch
  int rcp = one_chain.begin(f, fd, NL, /*small=*/true);
  if (rcp != ODEHIP_OK) return rcp;
  for (int l = NL - 1; l >= 0; --l) {
    ConvArgs a;
    if (l > 0) {
      memset(&a, 0, sizeof(a));
      a.combine = 2;
      a.bwd.mask_src = hidden[l - 1];  // ReLU output that fed conv l
      a.bwd.sc_c = 1.0f;
      a.dst = gp[l - 1];
    } else {
      a = last;
    }
    a.src1 = gp[l];  // gradient w.r.t. the output of conv l
    a.src2 = nullptr;
    a.q1 = a.qin = f->channels[l + 1] / 4;
    a.qout = f->channels[l] / 4;
    a.w_packed = fd->w_packed[l];
    a.w_wino = fd->w_wino[l];
    a.w_bf16 = fd->w_bf16[l];
    a.bias = nullptr;
    a.batch = batch;
    int r = launch_conv(a, f->ks, stream);
    if (r != ODEHIP_OK) return r;
  }
  return one_chain.finish(nullptr, nullptr, nullptr, batch, nullptr, f->ks, stream);
}

}  // namespace odehip

using namespace odehip;

extern "C" const char* odehip_last_error(void) { return g_err; }
extern "C" int odehip_version(void) { return ODEHIP_ABI_VERSION; }
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
  a.w_wino = d->w_wino;
  a.w_bf16 = d->w_bf16;
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
// diagnostic: n back-to-back evaluations of f on Q4 tensors (tools/fused_microbench.py); scratch = 2 hidden-sized buffers
extern "C" int odehip_debug_repeat_f(const odehip_convstack* f, const float* x_q4, float* out_q4, float* scratch, int batch, int n,
                                     void* stream) {
  int rc = check_stack(f);
  if (rc != ODEHIP_OK) return rc;
  const size_t hid = (size_t)batch * max_hidden(f) * kPix;
  for (int i = 0; i < n; ++i) {
    rc = enqueue_f(f, x_q4, batch, scratch, scratch + hid, nullptr, out_q4, nullptr, (hipStream_t)stream);
    if (rc != ODEHIP_OK) return rc;
  }
  return ODEHIP_OK;
}

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

