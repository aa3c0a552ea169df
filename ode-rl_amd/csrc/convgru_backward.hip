// convgru_backward.hip -- training path of the ODE-ConvGRU encoder (gfx950): forward with everything the reverse sweep
// needs kept in the workspace, and the reverse sweep itself.  What `loss.backward()` computes through
// /root/reference/modules/ODEConvGRUCell.py:32-78 and /root/reference/modules/ConvGRUCell.py:72-82 (train_test.py:204).
//
// Per observed frame (reverse order):
//   forward   k = f_enc(h); h_ode = h + dt k; G = conv5(cat(x, h_ode)); (z, r) = sigmoid(GN(G)); rh = r*h_ode;
//             Cr = conv5(cat(x, rh)); c = tanh(GN(Cr)); h' = (1-z) h_ode + z c
//   backward  gn_update_bwd : gh' -> gCr (through tanh + GroupNorm), gz_pre = gh' (c-h_ode) z(1-z), gh_ode = gh' (1-z)
//             conv5^T(gCr)  : -> gx_c (frame part), g_rh (state part)           MFMA ring kernel, transposed+flipped slices
//             gn_gates_bwd  : -> gG (through sigmoid + GroupNorm); gh_ode += g_rh r
//             conv5^T(gG)   : -> gx = gx_c + . (frame gradient), seed = gh_ode + .   (fused in the conv epilogues)
//             f_enc^T chain : gh = seed + dt J_f(h)^T seed                    (3x3 dgrad kernels, ReLU mask fused)
//   at the end one weight-gradient launch per (layer, 64x64 tile) over ALL frames (wgrad.hip), a fixed-order reduction of
//   the per-(frame, sample) GroupNorm affine partials, and the Q4 -> NCHW conversion of the frame gradients.
// GroupNorm statistics, sigmoid/tanh outputs are recomputed from the saved pre-normalisation tensors inside the fused
// backward kernels (a 32-channel group of one sample lives in the registers of one workgroup) -- nothing but conv
// outputs is stored.  Deterministic: no float atomics.
#include <string.h>

#include <vector>

#include "odehip_internal.h"
#include "persist.h"

namespace odehip {

typedef float f32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ float bsum256(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__device__ __forceinline__ void load_grp(const float* src, int b, int groups, int g, f32x4 (&v)[8]) {
  const f32x4* p = (const f32x4*)(src + ((size_t)(b * groups + g) * 8) * kPix * 4) + threadIdx.x;
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = p[q * kPix];
}
__device__ __forceinline__ void store_grp(float* dst, int b, int groups, int g, const f32x4 (&v)[8]) {
  f32x4* p = (f32x4*)(dst + ((size_t)(b * groups + g) * 8) * kPix * 4) + threadIdx.x;
#pragma unroll
  for (int q = 0; q < 8; ++q) p[q * kPix] = v[q];
}

// x -> xhat (in place); returns rstd.  Same two-pass statistics as the forward kernel (convgru.hip group_norm).
__device__ __forceinline__ float normalise(f32x4 (&v)[8], float* sh) {
  float s = 0.0f;
#pragma unroll
  for (int q = 0; q < 8; ++q) s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
  const float mean = bsum256(s, sh) * (1.0f / 8192.0f);
  float ss = 0.0f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 d = v[q] - mean;
    ss += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
  }
  const float var = bsum256(ss, sh) * (1.0f / 8192.0f);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = (v[q] - mean) * rstd;
  return rstd;
}

// GroupNorm backward over one (sample, group): gy = gradient w.r.t. the affine output, xh = xhat.  Writes the per-channel
// partial sums of this sample (dgamma = sum_p gy*xh, dbeta = sum_p gy) and turns gy into the gradient w.r.t. the input.
__device__ __forceinline__ void gn_backward(f32x4 (&gy)[8], const f32x4 (&xh)[8], const float* gamma, int g, float rstd,
                                            float* dgamma_part, float* dbeta_part, float* sh, float* shc) {
  // per-channel sums over the 256 pixels: wave shuffle, then the four waves meet in LDS (64 values)
  float pg[32], pb[32];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float a = gy[q][c] * xh[q][c], b = gy[q][c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
      }
      pg[q * 4 + c] = a;
      pb[q * 4 + c] = b;
    }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      shc[w * 64 + i] = pg[i];
      shc[w * 64 + 32 + i] = pb[i];
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const float v = (shc[threadIdx.x] + shc[64 + threadIdx.x]) + (shc[128 + threadIdx.x] + shc[192 + threadIdx.x]);
    if (threadIdx.x < 32) dgamma_part[g * 32 + threadIdx.x] = v;
    else dbeta_part[g * 32 + threadIdx.x - 32] = v;
  }
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 ga = *(const f32x4*)(gamma + g * 32 + q * 4);
    gy[q] *= ga;  // dxhat
    s1 += (gy[q].x + gy[q].y) + (gy[q].z + gy[q].w);
    const f32x4 m = gy[q] * xh[q];
    s2 += (m.x + m.y) + (m.z + m.w);
  }
  const float m1 = bsum256(s1, sh) * (1.0f / 8192.0f);
  const float m2 = bsum256(s2, sh) * (1.0f / 8192.0f);
#pragma unroll
  for (int q = 0; q < 8; ++q) gy[q] = (gy[q] - m1 - xh[q] * m2) * rstd;
}

// backward of  h' = (1-z) h_ode + z tanh(GN(cand_raw))  for one (sample, group of 32 hidden channels)
__global__ __launch_bounds__(256) void gn_update_bwd_kernel(const float* __restrict__ cand_raw, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ gh,
                                                            const float* __restrict__ z, const float* __restrict__ h_ode,
                                                            float* __restrict__ g_cand_raw, float* __restrict__ gz_pre,
                                                            float* __restrict__ gh_ode, float* __restrict__ dgamma_part,
                                                            float* __restrict__ dbeta_part, int hid_groups) {
  __shared__ float sh[4];
  __shared__ float shc[256];
  const int g = blockIdx.x, b = blockIdx.y;
  f32x4 xh[8], gy[8], t[8];
  load_grp(cand_raw, b, hid_groups, g, xh);
  const float rstd = normalise(xh, sh);
  load_grp(gh, b, hid_groups, g, gy);
  f32x4 zz[8], hh[8];
  load_grp(z, b, hid_groups, g, zz);
  load_grp(h_ode, b, hid_groups, g, hh);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 ga = *(const f32x4*)(gamma + g * 32 + q * 4), be = *(const f32x4*)(beta + g * 32 + q * 4);
    const f32x4 cn = xh[q] * ga + be;
    const f32x4 c = {tanhf(cn.x), tanhf(cn.y), tanhf(cn.z), tanhf(cn.w)};
    const f32x4 g0 = gy[q];
    t[q] = g0 * (c - hh[q]) * zz[q] * (1.0f - zz[q]);  // gradient w.r.t. the z pre-activation (after GroupNorm)
    hh[q] = g0 * (1.0f - zz[q]);                        // direct path to h_ode
    gy[q] = g0 * zz[q] * (1.0f - c * c);                // gradient w.r.t. GN(cand_raw)
  }
  store_grp(gz_pre, b, hid_groups, g, t);
  store_grp(gh_ode, b, hid_groups, g, hh);
  const int H = hid_groups * 32;
  gn_backward(gy, xh, gamma, g, rstd, dgamma_part + (size_t)b * H, dbeta_part + (size_t)b * H, sh, shc);
  store_grp(g_cand_raw, b, hid_groups, g, gy);
}

// backward of  (z, r) = sigmoid(GN(gates_raw)), rh = r*h_ode  for one (sample, group of the 2*hidden gate channels)
__global__ __launch_bounds__(256) void gn_gates_bwd_kernel(const float* __restrict__ gates_raw, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ gz_pre,
                                                           const float* __restrict__ g_rh, const float* __restrict__ h_ode,
                                                           float* __restrict__ gh_ode, float* __restrict__ g_gates_raw,
                                                           float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                                                           int hid_groups) {
  __shared__ float sh[4];
  __shared__ float shc[256];
  const int g = blockIdx.x, b = blockIdx.y;
  f32x4 xh[8], gy[8];
  load_grp(gates_raw, b, 2 * hid_groups, g, xh);
  const float rstd = normalise(xh, sh);
  const bool is_z = g < hid_groups;
  if (is_z) {
    load_grp(gz_pre, b, hid_groups, g, gy);
  } else {
    const int gh_i = g - hid_groups;
    f32x4 hh[8], acc[8];
    load_grp(g_rh, b, hid_groups, gh_i, gy);
    load_grp(h_ode, b, hid_groups, gh_i, hh);
    load_grp(gh_ode, b, hid_groups, gh_i, acc);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 ga = *(const f32x4*)(gamma + g * 32 + q * 4), be = *(const f32x4*)(beta + g * 32 + q * 4);
      const f32x4 n = xh[q] * ga + be;
      const f32x4 r = {1.0f / (1.0f + __expf(-n.x)), 1.0f / (1.0f + __expf(-n.y)), 1.0f / (1.0f + __expf(-n.z)),
                       1.0f / (1.0f + __expf(-n.w))};
      acc[q] += gy[q] * r;                       // rh = r*h_ode: path to h_ode
      gy[q] = gy[q] * hh[q] * r * (1.0f - r);    // path to the r pre-activation
    }
    store_grp(gh_ode, b, hid_groups, gh_i, acc);
  }
  const int G2 = 2 * hid_groups * 32;
  gn_backward(gy, xh, gamma, g, rstd, dgamma_part + (size_t)b * G2, dbeta_part + (size_t)b * G2, sh, shc);
  store_grp(g_gates_raw, b, 2 * hid_groups, g, gy);
}

// (grad_mean, grad_std) NCHW + sign of the head's std half -> gradient of the head output (B, 2*out_ch) Q4
__global__ __launch_bounds__(256) void merge_mean_std_grad_kernel(const float* __restrict__ gmean, const float* __restrict__ gstd,
                                                                  const float* __restrict__ head_out, float* __restrict__ dst,
                                                                  int total, int out_quads) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int p = idx & 255, bq = idx >> 8, q = bq % (2 * out_quads), b = bq / (2 * out_quads);
  f32x4 v;
  if (q < out_quads) {
    const float* s = gmean + ((size_t)b * out_quads + q) * 4 * kPix + p;
    v = {s[0], s[kPix], s[2 * kPix], s[3 * kPix]};
  } else {
    const float* s = gstd + ((size_t)b * out_quads + (q - out_quads)) * 4 * kPix + p;
    const f32x4 o = *(const f32x4*)(head_out + (size_t)idx * 4);
    v = {s[0], s[kPix], s[2 * kPix], s[3 * kPix]};
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = o[c] > 0.0f ? v[c] : (o[c] < 0.0f ? -v[c] : 0.0f);  // d|x|/dx, 0 at 0 as torch
  }
  *(f32x4*)(dst + (size_t)idx * 4) = v;
}

// out[c] = sum_k part[k][c] in a fixed order: one workgroup per column, strided partial sums + a fixed LDS tree
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, int n_rows, int n_cols,
                                                          float* __restrict__ out) {
  __shared__ float sh[256];
  const int c = blockIdx.x;
  float s = 0.0f;
  for (int k = threadIdx.x; k < n_rows; k += 256) s += part[(size_t)k * n_cols + c];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = sh[0];
}

struct U32Pack {
  unsigned v[64];
};
__global__ void fill_u32_kernel(unsigned* dst, U32Pack p, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = p.v[threadIdx.x];
}
static int upload_bytes(void* dst, const void* src, size_t bytes, hipStream_t stream) {  // bytes % 4 == 0
  const unsigned* s = (const unsigned*)src;
  const int n = (int)(bytes / 4);
  for (int o = 0; o < n; o += 64) {
    U32Pack p;
    const int m = n - o < 64 ? n - o : 64;
    memcpy(p.v, s + o, (size_t)m * 4);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(1), dim3(64), 0, stream, (unsigned*)dst + o, p, m);
  }
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

// forward pieces shared with the inference path (convgru.hip)
int cell_step_q4(const odehip_convgru_cell* c, const float* x, const float* h, float* h_out, float* h_out_nchw,
                 long long nchw_batch_stride, int batch, float* gates_raw, float* z, float* rh, float* cand_raw, hipStream_t stream);
int conv_layer_q4(const float* src1, const float* src2, int cin1, int cin, int cout, int ks, const float* wp, const float* bias,
                  float* dst, int relu, int batch, hipStream_t stream);
void launch_split_mean_std(const float* head_out, float* mean, float* stdv, int batch, int out_ch, hipStream_t stream);
int check_cell_desc(const odehip_convgru_cell* c);


// Workspace of the training path; one place knows where everything lives.
struct EncLayout {
  int T, B, C, NH, HH, OUT2;
  size_t hs, fh, hh, ho;
  size_t off_dts, off_frames, off_ping, off_pong, off_hstate, off_hidden, off_hode, off_gates, off_z, off_rh, off_cand, off_headhid,
      off_headout;
  size_t off_gp, off_ggates, off_gcand, off_gx, off_gzpre, off_ghode, off_gxc, off_grh, off_gh, off_gheadout, off_gheadhid, off_pgg,
      off_pgc, off_tab, off_slab, total;
  EncLayout(const odehip_encoder* e, int n_frames, int batch) {
    T = n_frames; B = batch; C = e->cell.hidden; NH = e->f_enc.n_convs - 1; HH = e->head_hidden; OUT2 = 2 * e->out_ch;
    hs = al256((size_t)B * C * kPix * 4);
    int cmax = 32;
    for (int i = 0; i <= e->f_enc.n_convs; ++i) cmax = e->f_enc.channels[i] > cmax ? e->f_enc.channels[i] : cmax;
    fh = al256((size_t)B * cmax * kPix * 4);
    hh = al256((size_t)B * HH * kPix * 4);
    ho = al256((size_t)B * OUT2 * kPix * 4);
    size_t o = 0;
    auto take = [&](size_t b) { size_t r = o; o += al256(b); return r; };
    off_dts = take((size_t)T * 4);
    off_frames = take((size_t)T * hs);
    off_ping = take(fh);
    off_pong = take(fh);
    off_hstate = take((size_t)(T + 1) * hs);
    off_hidden = take((size_t)T * (NH > 0 ? NH : 1) * fh);
    off_hode = take((size_t)T * hs);
    off_gates = take((size_t)T * 2 * hs);
    off_z = take((size_t)T * hs);
    off_rh = take((size_t)T * hs);
    off_cand = take((size_t)T * hs);
    off_headhid = take(hh);
    off_headout = take(ho);
    off_gp = take((size_t)T * (NH + 1) * fh);
    off_ggates = take((size_t)T * 2 * hs);
    off_gcand = take((size_t)T * hs);
    off_gx = take((size_t)T * hs);
    off_gzpre = take(hs);
    off_ghode = take(hs);
    off_gxc = take(hs);
    off_grh = take(hs);
    off_gh = take(2 * hs);
    off_gheadout = take(ho);
    off_gheadhid = take(hh);
    off_pgg = take((size_t)2 * T * B * 2 * C * 4);
    off_pgc = take((size_t)2 * T * B * C * 4);
    off_tab = take((size_t)(ODEHIP_MAX_LAYERS + 6) * T * sizeof(WgradPair));  // every weight-gradient table of a backward pass: ONE upload
    off_slab = take(((size_t)B * wgrad_esplit_max(B) + 1) * kWgradSlabFloats * 4);
    total = o;
  }
  float* p(void* ws, size_t off) const { return (float*)((char*)ws + off); }
  float* frame(void* ws, int i) const { return p(ws, off_frames + (size_t)i * hs); }
  float* hstate(void* ws, int k) const { return p(ws, off_hstate + (size_t)k * hs); }
  float* hidden(void* ws, int idx, int l) const { return p(ws, off_hidden + ((size_t)idx * NH + l) * fh); }
  float* gp(void* ws, int idx, int l) const { return p(ws, off_gp + ((size_t)idx * (NH + 1) + l) * fh); }
  float* per(void* ws, size_t off, int idx, size_t bytes) const { return p(ws, off + (size_t)idx * bytes); }
};

static int check_encoder(const odehip_encoder* e, const char* who) {
  ODEHIP_REQUIRE(e, "%s: null descriptor", who);
  int rc = check_stack(&e->f_enc);
  if (rc != ODEHIP_OK) return rc;
  rc = check_cell_desc(&e->cell);
  if (rc != ODEHIP_OK) return rc;
  const int C = e->cell.hidden;
  ODEHIP_REQUIRE(e->cell.input == C && e->f_enc.channels[0] == C && e->f_enc.channels[e->f_enc.n_convs] == C,
                 "%s: encoder dynamics and cell must share the channel count (%d)", who, C);
  ODEHIP_REQUIRE(e->w_head0 && e->b_head0 && e->w_head1 && e->b_head1, "%s: bad transform_z0 head", who);
  ODEHIP_REQUIRE(C % 64 == 0 && e->head_hidden % 64 == 0 && (2 * e->out_ch) % 64 == 0 && e->f_enc.ks == 3,
                 "%s: the training path needs channel counts that are multiples of 64 and 3x3 encoder dynamics", who);
  for (int l = 0; l <= e->f_enc.n_convs; ++l)
    ODEHIP_REQUIRE(e->f_enc.channels[l] % 64 == 0, "%s: encoder dynamics channels must be multiples of 64", who);
  return ODEHIP_OK;
}

// Euler step in front of the idx-th visited frame (ODEConvGRUCell.py:47,73): first t[-1] - (t[-1] + 0.01); after visiting frame j the
// loop sets (prev_t, t_i) = (t[j], t[j-1]) -- Python indexing, so j = 0 wraps to t[-1].  Frames are visited T-1 .. 0 (run_backwards,
// what forward() does) or 0 .. T-1.
static float frame_dt(const double* t_host, int n_frames, int idx, int run_backwards) {
  if (idx == 0) return (float)(t_host[n_frames - 1] - (t_host[n_frames - 1] + 0.01));
  const int j = run_backwards ? n_frames - idx : idx - 1;  // frame visited at iteration idx - 1
  return (float)(t_host[(j + n_frames - 1) % n_frames] - t_host[j]);
}
static int visited_frame(int n_frames, int idx, int run_backwards) { return run_backwards ? n_frames - 1 - idx : idx; }

// gh (Q4) += grad_latent[:, slot] (NCHW slice of a (B,T,C,16,16) tensor): the gradient that arrives through latent_ys
__global__ __launch_bounds__(256) void add_nchw_slice_to_q4_kernel(const float* __restrict__ src, long long batch_stride,
                                                                   float* __restrict__ dst, int total, int quads) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int p = idx & 255, bq = idx >> 8, b = bq / quads, q = bq - b * quads;
  const float* s = src + (size_t)b * batch_stride + (size_t)q * 4 * kPix + p;
  f32x4* d = (f32x4*)(dst + (size_t)idx * 4);
  *d += f32x4{s[0], s[kPix], s[2 * kPix], s[3 * kPix]};
}

}  // namespace odehip

using namespace odehip;

extern "C" size_t odehip_encoder_train_workspace_bytes(const odehip_encoder* e, int n_frames, int batch) {
  if (!e || n_frames <= 0 || batch <= 0 || e->f_enc.n_convs < 1) return 0;
  return EncLayout(e, n_frames, batch).total;
}

extern "C" int odehip_odeconvgru_encode_train(const odehip_encoder* e, const float* inputs_nchw, const double* t_host, int n_frames,
                                              int batch, int run_backwards, float* mean_nchw, float* std_nchw, float* latent_nchw,
                                              void* workspace, size_t workspace_bytes, void* stream_) {
  int rc = check_encoder(e, "odeconvgru_encode_train");
  if (rc != ODEHIP_OK) return rc;
  ODEHIP_REQUIRE(inputs_nchw && t_host && mean_nchw && std_nchw && workspace, "odeconvgru_encode_train: null pointer");
  ODEHIP_REQUIRE(n_frames >= 1 && n_frames <= 64 && batch > 0, "odeconvgru_encode_train: bad sizes (frames %d, batch %d)", n_frames, batch);
  const EncLayout L(e, n_frames, batch);
  ODEHIP_REQUIRE(workspace_bytes >= L.total, "odeconvgru_encode_train: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  void* ws = workspace;
  const int C = L.C, NH = L.NH;
  const size_t hs_b = (size_t)batch * C * kPix * 4;
  float dts_h[64];
  for (int idx = 0; idx < n_frames; ++idx) dts_h[idx] = frame_dt(t_host, n_frames, idx, run_backwards);
  if ((rc = upload_bytes(L.p(ws, L.off_dts), dts_h, (size_t)n_frames * 4, stream)) != ODEHIP_OK) return rc;
  rc = odehip_nchw_to_q4(inputs_nchw, L.frame(ws, 0), n_frames * batch, C, stream);
  if (rc != ODEHIP_OK) return rc;
  ODEHIP_CHECK_HIP(hipMemsetAsync(L.hstate(ws, 0), 0, hs_b, stream));
  float* hidv[ODEHIP_MAX_LAYERS];
  for (int idx = 0; idx < n_frames; ++idx) {
    const int i = visited_frame(n_frames, idx, run_backwards);
    CombineArgs c;
    memset(&c, 0, sizeof(c));
    c.k_scale = 1.0f;
    c.y = L.hstate(ws, idx);
    c.h_ptr = L.p(ws, L.off_dts) + idx;
    c.c1[0] = 1.0f;
    c.out1 = L.per(ws, L.off_hode, idx, L.hs);
    for (int l = 0; l < NH; ++l) hidv[l] = L.hidden(ws, idx, l);
    rc = enqueue_f_saving(&e->f_enc, L.hstate(ws, idx), batch, hidv, L.p(ws, L.off_ping), L.p(ws, L.off_pong), &c, nullptr, nullptr,
                          stream);
    if (rc != ODEHIP_OK) return rc;
    float* lat = latent_nchw ? latent_nchw + (size_t)idx * C * kPix : nullptr;  // latent_ys (B,T,C,H,W): slot idx of each sample
    rc = cell_step_q4(&e->cell, L.frame(ws, i), L.per(ws, L.off_hode, idx, L.hs), L.hstate(ws, idx + 1), lat, (long long)n_frames * C * kPix, batch,
                      L.per(ws, L.off_gates, idx, 2 * L.hs), L.per(ws, L.off_z, idx, L.hs), L.per(ws, L.off_rh, idx, L.hs),
                      L.per(ws, L.off_cand, idx, L.hs), stream);
    if (rc != ODEHIP_OK) return rc;
  }
  rc = conv_layer_q4(L.hstate(ws, n_frames), nullptr, C, C, e->head_hidden, 1, e->w_head0, e->b_head0, L.p(ws, L.off_headhid), 1, batch,
                     stream);
  if (rc != ODEHIP_OK) return rc;
  rc = conv_layer_q4(L.p(ws, L.off_headhid), nullptr, e->head_hidden, e->head_hidden, 2 * e->out_ch, 1, e->w_head1, e->b_head1,
                     L.p(ws, L.off_headout), 0, batch, stream);
  if (rc != ODEHIP_OK) return rc;
  launch_split_mean_std(L.p(ws, L.off_headout), mean_nchw, std_nchw, batch, e->out_ch, stream);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

extern "C" int odehip_odeconvgru_encode_backward(const odehip_encoder* e, const odehip_encoder_bwd* eb, const double* t_host,
                                                 int n_frames, int batch, int run_backwards, const float* grad_mean_nchw,
                                                 const float* grad_std_nchw, const float* grad_latent_nchw, float* grad_inputs_nchw,
                                                 const odehip_encoder_grads* gr, void* workspace, size_t workspace_bytes, void* stream_) {
  int rc = check_encoder(e, "odeconvgru_encode_backward");
  if (rc != ODEHIP_OK) return rc;
  ODEHIP_REQUIRE(eb && t_host && grad_mean_nchw && grad_std_nchw && grad_inputs_nchw && gr && workspace,
                 "odeconvgru_encode_backward: null pointer");
  ODEHIP_REQUIRE(eb->w_gates_dx && eb->w_gates_dh && eb->w_can_dx && eb->w_can_dh && eb->w_head0_t && eb->w_head1_t,
                 "odeconvgru_encode_backward: null transposed weight");
  ODEHIP_REQUIRE(n_frames >= 1 && n_frames <= 64 && batch > 0, "odeconvgru_encode_backward: bad sizes");
  const EncLayout L(e, n_frames, batch);
  ODEHIP_REQUIRE(workspace_bytes >= L.total, "odeconvgru_encode_backward: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  void* ws = workspace;
  const int C = L.C, NH = L.NH, NL = e->f_enc.n_convs, T = n_frames, HH = e->head_hidden, OUT2 = 2 * e->out_ch, ks = e->cell.ks;
  const int HG = C / 32;
  float* dts = L.p(ws, L.off_dts);
  float* gh = L.p(ws, L.off_gh);
  float* gh_next = L.p(ws, L.off_gh + L.hs);
  float* gz_pre = L.p(ws, L.off_gzpre);
  float* gh_ode = L.p(ws, L.off_ghode);
  float* gx_c = L.p(ws, L.off_gxc);
  float* g_rh = L.p(ws, L.off_grh);
  float* pgg = L.p(ws, L.off_pgg);  // [2][T*B][2C]: dgamma then dbeta partials of the gates GroupNorm
  float* pgc = L.p(ws, L.off_pgc);  // [2][T*B][C]
  const size_t pgg_half = (size_t)T * batch * 2 * C, pgc_half = (size_t)T * batch * C;

  auto conv_bwd = [&](const float* src, int cin, int cout, int k, const float* wt, int combine, const BwdArgs* bw, float* dst,
                      const void* wbf = nullptr, const float* wwino = nullptr) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.src1 = src;
    a.q1 = a.qin = cin / 4;
    a.qout = cout / 4;
    a.w_packed = wt;
    a.w_bf16 = wbf;
    a.w_wino = wbf ? nullptr : wwino;
    a.batch = batch;
    a.combine = combine;
    if (bw) a.bwd = *bw;
    a.dst = dst;
    return launch_conv(a, k, stream);
  };

  // ---- head: |std|, 1x1 -> ReLU -> 1x1
  {
    const int total = batch * (OUT2 / 4) * kPix;
    hipLaunchKernelGGL(merge_mean_std_grad_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, grad_mean_nchw, grad_std_nchw,
                       L.p(ws, L.off_headout), L.p(ws, L.off_gheadout), total, e->out_ch / 4);
    BwdArgs w;
    memset(&w, 0, sizeof(w));
    w.mask_src = L.p(ws, L.off_headhid);
    w.sc_c = 1.0f;
    if ((rc = conv_bwd(L.p(ws, L.off_gheadout), OUT2, HH, 1, eb->w_head1_t, 2, &w, L.p(ws, L.off_gheadhid))) != ODEHIP_OK) return rc;
    if ((rc = conv_bwd(L.p(ws, L.off_gheadhid), HH, C, 1, eb->w_head0_t, 0, nullptr, gh)) != ODEHIP_OK) return rc;
  }

  // ---- frames, last processed first
  for (int idx = T - 1; idx >= 0; --idx) {
    const int i = visited_frame(T, idx, run_backwards);
    if (grad_latent_nchw) {  // gh = gradient w.r.t. the state after the idx-th visited frame: + what arrives through latent_ys[:, idx]
      const int total = batch * (C / 4) * kPix;
      hipLaunchKernelGGL(add_nchw_slice_to_q4_kernel, dim3((total + 255) / 256), dim3(256), 0, stream,
                         grad_latent_nchw + (size_t)idx * C * kPix, (long long)T * C * kPix, gh, total, C / 4);
    }
    float* g_cand = L.per(ws, L.off_gcand, idx, L.hs);
    float* g_gates = L.per(ws, L.off_ggates, idx, 2 * L.hs);
    const float* h_ode = L.per(ws, L.off_hode, idx, L.hs);
    hipLaunchKernelGGL(gn_update_bwd_kernel, dim3(HG, batch), dim3(256), 0, stream, L.per(ws, L.off_cand, idx, L.hs), e->cell.gn_can_w,
                       e->cell.gn_can_b, gh, L.per(ws, L.off_z, idx, L.hs), h_ode, g_cand, gz_pre, gh_ode,
                       pgc + (size_t)idx * batch * C, pgc + pgc_half + (size_t)idx * batch * C, HG);
    if ((rc = conv_bwd(g_cand, C, C, ks, eb->w_can_dx, 0, nullptr, gx_c, eb->bf16[2], eb->wino[2])) != ODEHIP_OK) return rc;
    if ((rc = conv_bwd(g_cand, C, C, ks, eb->w_can_dh, 0, nullptr, g_rh, eb->bf16[3], eb->wino[3])) != ODEHIP_OK) return rc;
    hipLaunchKernelGGL(gn_gates_bwd_kernel, dim3(2 * HG, batch), dim3(256), 0, stream, L.per(ws, L.off_gates, idx, 2 * L.hs),
                       e->cell.gn_gates_w, e->cell.gn_gates_b, gz_pre, g_rh, h_ode, gh_ode, g_gates,
                       pgg + (size_t)idx * batch * 2 * C, pgg + pgg_half + (size_t)idx * batch * 2 * C, HG);
    BwdArgs w;
    memset(&w, 0, sizeof(w));
    w.n_targets = 1;
    w.tgt[0].out = L.per(ws, L.off_gx, i, L.hs);  // gradient of observed frame i
    w.tgt[0].srcA = gx_c;
    w.tgt[0].a_c = 1.0f;
    w.tgt[0].g_c = 1.0f;
    if ((rc = conv_bwd(g_gates, 2 * C, C, ks, eb->w_gates_dx, 3, &w, nullptr, eb->bf16[0], eb->wino[0])) != ODEHIP_OK) return rc;
    w.tgt[0].out = L.gp(ws, idx, NH);             // seed of the encoder-dynamics chain = total gradient of h_ode
    w.tgt[0].srcA = gh_ode;
    if ((rc = conv_bwd(g_gates, 2 * C, C, ks, eb->w_gates_dh, 3, &w, nullptr, eb->bf16[1], eb->wino[1])) != ODEHIP_OK) return rc;
    // h_ode = h + dt f(h):  gh = seed + dt J_f(h)^T seed
    {
      float* gpv[ODEHIP_MAX_LAYERS];
      const float* hv[ODEHIP_MAX_LAYERS];
      for (int l = 0; l < NL; ++l) gpv[l] = L.gp(ws, idx, l);
      for (int l = 0; l + 1 < NL; ++l) hv[l] = L.hidden(ws, idx, l);
      ConvArgs a;
      memset(&a, 0, sizeof(a));
      a.combine = 3;
      a.bwd.n_targets = 1;
      a.bwd.h_ptr = dts + idx;
      a.bwd.tgt[0].out = gh_next;
      a.bwd.tgt[0].srcA = L.gp(ws, idx, NH);
      a.bwd.tgt[0].a_c = 1.0f;
      a.bwd.tgt[0].g_h = 1.0f;
      if ((rc = enqueue_dgrad_chain(&e->f_enc, &eb->f_dgrad, batch, gpv, hv, a, stream)) != ODEHIP_OK) return rc;
    }
    float* t = gh; gh = gh_next; gh_next = t;
  }
  rc = odehip_q4_to_nchw(L.p(ws, L.off_gx), grad_inputs_nchw, T * batch, C, stream);
  if (rc != ODEHIP_OK) return rc;

  // ---- weight gradients: one launch per (layer, 64x64 tile) over all frames
  // All tables (NL dynamics layers, 2 x 2 ConvGRU halves, 2 head layers; T entries each) are built first and go up in ONE
  // asynchronous copy (a fill launch per 256 bytes before: 21 launches of 5 us per backward pass).
  WgradPair* const table0 = (WgradPair*)L.p(ws, L.off_tab);
  float* slabs = L.p(ws, L.off_slab);
  std::vector<WgradPair> host((size_t)(NL + 6) * T);
  memset(host.data(), 0, host.size() * sizeof(WgradPair));
  struct CatJob {
    size_t g_off; int g_ch; size_t a2_off; float* dw; float* db;
  } jobs[2] = {{L.off_ggates, 2 * C, L.off_hode, gr->w_gates, gr->b_gates}, {L.off_gcand, C, L.off_rh, gr->w_can, gr->b_can}};
  for (int l = 0; l < NL; ++l)   // encoder dynamics (3x3), weight of frame idx = its Euler dt
    for (int idx = 0; idx < T; ++idx) {
      WgradPair& h = host[(size_t)l * T + idx];
      h.g = L.gp(ws, idx, l);
      h.a = l == 0 ? L.hstate(ws, idx) : L.hidden(ws, idx, l - 1);
      h.scale = frame_dt(t_host, T, idx, run_backwards);
    }
  for (int j = 0; j < 2; ++j) {   // ConvGRU convs on cat(x, state): the two halves of the input are separate tensors
    const size_t gbytes = (size_t)(jobs[j].g_ch / C) * L.hs;
    for (int half = 0; half < 2; ++half)
      for (int idx = 0; idx < T; ++idx) {
        WgradPair& h = host[(size_t)(NL + 2 * j + half) * T + idx];
        h.g = L.per(ws, jobs[j].g_off, idx, gbytes);
        h.a = half == 0 ? L.frame(ws, visited_frame(T, idx, run_backwards)) : L.per(ws, jobs[j].a2_off, idx, L.hs);
        h.scale = 1.0f;
      }
  }
  {   // head 1x1 convs (one entry each)
    WgradPair& h1 = host[(size_t)(NL + 4) * T];
    h1.g = L.p(ws, L.off_gheadout);
    h1.a = L.p(ws, L.off_headhid);
    h1.scale = 1.0f;
    WgradPair& h0 = host[(size_t)(NL + 5) * T];
    h0.g = L.p(ws, L.off_gheadhid);
    h0.a = L.hstate(ws, T);
    h0.scale = 1.0f;
  }
  if ((rc = staged_upload(table0, host.data(), host.size() * sizeof(WgradPair), stream)) != ODEHIP_OK) return rc;
  for (int l = 0; l < NL; ++l) {
    // (fp32: 256 / batch workgroups per sample share the frames -- a batch of 4 with esplit 4 kept 240 CUs idle)
    rc = launch_wgrad(table0 + (size_t)l * T, T, batch, (&e->f_enc)->w_bf16[l] ? 4 : wgrad_esplit(batch, T), slabs, gr->f_w[l], gr->f_b[l],
                      e->f_enc.channels[l + 1], e->f_enc.channels[l], stream, (&e->f_enc)->w_bf16[l] != nullptr);
    if (rc != ODEHIP_OK) return rc;
  }
  for (int j = 0; j < 2; ++j) {
    const CatJob& J = jobs[j];
    for (int half = 0; half < 2; ++half) {
      const WgradPair* const table = table0 + (size_t)(NL + 2 * j + half) * T;
      for (int co0 = 0; co0 < J.g_ch; co0 += 64)
        for (int ci0 = 0; ci0 < C; ci0 += 64) {
          if (ks == 5 && e->cell.w_gates_bf16)  // bf16 compute mode: operands rounded to bf16, fp32 accumulation
            rc = launch_wgrad_tile_bf16_5x5(table, T, batch, 4, slabs, J.dw, J.db, 2 * C, co0, half * C + ci0, J.g_ch / 4, co0 / 4, C / 4,
                                            ci0 / 4, half == 0 && ci0 == 0, stream);
          else
            rc = launch_wgrad_tile(table, T, batch, wgrad_esplit(batch, T), slabs, J.dw, J.db, ks, 2 * C, co0, half * C + ci0, J.g_ch / 4, co0 / 4, C / 4,
                                   ci0 / 4, half == 0 && ci0 == 0, stream);
          if (rc != ODEHIP_OK) return rc;
        }
    }
  }
  {
    const WgradPair* table = table0 + (size_t)(NL + 4) * T;
    for (int co0 = 0; co0 < OUT2; co0 += 64)
      for (int ci0 = 0; ci0 < HH; ci0 += 64) {
        rc = launch_wgrad_tile(table, 1, batch, 4, slabs, gr->w_head1, gr->b_head1, 1, HH, co0, ci0, OUT2 / 4, co0 / 4, HH / 4, ci0 / 4,
                               ci0 == 0, stream);
        if (rc != ODEHIP_OK) return rc;
      }
    table = table0 + (size_t)(NL + 5) * T;
    for (int co0 = 0; co0 < HH; co0 += 64)
      for (int ci0 = 0; ci0 < C; ci0 += 64) {
        rc = launch_wgrad_tile(table, 1, batch, 4, slabs, gr->w_head0, gr->b_head0, 1, C, co0, ci0, HH / 4, co0 / 4, C / 4, ci0 / 4,
                               ci0 == 0, stream);
        if (rc != ODEHIP_OK) return rc;
      }
  }
  // GroupNorm affine parameters
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(2 * C), dim3(256), 0, stream, pgg, T * batch, 2 * C, gr->gn_gates_w);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(2 * C), dim3(256), 0, stream, pgg + pgg_half, T * batch, 2 * C, gr->gn_gates_b);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(C), dim3(256), 0, stream, pgc, T * batch, C, gr->gn_can_w);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(C), dim3(256), 0, stream, pgc + pgc_half, T * batch, C, gr->gn_can_b);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}

// ---- one ConvGRU step: backward of ConvGRUCell.forward(input_tensor, h_cur, seq_len = 1)   (modules/ConvGRUCell.py:55-86) ----
// Stateless: the step is recomputed from (x, h) inside the call (two convs), then reversed.
namespace odehip {
// Workspace of one ConvGRU-step backward; offsets are computed in ONE place for the size query and the call.
struct CellBwdLayout {
  size_t x, gx_c, gx_out, h, gates, z, rh, cand, hn, ghn, g_cand, g_gates, gz_pre, gh_ode, g_rh, gh, pg, table, slabs, total;
  CellBwdLayout(const odehip_convgru_cell* c, int batch) {
    const size_t hs = (size_t)batch * c->hidden * kPix * 4, xs = (size_t)batch * c->input * kPix * 4;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += al256(bytes); return r; };
    x = take(xs); gx_c = take(xs); gx_out = take(xs);
    h = take(hs); gates = take(2 * hs); z = take(hs); rh = take(hs); cand = take(hs); hn = take(hs); ghn = take(hs);
    g_cand = take(hs); g_gates = take(2 * hs); gz_pre = take(hs); gh_ode = take(hs); g_rh = take(hs); gh = take(hs);
    pg = take((size_t)6 * batch * c->hidden * 4);  // [dgamma_g | dbeta_g] (2H each) then [dgamma_c | dbeta_c] (H each), per sample
    table = take(sizeof(WgradPair));
    slabs = take(((size_t)batch * wgrad_esplit_max(batch) + 1) * kWgradSlabFloats * 4);
    total = o;
  }
};
}  // namespace odehip

extern "C" size_t odehip_convgru_cell_backward_workspace_bytes(const odehip_convgru_cell* c, int batch) {
  if (!c || batch <= 0 || c->hidden <= 0 || c->input <= 0) return 0;
  return CellBwdLayout(c, batch).total;
}

extern "C" int odehip_convgru_cell_backward(const odehip_convgru_cell* c, const odehip_convgru_cell_bwd* cb, const float* x_nchw,
                                            const float* h_nchw, const float* grad_h_next_nchw, float* grad_x_nchw,
                                            float* grad_h_nchw, const odehip_convgru_cell_grads* gr, int batch, void* workspace,
                                            size_t workspace_bytes, void* stream_) {
  int rc = check_cell_desc(c);
  if (rc != ODEHIP_OK) return rc;
  ODEHIP_REQUIRE(cb && x_nchw && h_nchw && grad_h_next_nchw && grad_x_nchw && grad_h_nchw && gr && workspace && batch > 0,
                 "convgru_cell_backward: bad argument");
  ODEHIP_REQUIRE(cb->w_gates_dx && cb->w_gates_dh && cb->w_can_dx && cb->w_can_dh, "convgru_cell_backward: null transposed weight");
  ODEHIP_REQUIRE(c->input % 64 == 0 && c->hidden % 64 == 0, "convgru_cell_backward: input_dim and hidden_dim must be multiples of 64");
  ODEHIP_REQUIRE(workspace_bytes >= odehip_convgru_cell_backward_workspace_bytes(c, batch), "convgru_cell_backward: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  const int H = c->hidden, I = c->input, ks = c->ks, HG = H / 32;
  const CellBwdLayout L(c, batch);
  auto at = [&](size_t off) { return (float*)((char*)workspace + off); };
  float *x = at(L.x), *gx_c = at(L.gx_c), *gx_out = at(L.gx_out), *h = at(L.h), *gates = at(L.gates), *z = at(L.z), *rh = at(L.rh),
        *cand = at(L.cand), *hn = at(L.hn), *ghn = at(L.ghn), *g_cand = at(L.g_cand), *g_gates = at(L.g_gates), *gz_pre = at(L.gz_pre),
        *gh_ode = at(L.gh_ode), *g_rh = at(L.g_rh), *gh = at(L.gh), *pg = at(L.pg), *slabs = at(L.slabs);
  WgradPair* table = (WgradPair*)at(L.table);
  float* pgg = pg;
  float* pgc = pg + (size_t)2 * batch * 2 * H;

  if ((rc = odehip_nchw_to_q4(x_nchw, x, batch, I, stream)) != ODEHIP_OK) return rc;
  if ((rc = odehip_nchw_to_q4(h_nchw, h, batch, H, stream)) != ODEHIP_OK) return rc;
  if ((rc = odehip_nchw_to_q4(grad_h_next_nchw, ghn, batch, H, stream)) != ODEHIP_OK) return rc;
  if ((rc = cell_step_q4(c, x, h, hn, nullptr, 0, batch, gates, z, rh, cand, stream)) != ODEHIP_OK) return rc;

  auto conv_bwd = [&](const float* src, int cin, int cout, const float* wt, const BwdArgs* bw, float* dst, const void* wbf,
                      const float* wwino) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.src1 = src;
    a.q1 = a.qin = cin / 4;
    a.qout = cout / 4;
    a.w_packed = wt;
    a.w_bf16 = wbf;
    a.w_wino = wbf ? nullptr : wwino;
    a.batch = batch;
    a.combine = bw ? 3 : 0;
    if (bw) a.bwd = *bw;
    a.dst = dst;
    return launch_conv(a, ks, stream);
  };
  hipLaunchKernelGGL(gn_update_bwd_kernel, dim3(HG, batch), dim3(256), 0, stream, cand, c->gn_can_w, c->gn_can_b, ghn, z, h, g_cand,
                     gz_pre, gh_ode, pgc, pgc + (size_t)batch * H, HG);
  if ((rc = conv_bwd(g_cand, H, I, cb->w_can_dx, nullptr, gx_c, cb->bf16[2], cb->wino[2])) != ODEHIP_OK) return rc;
  if ((rc = conv_bwd(g_cand, H, H, cb->w_can_dh, nullptr, g_rh, cb->bf16[3], cb->wino[3])) != ODEHIP_OK) return rc;
  hipLaunchKernelGGL(gn_gates_bwd_kernel, dim3(2 * HG, batch), dim3(256), 0, stream, gates, c->gn_gates_w, c->gn_gates_b, gz_pre, g_rh, h,
                     gh_ode, g_gates, pgg, pgg + (size_t)batch * 2 * H, HG);
  BwdArgs w;
  memset(&w, 0, sizeof(w));
  w.n_targets = 1;
  w.tgt[0].out = gx_out;  // gradient of x = can-conv part + gates-conv part
  w.tgt[0].srcA = gx_c;
  w.tgt[0].a_c = 1.0f;
  w.tgt[0].g_c = 1.0f;
  if ((rc = conv_bwd(g_gates, 2 * H, I, cb->w_gates_dx, &w, nullptr, cb->bf16[0], cb->wino[0])) != ODEHIP_OK) return rc;
  w.tgt[0].out = gh;
  w.tgt[0].srcA = gh_ode;
  if ((rc = conv_bwd(g_gates, 2 * H, H, cb->w_gates_dh, &w, nullptr, cb->bf16[1], cb->wino[1])) != ODEHIP_OK) return rc;
  if ((rc = odehip_q4_to_nchw(gx_out, grad_x_nchw, batch, I, stream)) != ODEHIP_OK) return rc;
  if ((rc = odehip_q4_to_nchw(gh, grad_h_nchw, batch, H, stream)) != ODEHIP_OK) return rc;

  WgradPair pr;
  memset(&pr, 0, sizeof(pr));
  pr.scale = 1.0f;
  struct Job { const float* g; int g_ch; const float* a2; float* dw; float* db; } jobs[2] = {{g_gates, 2 * H, h, gr->w_gates, gr->b_gates},
                                                                                          {g_cand, H, rh, gr->w_can, gr->b_can}};
  for (int j = 0; j < 2; ++j)
    for (int half = 0; half < 2; ++half) {
      pr.g = jobs[j].g;
      pr.a = half == 0 ? x : jobs[j].a2;
      if ((rc = upload_bytes(table, &pr, sizeof(pr), stream)) != ODEHIP_OK) return rc;
      const int a_ch = half == 0 ? I : H;
      for (int co0 = 0; co0 < jobs[j].g_ch; co0 += 64)
        for (int ci0 = 0; ci0 < a_ch; ci0 += 64) {
          if (ks == 5 && c->w_gates_bf16)
            rc = launch_wgrad_tile_bf16_5x5(table, 1, batch, 4, slabs, jobs[j].dw, jobs[j].db, I + H, co0, half * I + ci0,
                                            jobs[j].g_ch / 4, co0 / 4, a_ch / 4, ci0 / 4, half == 0 && ci0 == 0, stream);
          else
            rc = launch_wgrad_tile(table, 1, batch, 4, slabs, jobs[j].dw, jobs[j].db, ks, I + H, co0, half * I + ci0, jobs[j].g_ch / 4,
                                   co0 / 4, a_ch / 4, ci0 / 4, half == 0 && ci0 == 0, stream);
          if (rc != ODEHIP_OK) return rc;
        }
    }
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(2 * H), dim3(256), 0, stream, pgg, batch, 2 * H, gr->gn_gates_w);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(2 * H), dim3(256), 0, stream, pgg + (size_t)batch * 2 * H, batch, 2 * H,
                     gr->gn_gates_b);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(H), dim3(256), 0, stream, pgc, batch, H, gr->gn_can_w);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(H), dim3(256), 0, stream, pgc + (size_t)batch * H, batch, H, gr->gn_can_b);
  ODEHIP_CHECK_HIP(hipGetLastError());
  return ODEHIP_OK;
}
