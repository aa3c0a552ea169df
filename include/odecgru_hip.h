/* odecgru_hip.h -- C ABI of the MI355X (gfx950) Neural-ODE hot path.
 *
 * The reference (jithendaraa/ODE-RL) has no FFI: its hot path is Python calling
 * `torchdiffeq.odeint(func, y0, t, rtol=, atol=, method=)` (modules/DiffEqSolver.py:37,45-46)
 * on a conv dynamics `f` built by `utils.create_convnet` (helpers/utils.py:158-183), plus the
 * Euler+ConvGRU encoder loop (modules/ODEConvGRUCell.py:39-78, modules/ConvGRUCell.py:55-86).
 * These entry points are what a native binding for that path would bind:
 * plain pointers and sizes, device pointers are HIP device memory, `stream` is a hipStream_t.
 *
 * Conventions
 *   - every function returns 0 on success, a negative ODEHIP_E* code on failure;
 *     odehip_last_error() returns a static string for the calling thread.
 *   - scratch comes from caller-provided workspaces sized by the *_bytes() queries, and the compute entry points
 *     are enqueue-only on the caller's stream, WITH THESE EXCEPTIONS (each is stated again at its entry point):
 *       * the persistent-launch layer table: the library keeps up to 8 layer tables in device memory it owns; a call
 *         whose table is not cached yet synchronises the stream once, may hipMalloc, and uploads with a blocking copy
 *         (a steady loop re-uses its table: nothing is allocated, copied or synchronised);
 *       * odehip_odeint_dopri5 returns after the device-side controller has reported completion: it polls a pinned
 *         64 KiB host mailbox that the library allocates on first use (hipHostMalloc);
 *       * odehip_odeint_adjoint_dopri5_backward reads one 8-byte verdict per attempted step from pinned host memory it
 *         allocates on first use;
 *       * tables that follow the accepted steps of an adaptive solve (odehip_odeint_dopri5_backward[_saved]: its layer table and its
 *         weight-gradient tables) travel through a library-owned ring of pinned staging buffers and device slots with asynchronous
 *         copies on the caller's stream -- no stream synchronisation (hipHostMalloc / hipMalloc when a slot has to grow);
 *       * the "small" persistent launches (ODEHIP_PERSISTENT_SMALL=1 only) own a flag area in device memory;
 *       * odehip_odeint_fixed_backward with saved_format 1 (bf16 whole-trajectory path) runs its weight-gradient launches on a
 *         library-owned side stream, ordered against the caller's stream by events in both directions (the caller's stream
 *         continues only when the side stream is done with the workspace and the gradients); ODEHIP_BF16_OVERLAP=0 keeps
 *         everything on the caller's stream.
 *   - library state (table cache, mailbox, error word, flag areas) is process-global and serves ONE stream at a time:
 *     the library assumes one process per GPU driving it from a single stream (the Python binding passes torch's
 *     current stream); concurrent calls from several streams or threads are not supported.
 *   - a persistent launch whose in-kernel wait gives up (never observed; every wait is capped) sets a sticky error
 *     word: the outputs of the call that was running are filled with NaN, odehip_persistent_error() reports the code,
 *     and the next call that could use the persistent path fails with ODEHIP_EINVAL.
 *   - boundary tensors are fp32 NCHW, contiguous, H = W = 16 (the reference's latent map,
 *     models/ODEConvGRU.py:18-20 with resolution 64, n_downs 2).  Internally activations use
 *     the "Q4" layout [B][C/4][256 pixels][4 channels].
 */
#ifndef ODECGRU_HIP_H
#define ODECGRU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ODEHIP_ABI_VERSION 12 /* == odehip_version(); bumped whenever a struct layout or a signature below changes */
#define ODEHIP_MAX_LAYERS 8
#define ODEHIP_MAX_STAGES 7

enum {
  ODEHIP_OK = 0,
  ODEHIP_EINVAL = -1,   /* bad shape / argument (the Python side raises ValueError) */
  ODEHIP_EHIP = -2,     /* a HIP runtime call failed */
  ODEHIP_ENOTCONV = -3, /* adaptive solver hit max_num_steps / dt underflow (AssertionError in torchdiffeq) */
  ODEHIP_ENAN = -4,     /* non-finite state (the reference asserts: modules/ODEConvGRUCell.py:56,59) */
  ODEHIP_ETRUNC = -5    /* asynchronous dopri5: the attempts enqueued by _start did not finish the solve (see there) */
};

/* fixed-grid methods of torchdiffeq (_impl/fixed_grid.py); rk4 is the 3/8 rule */
enum { ODEHIP_EULER = 0, ODEHIP_MIDPOINT = 1, ODEHIP_RK4 = 2, ODEHIP_DOPRI5 = 3 };

const char* odehip_last_error(void);
int odehip_version(void);
/* diagnostic ablation bits for tools/conv_microbench.py (1 skip LDS-DMA, 2 skip MFMA, 4 skip epilogue); 0 in production */
void odehip_set_debug_flags(int flags);
/* device buffer of 8 x uint64 per workgroup for in-kernel stamps (flag 8); NULL disables */
void odehip_set_debug_buffer(void* device_buffer);

/* ---- layout + weight packing -------------------------------------------------------------- */

/* number of floats of the MFMA-ordered image of a (cout, cin, ks, ks) conv weight */
size_t odehip_packed_weight_floats(int cout, int cin, int ks);

/* OIHW fp32 -> packed image.  transpose_flip != 0 packs the weights of the input-gradient
 * convolution (dgrad): W'[ci][co][ks-1-dy][ks-1-dx], i.e. a conv with cin/cout swapped.
 * Replaces: the implicit cuDNN/MIOpen weight handling behind nn.Conv2d (helpers/utils.py:167-177). */
int odehip_pack_conv_weight(const float* w_oihw, float* w_packed, int cout, int cin, int ks,
                            int transpose_flip, void* stream);

/* Winograd F(2x2,3x3) form of a 3x3 conv weight (U = G g G^T, 16 values per (cout, cin) pair) in the LDS image of the
 * Winograd kernel.  Optional: a conv with w_wino == NULL runs the direct kernel. */
size_t odehip_winograd_weight_floats(int cout, int cin);
int odehip_pack_conv_weight_winograd(const float* w_oihw, float* w_wino, int cout, int cin, int transpose_flip, void* stream);
/* Several of the two packs above in ONE launch (a training step repacks every weight tensor after the optimizer's update).
 * kind 0 = odehip_pack_conv_weight (cout, cin, ks, transpose_flip), kind 1 = odehip_pack_conv_weight_winograd (ks must be 3),
 * kind 2 = odehip_pack_conv_weight_winograd5 (ks must be 5); the results are identical to those calls'. */
#define ODEHIP_MAX_PACK_JOBS 32
typedef struct odehip_pack_job {
  const float* w; /* (cout, cin, ks, ks) as nn.Conv2d holds it */
  float* out;
  int cout, cin, ks, kind, transpose_flip;
} odehip_pack_job;
int odehip_pack_conv_weights(const odehip_pack_job* jobs, int n_jobs, void* stream);
/* bf16 A-operand image of a 3x3 weight (round to nearest even): cout % 32 == 0, cin % 16 == 0, cin <= 128 is what the
 * bf16 kernel serves; odehip_bf16_weight_bytes(cout, cin) bytes.  transpose_flip as odehip_pack_conv_weight. */
/* Winograd F(2x2,5x5) form of a 5x5 weight (conv_wino5.hip): U = G g G^T for the points 0, +-1, +-2, inf; cout % 32 == 0,
 * cin % 8 == 0.  Set as odehip_conv_desc.w_wino of a ks = 5 layer, or as odehip_convgru_cell.w_gates_wino / w_can_wino. */
size_t odehip_winograd5_weight_floats(int cout, int cin);
int odehip_pack_conv_weight_winograd5(const float* w_oihw, float* w_wino, int cout, int cin, int transpose_flip, void* stream);

size_t odehip_bf16_weight_bytes(int cout, int cin);
int odehip_pack_conv_weight_bf16(const float* w_oihw, void* w_bf16, int cout, int cin, int transpose_flip, void* stream);
/* fused image of a 64 -> 64 3x3 stack: call once per layer with its position `exec_index` in execution order (forward: the
 * layer index; input-gradient chain: n_convs-1-layer with transpose_flip = 1); odehip_fused_bf16_weight_bytes(n_layers) bytes */
size_t odehip_fused_bf16_weight_bytes(int n_layers);
int odehip_pack_convstack_fused_bf16(const float* w_oihw, void* w_fused, int exec_index, int transpose_flip, void* stream);
/* the same for the 5x5 convs of the ConvGRU cell (block-major image, cout*cin*25*2 bytes) */
int odehip_pack_conv_weight_bf16_ks(const float* w_oihw, void* w_bf16, int cout, int cin, int ks, int transpose_flip, void* stream);

int odehip_nchw_to_q4(const float* src_nchw, float* dst_q4, int batch, int channels, void* stream);
int odehip_q4_to_nchw(const float* src_q4, float* dst_nchw, int batch, int channels, void* stream);

/* ---- one convolution layer on 16x16 maps (building block; exposed for tests) ---------------- */

typedef struct odehip_conv_desc {
  const float* src1;      /* Q4 input, channels [0, cin1)                                     */
  const float* src2;      /* Q4 input, channels [cin1, cin); NULL when cin1 == cin (torch.cat) */
  int cin1, cin, cout, ks, batch;
  const float* w_packed;  /* from odehip_pack_conv_weight                                      */
  const float* w_wino;    /* from odehip_pack_conv_weight_winograd (ks 3) / _winograd5 (ks 5), or NULL (direct kernel) */
  const void* w_bf16;     /* from odehip_pack_conv_weight_bf16: non-NULL = bf16 operands, fp32 accumulate (3x3) */
  const float* bias;      /* cout floats or NULL                                               */
  float* dst;             /* Q4 output                                                        */
  int relu;               /* fuse ReLU into the epilogue                                      */
} odehip_conv_desc;

int odehip_conv_q4(const odehip_conv_desc* d, void* stream);
/* diagnostic: n back-to-back evaluations of f on Q4 tensors; scratch holds 2 hidden activations (tools/fused_microbench.py) */
struct odehip_convstack;
int odehip_debug_repeat_f(const struct odehip_convstack* f, const float* x_q4, float* out_q4, float* scratch, int batch, int n,
                          void* stream);
/* diagnostic: n back-to-back launches of the same layer (tools/conv_microbench.py) */
int odehip_debug_repeat_conv(const odehip_conv_desc* d, int n, void* stream);

/* ---- the dynamics f(t, y) = gradient_net(y)  (modules/DiffEqSolver.py:71-80) ---------------- */

typedef struct odehip_convstack {
  int n_convs;                           /* create_convnet: n_layers + 2 (helpers/utils.py:166-177) */
  int ks;                                /* 3                                                       */
  int channels[ODEHIP_MAX_LAYERS + 1];   /* channels[i] -> channels[i+1]                            */
  const float* w_packed[ODEHIP_MAX_LAYERS];
  const float* w_wino[ODEHIP_MAX_LAYERS];  /* optional Winograd forms (3x3 layers), NULL entries run the direct kernel */
  const void* w_bf16[ODEHIP_MAX_LAYERS];   /* optional bf16 forms: a non-NULL entry runs that layer with bf16 operands and fp32
                                              accumulation (BASELINE.json configs[4]); state and stage combines stay fp32 */
  const void* w_fused;                     /* optional (odehip_pack_convstack_fused_bf16): when every layer is 3x3, 64 -> 64, the
                                              whole stack runs as ONE bf16 launch, one workgroup per sample (fstack_bf16.hip)   */
  const float* bias[ODEHIP_MAX_LAYERS];
  int final_tanh;                        /* final_act=True appends Tanh (helpers/utils.py:179-181)  */
} odehip_convstack;

size_t odehip_convstack_workspace_bytes(const odehip_convstack* f, int batch);

/* y, out: NCHW fp32 (batch, channels[0]|channels[n], 16, 16).  negate != 0 returns -f (backwards=True). */
int odehip_convstack_forward(const odehip_convstack* f, const float* y_nchw, float* out_nchw, int batch,
                             int negate, void* workspace, size_t workspace_bytes, void* stream);

/* ---- odeint, fixed grid: euler / midpoint / rk4(3/8)  (torchdiffeq FixedGridODESolver) ------ */

size_t odehip_odeint_workspace_bytes(const odehip_convstack* f, int batch, int n_times, int method,
                                     int save_for_backward);

/* z0: (B,C,16,16) NCHW; t_host: n_times float64 on the HOST, strictly increasing;
 * out: (n_times,B,C,16,16) NCHW, out[0] = z0.  One step per output interval.
 * negate != 0 integrates dz/dt = -f(z): torchdiffeq's handling of a strictly DEcreasing t is to flip the sign of t
 * and of the dynamics (_impl/odeint.py _check_inputs); the host passes -t here.
 * saved_format_out (required with save_for_backward, else may be NULL) receives how the workspace holds what the backward
 * pass needs: 0 = one fp32 Q4 tensor per evaluation and layer (per-layer / per-evaluation launches), 1 = bf16 "Q4h" tensors
 * written by the whole-trajectory bf16 launch (bf16 fused stack, rk4).  Hand it to odehip_odeint_fixed_backward. */
int odehip_odeint_fixed(const odehip_convstack* f, int method, const float* z0_nchw, const double* t_host,
                        int n_times, int batch, float* out_nchw, int save_for_backward, int negate, void* workspace,
                        size_t workspace_bytes, int* saved_format_out, void* stream);

/* Backward of odehip_odeint_fixed(save_for_backward = 1), on the SAME workspace (untouched in between).
 * This is what `loss.backward()` (train_test.py:204) computes through torchdiffeq's fixed-grid ops: the exact gradient
 * of the discrete solver.  f_dgrad->w_packed[l] = odehip_pack_conv_weight(W_l, transpose_flip = 1) in forward layer
 * order (its bias pointers are ignored).  grad_out (T,B,C,16,16) -> grad_z0 (B,C,16,16), grad_w[l] (OIHW), grad_b[l].
 * Deterministic (no float atomics).  3x3 dynamics with channel counts that are multiples of 64.
 * saved_format: what the forward call reported (0 or 1). */
int odehip_odeint_fixed_backward(const odehip_convstack* f, const odehip_convstack* f_dgrad, int method, const double* t_host,
                                 int n_times, int batch, const float* grad_out_nchw, float* grad_z0_nchw, float* const* grad_w,
                                 float* const* grad_b, int saved_format, void* workspace, size_t workspace_bytes, void* stream);

/* Adjoint backward: torchdiffeq `odeint_adjoint` semantics (new capability; the reference uses plain autograd).  For
 * i = T-1..1 the augmented state (y, a_y, a_theta) is integrated from t[i] back to t[i-1] with one step of `method`,
 * y is reset to the stored y[i-1], a_y += grad_out[i-1].  y_traj = the forward output (T,B,C,16,16).  Needs no saved
 * activations; workspace size = odehip_odeint_workspace_bytes(..., save_for_backward = 1).  3x3 dynamics, channels % 64 == 0. */
int odehip_odeint_adjoint_backward(const odehip_convstack* f, const odehip_convstack* f_dgrad, int method, const double* t_host,
                                   int n_times, int batch, const float* y_traj_nchw, const float* grad_out_nchw,
                                   float* grad_z0_nchw, float* const* grad_w, float* const* grad_b, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* Adaptive adjoint backward: torchdiffeq `odeint_adjoint(..., method="dopri5", adjoint_options={"norm": "seminorm"})`
 * (BASELINE.json configs[2]; new capability, the reference never calls the adjoint -- modules/DiffEqSolver.py:9).  Each
 * interval t[i] -> t[i-1] is a fresh dopri5 solve of the augmented system with the error norm max(rms_y, rms_a_y); the
 * parameter adjoint does not steer the step size (seminorm) and is accumulated by one weight-gradient launch per layer
 * at the end.  Synchronises `stream` once per attempted step (host-side step control).  `max_accept` bounds the number
 * of accepted steps over the whole backward pass (each keeps its activations); exceeding it returns ODEHIP_EINVAL.
 * stats_host (may be null): {f evaluations of the augmented system, accepted steps, rejected steps}.
 * mixed_norm = 1 is torchdiffeq's DEFAULT adjoint norm: every parameter tensor's RMS error ratio also steers the steps; the
 * parameter block is then integrated step by step (two batched weight-gradient launches per layer and attempt: its error
 * estimate and its increment are both linear in the seven stages). */
size_t odehip_adjoint_dopri5_workspace_bytes(const odehip_convstack* f, int batch, int n_times, int max_accept);
int odehip_odeint_adjoint_dopri5_backward(const odehip_convstack* f, const odehip_convstack* f_dgrad, const double* t_host,
                                          int n_times, int batch, float rtol, float atol, const float* y_traj_nchw,
                                          const float* grad_out_nchw, float* grad_z0_nchw, float* const* grad_w,
                                          float* const* grad_b, int max_accept, int mixed_norm, int* stats_host, void* workspace,
                                          size_t workspace_bytes, void* stream);

/* ---- ConvGRU cell and the ODE-ConvGRU encoder (modules/ConvGRUCell.py:55-86, modules/ODEConvGRUCell.py:32-78) ---- */

typedef struct odehip_convgru_cell {
  int input, hidden, ks;       /* ConvGRUCell(input_dim, hidden_dim, kernel_size); GroupNorm groups of 32 channels        */
  const float* w_gates;        /* packed conv_gates.0.weight (2*hidden, input+hidden, ks, ks)                              */
  const float* b_gates;        /* conv_gates.0.bias                                                                        */
  const float* gn_gates_w;     /* conv_gates.1.weight (GroupNorm gamma, 2*hidden)                                           */
  const float* gn_gates_b;     /* conv_gates.1.bias                                                                        */
  const float* w_can;          /* packed conv_can.0.weight (hidden, input+hidden, ks, ks)                                   */
  const float* b_can;
  const float* gn_can_w;
  const float* gn_can_b;
  const void* w_gates_bf16;    /* optional: odehip_pack_conv_weight_bf16_ks images; non-NULL = bf16 operands, fp32 accumulation */
  const void* w_can_bf16;      /*           (5x5 cells with input + hidden <= 128 channels)                                   */
  const float* w_gates_wino;   /* optional: odehip_pack_conv_weight_winograd5 forms of the two 5x5 weights; non-NULL = Winograd       */
  const float* w_can_wino;     /*           F(2x2,5x5) in fp32 (36 instead of 100 multiplies per 2x2 outputs; ignored in bf16 mode)  */
} odehip_convgru_cell;

size_t odehip_convgru_cell_workspace_bytes(const odehip_convgru_cell* c, int batch);
/* one step: x (B,input,16,16), h (B,hidden,16,16) -> h_next, all NCHW          (ConvGRUCell.forward, seq_len = 1) */
int odehip_convgru_cell_forward(const odehip_convgru_cell* c, const float* x_nchw, const float* h_nchw, float* h_next_nchw,
                                int batch, void* workspace, size_t workspace_bytes, void* stream);

typedef struct odehip_encoder {
  odehip_convstack f_enc;      /* ode_encoder_func.gradient_net                                                            */
  odehip_convgru_cell cell;    /* cgru_cell                                                                                */
  int head_hidden, out_ch;     /* transform_z0: Conv1x1(ch, head_hidden) -> ReLU -> Conv1x1(head_hidden, 2*out_ch)          */
  const float* w_head0;        /* packed 1x1 weights / biases                                                              */
  const float* b_head0;
  const float* w_head1;
  const float* b_head1;
} odehip_encoder;

size_t odehip_encoder_workspace_bytes(const odehip_encoder* e, int n_frames, int batch);
/* inputs (T,B,C,16,16) time-first NCHW, t_host[T] float64 -> mean_z0, std_z0 (B,out_ch,16,16); latent (B,T,C,16,16) or NULL
 * (slot k of a sample = the state after the k-th VISITED frame, as the reference stacks them, ODEConvGRUCell.py:74,76).
 * ODEConvGRUCell.forward / run_ode_conv_gru: run_backwards != 0 visits the frames T-1 .. 0 (what forward() does, :33),
 * 0 visits them 0 .. T-1 with the step sizes the reference's loop then produces (:47,73); `mask` is ignored as in the
 * reference (ConvGRUCell.py:55-86). */
int odehip_odeconvgru_encode(const odehip_encoder* e, const float* inputs_nchw, const double* t_host, int n_frames, int batch,
                             int run_backwards, float* mean_nchw, float* std_nchw, float* latent_nchw, void* workspace,
                             size_t workspace_bytes, void* stream);

/* Backward of one ConvGRU step (ConvGRUCell.forward with seq_len = 1): grad_h_next -> grad_x, grad_h and the gradients of
 * the cell's eight parameters.  Stateless: the step is recomputed from (x, h) inside the call.  input_dim and hidden_dim
 * must be multiples of 64. */
typedef struct odehip_convgru_cell_bwd {
  const float* w_gates_dx;     /* odehip_pack_conv_weight(conv_gates.0.weight[:, :input], transpose_flip = 1)               */
  const float* w_gates_dh;     /* ... conv_gates.0.weight[:, input:]                                                       */
  const float* w_can_dx;       /* ... conv_can.0.weight[:, :input]                                                         */
  const float* w_can_dh;       /* ... conv_can.0.weight[:, input:]                                                         */
  const void* bf16[4];         /* optional bf16 images of the same four (odehip_pack_conv_weight_bf16_ks, transpose_flip = 1)  */
  const float* wino[4];        /* optional F(2x2,5x5) forms of the same four (odehip_pack_conv_weight_winograd5, transpose_flip = 1) */
} odehip_convgru_cell_bwd;
typedef struct odehip_convgru_cell_grads {
  float *w_gates, *b_gates, *gn_gates_w, *gn_gates_b, *w_can, *b_can, *gn_can_w, *gn_can_b;
} odehip_convgru_cell_grads;
size_t odehip_convgru_cell_backward_workspace_bytes(const odehip_convgru_cell* c, int batch);
int odehip_convgru_cell_backward(const odehip_convgru_cell* c, const odehip_convgru_cell_bwd* cb, const float* x_nchw,
                                 const float* h_nchw, const float* grad_h_next_nchw, float* grad_x_nchw, float* grad_h_nchw,
                                 const odehip_convgru_cell_grads* grads, int batch, void* workspace, size_t workspace_bytes,
                                 void* stream);

/* ---- training path of the encoder: `loss.backward()` through ODEConvGRUCell.forward (train_test.py:204) ---------------- */

typedef struct odehip_encoder_bwd {
  odehip_convstack f_dgrad;    /* encoder dynamics with weights packed transpose_flip = 1 (bias pointers ignored)          */
  const float* w_gates_dx;     /* odehip_pack_conv_weight(conv_gates.0.weight[:, :input],  transpose_flip = 1)              */
  const float* w_gates_dh;     /* ... conv_gates.0.weight[:, input:]                                                       */
  const float* w_can_dx;       /* ... conv_can.0.weight[:, :input]                                                         */
  const float* w_can_dh;       /* ... conv_can.0.weight[:, input:]                                                         */
  const float* w_head0_t;      /* ... transform_z0.0.weight, transform_z0.2.weight                                         */
  const float* w_head1_t;
  const void* bf16[4];         /* optional bf16 images of w_gates_dx, w_gates_dh, w_can_dx, w_can_dh                          */
  const float* wino[4];        /* optional F(2x2,5x5) forms of the same four (ks = 5, fp32 mode)                                */
} odehip_encoder_bwd;

typedef struct odehip_encoder_grads {  /* outputs, each shaped like its parameter (conv weights OIHW) */
  float* f_w[ODEHIP_MAX_LAYERS];
  float* f_b[ODEHIP_MAX_LAYERS];
  float *w_gates, *b_gates, *gn_gates_w, *gn_gates_b, *w_can, *b_can, *gn_can_w, *gn_can_b;
  float *w_head0, *b_head0, *w_head1, *b_head1;
} odehip_encoder_grads;

/* Forward that keeps the conv outputs of every frame in `workspace` (which the caller must leave untouched until the
 * backward call), then the reverse sweep: grad_mean / grad_std (B,out_ch,16,16) -> grad_inputs (T,B,C,16,16) and the
 * gradient of every parameter of the encoder dynamics, the ConvGRU cell (incl. GroupNorm affine) and the 1x1 head.
 * run_backwards as in odehip_odeconvgru_encode (the same value in both calls); latent_nchw (B,T,C,16,16) or NULL receives
 * run_ode_conv_gru's latent_ys, grad_latent_nchw (same shape) or NULL is the gradient that arrives through it.
 * Channel counts must be multiples of 64, encoder dynamics 3x3.  Deterministic (no float atomics). */
size_t odehip_encoder_train_workspace_bytes(const odehip_encoder* e, int n_frames, int batch);
int odehip_odeconvgru_encode_train(const odehip_encoder* e, const float* inputs_nchw, const double* t_host, int n_frames, int batch,
                                   int run_backwards, float* mean_nchw, float* std_nchw, float* latent_nchw, void* workspace,
                                   size_t workspace_bytes, void* stream);
int odehip_odeconvgru_encode_backward(const odehip_encoder* e, const odehip_encoder_bwd* eb, const double* t_host, int n_frames,
                                      int batch, int run_backwards, const float* grad_mean_nchw, const float* grad_std_nchw,
                                      const float* grad_latent_nchw, float* grad_inputs_nchw, const odehip_encoder_grads* grads,
                                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- odeint, adaptive dopri5 (torchdiffeq Dopri5Solver; the reference's default method, configs.yaml:79) ------ */

/* Exact-global step control when the batch is sharded over ranks: torchdiffeq's error norm is one RMS
 * over the WHOLE batch.  With a callback installed, odehip_odeint_dopri5 hands every sum of squares (n <= 4 floats in
 * `scratch_dev`, device memory the caller owns) to `cb`, which must enqueue an in-place SUM all-reduce of scratch_dev[0..n)
 * over the ranks on `stream` (RCCL) and return 0; element counts are multiplied by world_size.  All ranks then take the same
 * accept/reject decisions as one device holding the whole batch (up to summation order).  One collective per attempted step
 * (plus two for the initial step): latency-bound, and the host no longer runs one attempt ahead.  cb = NULL restores
 * per-shard control. */
typedef int (*odehip_allreduce_fn)(float* scratch_dev, int n, void* stream, void* user);
int odehip_set_norm_allreduce(odehip_allreduce_fn cb, void* user, int world_size, float* scratch_dev);

size_t odehip_dopri5_workspace_bytes(const odehip_convstack* f, int batch, int n_times);

/* Same tensors as odehip_odeint_fixed.  rtol/atol as passed by DiffEqSolver (modules/DiffEqSolver.py:13: 1e-4, 1e-5).
 * Step control runs on the device (global RMS norm over the whole batch, as torchdiffeq); outputs are the quartic
 * dense output at t[i].  The call returns after the controller has reported completion (it polls a pinned host
 * mailbox; it does not synchronise the stream otherwise).  stats_host[4] (may be NULL) = {nfe, n_accept, n_reject,
 * attempts enqueued}.  first_step > 0 is torchdiffeq's options={'first_step': dt} (skips the initial-step heuristic);
 * max_steps <= 0 means unlimited (torchdiffeq max_num_steps = 2^31-1); like torchdiffeq's it bounds the attempted steps (accepted
 * or rejected) spent on ONE output time -- the counter is a local of `_advance(next_t)` and restarts with every output.
 * accepted_host (may be NULL): receives (t0, dt) of the accepted steps, in order, at most accepted_cap pairs -- the
 * input of odehip_odeint_dopri5_backward (stats_host[1] > accepted_cap means the log was cut).
 * Errors: ODEHIP_ENOTCONV (dt underflow / max_steps), ODEHIP_ENAN (non-finite error ratio). */
int odehip_odeint_dopri5(const odehip_convstack* f, const float* z0_nchw, const double* t_host, int n_times, int batch,
                         float rtol, float atol, double first_step, int max_steps, int negate, float* out_nchw,
                         int* stats_host, double* accepted_host, int accepted_cap, void* workspace, size_t workspace_bytes,
                         void* stream);

/* `loss.backward()` through odeint(method="dopri5") -- the reference's default training path (configs.yaml:79,
 * modules/DiffEqSolver.py:9, train_test.py:204): the gradient of the arithmetic of the accepted steps (what autograd
 * through torchdiffeq differentiates; step sizes are constants, as under torchdiffeq's no_grad step-size update).
 * Re-integrates the `n_steps` accepted steps of the forward call from z0 keeping activations (n_steps * ~3.5 state-sized
 * tensors per conv layer), walks them backwards, one weight-gradient launch per layer at the end.  Enqueue-only.  For 64-channel
 * fp32 stacks the re-integration and the whole reverse sweep are ONE launch of the adaptive persistent walk (DESIGN.md section
 * 4.4a).  3x3 dynamics, channels % 64 == 0, increasing t. */
size_t odehip_dopri5_backward_workspace_bytes(const odehip_convstack* f, int batch, int n_times, int n_steps);
int odehip_odeint_dopri5_backward(const odehip_convstack* f, const odehip_convstack* f_dgrad, const double* t_host, int n_times,
                                  int batch, const double* accepted_host, int n_steps, const float* z0_nchw,
                                  const float* grad_out_nchw, float* grad_z0_nchw, float* const* grad_w, float* const* grad_b,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* The same pair WITHOUT the re-integration: the forward of a training step keeps the stage inputs and hidden activations of every
 * accepted step (an attempt writes them into a slot chosen on the device; a rejected attempt's slot is reused), the backward walks
 * them in reverse.  odehip_odeint_dopri5_saving = odehip_odeint_dopri5 (same arguments; no `negate`) + max_accept slots of
 * workspace (odehip_dopri5_saving_workspace_bytes; ~140 MB per slot at B=64) + *saved_out: 1 if the activations are in the
 * workspace, 0 if nothing was saved (the stack is not a 64-channel fp32 stack, the persistent walk is unavailable, exact-global
 * step control is on, or more than max_accept steps were accepted) -- the caller then uses odehip_odeint_dopri5_backward.
 * odehip_odeint_dopri5_backward_saved takes that workspace (untouched since the forward) with the forward's accepted-step log. */
size_t odehip_dopri5_saving_workspace_bytes(const odehip_convstack* f, int batch, int n_times, int max_accept);
int odehip_odeint_dopri5_saving(const odehip_convstack* f, const float* z0_nchw, const double* t_host, int n_times, int batch,
                                float rtol, float atol, double first_step, int max_steps, float* out_nchw, int* stats_host,
                                double* accepted_host, int accepted_cap, int max_accept, int* saved_out, void* workspace,
                                size_t workspace_bytes, void* stream);
int odehip_odeint_dopri5_backward_saved(const odehip_convstack* f, const odehip_convstack* f_dgrad, const double* t_host, int n_times,
                                        int batch, const double* accepted_host, int n_steps, const float* grad_out_nchw,
                                        float* grad_z0_nchw, float* const* grad_w, float* const* grad_b, int max_accept,
                                        void* saved_workspace, size_t saved_workspace_bytes, void* stream);

/* The ASYNCHRONOUS pair (ABI 8).  odehip_odeint_dopri5 / _saving return only when the device-side controller has reported
 * completion -- the host cannot enqueue the work BEHIND the solver (decoder, loss, the backward pass) meanwhile, and in a whole
 * training step the device then idles while ~600 launches are enqueued.  odehip_odeint_dopri5_start takes _saving's arguments
 * (max_accept = 0: nothing is kept for a backward pass), enqueues `attempts` attempted steps (those queued behind completion
 * return at once) and comes back WITHOUT waiting: no stats, no status.  The caller enqueues its consumers of `out` behind this
 * call, so the attempts enqueued HERE are all the solve ever gets: behind the last one sits a seal kernel which, if the solve is
 * not done, fills the frames it has not reached with NaN (ABI 10; before, collect enqueued the missing attempts behind the
 * consumers, which had then read uninitialised frames).  odehip_odeint_dopri5_collect(token) waits for the device (normally long
 * done) and reports what the synchronous call reports -- including ODEHIP_ENOTCONV / ODEHIP_ENAN, i.e. an error surfaces at
 * collect time -- or ODEHIP_ETRUNC for a sealed, unfinished solve (stats_host[3] = the attempts that were enqueued; stats are
 * filled in that case too, so the caller can size its retry).  Everything start was given (workspace, out, z0) must stay
 * untouched until collect; at most 4 solves may be pending; not available under exact-global step control. */
int odehip_odeint_dopri5_start(const odehip_convstack* f, const float* z0_nchw, const double* t_host, int n_times, int batch,
                               float rtol, float atol, double first_step, int max_steps, float* out_nchw, int max_accept,
                               int attempts, int* token_out, void* workspace, size_t workspace_bytes, void* stream);
int odehip_odeint_dopri5_collect(int token, int* stats_host, double* accepted_host, int accepted_cap, int* saved_out);

/* ---- optimizer step of the training loop (train_test.py:24,205: optim.Adam(model.parameters(), lr)) ------------------------ */

/* One launch for every parameter tensor: torch.optim.Adam arithmetic (amsgrad off; weight_decay is the L2 form), `step` counts
 * from 1.  Host arrays of n_tensors device pointers / element counts. */
int odehip_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                     const long long* numel, int n_tensors, float lr, float beta1, float beta2, float eps, float weight_decay,
                     int step, void* stream);

/* ---- VidODE's warp chain + mask compositing (models/VidODE.py:119-140, get_warped_images :160-186) -------------------------- */

/* pred_outputs (B,T,c+3,H,W) = the flow decoder's output per predicted frame: channels [0:2] optical flow (x, y) in pixels,
 * [2:2+c] the "intermediate" frame, [2+c] the mask logit.  start_image (B,c,H,W) = the last observed frame.  grid_x[W], grid_y[H]
 * = torch.linspace(-1, 1, n) (VidODE.py:122-123; handed over so that they are the caller's torch's values).  For t = 0..T-1:
 *   warped_t = grid_sample(warped_{t-1}, grid + flow_t / ((n-1)/2), bilinear, padding_mode="border", align_corners=False)
 *   masks_t = sigmoid(logit_t);  pred_x_t = masks_t * warped_t + (1 - masks_t) * intermediate_t
 * with warped_{-1} = start_image -- the whole chain in ONE launch (one workgroup per sample, the image stays in LDS).
 * Outputs: pred_x, warped (B,T,c,H,W), masks (B,T,1,H,W).  c <= 4, 2*c*H*W*4 bytes must fit in LDS (160 KiB). */
int odehip_warp_composite(const float* pred_outputs, const float* start_image, const float* grid_x, const float* grid_y, int batch,
                          int n_times, int channels, int height, int width, float* pred_x, float* warped, float* masks,
                          void* stream);
/* Backward of the above: gradients w.r.t. pred_outputs (flow through the bilinear weights -- zero where a coordinate was clamped
 * to the border --, intermediate frames, mask logits) and, if grad_start_image != NULL, the start image.  `warped` is the forward's
 * output; grad_warped / grad_masks may be NULL (no gradient arrives through those outputs).  The scatter into the previous
 * image's gradient uses LDS float atomics: reproducible to rounding, not bitwise. */
int odehip_warp_composite_backward(const float* pred_outputs, const float* start_image, const float* warped, const float* grid_x,
                                   const float* grid_y, const float* grad_pred_x, const float* grad_warped, const float* grad_masks,
                                   int batch, int n_times, int channels, int height, int width, float* grad_pred_outputs,
                                   float* grad_start_image, void* stream);

/* nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) of VidODE's flow decoder (models/VidODE.py:34, applied per
 * predicted frame by get_flowmaps :143-158) on `planes` = N * C images of (height, width) -> (2 height, 2 width), ATen's arithmetic
 * (source index max(0, (dst + 0.5) / 2 - 0.5), the four-point blend in its order of operations).  HBM-bound data movement: torch's
 * own kernel runs it at 20 GB/s on this stack (4.2 ms per call, 85 % of a VidODE forward at batch 64).  width must be even.
 * Backward: the transpose as a gather (deterministic). */
int odehip_upsample2x_bilinear(const float* in, float* out, long long planes, int height, int width, void* stream);
int odehip_upsample2x_bilinear_backward(const float* grad_out, float* grad_in, long long planes, int height, int width, void* stream);

/* BatchNorm2d -> ReLU (-> that upsampling) of the same decoder (models/VidODE.py:35-36) as one pass over the convolution's output
 * x (batch, channels, height, width) NCHW fp32; width % 4 == 0.  training != 0: the statistics of THIS call (biased variance for the
 * normalisation; running_mean / running_var, if given, updated with `momentum` and the unbiased variance, as nn.BatchNorm2d does);
 * else the running statistics.  weight / bias may be NULL (affine=False).  out = relu(bn(x)), upsampled x2 (-> 2 height x 2 width) when
 * upsample != 0.  mean_out, invstd_out, scale_out, shift_out [channels]: what the backward needs (scale = weight * invstd,
 * shift = bias - mean * scale).  workspace: odehip_bn_workspace_bytes(channels) (forward), + 2 * channels floats (backward).
 * Backward: grad_out has out's shape; g_pre (x's shape) is scratch the caller provides; grad_x, grad_weight, grad_bias are written.
 * Every reduction has a fixed order: bitwise reproducible. */
size_t odehip_bn_workspace_bytes(int channels);
int odehip_bn_relu_up2x_forward(const float* x, int batch, int channels, int height, int width, const float* weight, const float* bias,
                                float* running_mean, float* running_var, int training, float momentum, float eps, int upsample, float* out,
                                float* mean_out, float* invstd_out, float* scale_out, float* shift_out, void* workspace,
                                size_t workspace_bytes, void* stream);
int odehip_bn_relu_up2x_backward(const float* grad_out, const float* x, int batch, int channels, int height, int width, const float* mean,
                                 const float* invstd, const float* scale, const float* shift, int training, int upsample, float* grad_x,
                                 float* grad_weight, float* grad_bias, float* g_pre, void* workspace, size_t workspace_bytes, void* stream);

/* ---- the conv encoder / decoder either side of the path (models/ODEConvGRU.py:101-118 Encoder, :121-140 Decoder; n_downs = 2) --
 * Each is ONE fused launch: the 32x32 intermediate stays in LDS, the 16 -> out_ch and in_ch -> 32 layers run on the fp32 MFMA.
 * Frames are 64x64, latents 16x16 (the path's fixed map size).  LeakyReLU slope: the reference's 0.2.
 *
 * Encoder: Conv2d(in_ch, 16, 3, 2, 1) -> LeakyReLU -> Conv2d(16, out_ch, 3, 2, 1) -> LeakyReLU.  w1 (16, in_ch, 3, 3), b1 (16),
 * w2 (out_ch, 16, 3, 3), b2 (out_ch) as nn.Conv2d holds them; in_ch 1..4, out_ch 32 / 64 / 128.  `pack` (device,
 * odehip_frame_encoder_pack_floats floats) is refreshed by odehip_pack_frame_encoder whenever the parameters change.
 * frames (B, T, in_ch, 64, 64) -- the reference's inputs.view(b*t, c, h, w) (:63) -- and the result is written TIME-FIRST,
 * (T, B, out_ch, 16, 16) contiguous: what :64-68 produce as a permuted view and odehip_odeconvgru_encode consumes. */
size_t odehip_frame_encoder_pack_floats(int in_ch, int out_ch);
int odehip_pack_frame_encoder(const float* w1, const float* b1, const float* w2, const float* b2, int in_ch, int out_ch, float* pack,
                              void* stream);
int odehip_frame_encode(const float* pack, const float* frames, int batch, int n_frames, int in_ch, int out_ch, float negative_slope,
                        float* out_time_first, void* stream);
/* Decoder: ConvTranspose2d(in_ch, 32, 4, 2, 1) -> LeakyReLU -> ConvTranspose2d(32, out_ch, 4, 2, 1) [-> sigmoid if apply_sigmoid:
 * the F.sigmoid of :85].  w1 (in_ch, 32, 4, 4), b1 (32), w2 (32, out_ch, 4, 4), b2 (out_ch) as nn.ConvTranspose2d holds them;
 * in_ch 32 / 64 / 128, out_ch 1..4.  latents (N, in_ch, 16, 16) -- the solver's (T, B, C, 16, 16) as it lies, N = T*B (:84) --
 * -> out (N, out_ch, 64, 64). */
size_t odehip_frame_decoder_pack_floats(int in_ch, int out_ch);
int odehip_pack_frame_decoder(const float* w1, const float* b1, const float* w2, const float* b2, int in_ch, int out_ch, float* pack,
                              void* stream);
int odehip_frame_decode(const float* pack, const float* latents, int n_images, int in_ch, int out_ch, float negative_slope,
                        int apply_sigmoid, float* out, void* stream);
/* Backward of the two fused launches above (frame_codec_backward.hip) -- what `loss.backward()` needs from the Encoder / Decoder of
 * models/ODEConvGRU.py:101-140 in a training step, for ONE frame channel and 32 / 64 latent channels (other shapes: rc != 0, the
 * caller keeps the library's backward).  The 32x32 intermediates are recomputed in LDS from the forward's inputs; weight gradients
 * are summed in a fixed order (bitwise reproducible).  All gradients are WRITTEN (not accumulated), in the parameters' own layouts.
 * `workspace`: device scratch of odehip_frame_*_backward_workspace_floats floats (0 = unsupported shape).
 *
 * Decoder: pack = the forward's pack, w1 = the first ConvTranspose2d's weight (in_ch, 32, 4, 4) as it lies; latents (N, in_ch, 16,
 * 16) and pred (N, 1, 64, 64) = the forward's input and output, g_out = dL/d pred (sigmoid_applied: pred is after the sigmoid and
 * its derivative pred (1 - pred) is applied here).  Out: g_latents (N, in_ch, 16, 16), dw1 (in_ch, 32, 4, 4), db1 (32), dw2 (32, 1,
 * 4, 4), db2 (1). */
/* odehip_frame_decode_train = odehip_frame_decode that also writes the 32-channel intermediate (after its LeakyReLU) to mid_save
 * (n_images * 32 * 32 * 32 floats, channel quads: [N][8][32][32] x 4; may be NULL): handed to the backward as mid_saved it replaces the
 * recomputation there (mid_saved NULL: recomputed from the latents, nothing but inputs and outputs kept). */
int odehip_frame_decode_train(const float* pack, const float* latents, int n_images, int in_ch, int out_ch, float negative_slope,
                              int apply_sigmoid, float* out, float* mid_save, void* stream);
size_t odehip_frame_decode_backward_workspace_floats(int n_images, int in_ch, int out_ch);
int odehip_frame_decode_backward(const float* pack, const float* w1, const float* latents, const float* mid_saved, const float* pred,
                                 const float* g_out, int n_images, int in_ch, int out_ch, float negative_slope, int sigmoid_applied,
                                 float* g_latents, float* dw1, float* db1, float* dw2, float* db2, float* workspace,
                                 size_t workspace_floats, void* stream);
/* Encoder: pack = the forward's pack, w2 = the second Conv2d's weight (out_ch, 16, 3, 3); frames (B, T, 1, 64, 64),
 * out_time_first = the forward's result (T, B, out_ch, 16, 16), g_out_time_first = dL/d of it in the same layout.  The frames
 * receive no gradient.  Out: dw1 (16, 1, 3, 3), db1 (16), dw2 (out_ch, 16, 3, 3), db2 (out_ch). */
size_t odehip_frame_encode_backward_workspace_floats(int batch, int n_frames, int in_ch, int out_ch);
int odehip_frame_encode_backward(const float* pack, const float* w2, const float* frames, const float* out_time_first,
                                 const float* g_out_time_first, int batch, int n_frames, int in_ch, int out_ch, float negative_slope,
                                 float* dw1, float* db1, float* dw2, float* db2, float* workspace, size_t workspace_floats, void* stream);

/* odehip_odeint_fixed runs a forward-only trajectory of a 64-channel fp32 stack as ONE persistent launch (the four workgroups of a
 * sample hand layers to each other through L2 instead of through launch boundaries).  On by default
 * (environment ODEHIP_PERSISTENT=0 turns it off); this switch is for A/B measurements and tests.  Returns the previous
 * setting.  odehip_persistent_trajectory_launches: how many trajectories have taken that path in this process. */
int odehip_set_persistent_trajectory(int enable);
long long odehip_persistent_trajectory_launches(void);
/* Sticky error word of the persistent launches: 0 = none, else the code a capped in-kernel wait left when it gave up (2: a
 * partner workgroup never announced itself, 3: a partner's layer never arrived).  Host-side read of mapped memory, no
 * synchronisation.  clear != 0 resets the word and switches the persistent path off for the process. */
int odehip_persistent_error(int clear);

/* Moving-MNIST-shaped frames rendered on the device (replaces the host generator dataloader.py:47-103 + the normalisation of
 * __getitem__ :217-218).  init: [batch][n_digits][4] doubles = x, y, v_x, v_y in the unit square (drawn by the host as
 * dataloader.py:50-54 does); digit_ids: [batch][n_digits] indices into glyphs [n_glyphs][28][28] (uint8); lut256[v] =
 * float32(v) / 255 - 0.5 evaluated in float32 as numpy does for the reference's float32 frames.  Frames 0..t_in-1 go to out_in (batch, t_in, 1, 64, 64), the rest to out_pred (batch, t_out, 1,
 * 64, 64).  The digit ids are NOT range-checked on the device: the caller guarantees 0 <= id < n_glyphs. */
int odehip_mmnist_render(const double* init, const int* digit_ids, const unsigned char* glyphs, int n_glyphs,
                         const float* lut256, int batch, int n_digits, int t_in, int t_out, float* out_in, float* out_pred,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ODECGRU_HIP_H */
