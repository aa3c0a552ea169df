// conv_common.h -- device helpers shared by the convolution kernels (conv_q4.hip, conv_wino.hip).
#pragma once
#include "odehip_internal.h"

namespace odehip {

// ReLU that PROPAGATES NaN, as torch.relu does: v_max_f32(v, 0) returns 0 for a NaN input, which would launder a non-finite
// activation into a plausible zero (found by round 4's fault-injection test: NaN written by a walk that had given up disappeared in the
// ReLUs of the encoder's 1x1 head).  v < 0 is false for NaN, so NaN passes through; finite values are unchanged.
__device__ __forceinline__ float relu_f(float v) { return v < 0.0f ? 0.0f : v; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ODEHIP_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int kOobOffset = 0x7fff0000;  // beyond any buffer's num_records -> DMA writes zeros

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  // raw buffer (stride 0), DATA_FORMAT=32 so that the range check is enabled: flags 0x00020000
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, int voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, ODEHIP_LDS_PTR(lds), 16, voffset, soffset, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// o + a * c with ONE rounding per element, stated explicitly: the reverse Runge-Kutta bookkeeping exists in several kernels
// (these epilogues and the register-resident sweep of btraj_bf16.hip) whose results are compared bit for bit -- left to the
// compiler, `o += a * c` is contracted into an fma in one kernel and not in another
__device__ __forceinline__ f32x4 fma4(f32x4 a, float c, f32x4 o) {
  return f32x4{__builtin_fmaf(a.x, c, o.x), __builtin_fmaf(a.y, c, o.y), __builtin_fmaf(a.z, c, o.z), __builtin_fmaf(a.w, c, o.w)};
}


// ---- order-1 stage combine (CombineArgs::order == 1; the adaptive solver's drivers) of one channel quad of one pixel.  Split in
// two so that the adaptive persistent walk can run the first half ahead of the matrix work; emit_quad / epilogue() run both back to
// back.  Every multiply-add is an explicit fma: all kernels round alike.
struct CombinePartial {
  f32x4 sa, sb, se;   // sums over k_prev of c1 / c2 / ce
};
__device__ __forceinline__ CombinePartial combine1_prev(const CombineArgs& m, size_t off) {
  CombinePartial p;
  p.sa = p.sb = p.se = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < m.n_prev; ++j) {
    const f32x4 kp = *(const f32x4*)(m.k_prev[j] + off);
    p.sa = fma4(kp, m.c1[j], p.sa);
    p.sb = fma4(kp, m.c2[j], p.sb);
    p.se = fma4(kp, m.ce[j], p.se);
  }
  return p;
}
// error-norm contribution of one quad: err = h * se, tol = atol + rtol * max(|y|, |y1|)
__device__ __forceinline__ float combine1_err(f32x4 se, float h, f32x4 yv, f32x4 y1, float rtol, float atol, float esum) {
  const f32x4 e = se * h;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float tol = __builtin_fmaf(rtol, fmaxf(fabsf(yv[i]), fabsf(y1[i])), atol);
    const float r = e[i] / tol;
    esum = __builtin_fmaf(r, r, esum);
  }
  return esum;
}
__device__ __forceinline__ void combine1_tail(const CombineArgs& m, int b, int qout, int Q, int P, size_t off, f32x4 kc, f32x4 yv, float h,
                                              CombinePartial p, float& esum, float* nchw) {
  const int n = m.n_prev;
  if (m.out1) *(f32x4*)(m.out1 + off) = fma4(fma4(kc, m.c1[n], p.sa), h, yv);
  if (m.out2 || nchw) {
    const f32x4 o2 = fma4(fma4(kc, m.c2[n], p.sb), h, yv);
    if (m.out2) *(f32x4*)(m.out2 + off) = o2;
    if (nchw) {
      float* o = nchw + ((size_t)b * qout * 4 + Q * 4) * kPix + P;
      o[0] = o2.x; o[kPix] = o2.y; o[2 * kPix] = o2.z; o[3 * kPix] = o2.w;
    }
  }
  if (m.err_partials) esum = combine1_err(fma4(kc, m.ce[n], p.se), h, yv, *(const f32x4*)(m.err_y1 + off), m.rtol, m.atol, esum);
}

// ---- elementwise row (ConvArgs::combine == 4) on one quad: out = (y ? y : 0) + sum_j (c[j] * hs) * k_prev[j], explicit fmas
__device__ __forceinline__ void ew_quad(const CombineArgs& m, size_t off, float hs) {
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f};
  if (m.y) s1 = *(const f32x4*)(m.y + off);
  f32x4 s2 = s1;
  for (int j = 0; j < m.n_prev; ++j) {
    const f32x4 kp = *(const f32x4*)(m.k_prev[j] + off);
    s1 = fma4(kp, (m.c_dev ? m.c_dev[j] : m.c1[j]) * hs, s1);
    if (m.out2) s2 = fma4(kp, m.c2[j] * hs, s2);
  }
  if (m.out1) *(f32x4*)(m.out1 + off) = s1;
  if (m.out2) *(f32x4*)(m.out2 + off) = s2;
}

// ---- per-(channel quad Q, pixel P) epilogue: plain / ReLU store, Runge-Kutta stage combine (+ adaptive error
// partial), ReLU-mask backward, reverse-sweep targets.  `v` is the conv output (bias included) of 4 channels.
// nchw_override: persistent trajectory kernel only -- where this layer's NCHW result frame goes (its table cannot hold the
// pointer: the output tensor changes from call to call)
// ORDER1 = false: instantiations that can never meet an order-1 stage combine (the fixed-grid persistent walks: tables with such rows
// go to the adaptive walk) leave the branch out -- its uniform operands would otherwise cost the headline kernel scalar registers
template <bool ORDER1 = true>
__device__ __forceinline__ void emit_quad(const ConvArgs& a, int b, int Q, int P, f32x4 v, float& esum, float* nchw_override = nullptr) {
  const size_t off = (((size_t)b * a.qout + Q) * kPix + P) * 4;
  if (a.combine == 0) {
    if (a.relu) {
      v.x = relu_f(v.x); v.y = relu_f(v.y); v.z = relu_f(v.z); v.w = relu_f(v.w);
    }
    *(f32x4*)(a.dst + off) = v;
    return;
  }
  if (a.combine == 2) {
    const BwdArgs& w = a.bwd;
    v *= w.sc_c + w.sc_h * (w.h_ptr ? *w.h_ptr : 0.0f);
    if (w.mask_src) {
      const f32x4 mk = *(const f32x4*)(w.mask_src + off);
      v.x = mk.x > 0.0f ? v.x : 0.0f; v.y = mk.y > 0.0f ? v.y : 0.0f;
      v.z = mk.z > 0.0f ? v.z : 0.0f; v.w = mk.w > 0.0f ? v.w : 0.0f;
    }
    *(f32x4*)(a.dst + off) = v;
    return;
  }
  if (a.combine == 3) {
    const BwdArgs& w = a.bwd;
    const float hb = w.h_ptr ? *w.h_ptr : 0.0f;
    for (int t = 0; t < w.n_targets; ++t) {
      const BwdTarget& T = w.tgt[t];
      f32x4 o = v * (T.g_c + T.g_h * hb);
      if (T.srcA) o = fma4(*(const f32x4*)(T.srcA + off), T.a_c + T.a_h * hb, o);
      if (T.srcB) o = fma4(*(const f32x4*)(T.srcB + off), T.b_c + T.b_h * hb, o);
      *(f32x4*)(T.out + off) = o;
    }
    return;
  }
  const CombineArgs& m = a.cmb;
  const float h = m.h_ptr ? *m.h_ptr : 1.0f;
  const f32x4 kc = v * m.k_scale;
  if (m.k_out) *(f32x4*)(m.k_out + off) = kc;
  if constexpr (ORDER1) {
    if (m.y && m.order) {
      combine1_tail(m, b, a.qout, Q, P, off, kc, *(const f32x4*)(m.y + off), h, combine1_prev(m, off), esum,
                    nchw_override ? nchw_override : m.out2_nchw);
      return;
    }
  }
  if (m.y) {
    const f32x4 yv = *(const f32x4*)(m.y + off);
    f32x4 sa = kc * m.c1[m.n_prev];
    f32x4 sb = kc * m.c2[m.n_prev];
    f32x4 se = kc * m.ce[m.n_prev];
    for (int j = 0; j < m.n_prev; ++j) {
      const f32x4 kp = *(const f32x4*)(m.k_prev[j] + off);
      sa += kp * m.c1[j];
      sb += kp * m.c2[j];
      se += kp * m.ce[j];
    }
    if (m.out1) *(f32x4*)(m.out1 + off) = yv + sa * h;
    const f32x4 o2 = yv + sb * h;
    if (m.out2) *(f32x4*)(m.out2 + off) = o2;
    float* const nchw = nchw_override ? nchw_override : m.out2_nchw;
    if (nchw) {
      float* o = nchw + ((size_t)b * a.qout * 4 + Q * 4) * kPix + P;
      o[0] = o2.x; o[kPix] = o2.y; o[2 * kPix] = o2.z; o[3 * kPix] = o2.w;
    }
    if (m.err_partials) {
      const f32x4 y1 = *(const f32x4*)(m.err_y1 + off);
      const f32x4 e = se * h;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float tol = m.atol + m.rtol * fmaxf(fabsf(yv[i]), fabsf(y1[i]));
        const float r = e[i] / tol;
        esum += r * r;
      }
    }
  }
}

// diagnostic stamps (debug & 8), shader cycles (s_memtime) relative to the workgroup's start:
// [1] first DMAs issued, [2] stage 0 landed, [3..6] end of stage 0..3 MFMAs, [7] end; [0] = start in 100 MHz ticks
struct Stamps {
  unsigned long long cy[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rt0 = 0;
  bool on;
  __device__ __forceinline__ explicit Stamps(const ConvArgs& a, int who = 0) : on((a.debug & 8) && (int)threadIdx.x == who) {}
  __device__ __forceinline__ void take(int i) {
    if (on) {
      cy[i] = __builtin_amdgcn_s_memtime();
      if (i == 0) rt0 = __builtin_amdgcn_s_memrealtime();
    }
  }
  __device__ __forceinline__ void flush(const ConvArgs& a) {
    if (on) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      take(7);
      unsigned long long* o = a.dbg + (size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8;
      o[0] = rt0;
      for (int i = 1; i < 8; ++i) o[i] = cy[i] ? cy[i] - cy[0] : 0;
    }
  }
};

// one partial of the adaptive error norm per wave (fixed order downstream: no atomics)
__device__ __forceinline__ void finish_err(const ConvArgs& a, float esum, int wave) {
  if (a.combine == 1 && a.cmb.err_partials) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) esum += __shfl_xor(esum, o, 64);
    if ((threadIdx.x & 63) == 0) a.cmb.err_partials[(blockIdx.x + blockIdx.y * gridDim.x) * 4 + wave] = esum;
  }
}

// ---- shared by the 32x32-accumulator kernels (conv_q4.hip, conv_bf16.hip)
__device__ __forceinline__ int xcd_block_id() {
  // blocks p and p+8 share an XCD (round-robin dispatch; speed only, never correctness): give each XCD a
  // contiguous range of logical ids so the workgroups of one sample hit the same L2.
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
  return bid;
}

// 32 biases of the tile by two wave-uniform 64-B scalar loads; lane half kq picks its 16
__device__ __forceinline__ f32x16 bias_init(const float* bias, int ct, int kq) {
  f32x16 acc;
  if (bias) {
    const f32x16 lo16 = *(const f32x16*)(bias + ct * 32);
    const f32x16 hi16 = *(const f32x16*)(bias + ct * 32 + 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c0 = (r & 3) + 8 * (r >> 2);  // 0..27, +4 for the upper lane half
      const float lo = c0 < 16 ? lo16[c0 & 15] : hi16[c0 & 15];
      const float hi = (c0 + 4) < 16 ? lo16[(c0 + 4) & 15] : hi16[(c0 + 4) & 15];
      acc[r] = kq ? hi : lo;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  }
  return acc;
}

// ---- epilogue: lane (pixel i32, half kq) holds channel quads 2g+kq of this 32-channel tile
// part_idx >= 0 overrides the slot of this wave's error-norm partial (kernels whose grid is not (tile, sample))
__device__ __forceinline__ void epilogue(const ConvArgs& a, const f32x16& acc, int b, int ct, int P, int kq, int wave, int part_idx = -1) {
  if (!a.combine) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
      if (a.relu) {
        v.x = relu_f(v.x); v.y = relu_f(v.y); v.z = relu_f(v.z); v.w = relu_f(v.w);
      }
      const size_t off = (((size_t)b * a.qout + ct * 8 + 2 * g + kq) * kPix + P) * 4;
      *(f32x4*)(a.dst + off) = v;
    }
    return;
  }
  if (a.combine >= 2) {
    const BwdArgs& w = a.bwd;
    const float hb = w.h_ptr ? *w.h_ptr : 0.0f;
    if (a.combine == 2) {
      const float sc = w.sc_c + w.sc_h * hb;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const size_t off = (((size_t)b * a.qout + ct * 8 + 2 * g + kq) * kPix + P) * 4;
        f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        v *= sc;
        if (w.mask_src) {
          const f32x4 mk = *(const f32x4*)(w.mask_src + off);
          v.x = mk.x > 0.0f ? v.x : 0.0f; v.y = mk.y > 0.0f ? v.y : 0.0f;
          v.z = mk.z > 0.0f ? v.z : 0.0f; v.w = mk.w > 0.0f ? v.w : 0.0f;
        }
        *(f32x4*)(a.dst + off) = v;
      }
      return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const size_t off = (((size_t)b * a.qout + ct * 8 + 2 * g + kq) * kPix + P) * 4;
      const f32x4 gx = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
      for (int t = 0; t < w.n_targets; ++t) {
        const BwdTarget& T = w.tgt[t];
        f32x4 o = gx * (T.g_c + T.g_h * hb);
        if (T.srcA) o = fma4(*(const f32x4*)(T.srcA + off), T.a_c + T.a_h * hb, o);
        if (T.srcB) o = fma4(*(const f32x4*)(T.srcB + off), T.b_c + T.b_h * hb, o);
        *(f32x4*)(T.out + off) = o;
      }
    }
    return;
  }
  const CombineArgs& m = a.cmb;
  const float h = m.h_ptr ? *m.h_ptr : 1.0f;
  float esum = 0.0f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int Q = ct * 8 + 2 * g + kq;
    const size_t off = (((size_t)b * a.qout + Q) * kPix + P) * 4;
    f32x4 kc = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    kc *= m.k_scale;
    if (m.k_out) *(f32x4*)(m.k_out + off) = kc;
    if (m.y && m.order) {
      combine1_tail(m, b, a.qout, Q, P, off, kc, *(const f32x4*)(m.y + off), h, combine1_prev(m, off), esum, m.out2_nchw);
      continue;
    }
    if (m.y) {
      const f32x4 yv = *(const f32x4*)(m.y + off);
      f32x4 sa = kc * m.c1[m.n_prev];
      f32x4 sb = kc * m.c2[m.n_prev];
      f32x4 se = kc * m.ce[m.n_prev];
      for (int j = 0; j < m.n_prev; ++j) {
        const f32x4 kp = *(const f32x4*)(m.k_prev[j] + off);
        sa += kp * m.c1[j];
        sb += kp * m.c2[j];
        se += kp * m.ce[j];
      }
      if (m.out1) *(f32x4*)(m.out1 + off) = yv + sa * h;
      const f32x4 o2 = yv + sb * h;
      if (m.out2) *(f32x4*)(m.out2 + off) = o2;
      if (m.out2_nchw) {
        float* o = m.out2_nchw + ((size_t)b * a.qout * 4 + Q * 4) * kPix + P;
        o[0] = o2.x; o[kPix] = o2.y; o[2 * kPix] = o2.z; o[3 * kPix] = o2.w;
      }
      if (m.err_partials) {
        const f32x4 y1 = *(const f32x4*)(m.err_y1 + off);
        const f32x4 e = se * h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float tol = m.atol + m.rtol * fmaxf(fabsf(yv[i]), fabsf(y1[i]));
          const float r = e[i] / tol;
          esum += r * r;
        }
      }
    }
  }
  if (m.err_partials) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) esum += __shfl_xor(esum, o, 64);
    if ((threadIdx.x & 63) == 0) m.err_partials[part_idx >= 0 ? part_idx : (int)(blockIdx.x + blockIdx.y * gridDim.x) * 4 + wave] = esum;
  }
}


}  // namespace odehip
