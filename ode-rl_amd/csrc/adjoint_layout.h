// adjoint_layout.h -- workspace layout of the adaptive adjoint (adjoint_dopri5.hip: host-driven; adjoint_device.hip: device-driven)
#pragma once
#include "odehip_internal.h"
#include "persist.h"

namespace odehip {

struct AdjLayout {
  int T, B, C, NH, max_slots, n_part;
  size_t st, hid, slot_bytes;
  size_t off_h, off_part, off_sums, off_ping, off_pong, off_y, off_go, off_a2, off_ky, off_ka, off_slots, off_tab, off_slab, off_theta, off_state, off_psync, off_wtab, total;
  int P;  // floats of the flattened parameter vector (w0, b0, w1, b1, ...)
  AdjLayout(const odehip_convstack* f, int batch, int n_times, int max_accept) {
    T = n_times; B = batch; C = f->channels[0]; NH = f->n_convs - 1; max_slots = max_accept + 1;
    st = al256((size_t)B * C * kPix * 4);
    int cmax = 32;
    for (int i = 0; i <= f->n_convs; ++i) cmax = f->channels[i] > cmax ? f->channels[i] : cmax;
    hid = al256((size_t)B * cmax * kPix * 4);
    n_part = B * (C / 32) * 2 * 4;
    slot_bytes = 7 * (st + (size_t)NH * hid + (size_t)(NH + 1) * hid);
    size_t o = 0;
    auto take = [&](size_t b) { size_t r = o; o += al256(b); return r; };
    off_h = take(256);
    off_part = take(8 * (size_t)part_stride() * 4);
    off_sums = take(256);
    off_ping = take(hid);
    off_pong = take(hid);
    off_y = take((size_t)T * st);
    off_go = take((size_t)T * st);
    off_a2 = take(2 * st);
    off_ky = take(7 * st);
    off_ka = take(7 * st);
    off_slots = take((size_t)max_slots * slot_bytes);
    off_tab = take((size_t)max_slots * 7 * sizeof(WgradPair));
    off_slab = take(((size_t)B * wgrad_esplit_max(B) + 1) * kWgradSlabFloats * 4);
    P = 0;
    for (int l = 0; l < f->n_convs; ++l) P += f->channels[l + 1] * f->channels[l] * 9 + f->channels[l + 1];
    off_theta = take((size_t)5 * P * 4);  // mixed norm: running a_theta, error estimate, increment, K^theta at the two initial-step points
    // device-driven path (adjoint_device.hip): controller state, the walk's flag area, one weight-gradient table per layer
    off_state = take(4096);
    off_psync = take(persist_sync_bytes(B));
    off_wtab = take((size_t)ODEHIP_MAX_LAYERS * max_slots * 7 * sizeof(WgradPair));
    total = o;
  }
  float* p(const void* ws, size_t off) const { return (float*)((char*)const_cast<void*>(ws) + off); }
  float* xin(const void* ws, int slot, int s) const { return p(ws, off_slots + (size_t)slot * slot_bytes + (size_t)s * st); }
  float* hidden(const void* ws, int slot, int s, int l) const {
    return p(ws, off_slots + (size_t)slot * slot_bytes + 7 * st + ((size_t)s * NH + l) * hid);
  }
  float* gp(const void* ws, int slot, int s, int l) const {
    return p(ws, off_slots + (size_t)slot * slot_bytes + 7 * st + 7 * (size_t)NH * hid + ((size_t)s * (NH + 1) + l) * hid);
  }
  // floats per partial array: the per-layer kernels write n_part, the sixteen-workgroup walk 64 per sample (batch <= 16)
  int part_stride() const { return n_part > 1024 ? n_part : 1024; }
  float* part(const void* ws, int j) const { return p(ws, off_part + (size_t)j * part_stride() * 4); }
};


// byte offsets inside a slot (relocatable pointers of the device-driven path are class << 56 | one of these)
inline size_t adj_off_xin(const AdjLayout& L, int s) { return (size_t)s * L.st; }
inline size_t adj_off_hidden(const AdjLayout& L, int s, int l) { return 7 * L.st + ((size_t)s * L.NH + l) * L.hid; }
inline size_t adj_off_gp(const AdjLayout& L, int s, int l) { return 7 * L.st + 7 * (size_t)L.NH * L.hid + ((size_t)s * (L.NH + 1) + l) * L.hid; }

// adjoint_device.hip: the seminorm adjoint steered by a device-side controller on the adaptive persistent walk; returns
// ODEHIP_OK and sets *ran = 1 when it took the call, *ran = 0 when the path is not available (the caller then runs the host loop)
int adjoint_dopri5_device(const odehip_convstack* f, const odehip_convstack* f_dgrad, const double* t_host, int n_times, int batch, float rtol,
                          float atol, const float* y_traj_nchw, const float* grad_out_nchw, float* grad_z0_nchw, float* const* grad_w,
                          float* const* grad_b, int max_accept, int* stats_host, void* workspace, size_t workspace_bytes, hipStream_t stream,
                          int* ran);

}  // namespace odehip
