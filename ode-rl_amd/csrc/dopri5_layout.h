// dopri5_layout.h -- workspace layout of the backward pass through dopri5 (dopri5_backward.hip); shared with the forward solver
// (dopri5.hip), whose SAVING mode writes the stage inputs and hidden activations of every accepted step straight into these slots
// so that the backward pass needs no re-integration.
#pragma once
#include "odehip_internal.h"
#include "persist.h"

namespace odehip {

struct BwdLayout {
  int T, B, C, NH, N;
  size_t st, hid, slot_bytes;
  size_t off_h, off_ping, off_pong, off_go, off_k, off_gY, off_gy, off_gk1, off_slots, off_tab, off_psync, off_slab, total;
  BwdLayout(const odehip_convstack* f, int batch, int n_times, int n_steps) {
    T = n_times; B = batch; C = f->channels[0]; NH = f->n_convs - 1; N = n_steps;
    st = al256((size_t)B * C * kPix * 4);
    int cmax = 32;
    for (int i = 0; i <= f->n_convs; ++i) cmax = f->channels[i] > cmax ? f->channels[i] : cmax;
    hid = al256((size_t)B * cmax * kPix * 4);
    slot_bytes = 7 * (st + (size_t)NH * hid + (size_t)(NH + 1) * hid);
    size_t o = 0;
    auto take = [&](size_t b) { size_t r = o; o += al256(b); return r; };
    off_h = take((size_t)(N > 0 ? N : 1) * 4);
    off_ping = take(hid);
    off_pong = take(hid);
    off_go = take((size_t)T * st);
    off_k = take(7 * st);
    off_gY = take(7 * st);
    off_gy = take(2 * st);
    off_gk1 = take(2 * st);
    off_slots = take((size_t)(N > 0 ? N : 1) * slot_bytes);
    off_tab = take(((size_t)N * 6 + 1) * sizeof(WgradPair) * ODEHIP_MAX_LAYERS);
    off_psync = take(persist_sync_bytes(B));
    off_slab = take(((size_t)B * wgrad_esplit_max(B) + 1) * kWgradSlabFloats * 4);
    total = o;
  }
  float* p(const void* ws, size_t off) const { return (float*)((char*)const_cast<void*>(ws) + off); }
  float* xin(const void* ws, int n, int s) const { return p(ws, off_slots + (size_t)n * slot_bytes + (size_t)s * st); }
  float* hidden(const void* ws, int n, int s, int l) const {
    return p(ws, off_slots + (size_t)n * slot_bytes + 7 * st + ((size_t)s * NH + l) * hid);
  }
  float* gp(const void* ws, int n, int s, int l) const {
    return p(ws, off_slots + (size_t)n * slot_bytes + 7 * st + 7 * (size_t)NH * hid + ((size_t)s * (NH + 1) + l) * hid);
  }
};


}  // namespace odehip
