// persist.hip -- host side of the persistent trajectory launch: eligibility, the library-owned table cache, PersistScope.
#include <stdlib.h>
#include <string.h>

#include "persist.h"

namespace odehip {

// The layer sequence of a call is recorded (launch_conv under g_conv_recorder) and handed over as a device table.  Tables are
// kept in a small LIBRARY-OWNED device cache keyed by content (the one allocation this library makes on its own: a caller's
// workspace may be reused by other entry points between calls, which would silently invalidate a table parked there); a
// steady loop re-uses its table without any upload.  Result pointers are stored as offsets so that a fresh output tensor per
// call does not change the table.
struct PersistState {
  std::mutex mu;
  struct Entry {
    std::vector<char> host;
    ConvArgs* dev = nullptr;
    size_t cap = 0;
    unsigned long long stamp = 0;
  } tab[8];
  unsigned long long clock = 0;
  unsigned* host_err = nullptr;      // mapped, pinned: written by the kernel if a capped wait gives up
  unsigned* host_err_dev = nullptr;
  int enabled = -1;                  // -1 unknown, 0 off (env, device, or a failed launch), 1 on
  long long launches = 0;
  // Tables that differ from call to call (the backward passes of the adaptive solver: buffers, step sizes and dense-output weights
  // follow the accepted steps) bypass the content cache: a ring of device slots filled through staged_upload() -- an
  // ASYNCHRONOUS transfer on the caller's stream, no stream synchronisation.  A slot is reused kVolSlots uploads later;
  // the event recorded behind the launch that read it says when that is safe (long past by then: the wait is a formality).
  static constexpr int kVolSlots = 4;
  struct VolSlot {
    ConvArgs* dev = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
  } vol[kVolSlots];
  int vol_next = 0;
  int vol_last = -1;   // slot of the upload that has not been followed by its launch yet
  // single evaluations on the trajectory walks (ODEHIP_EVAL_WALK=1): a flag area of the library, zeroed in front of every launch
  unsigned* eval_sync = nullptr;
  int eval_batch_cap = 0;
  // small launches: flag area that is never zeroed between launches (words are tagged with the launch's epoch)
  unsigned* small_flags = nullptr;
  int small_batch_cap = 0;
  unsigned small_epoch = 0;
};
static PersistState g_persist;

static bool persist_available() {
  PersistState& P = g_persist;
  if (P.enabled >= 0) return P.enabled == 1;
  P.enabled = 0;
  const char* env = getenv("ODEHIP_PERSISTENT");
  if (env && env[0] == '0') return false;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < kPersistGrid) return false;
  if (hipHostMalloc((void**)&P.host_err, 64, hipHostMallocMapped) != hipSuccess) return false;
  *P.host_err = 0;
  if (hipHostGetDevicePointer((void**)&P.host_err_dev, P.host_err, 0) != hipSuccess) return false;
  P.enabled = 1;
  return true;
}

// 0: not for a persistent walk; 1: a 64 -> 64 layer (the headline kernel); 2: a layer with a 128-channel side (128 -> 64 or
// 64 -> 128: the wide walk)
// 3: a row only the ADAPTIVE walk takes (wino_persist_d_kernel): an elementwise row, or a 64 -> 64 layer whose stage combine is
//    written in order 1 (the adaptive solver's drivers)
static int persist_layer_kind(const ConvArgs& a) {
  if (a.combine == 4 || a.combine == 5) return a.qout == 16 && !a.src2 ? 3 : 0;   // elementwise / norm rows
  if (!a.w_wino || a.w_bf16 || a.src2 || a.q1 != a.qin || a.combine < 0 || a.combine > 3) return 0;
  if (a.qin == 16 && a.qout == 16) return (a.combine == 1 && a.cmb.order == 1) ? 3 : 1;
  if ((a.qin == 32 && a.qout == 16) || (a.qin == 16 && a.qout == 32)) return 2;
  return 0;
}

// device copy of `items` (identical content -> the cached copy); null on failure
static const ConvArgs* persist_table(const ConvArgs* items, int n, hipStream_t stream) {
  PersistState& P = g_persist;
  const size_t bytes = (size_t)n * sizeof(ConvArgs);
  PersistState::Entry* lru = &P.tab[0];
  for (auto& e : P.tab) {
    if (e.dev && e.host.size() == bytes && memcmp(e.host.data(), items, bytes) == 0) {
      e.stamp = ++P.clock;
      return e.dev;
    }
    if (e.stamp < lru->stamp) lru = &e;
  }
  if (hipStreamSynchronize(stream) != hipSuccess) return nullptr;  // a running launch may still read the entry being replaced
  if (lru->cap < bytes) {
    if (lru->dev) (void)hipFree(lru->dev);
    lru->dev = nullptr;
    lru->cap = 0;
    if (hipMalloc((void**)&lru->dev, bytes) != hipSuccess) return nullptr;
    lru->cap = bytes;
  }
  lru->host.assign((const char*)items, (const char*)items + bytes);
  if (hipMemcpy(lru->dev, items, bytes, hipMemcpyHostToDevice) != hipSuccess) {
    lru->host.clear();
    return nullptr;
  }
  lru->stamp = ++P.clock;
  return lru->dev;
}

// Host -> device copy WITHOUT a stream synchronisation and without pageable memory on the stream: the bytes are copied into a
// pinned staging slot (a ring; a slot is free again once the event recorded behind ITS copy has passed) and transferred
// asynchronously on the caller's stream.
struct StageSlot {
  char* host = nullptr;
  size_t cap = 0;
  hipEvent_t ev = nullptr;
  bool pending = false;
};
static std::mutex g_stage_mu;
static StageSlot g_stage[8];
static int g_stage_next = 0;
int staged_upload(void* dst_dev, const void* src, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return ODEHIP_OK;
  std::lock_guard<std::mutex> lk(g_stage_mu);
  StageSlot& v = g_stage[g_stage_next];
  g_stage_next = (g_stage_next + 1) % 8;
  if (v.pending) {
    ODEHIP_CHECK_HIP(hipEventSynchronize(v.ev));
    v.pending = false;
  }
  if (!v.ev) ODEHIP_CHECK_HIP(hipEventCreateWithFlags(&v.ev, hipEventDisableTiming));
  if (v.cap < bytes) {
    if (v.host) (void)hipHostFree(v.host);
    v.host = nullptr;
    v.cap = 0;
    const size_t cap = (bytes + 65535) / 65536 * 65536;
    ODEHIP_CHECK_HIP(hipHostMalloc((void**)&v.host, cap, hipHostMallocDefault));
    v.cap = cap;
  }
  memcpy(v.host, src, bytes);
  ODEHIP_CHECK_HIP(hipMemcpyAsync(dst_dev, v.host, bytes, hipMemcpyHostToDevice, stream));
  ODEHIP_CHECK_HIP(hipEventRecord(v.ev, stream));
  v.pending = true;
  return ODEHIP_OK;
}

// device copy of a call-specific table: asynchronous upload into a ring of device slots (see PersistState::VolSlot); null on failure
static const ConvArgs* persist_table_async(const ConvArgs* items, int n, hipStream_t stream) {
  PersistState& P = g_persist;
  const size_t bytes = (size_t)n * sizeof(ConvArgs);
  PersistState::VolSlot& v = P.vol[P.vol_next];
  if (v.pending) {
    if (hipEventSynchronize(v.ev) != hipSuccess) return nullptr;
    v.pending = false;
  }
  if (!v.ev && hipEventCreateWithFlags(&v.ev, hipEventDisableTiming) != hipSuccess) return nullptr;
  if (v.cap < bytes) {
    if (v.dev) (void)hipFree(v.dev);
    v.dev = nullptr;
    v.cap = 0;
    const size_t cap = (bytes + 65535) / 65536 * 65536;
    if (hipMalloc((void**)&v.dev, cap) != hipSuccess) return nullptr;
    v.cap = cap;
  }
  if (staged_upload(v.dev, items, bytes, stream) != ODEHIP_OK) return nullptr;
  P.vol_last = P.vol_next;
  P.vol_next = (P.vol_next + 1) % PersistState::kVolSlots;
  return v.dev;
}
// behind the launch (or the replay) that read the table of the last persist_table_async()
static void persist_table_async_done(hipStream_t stream) {
  PersistState& P = g_persist;
  if (P.vol_last < 0) return;
  PersistState::VolSlot& v = P.vol[P.vol_last];
  P.vol_last = -1;
  if (hipEventRecord(v.ev, stream) == hipSuccess) v.pending = true;
  else (void)hipStreamSynchronize(stream);
}

constexpr int kGuardRegions = 24;
struct GuardArgs {
  const unsigned* abort_word;   // DEVICE word of the launch's flag area: 1 if a wait of that launch gave up (the mapped host word
                                // says the same, but a read over PCIe by every thread costs tens of microseconds)
  int n;
  float* p[kGuardRegions];
  unsigned long long floats[kGuardRegions];
};

__global__ void persist_guard_kernel(const GuardArgs a) {
  if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
  for (int r = 0; r < a.n; ++r)
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.floats[r]; i += (unsigned long long)gridDim.x * blockDim.x)
      a.p[r][i] = __builtin_nanf("");
}

static bool walk16_on() {
  static const bool on16 = [] { const char* e = getenv("ODEHIP_PERSIST16"); return !(e && e[0] == '0'); }();
  return on16;
}

// (see odehip_internal.h) -- takes the lock: not to be called inside a PersistScope
int persist_partials_per_sample(int batch) {
  std::lock_guard<std::mutex> g(g_persist.mu);
  return batch <= 16 && walk16_on() && persist_available() ? 64 : 16;
}

bool persist_switch_on() {
  std::lock_guard<std::mutex> g(g_persist.mu);
  if (g_persist.enabled == 0) return false;
  static const bool env_off = [] { const char* e = getenv("ODEHIP_PERSISTENT"); return e && e[0] == '0'; }();
  return !env_off;
}

void persist_count_launch() {
  std::lock_guard<std::mutex> g(g_persist.mu);
  ++g_persist.launches;
}

// caller holds g_persist.mu
static unsigned persist_error_locked(bool clear) {
  PersistState& P = g_persist;
  if (!P.host_err) return 0;
  const unsigned code = *P.host_err;
  if (code && clear) {
    *P.host_err = 0;
    P.enabled = 0;
  }
  return code;
}

unsigned persist_error(bool clear) {
  std::lock_guard<std::mutex> g(g_persist.mu);
  return persist_error_locked(clear);
}

PersistScope::PersistScope() : lock_(g_persist.mu, std::defer_lock) {}

int PersistScope::guard(float* const* regions, const size_t* floats, int n, hipStream_t stream) {
  if (!launched_ || !abort_word_) return ODEHIP_OK;
  for (int base = 0; base < n; base += kGuardRegions) {
    GuardArgs a;
    memset(&a, 0, sizeof(a));
    a.abort_word = abort_word_;
    a.n = n - base < kGuardRegions ? n - base : kGuardRegions;
    for (int i = 0; i < a.n; ++i) {
      a.p[i] = regions[base + i];
      a.floats[i] = regions[base + i] ? floats[base + i] : 0;
    }
    hipLaunchKernelGGL(persist_guard_kernel, dim3(64), dim3(256), 0, stream, a);
    ODEHIP_CHECK_HIP(hipGetLastError());
  }
  return ODEHIP_OK;
}
PersistScope::~PersistScope() {
  if (g_conv_recorder == &rec_) g_conv_recorder = nullptr;  // only the scope that installed the recorder removes it
}

int PersistScope::begin(const odehip_convstack* f, const odehip_convstack* f2, int max_layers, bool small) {
    small_ = small;
    if (small && (g_conv_recorder || max_layers > 5)) return ODEHIP_OK;  // inside an outer scope: its recorder takes the layers
    // ODEHIP_EVAL_WALK=1: the <= 5 layers of a single evaluation / input-gradient chain (the encoder loop's Euler step and its
    // backward) as ONE launch of the TRAJECTORY walks -- wino_persist_kernel, or the sixteen-workgroup walk up to batch 16 -- instead
    // of the round-2 single-evaluation kernel below (9.5 us per layer against 7.5 / 3.3): the table goes up through the volatile
    // ring, the flags live in a library-owned area zeroed per launch.  No host-side guard launch behind it; the sixteen-workgroup walk
    // NaN-fills its own outputs when a wait of the launch gave up (nan_fill_row16, conv_wino.hip), and the sticky word raises at the next call.
    // Round 4: ON for batches up to 16 (the sixteen-workgroup walk: the reference's batch 4 trains 5.16 -> 4.92 ms per step with it),
    // off above (B = 64: the cell loop 2.81 -> 2.93 ms) unless ODEHIP_EVAL_WALK=1 forces it; ODEHIP_EVAL_WALK=0 switches it off.
    // The batch is only known in finish(): a scope that turns out too large replays its rows as ordinary launches there.
    static const int eval_walk_env = [] { const char* e = getenv("ODEHIP_EVAL_WALK"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    eval_walk_small_only_ = eval_walk_env < 0;
    if (small && eval_walk_env != 0) {
      small_ = false;
      eval_walk_ = true;
      volatile_ = true;
      small = false;
      for (int l = 0; l <= f->n_convs; ++l)
        if (f->channels[l] != 64) return ODEHIP_OK;   // 64-channel stacks only here
    }
    if (small) {
      // Measured (dopri5 forward + backward, B=64): 55 us per 5-layer launch = 11 us per layer, no better than five launches --
      // the last layer of an adaptive solver's evaluation combines up to six earlier stages through the shared epilogue
      // (operands fetched AFTER the last MFMA, one dependent load after the other).  Off unless asked for, until that epilogue
      // gets the prefetched-operand treatment of the fixed-grid one.
      static const bool small_on = [] { const char* e = getenv("ODEHIP_PERSISTENT_SMALL"); return e && e[0] == '1'; }();
      if (!small_on) return ODEHIP_OK;
    }
    if (f->ks != 3 || f->w_fused || (f2 && f2->w_fused) || g_debug_flags) return ODEHIP_OK;
    for (int l = 0; l <= f->n_convs; ++l)
      if (f->channels[l] != 64 && (small || f->channels[l] != 128)) return ODEHIP_OK;  // (128-channel sides: the wide walk, finish())
    for (int l = 0; l < f->n_convs; ++l)
      if (f->channels[l] == 128 && f->channels[l + 1] == 128) return ODEHIP_OK;
    for (int l = 0; l < f->n_convs; ++l)
      if (!f->w_wino[l] || f->w_bf16[l] || (f2 && (!f2->w_wino[l] || f2->w_bf16[l]))) return ODEHIP_OK;
    lock_.lock();
    if (!persist_available()) {
      lock_.unlock();
      return ODEHIP_OK;
    }
    if (*g_persist.host_err) {
      const unsigned code = *g_persist.host_err;
      *g_persist.host_err = 0;
      g_persist.enabled = 0;
      set_error("an earlier persistent launch gave up waiting for a partner workgroup (code %u); its result is invalid.  "
                "Persistent launches are now disabled for this process", code);
      return ODEHIP_EINVAL;
    }
    items_.resize((size_t)max_layers);
    rec_.items = items_.data();
    rec_.count = 0;
    rec_.capacity = max_layers;
    g_conv_recorder = &rec_;
    active_ = true;
    return ODEHIP_OK;
  }

int PersistScope::finish(const float* hbuf, const float* hdev, float* out_nchw, int batch, unsigned* sync, int ks, hipStream_t stream,
                         bool sync_is_zero) {
    if (g_conv_recorder == &rec_) g_conv_recorder = nullptr;
    if (!active_) return ODEHIP_OK;
    active_ = false;
    const bool eval_skip = eval_walk_ && eval_walk_small_only_ && (batch > 16 || !walk16_on());   // default: small batches only
    if (eval_walk_ && !sync && !eval_skip) {   // the library's flag area (grown synchronously, rarely)
      PersistState& P = g_persist;
      if (batch > P.eval_batch_cap) {
        bool ok = hipStreamSynchronize(stream) == hipSuccess;
        if (ok && P.eval_sync) (void)hipFree(P.eval_sync);
        P.eval_sync = nullptr;
        P.eval_batch_cap = 0;
        const int cap = batch < 64 ? 64 : batch;
        ok = ok && hipMalloc((void**)&P.eval_sync, persist_sync_bytes(cap)) == hipSuccess;
        if (ok) P.eval_batch_cap = cap;
      }
      sync = P.eval_sync;   // null: the recorded layers are replayed below
      sync_is_zero = false;
    }
    bool all_ok = rec_.count > 0 && !(eval_walk_ && !sync), wide = false, adaptive = adaptive_;
    if (eval_skip) all_ok = false;
    for (int i = 0; i < rec_.count && all_ok; ++i) {
      const int kind = persist_layer_kind(rec_.items[i]);
      all_ok = kind != 0;
      wide = wide || kind == 2;
      adaptive = adaptive || kind == 3;
    }
    if (wide && (small_ || adaptive)) all_ok = false;   // (order-1 rows of a 128-channel-ended stack: one launch per layer)
    if (adaptive && small_) all_ok = false;
    if (small_ && all_ok && rec_.count <= 5) {
      PersistState& P = g_persist;
      bool ready = true;
      if (batch > P.small_batch_cap || P.small_epoch >= (1u << 21)) {  // (re)allocate / re-zero the flag area: rare, synchronous
        ready = hipStreamSynchronize(stream) == hipSuccess;
        if (ready && batch > P.small_batch_cap) {
          if (P.small_flags) (void)hipFree(P.small_flags);
          P.small_flags = nullptr;
          P.small_batch_cap = 0;
          const int cap = batch < 64 ? 64 : batch;
          ready = hipMalloc((void**)&P.small_flags, persist_sync_bytes(cap)) == hipSuccess;
          if (ready) P.small_batch_cap = cap;
        }
        ready = ready && hipMemset(P.small_flags, 0, persist_sync_bytes(P.small_batch_cap)) == hipSuccess;
        P.small_epoch = 0;
      }
      if (ready) {
        for (int i = 0; i < rec_.count; ++i) {
          rec_.items[i].dbg = nullptr;
          rec_.items[i].h_by_value = 0;
        }
        const int rcs = launch_wino_persist_small(rec_.items, rec_.count, batch, P.small_flags,
                                                  P.small_flags + (size_t)P.small_batch_cap * kPersistDoneStride, ++P.small_epoch,
                                                  P.host_err_dev, kPersistGrid, stream);
        if (rcs == ODEHIP_OK) {
          ++P.launches;
          // Small launches carry NO NaN guard: their flag area is epoch-tagged and shared, so there is no per-launch abort word a
          // guard kernel could read.  Their callers (one evaluation of f, an input-gradient chain, the encoder's Euler step) are
          // enqueue-only, so a give-up in one of them is reported LATE: the sticky host word is read by the binding at the next
          // library call after the kernel has run (or by odehip_persistent_error), which raises "an earlier call ..." -- the data
          // that earlier call returned is invalid.  This path is off by default (ODEHIP_PERSISTENT_SMALL=1 enables it).
          return rcs;
        }
        P.enabled = 0;
        (void)hipGetLastError();
      }
      all_ok = false;  // fall through to the replay
    }
    const ConvArgs* table = nullptr;
    if (all_ok) {
      for (int i = 0; i < rec_.count; ++i) {
        ConvArgs& a = rec_.items[i];
        a.dbg = nullptr;
        a.h_by_value = 0;
        if (hbuf) {  // h by value, in a field no fixed-grid layer uses (read instead of *h_ptr)
          const float* hp = a.combine == 1 ? a.cmb.h_ptr : (a.combine >= 2 ? a.bwd.h_ptr : nullptr);
          a.cmb.atol = hp ? hbuf[hp - hdev] : (a.combine == 1 ? 1.0f : 0.0f);
          a.h_by_value = 1;
        }
        if (a.combine == 1 && a.cmb.out2_nchw) {  // result frames as offsets: the output tensor is new every call
          a.dbg = (unsigned long long*)(uintptr_t)((size_t)(a.cmb.out2_nchw - out_nchw) + 1);
          a.cmb.out2_nchw = nullptr;
        }
      }
      table = volatile_ ? persist_table_async(rec_.items, rec_.count, stream) : persist_table(rec_.items, rec_.count, stream);
    }
    int rc;
    // batches up to 16: forward tables of a 64-channel stack take the sixteen-workgroups-per-sample walk (a layer's matrix work is
    // split four times finer; bit-identical results)
    bool small16 = false;
    if (table && !wide && batch <= 16 && walk16_on()) {
      small16 = true;
      for (int i = 0; i < rec_.count && small16; ++i) {
        const ConvArgs& a = rec_.items[i];
        if (a.combine >= 4) continue;   // elementwise / norm rows
        small16 = a.qin == 16 && a.qout == 16 &&
                  (a.combine == 0 || a.combine == 2 || a.combine == 3 ||
                   (a.combine == 1 && (((a.h_by_value || eval_walk_) && a.cmb.n_prev <= 3 && !a.cmb.err_partials && !a.cmb.order) ||
                                       (a.cmb.order == 1 && !(a.cmb.err_partials && (a.cmb.out2 || a.cmb.out2_nchw))))));
        // (a reverse-sweep row that falls to the shared epilogue must not carry relocatable pointers: the adaptive drivers keep to two
        // constant-coefficient targets)
      }
      if (adaptive && !small16) {   // an adaptive table counts on 64 partials per sample here (persist_partials_per_sample)
        set_error("persistent walk: an adaptive table of batch %d has a row the sixteen-workgroup walk does not take", batch);
        return ODEHIP_EINVAL;
      }
    }
    if (table) {
      if (!sync_is_zero) ODEHIP_CHECK_HIP(hipMemsetAsync(sync, 0, persist_sync_bytes(batch), stream));
      if (small16) rc = launch_wino_persist16(table, rec_.count, batch, sync, sync + (size_t)batch * kPersistDoneStride, g_persist.host_err_dev, out_nchw, stream,
                                              rows_dev_, reloc_dev_);
      else
      rc = launch_wino_persist(table, rec_.count, batch, sync, sync + (size_t)batch * kPersistDoneStride, g_persist.host_err_dev, out_nchw,
                               kPersistGrid, stream, wide, adaptive, rows_dev_, reloc_dev_);
      if (volatile_) persist_table_async_done(stream);
      if (rc == ODEHIP_OK) {
        ++g_persist.launches;
        launched_ = true;
        abort_word_ = sync + (size_t)batch * kPersistDoneStride + kPersistGrid;   // xcc_of[grid]: zeroed above, epoch 0 => tag 1
        table_ = table;
        table_rows_ = rec_.count;
        table_wide_ = wide;
        table_adaptive_ = adaptive;
        table_small16_ = small16;
        return rc;
      }
      g_persist.enabled = 0;  // the launch was refused: one launch per layer from now on
      (void)hipGetLastError();
    }
    if (rows_dev_ || reloc_dev_) {
      set_error("the persistent walk refused a device-steered table (%d rows): it cannot be replayed as ordinary launches", rec_.count);
      return ODEHIP_EHIP;
    }
    for (int i = 0; i < rec_.count; ++i) {  // replay the recorded layers as ordinary launches
      ConvArgs a = rec_.items[i];
      if (a.dbg) a.cmb.out2_nchw = out_nchw + ((size_t)(uintptr_t)a.dbg - 1);
      a.dbg = nullptr;
      if (a.h_by_value && a.combine != 1) a.cmb.atol = 0.0f;
      a.h_by_value = 0;
      if ((rc = launch_conv(a, ks, stream)) != ODEHIP_OK) return rc;
    }
    return ODEHIP_OK;
  }

int PersistScope::relaunch(int batch, unsigned* sync, hipStream_t stream, bool sync_is_zero) {
  ODEHIP_REQUIRE(launched_ && table_ && !volatile_, "persistent relaunch without a table");
  if (!sync_is_zero) ODEHIP_CHECK_HIP(hipMemsetAsync(sync, 0, persist_sync_bytes(batch), stream));
  const int rc = table_small16_
                     ? launch_wino_persist16(table_, table_rows_, batch, sync, sync + (size_t)batch * kPersistDoneStride, g_persist.host_err_dev, nullptr,
                                             stream, rows_dev_, reloc_dev_)
                     : launch_wino_persist(table_, table_rows_, batch, sync, sync + (size_t)batch * kPersistDoneStride, g_persist.host_err_dev, nullptr,
                                           kPersistGrid, stream, table_wide_, table_adaptive_, rows_dev_, reloc_dev_);
  if (rc == ODEHIP_OK) ++g_persist.launches;
  return rc;
}

}  // namespace odehip

using namespace odehip;

extern "C" int odehip_set_persistent_trajectory(int enable) {
  std::lock_guard<std::mutex> g(g_persist.mu);
  const int was = g_persist.enabled != 0;
  g_persist.enabled = enable ? (g_persist.host_err ? 1 : -1) : 0;
  return was;
}

extern "C" int odehip_persistent_error(int clear) {
  return (int)persist_error(clear != 0);
}

extern "C" long long odehip_persistent_trajectory_launches(void) {
  std::lock_guard<std::mutex> g(g_persist.mu);
  return g_persist.launches;
}

