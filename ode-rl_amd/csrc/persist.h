// persist.h -- host side of the persistent trajectory launch (conv_wino.hip: wino_persist_kernel), shared by the drivers.
#pragma once
#include <mutex>
#include <vector>

#include "odehip_internal.h"

namespace odehip {

constexpr int kPersistDoneStride = 64;  // words between per-sample flag lines (= kDoneStride in conv_wino.hip)
constexpr int kPersistGrid = 256;       // the persistent kernel holds every CU of an MI355X (one workgroup each)
inline size_t persist_sync_bytes(int batch) { return ((size_t)batch * kPersistDoneStride + kPersistGrid + 64) * 4; }

// Records the conv launches of a driver between begin() and finish(); finish() runs them as one persistent launch, or replays
// them one by one when the persistent path is unavailable.  Nothing but launch_conv calls may be enqueued in between.
class PersistScope {
 public:
  PersistScope();
  ~PersistScope();
  // rc != OK: a sticky error of an earlier launch was found.  active(): the recorder is on.
  // small: a sequence of at most 5 layers that is launched with its table in the kernel arguments (no cache, library-owned
  // flags); such a scope stays inactive inside an outer scope, whose recorder then simply sees the layers
  int begin(const odehip_convstack* f, const odehip_convstack* f2, int max_layers, bool small = false);
  bool active() const { return active_; }
  // volatile_table: the recorded table differs from call to call (buffers / coefficients that follow the accepted steps of an
  // adaptive solve): it is uploaded asynchronously through a ring instead of entering the content cache (whose misses cost a
  // stream synchronisation).  Call between begin() and finish().
  void set_volatile_table(bool v) { volatile_ = v; }
  // The recorded rows are a TEMPLATE a device-side controller steers: rows_dev = {first row, number of rows} of the section the
  // launch walks, reloc_dev = the base addresses of the relocatable pointers in the rows (class << 56 | offset; conv_wino.hip,
  // rel()).  Such a table only runs on the adaptive walk (callers check recording() first and take their host-driven path
  // otherwise): finish() fails with ODEHIP_EHIP rather than replaying it.
  void set_device_steering(const int* rows_dev, const unsigned long long* reloc_dev) {
    rows_dev_ = rows_dev;
    reloc_dev_ = reloc_dev;
    adaptive_ = true;
  }
  // true if begin() found the persistent path usable for this stack (the recorder is on)
  bool recording() const { return active_; }
  // hbuf / hdev: host copy and device array of the step sizes (fixed grids: the table gets h by value), or null / null when the
  // step size only exists on the device (dopri5); out_nchw may be null; sync: persist_sync_bytes(batch) of workspace
  // sync_is_zero: the caller has already zeroed the flag area on this stream (traj_prologue)
  int finish(const float* hbuf, const float* hdev, float* out_nchw, int batch, unsigned* sync, int ks, hipStream_t stream,
             bool sync_is_zero = false);
  // Enqueued behind a call's last kernel when finish() took the persistent path: if a capped wait of a persistent launch has
  // given up (the mapped error word is set; never expected) the regions are filled with NaN, so the call that produced the
  // invalid result cannot hand plausible-looking numbers to its caller.  No-op when nothing was launched persistently.
  int guard(float* const* regions, const size_t* floats, int n, hipStream_t stream);
  bool launched() const { return launched_; }
  // Launch the table of the last finish() AGAIN (a device-steered table is walked once per tick of its controller).  Only after a
  // finish() that took the persistent path; sync as for finish().
  int relaunch(int batch, unsigned* sync, hipStream_t stream, bool sync_is_zero);

 private:
  std::unique_lock<std::mutex> lock_;
  std::vector<ConvArgs> items_;
  ConvRecorder rec_ = {nullptr, 0, 0};
  bool active_ = false;
  bool small_ = false;
  bool eval_walk_small_only_ = false;   // ODEHIP_EVAL_WALK unset: only batches the sixteen-workgroup walk takes
  bool eval_walk_ = false;   // a small scope that runs on the trajectory walks (ODEHIP_EVAL_WALK=1): library-owned flags, volatile table
  bool adaptive_ = false;
  bool volatile_ = false;
  const int* rows_dev_ = nullptr;
  const unsigned long long* reloc_dev_ = nullptr;
  bool launched_ = false;
  const unsigned* abort_word_ = nullptr;
  const ConvArgs* table_ = nullptr;   // device table of the last persistent launch
  int table_rows_ = 0;
  bool table_wide_ = false, table_adaptive_ = false, table_small16_ = false;
};

// Whole-trajectory launches that need no cross-workgroup hand-off (bf16: one workgroup per sample) obey the same on/off switch
// (odehip_set_persistent_trajectory / ODEHIP_PERSISTENT=0) and are counted by odehip_persistent_trajectory_launches().
bool persist_switch_on();
void persist_count_launch();

// asynchronous host -> device copy through a ring of pinned staging slots (no stream synchronisation, no pageable memory)
int staged_upload(void* dst_dev, const void* src, size_t bytes, hipStream_t stream);

// host-side look at the sticky error word of the persistent launches (0 = none); clear != 0 resets it and disables the path
unsigned persist_error(bool clear);

}  // namespace odehip
