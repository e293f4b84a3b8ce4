// Version / error plumbing of libevae_hip.so (see include/evae_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#include "../../include/evae_hip.h"

namespace evae {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace evae

extern "C" int evae_version(void) { return EVAE_ABI_VERSION; }
extern "C" const char* evae_last_error(void) { return evae::g_err; }

// Double-buffered upload of a training step's control block (evae/graph.py::_refresh) as ONE call: the six stream / event / copy
// operations cost ~65 us of host time through the Python bindings and ~10 us here, and the replayed step of a small exemplar set
// is bound by the host.  up: upload stream, step: the stream the graph is launched on; ev_used / ev_up: hipEvent_t handles.
extern "C" int evae_ctl_upload(void* d_stage, const void* h_pinned, void* d_ctl, size_t bytes, evae_stream_t up_, evae_stream_t step_,
                               void* ev_used_, void* ev_up_) {
  hipStream_t up = (hipStream_t)up_, step = (hipStream_t)step_;
  hipEvent_t ev_used = (hipEvent_t)ev_used_, ev_up = (hipEvent_t)ev_up_;
  if (!d_stage || !h_pinned || !ev_used || !ev_up) { evae::set_error("ctl_upload: null argument"); return EVAE_EINVAL; }
  hipError_t e = hipStreamWaitEvent(up, ev_used, 0);                                   // staging block consumed two steps ago
  if (e == hipSuccess) e = hipMemcpyAsync(d_stage, h_pinned, bytes, hipMemcpyHostToDevice, up);
  if (e == hipSuccess) e = hipEventRecord(ev_up, up);                                   // host block reusable once this ran
  if (e == hipSuccess) e = hipStreamWaitEvent(step, ev_up, 0);
  if (d_ctl) {
    if (e == hipSuccess) e = hipMemcpyAsync(d_ctl, d_stage, bytes, hipMemcpyDeviceToDevice, step);
    if (e == hipSuccess) e = hipEventRecord(ev_used, step);
  }
  if (e != hipSuccess) { evae::set_error("ctl_upload: %s", hipGetErrorString(e)); return EVAE_ELAUNCH; }
  return EVAE_OK;
}

// Duplicates among the exemplar draw of a step (reference models/BaseModel.py:245: torch.randint WITH replacement -- 25 000 draws
// from 50 000 rows name ~19 700 distinct images): the distinct rows in first-occurrence order, every draw's position among them,
// one draw per distinct row and the multiplicities.  Host-side, one pass, O(draws) with a stamp table of n_rows words kept
// between calls (thread-local).  rows / rep / mult have `cap` entries: the tail behind the U distinct rows is padded with
// (rows[0], 0, 0.0f).  Returns U, or -1 when U > cap (nothing usable was written) or an index is out of range.
extern "C" int evae_host_dedup(const int64_t* draws, int n_draws, int64_t n_rows, int cap, int64_t* rows, int64_t* inv, int64_t* rep,
                               float* mult) {
  if (!draws || !rows || !inv || !rep || !mult || n_draws <= 0 || n_rows <= 0 || cap <= 0) { evae::set_error("host_dedup: bad arguments"); return -1; }
  // one table entry per dataset row: (generation of the call that last saw it) << 32 | its position among the distinct rows
  static thread_local std::vector<uint64_t> tab;
  static thread_local uint32_t gen = 0;
  if ((int64_t)tab.size() < n_rows) { tab.assign((size_t)n_rows, 0ull); gen = 0; }
  if (++gen == 0) { std::fill(tab.begin(), tab.end(), 0ull); gen = 1; }
  const uint64_t tag = (uint64_t)gen << 32;
  int U = 0;
  for (int j = 0; j < n_draws; ++j) {
    const int64_t i = draws[j];
    if (i < 0 || i >= n_rows) { evae::set_error("host_dedup: index %lld outside [0, %lld)", (long long)i, (long long)n_rows); return -1; }
    const uint64_t e = tab[(size_t)i];
    if ((e >> 32) != gen) {
      if (U == cap) { evae::set_error("host_dedup: more than %d distinct rows among %d draws", cap, n_draws); return -1; }
      tab[(size_t)i] = tag | (uint32_t)U;
      rows[U] = i; rep[U] = j; mult[U] = 1.0f; inv[j] = U; ++U;
    } else {
      const uint32_t u = (uint32_t)e;
      inv[j] = u; mult[u] += 1.0f;
    }
  }
  for (int u = U; u < cap; ++u) { rows[u] = rows[0]; rep[u] = 0; mult[u] = 0.0f; }
  return U;
}
