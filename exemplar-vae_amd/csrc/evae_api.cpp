// Version / error plumbing of libevae_hip.so (see include/evae_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

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
  if (!d_stage || !h_pinned || !d_ctl || !ev_used || !ev_up) { evae::set_error("ctl_upload: null argument"); return EVAE_EINVAL; }
  hipError_t e = hipStreamWaitEvent(up, ev_used, 0);                                   // staging block consumed two steps ago
  if (e == hipSuccess) e = hipMemcpyAsync(d_stage, h_pinned, bytes, hipMemcpyHostToDevice, up);
  if (e == hipSuccess) e = hipEventRecord(ev_up, up);                                   // host block reusable once this ran
  if (e == hipSuccess) e = hipStreamWaitEvent(step, ev_up, 0);
  if (e == hipSuccess) e = hipMemcpyAsync(d_ctl, d_stage, bytes, hipMemcpyDeviceToDevice, step);
  if (e == hipSuccess) e = hipEventRecord(ev_used, step);
  if (e != hipSuccess) { evae::set_error("ctl_upload: %s", hipGetErrorString(e)); return EVAE_ELAUNCH; }
  return EVAE_OK;
}
