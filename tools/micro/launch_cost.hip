// Host cost of issuing the 33 small launches of a thin training step three ways: eager hipLaunchKernel on two streams with the
// step's ~12 cross-stream event pairs, and hipGraphLaunch of the same launches captured.  hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Args { float* p; int n; float a, b, c, d; long s0, s1, s2, s3; void* q0; void* q1; void* q2; void* q3; };   // ~100 bytes, like the step's
__global__ void small_kernel(Args a) { if (threadIdx.x == 0 && blockIdx.x == 0 && a.n < 0) a.p[0] = a.a; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void issue(hipStream_t s0, hipStream_t s1, std::vector<hipEvent_t>& ev, const Args& a, int nodes, int joins) {
  // nodes launches alternating in runs between the two streams; `joins` event record / wait pairs spread over them
  int e = 0;
  const int every = joins ? nodes / joins : nodes + 1;
  for (int i = 0; i < nodes; ++i) {
    hipStream_t s = ((i / 4) & 1) ? s1 : s0;
    if (joins && i % every == 0 && e < joins) {
      hipStream_t o = (s == s0) ? s1 : s0;
      hipEventRecord(ev[e], o);
      hipStreamWaitEvent(s, ev[e], 0);
      ++e;
    }
    small_kernel<<<64, 256, 0, s>>>(a);
  }
  hipEventRecord(ev[joins], s1);
  hipStreamWaitEvent(s0, ev[joins], 0);
}

int main(int argc, char** argv) {
  const int nodes = argc > 1 ? atoi(argv[1]) : 33, joins = argc > 2 ? atoi(argv[2]) : 12, reps = 300;
  float* p; CK(hipMalloc(&p, 4096));
  Args a = {p, 1, 1.f, 2.f, 3.f, 4.f, 1, 2, 3, 4, p, p, p, p};
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(joins + 1);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int w = 0; w < 20; ++w) issue(s0, s1, ev, a, nodes, joins);
  CK(hipDeviceSynchronize());
  double t = 0;
  for (int r = 0; r < reps; ++r) {
    const double t0 = now();
    issue(s0, s1, ev, a, nodes, joins);
    t += now() - t0;
    if (r % 8 == 7) CK(hipStreamSynchronize(s0));
  }
  CK(hipDeviceSynchronize());
  printf("eager, two streams, %d launches + %d event pairs: %.1f us per step (%.2f us per launch)\n", nodes, joins + 1, 1e6 * t / reps, 1e6 * t / reps / nodes);
  // one stream, no events
  t = 0;
  for (int r = 0; r < reps; ++r) {
    const double t0 = now();
    for (int i = 0; i < nodes; ++i) small_kernel<<<64, 256, 0, s0>>>(a);
    t += now() - t0;
    if (r % 8 == 7) CK(hipStreamSynchronize(s0));
  }
  CK(hipDeviceSynchronize());
  printf("eager, one stream, %d launches: %.1f us per step (%.2f us per launch)\n", nodes, 1e6 * t / reps, 1e6 * t / reps / nodes);
  // the same launches captured
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
  issue(s0, s1, ev, a, nodes, joins);
  CK(hipStreamEndCapture(s0, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(ge, s0));
  CK(hipDeviceSynchronize());
  t = 0;
  for (int r = 0; r < reps; ++r) {
    const double t0 = now();
    hipGraphLaunch(ge, s0);
    t += now() - t0;
    if (r % 8 == 7) CK(hipStreamSynchronize(s0));
  }
  CK(hipDeviceSynchronize());
  printf("hipGraphLaunch of the captured two-stream step: %.1f us per step (%.2f us per node)\n", 1e6 * t / reps, 1e6 * t / reps / nodes);
  // wall clock per step when issued back to back (device included)
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipDeviceSynchronize());
    const double t0 = now();
    for (int r = 0; r < reps; ++r) { if (mode) hipGraphLaunch(ge, s0); else issue(s0, s1, ev, a, nodes, joins); }
    CK(hipDeviceSynchronize());
    printf("%s: %.1f us per step wall clock, %d steps back to back\n", mode ? "graph" : "eager", 1e6 * (now() - t0) / reps, reps);
  }
  return 0;
}
