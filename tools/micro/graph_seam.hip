// How long does the device idle between two replays of a GPU-bound two-branch graph, and inside it at each fork / join?
// main chain: NM spins of TM us; side chain: NS spins of TS us forked after the first main spin, `joins` times the main chain waits for
// the side chain's progress (and the side chain for main's, alternating), one join at the end.  Wall clock per replay against the
// main chain's own time (NM + 1) x TM.      hipcc --offload-arch=gfx950 -O2 ;  ./graph_seam [joins] [NM] [TM] [NS] [TS]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(long long ticks, int* sink) {      // wall_clock64: 100 MHz
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && ticks < 0) sink[0] = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int joins = argc > 1 ? atoi(argv[1]) : 4, NM = argc > 2 ? atoi(argv[2]) : 12, TM = argc > 3 ? atoi(argv[3]) : 30;
  const int NS = argc > 4 ? atoi(argv[4]) : 20, TS = argc > 5 ? atoi(argv[5]) : 10, reps = 200;
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(2 * joins + 4);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
  int e = 0;
  spin<<<1, 64, 0, s0>>>(TM * 100, nullptr);
  if (NS > 0) { hipEventRecord(ev[e], s0); hipStreamWaitEvent(s1, ev[e], 0); ++e; }        // fork
  int si = 0;
  const int every = NM / (joins + 1) > 0 ? NM / (joins + 1) : 1;
  for (int i = 1; i < NM; ++i) {
    const int upto = (int)((long long)NS * i / NM);     // the side chain's share up to this point of the main chain
    for (; si < upto; ++si) spin<<<1, 64, 0, s1>>>(TS * 100, nullptr);
    if (NS > 0 && joins && i % every == 0 && e <= 2 * joins) {
      if ((i / every) & 1) { hipEventRecord(ev[e], s1); hipStreamWaitEvent(s0, ev[e], 0); }   // main waits for side
      else { hipEventRecord(ev[e], s0); hipStreamWaitEvent(s1, ev[e], 0); }                    // side waits for main
      ++e;
    }
    spin<<<1, 64, 0, s0>>>(TM * 100, nullptr);
  }
  for (; si < NS; ++si) spin<<<1, 64, 0, s1>>>(TS * 100, nullptr);
  if (NS > 0) { hipEventRecord(ev[e], s1); hipStreamWaitEvent(s0, ev[e], 0); }               // final join
  spin<<<1, 64, 0, s0>>>(TM * 100, nullptr);                                                    // the tail (the optimizer)
  CK(hipStreamEndCapture(s0, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 10; ++w) CK(hipGraphLaunch(ge, s0));
  CK(hipDeviceSynchronize());
  const double t0 = now();
  double host = 0;
  for (int r = 0; r < reps; ++r) { const double a = now(); hipGraphLaunch(ge, s0); host += now() - a; }
  CK(hipDeviceSynchronize());
  const double wall = 1e6 * (now() - t0) / reps, chain = (double)(NM + 1) * TM;
  printf("joins %d: main chain %d x %d us (+ tail) = %.0f us, side chain %d x %d us; wall clock per replay %.1f us (host %.1f us per launch): %.1f us idle per replay\n",
         joins, NM, TM, chain, NS, TS, wall, 1e6 * host / reps, wall - chain);
  return 0;
}
