// Stand-alone check + timing of the pre-split bf16 GEMM (csrc/evae_gemm_p6.h) against the in-kernel-split one (evae_gemm_x6.h)
// at the c2 step's shapes.  Build (tools/micro/build.sh):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -I exemplar-vae_amd/csrc tools/micro/p6_bench.hip -o tools/micro/p6_bench
// Run on the GPU box: tools/micro/p6_bench [reps]
#include "evae_gemm_p6.h"
#include "evae_gemm_x6.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>

namespace evae {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int g_x6_enabled = 1, g_x6_min_rows = 0;
void gemm_x6_init_policy() {}
}
using namespace evae;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static std::vector<float> rnd(size_t n, unsigned seed, float scale = 1.f) {
  std::mt19937 g(seed); std::normal_distribution<float> d(0.f, scale);
  std::vector<float> v(n); for (auto& x : v) x = d(g); return v;
}
template <class T> static T* dev(const std::vector<T>& h) { T* p; CK(hipMalloc(&p, h.size() * sizeof(T))); CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }
template <class T> static T* devz(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }

template <class F> static float time_us(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.f / reps;
}

static GemmArgs blank() { GemmArgs g; std::memset(&g, 0, sizeof(g)); g.npairs = 1; g.ones_col = -1; return g; }

// C [M x N] = A [M x K] B[N x K]^T: p6 over A's image, p6 over A^T's image (transpose reads), x6; checked on sampled rows against fp64
static void case_nt(const char* name, int M, int N, int K, int reps) {
  const int Kp = (K + 15) / 16 * 16, nks = p6_nks(Kp), nms = p6_nks_rows(M);
  auto hA = rnd((size_t)M * K, 1), hB = rnd((size_t)N * K, 2, 0.05f);
  float* dA = dev(hA); float* dB = dev(hB);
  const int Mi = (M + 127) / 128 * 128, Ni = (N + 127) / 128 * 128, Ci = (Kp + 127) / 128 * 128;
  unsigned char* iA = devz<unsigned char>(p6_image_bytes(M, nks)); unsigned char* iB = devz<unsigned char>(p6_image_bytes(N, nks));
  unsigned char* iAT = devz<unsigned char>(p6_image_bytes(Kp, nms));
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Mi * nks * 2 + 255) / 256), 256>>>(dA, nullptr, M, K, K, 0, Mi, nks, iA);
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Ni * nks * 2 + 255) / 256), 256>>>(dB, nullptr, N, K, K, 0, Ni, nks, iB);
  p6_pack_cols_kernel<<<(unsigned)(((size_t)Ci * nms * 2 + 255) / 256), 256>>>(dA, nullptr, M, K, K, -1, Ci, nms, iAT);
  CK(hipDeviceSynchronize());
  float* dC[5]; for (auto& p : dC) p = devz<float>((size_t)M * N);
  GemmArgs g = blank();
  g.A[0] = (const float*)iA; g.B[0] = (const float*)iB; g.Kc[0] = Kp; g.M = M; g.N = N; g.ldo = N;
  GemmArgs g0 = g; g0.out0 = dC[0];
  GemmArgs g1 = g; g1.out0 = dC[1];
  GemmArgs g2 = g; g2.out0 = dC[2]; g2.A[0] = (const float*)iAT; g2.lda[0] = nms;
  GemmArgs g3 = g2; g3.out0 = dC[3];
  GemmArgs gx = blank();
  gx.A[0] = dA; gx.B[0] = dB; gx.lda[0] = K; gx.ldb[0] = K; gx.Kc[0] = K; gx.M = M; gx.N = N; gx.out0 = dC[4]; gx.ldo = N;
  const float t0 = time_us([&] { launch_gemm_p6<EPI_LINEAR, 128>(g0, 1, 0, "p6"); }, reps);
  const float t1 = time_us([&] { launch_gemm_p6<EPI_LINEAR, 64>(g1, 1, 0, "p6/64"); }, reps);
  const float t2 = time_us([&] { launch_gemm_p6<EPI_LINEAR, 128, true>(g2, 1, 0, "p6 TA"); }, reps);
  const float t3 = time_us([&] { launch_gemm_p6<EPI_LINEAR, 64, true>(g3, 1, 0, "p6 TA/64"); }, reps);
  GemmArgs gne = g2; gne.dbg = 4;
  const float tne = time_us([&] { launch_gemm_p6<EPI_LINEAR, 128, true>(gne, 1, 0, "p6 TA no epilogue"); }, reps);
  launch_gemm_p6<EPI_LINEAR, 128, true>(g2, 1, 0, "p6 TA");
  const float tx = time_us([&] { launch_gemm_x6<EPI_LINEAR, 0, 128>(gx, 1, 0, "x6"); }, reps);
  const float tx64 = time_us([&] { launch_gemm_x6<EPI_LINEAR, 0, 64>(gx, 1, 0, "x6/64"); }, reps);
  launch_gemm_x6<EPI_LINEAR, 0, 128>(gx, 1, 0, "x6");
  CK(hipDeviceSynchronize());
  std::vector<float> hC[5];
  for (int v = 0; v < 5; ++v) { hC[v].resize((size_t)M * N); CK(hipMemcpy(hC[v].data(), dC[v], hC[v].size() * 4, hipMemcpyDeviceToHost)); }
  double e[5] = {0, 0, 0, 0, 0}, ref_max = 0;
  for (int s = 0; s < 96; ++s) {
    const int m = (int)(((long long)s * 2654435761ll) % M);
    const int mm = s < 4 ? (s & 1 ? M - 1 - s : s) : m;
    for (int n = 0; n < N; ++n) {
      double r = 0; for (int k = 0; k < K; ++k) r += (double)hA[(size_t)mm * K + k] * hB[(size_t)n * K + k];
      ref_max = fmax(ref_max, fabs(r));
      for (int v = 0; v < 5; ++v) e[v] = fmax(e[v], fabs(hC[v][(size_t)mm * N + n] - r));
    }
  }
  const double gf = 2.0 * M * N * K * 1e-9;
  printf("%-8s M=%d N=%d K=%d  p6 %.1f us %.0f TF | p6/64 %.1f | p6-T %.1f us %.0f TF | p6-T/64 %.1f | p6-T no epilogue %.1f | x6 %.1f us %.0f TF | x6/64 %.1f | err/max: %.2e %.2e T %.2e %.2e x6 %.2e\n",
         name, M, N, K, t0, gf / t0 * 1e3, t1, t2, gf / t2 * 1e3, t3, tne, tx, gf / tx * 1e3, tx64, e[0] / ref_max, e[1] / ref_max, e[2] / ref_max, e[3] / ref_max, e[4] / ref_max);
  hipFree(dA); hipFree(dB); hipFree(iA); hipFree(iB); hipFree(iAT); for (auto p : dC) hipFree(p);
}

// gated forward: out = (A Wh^T + bh) * sigmoid(A Wg^T + bg)
static void case_gated(int M, int N, int K, int reps) {
  const int Kp = (K + 15) / 16 * 16, nks = p6_nks(Kp), nms = p6_nks_rows(M);
  auto hA = rnd((size_t)M * K, 3), hWh = rnd((size_t)N * K, 4, 0.05f), hWg = rnd((size_t)N * K, 5, 0.05f), hbh = rnd(N, 6), hbg = rnd(N, 7);
  float* dA = dev(hA); float* dWh = dev(hWh); float* dWg = dev(hWg); float* dbh = dev(hbh); float* dbg = dev(hbg);
  const int Mi = (M + 127) / 128 * 128, tiles_n = (N + 63) / 64, Ni = tiles_n * 128, Ci = (Kp + 127) / 128 * 128;
  unsigned char* iA = devz<unsigned char>(p6_image_bytes(M, nks)); unsigned char* iB = devz<unsigned char>(p6_image_bytes(Ni, nks));
  unsigned char* iAT = devz<unsigned char>(p6_image_bytes(Kp, nms));
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Mi * nks * 2 + 255) / 256), 256>>>(dA, nullptr, M, K, K, 0, Mi, nks, iA);
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Ni * nks * 2 + 255) / 256), 256>>>(dWh, dWg, N, K, K, 1, Ni, nks, iB);
  p6_pack_cols_kernel<<<(unsigned)(((size_t)Ci * nms * 2 + 255) / 256), 256>>>(dA, nullptr, M, K, K, -1, Ci, nms, iAT);
  float* dO[3]; float* dS[3]; for (int v = 0; v < 3; ++v) { dO[v] = devz<float>((size_t)M * N); dS[v] = devz<float>((size_t)M * N); }
  GemmArgs g = blank();
  g.A[0] = (const float*)iA; g.B[0] = (const float*)iB; g.Kc[0] = Kp; g.M = M; g.N = N; g.out0 = dO[0]; g.out2 = dS[0]; g.ldo = N; g.bias0 = dbh; g.bias1 = dbg;
  GemmArgs gt = g; gt.A[0] = (const float*)iAT; gt.lda[0] = nms; gt.out0 = dO[1]; gt.out2 = dS[1];
  GemmArgs gx = blank();
  gx.A[0] = dA; gx.B[0] = dWh; gx.Bg = dWg; gx.lda[0] = K; gx.ldb[0] = K; gx.Kc[0] = K; gx.M = M; gx.N = N; gx.out0 = dO[2]; gx.out2 = dS[2]; gx.ldo = N;
  gx.bias0 = dbh; gx.bias1 = dbg;
  const float t_p = time_us([&] { launch_gemm_p6<EPI_GATED, 128>(g, 1, 0, "p6 gated"); }, reps);
  const float t_t = time_us([&] { launch_gemm_p6<EPI_GATED, 128, true>(gt, 1, 0, "p6 TA gated"); }, reps);
  launch_gemm_p6<EPI_GATED, 128, true>(gt, 1, 0, "p6 TA gated");
  const float t_x = time_us([&] { launch_gemm_x6<EPI_GATED, 0, 128>(gx, 1, 0, "x6 gated"); }, reps);
  std::vector<float> hO[3];
  for (int v = 0; v < 3; ++v) { hO[v].resize((size_t)M * N); CK(hipMemcpy(hO[v].data(), dO[v], hO[v].size() * 4, hipMemcpyDeviceToHost)); }
  double e[3] = {0, 0, 0}, ref_max = 0;
  for (int s = 0; s < 64; ++s) {
    const int m = s < 2 ? (s ? M - 1 : 0) : (int)(((long long)s * 2654435761ll) % M);
    for (int n = 0; n < N; ++n) {
      double h = hbh[n], gg = hbg[n];
      for (int k = 0; k < K; ++k) { h += (double)hA[(size_t)m * K + k] * hWh[(size_t)n * K + k]; gg += (double)hA[(size_t)m * K + k] * hWg[(size_t)n * K + k]; }
      const double r = h / (1.0 + exp(-gg));
      ref_max = fmax(ref_max, fabs(r));
      for (int v = 0; v < 3; ++v) e[v] = fmax(e[v], fabs(hO[v][(size_t)m * N + n] - r));
    }
  }
  const double gf = 4.0 * M * N * K * 1e-9;
  printf("gated    M=%d N=%d K=%d  p6 %.1f us %.0f TF | p6-T %.1f us %.0f TF | x6 %.1f us %.0f TF | err/max: p6 %.2e p6-T %.2e x6 %.2e\n", M, N, K, t_p, gf / t_p * 1e3,
         t_t, gf / t_t * 1e3, t_x, gf / t_x * 1e3, e[0] / ref_max, e[1] / ref_max, e[2] / ref_max);
}

// weight gradient dW [N x K] = dy[M x N]^T x[M x K] (+ ones column), split-K over the M rows
static void case_wgrad(int M, int N, int K, int reps) {
  auto hdy = rnd((size_t)M * N, 8, 0.01f), hx = rnd((size_t)M * K, 9);
  float* ddy = dev(hdy); float* dx = dev(hx);
  const int Kc = K + 1, nms = p6_nks_rows(M), Mp = nms * 16;          // output columns: K + the ones column
  const int Ni = (N + 127) / 128 * 128, Ki = (Kc + 127) / 128 * 128;
  unsigned char* iA = devz<unsigned char>(p6_image_bytes(N, nms)); unsigned char* iB = devz<unsigned char>(p6_image_bytes(Kc, nms));
  const float t_pa = time_us([&] { p6_pack_cols_kernel<<<(unsigned)(((size_t)Ni * nms * 2 + 255) / 256), 256>>>(ddy, nullptr, M, N, N, -1, Ni, nms, iA); }, 5);
  const float t_pb = time_us([&] { p6_pack_cols_kernel<<<(unsigned)(((size_t)Ki * nms * 2 + 255) / 256), 256>>>(dx, nullptr, M, K, K, K, Ki, nms, iB); }, 5);
  for (int target : {256, 512}) {
    for (int bn : {128, 64}) {
      const int tiles = cdiv(N, 128) * cdiv(Kc, bn);
      int nz = std::max(1, std::min(target / tiles, nms / 8));
      const int ksplit = cdiv(nms, nz); nz = cdiv(nms, ksplit);
      float* part = devz<float>((size_t)nz * N * Kc);
      float* dW = devz<float>((size_t)N * K); float* db = devz<float>(N);
      GemmArgs g = blank();
      g.A[0] = (const float*)iA; g.B[0] = (const float*)iB; g.Kc[0] = Mp; g.M = N; g.N = Kc; g.out0 = part; g.ldo = Kc; g.ksplit = ksplit;
      FinishArgs f; std::memset(&f, 0, sizeof(f));
      f.part = part; f.nz = nz; f.M = N; f.N = Kc; f.ldo = Kc; f.epi = EPI_RAW; f.out0 = dW; f.ones_col = K; f.out_db = db;
      const float t = time_us([&] { if (bn == 128) launch_gemm_p6<EPI_RAW, 128>(g, nz, 0, "p6 wgrad"); else launch_gemm_p6<EPI_RAW, 64>(g, nz, 0, "p6 wgrad"); }, reps);
      const float tf = time_us([&] { launch_finish(f, 0); }, reps);
      std::vector<float> hW((size_t)N * K), hb(N);
      CK(hipMemcpy(hW.data(), dW, hW.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), db, hb.size() * 4, hipMemcpyDeviceToHost));
      double e = 0, rm = 0, eb = 0;
      for (int s = 0; s < 24; ++s) {
        const int n = s < 2 ? (s ? N - 1 : 0) : (int)(((long long)s * 2654435761ll) % N);
        for (int k = 0; k < K; ++k) {
          double r = 0; for (int m = 0; m < M; ++m) r += (double)hdy[(size_t)m * N + n] * hx[(size_t)m * K + k];
          rm = fmax(rm, fabs(r)); e = fmax(e, fabs(hW[(size_t)n * K + k] - r));
        }
        double rb = 0; for (int m = 0; m < M; ++m) rb += hdy[(size_t)m * N + n];
        eb = fmax(eb, fabs(hb[n] - rb));
      }
      const double gf = 2.0 * M * N * Kc * 1e-9;
      printf("wgrad    M=%d N=%d K=%d  bn %d nz %d (%d blocks)  p6 %.1f us %.0f TF + finish %.1f us | err/max %.2e db err %.2e | pack dy^T %.1f us x^T %.1f us\n",
             M, N, K, bn, nz, tiles * nz, t, gf / t * 1e3, tf, e / rm, eb, t_pa, t_pb);
      hipFree(part); hipFree(dW); hipFree(db);
    }
  }
  // the x6t kernel (in-kernel split) the step could run today
  {
    GemmArgs g = blank();
    const int Kp = (K + 4) / 4 * 4;
    g.A[0] = ddy; g.B[0] = dx; g.lda[0] = N; g.ldb[0] = K; g.Kc[0] = M; g.M = N; g.N = Kp; g.ones_col = K;
    const X6tSplit sp = x6t_split(M, N, Kp);
    float* part = devz<float>((size_t)sp.nz * N * Kp);
    g.out0 = part; g.ldo = Kp; g.ksplit = sp.ksplit;
    if (gemm_x6t_ok(g)) {
      const float t = time_us([&] { launch_gemm_x6t<EPI_RAW>(g, sp.nz, 0, "x6t"); }, reps);
      printf("wgrad    x6t (in-kernel split) nz %d: %.1f us %.0f TF\n", sp.nz, t, 2.0 * M * N * Kp * 1e-9 / t * 1e3);
    } else printf("wgrad    x6t not applicable\n");
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const char* only = argc > 2 ? argv[2] : "";
  if (!*only || !strcmp(only, "nt")) {
    case_nt("small", 300, 200, 48, 2);
    case_nt("ragged", 1000, 333, 100, 2);
    case_nt("fwd2", 25100, 600, 300, reps);
    case_nt("dgrad2", 25100, 300, 600, reps);
  }
  if (!*only || !strcmp(only, "gated")) { case_gated(500, 70, 52, 2); case_gated(25100, 300, 300, reps); }
  if (!*only || !strcmp(only, "wgrad")) { case_wgrad(777, 130, 90, 2); case_wgrad(25100, 600, 300, reps); }
  return 0;
}
