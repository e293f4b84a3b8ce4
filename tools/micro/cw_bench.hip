// Stand-alone check + timing of the window convolution kernels (csrc/evae_conv_win.h) at convhvae_2level's layer shapes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -I exemplar-vae_amd/csrc tools/micro/cw_bench.hip -o tools/micro/cw_bench
// Run on the GPU box: tools/micro/cw_bench [images] [reps]
#include "evae_conv_win.h"
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
static float bf16f(unsigned short u) { unsigned v = (unsigned)u << 16; float f; memcpy(&f, &v, 4); return f; }
// element (r, k) of a p6 image on the host: the sum of its three planes
static double img_at(const std::vector<unsigned char>& img, int r, int k, int nks) {
  const size_t o = p6_off(r, k, nks);
  double s = 0;
  for (int p = 0; p < 3; ++p) { unsigned short u; memcpy(&u, &img[o + p * P6_CHUNK], 2); s += bf16f(u); }
  return s;
}

// gated forward, stride 1, 'same' padding
static void case_fwd(int N, int C, int H, int Co, int K, int reps) {
  const int W = H, pad = (K - 1) / 2, taps = K * K, M = N * H * W;
  typedef CwGeom<2, 2, 320> G;
  auto hx = rnd((size_t)M * C, 1), hwh = rnd((size_t)Co * C * taps, 2, 0.05f), hwg = rnd((size_t)Co * C * taps, 3, 0.05f), hbh = rnd(Co, 4, 0.3f), hbg = rnd(Co, 5, 0.3f);
  float* dx = dev(hx); float* dwh = dev(hwh); float* dwg = dev(hwg); float* dbh = dev(hbh); float* dbg = dev(hbg);
  const int nks_in = C / 16, ncg = C / 16, nks_w = ncg * taps, tiles_n = (Co + 63) / 64, nks_o = Co / 16;
  const int Mi = (M + 127) / 128 * 128;
  unsigned char* ix = devz<unsigned char>(p6_image_bytes(M, nks_in));
  unsigned char* iw = devz<unsigned char>(p6_image_bytes(tiles_n * 128, nks_w) + 4096);
  unsigned char* io = devz<unsigned char>(p6_image_bytes(M, nks_o));
  float* ds = devz<float>((size_t)M * Co); float* dout = devz<float>((size_t)M * Co);
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Mi * nks_in * 2 + 255) / 256), 256>>>(dx, nullptr, M, C, C, 0, Mi, nks_in, ix);
  cw_pack_filter_kernel<<<(unsigned)(((size_t)tiles_n * 128 * nks_w * 2 + 255) / 256), 256>>>(dwh, dwg, Co, C, taps, 0, 128, tiles_n * 128, nks_w, iw);
  CK(hipDeviceSynchronize());
  const int slots = cw_window_slots(H, W, K, K, pad, G::R);
  ConvWinArgs g; memset(&g, 0, sizeof(g));
  g.xin = ix; g.nks_in = nks_in; g.ncg = ncg; g.cg0 = 0; g.N = N; g.H = H; g.W = W; g.KH = K; g.KW = K; g.pad = pad;
  g.wimg = iw; g.nks_w = nks_w; g.Co = Co; g.tiles_n = tiles_n; g.bias0 = dbh; g.bias1 = dbg;
  g.oimg = io; g.nks_o = nks_o; g.och0 = 0; g.out_s = ds; g.out_f = dout; g.ldo = Co;
  if (slots > 320) { printf("fwd C=%d H=%d Co=%d k=%d: window of %d slots does not fit\n", C, H, Co, K, slots); return; }
  auto run = [&](ConvWinArgs& a) { launch_conv_win<CW_FWD_GATED, 2, 2, 320>(a, 0, "cw fwd"); };
  run(g);
  CK(hipDeviceSynchronize());
  const float t = time_us([&] { run(g); }, reps);
  ConvWinArgs g2 = g; g2.out_f = nullptr;
  const float t2 = time_us([&] { run(g2); }, reps);
  ConvWinArgs gn = g; gn.dbg = 4;
  const float tn = time_us([&] { run(gn); }, reps);
  run(g);
  CK(hipDeviceSynchronize());
  std::vector<float> ho((size_t)M * Co), hs((size_t)M * Co);
  std::vector<unsigned char> himg(p6_image_bytes(M, nks_o));
  CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs.data(), ds, hs.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(himg.data(), io, himg.size(), hipMemcpyDeviceToHost));
  double eo = 0, es = 0, ei = 0, rmax = 0;
  for (int sidx = 0; sidx < 160; ++sidx) {
    int m = (int)(((long long)sidx * 2654435761ll) % M);
    if (sidx < 8) m = (sidx & 1) ? M - 1 - sidx : sidx * 13;              // corners / first and last images
    if (sidx >= 8 && sidx < 24) m = (sidx - 8) * (H * W / 16) + (sidx & 1 ? H * W * (N - 1) : 0);
    if (m >= M) m = M - 1;
    const int n = m / (H * W), y = (m / W) % H, x = m % W;
    for (int co = 0; co < Co; ++co) {
      double h = hbh[co], gg = hbg[co];
      for (int kh = 0; kh < K; ++kh)
        for (int kw = 0; kw < K; ++kw) {
          const int yy = y + kh - pad, xx = x + kw - pad;
          if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
          const float* px = &hx[((size_t)(n * H + yy) * W + xx) * C];
          for (int c = 0; c < C; ++c) {
            h += (double)px[c] * hwh[((size_t)co * C + c) * taps + kh * K + kw];
            gg += (double)px[c] * hwg[((size_t)co * C + c) * taps + kh * K + kw];
          }
        }
      const double s = 1.0 / (1.0 + exp(-gg)), r = h * s;
      rmax = fmax(rmax, fabs(r));
      eo = fmax(eo, fabs(ho[(size_t)m * Co + co] - r));
      es = fmax(es, fabs(hs[(size_t)m * Co + co] - s));
      ei = fmax(ei, fabs(img_at(himg, m, co, nks_o) - r));
    }
  }
  const double gf = 4.0 * M * Co * C * taps * 1e-9;
  printf("cw fwd  N=%d C=%d H=%d Co=%d k=%d (window %d slots, %d blocks): %.1f us %.0f TF (%.3f of the 6-product ceiling 417) | without the fp32 copy %.1f us | main loop only %.1f us | err/max: out %.2e s %.2e image %.2e\n",
         N, C, H, Co, K, slots, (M + G::R - 1) / G::R * tiles_n, t, gf / t * 1e3, gf / t * 1e3 / 417.0, t2, tn, eo / rmax, es, ei / rmax);
  hipFree(dx); hipFree(dwh); hipFree(dwg); hipFree(ix); hipFree(iw); hipFree(io); hipFree(ds); hipFree(dout);
}

// data gradient of a gated layer C -> Co (stride 1), with the gate derivative of the layer below (C channels) in the epilogue:
// v = conv_transpose([dh | dg], [wh | wg]);  [dh' | dg'] = [v s' | v out' (1 - s')]
static void case_dgrad(int N, int C, int H, int Co, int K, int reps) {
  const int W = H, pad = (K - 1) / 2, taps = K * K, M = N * H * W, ctot = 2 * Co;
  typedef CwGeom<4, 1, 576> G;
  if (C != 32) { printf("dgrad bench: C = 32 only\n"); return; }
  auto hdy = rnd((size_t)M * ctot, 11, 0.1f), hwh = rnd((size_t)Co * C * taps, 2, 0.05f), hwg = rnd((size_t)Co * C * taps, 3, 0.05f);
  auto ho_ = rnd((size_t)M * C, 12), hs_ = rnd((size_t)M * C, 13);
  for (auto& v : hs_) v = 1.f / (1.f + expf(-v));
  float* ddy = dev(hdy); float* dwh = dev(hwh); float* dwg = dev(hwg); float* dob = dev(ho_); float* dsb = dev(hs_);
  const int nks_in = ctot / 16, ncg = ctot / 16, nks_w = ncg * taps, nks_e = C / 16, nks_o = 2 * C / 16;
  const int Mi = (M + 127) / 128 * 128;
  unsigned char* idy = devz<unsigned char>(p6_image_bytes(M, nks_in));
  unsigned char* iw = devz<unsigned char>(p6_image_bytes(128, nks_w) + 4096);
  unsigned char* ie = devz<unsigned char>(p6_image_bytes(M, nks_e));
  unsigned char* io = devz<unsigned char>(p6_image_bytes(M, nks_o));
  float* dout = devz<float>((size_t)M * 2 * C);
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Mi * nks_in * 2 + 255) / 256), 256>>>(ddy, nullptr, M, ctot, ctot, 0, Mi, nks_in, idy);
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Mi * nks_e * 2 + 255) / 256), 256>>>(dob, nullptr, M, C, C, 0, Mi, nks_e, ie);
  cw_pack_filter_kernel<<<(unsigned)(((size_t)128 * nks_w * 2 + 255) / 256), 256>>>(dwh, dwg, Co, C, taps, 1, 32, 128, nks_w, iw);
  CK(hipDeviceSynchronize());
  const int slots = cw_window_slots(H, W, K, K, pad, G::R);
  ConvWinArgs g; memset(&g, 0, sizeof(g));
  g.xin = idy; g.nks_in = nks_in; g.ncg = ncg; g.cg0 = 0; g.N = N; g.H = H; g.W = W; g.KH = K; g.KW = K; g.pad = pad;
  g.wimg = iw; g.nks_w = nks_w; g.Co = C; g.tiles_n = 1;
  g.oimg = io; g.nks_o = nks_o; g.och0 = 0; g.out_f = dout; g.ldo = 2 * C;
  g.eimg = ie; g.nks_e = nks_e; g.ech0 = 0; g.e_s = dsb;
  if (slots > 576) { printf("dgrad: window of %d slots does not fit\n", slots); return; }
  auto run = [&](ConvWinArgs& a) { launch_conv_win<CW_DGRAD_GATE, 4, 1, 576>(a, 0, "cw dgrad"); };
  run(g);
  CK(hipDeviceSynchronize());
  const float t = time_us([&] { run(g); }, reps);
  ConvWinArgs g2 = g; g2.out_f = nullptr;
  const float t2 = time_us([&] { run(g2); }, reps);
  ConvWinArgs gn = g; gn.dbg = 4;
  const float tn = time_us([&] { run(gn); }, reps);
  run(g);
  CK(hipDeviceSynchronize());
  std::vector<float> ho((size_t)M * 2 * C);
  std::vector<unsigned char> himg(p6_image_bytes(M, nks_o));
  CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(himg.data(), io, himg.size(), hipMemcpyDeviceToHost));
  double eo = 0, ei = 0, rmax = 0;
  for (int sidx = 0; sidx < 120; ++sidx) {
    int m = (int)(((long long)sidx * 2654435761ll) % M);
    if (sidx < 8) m = (sidx & 1) ? M - 1 - sidx : sidx * 13;
    if (m >= M) m = M - 1;
    const int n = m / (H * W), y = (m / W) % H, x = m % W;
    for (int c = 0; c < C; ++c) {
      double v = 0;
      for (int kh = 0; kh < K; ++kh)
        for (int kw = 0; kw < K; ++kw) {
          const int yy = y - kh + pad, xx = x - kw + pad;          // output pixel whose tap (kh, kw) reads (y, x)
          if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
          const float* py = &hdy[((size_t)(n * H + yy) * W + xx) * ctot];
          for (int co = 0; co < Co; ++co) {
            v += (double)py[co] * hwh[((size_t)co * C + c) * taps + kh * K + kw];
            v += (double)py[Co + co] * hwg[((size_t)co * C + c) * taps + kh * K + kw];
          }
        }
      const double s = hs_[(size_t)m * C + c], o = ho_[(size_t)m * C + c];
      const double dh = v * s, dg = v * o * (1.0 - s);
      rmax = fmax(rmax, fmax(fabs(dh), fabs(dg)));
      eo = fmax(eo, fmax(fabs(ho[(size_t)m * 2 * C + c] - dh), fabs(ho[(size_t)m * 2 * C + C + c] - dg)));
      ei = fmax(ei, fmax(fabs(img_at(himg, m, c, nks_o) - dh), fabs(img_at(himg, m, C + c, nks_o) - dg)));
    }
  }
  const double gf = 2.0 * M * ctot * C * taps * 1e-9;
  printf("cw dgrad N=%d C=%d H=%d Co=%d k=%d (window %d slots, %d blocks): %.1f us %.0f TF (%.3f of 417) | without the fp32 copy %.1f us | main loop only %.1f us | err/max: fp32 %.2e image %.2e\n",
         N, C, H, Co, K, slots, (M + G::R - 1) / G::R, t, gf / t * 1e3, gf / t * 1e3 / 417.0, t2, tn, eo / rmax, ei / rmax);
  hipFree(ddy); hipFree(dwh); hipFree(dwg); hipFree(dob); hipFree(dsb); hipFree(idy); hipFree(iw); hipFree(ie); hipFree(io); hipFree(dout);
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 20224;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  const char* only = argc > 3 ? argv[3] : "";
  if (!*only) { case_fwd(37, 32, 14, 64, 5, 2); case_dgrad(37, 32, 14, 64, 5, 2); }
  if (!*only || !strcmp(only, "fwd5")) case_fwd(N, 32, 14, 64, 5, reps);
  if (!*only || !strcmp(only, "fwd3")) case_fwd(N, 32, 14, 64, 3, reps);
  if (!*only || !strcmp(only, "dgrad5")) case_dgrad(N, 32, 14, 64, 5, reps);
  return 0;
}
