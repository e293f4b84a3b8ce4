// Stand-alone check + timing of the window convolution kernels (csrc/evae_conv_win.h) at convhvae_2level's layer shapes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -I exemplar-vae_amd/csrc tools/micro/cw_bench.hip -o tools/micro/cw_bench
// Run on the GPU box: tools/micro/cw_bench [images] [reps]
#ifndef CW_PLAIN_BUILD           // -DCW_PLAIN_BUILD: the kernels as the library builds them (no ablation branches): the timings that count
#define EVAE_CW_ABL 1
#endif
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
  for (int i = 0; i < 3 + reps; ++i) f();           // (warm clocks: as many launches as are timed)
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

struct Layer { int C, Hin, Co, K, stride, pad; };

// rows of an Hin x Hin activation in the order a consumer of stride `cs` wants them
static size_t row_of(int n, int y, int x, int Hin, int cs) {
  return cs == 2 ? (size_t)n * Hin * Hin + cw_planar(y, x, Hin, Hin) : ((size_t)n * Hin + y) * Hin + x;
}

template <int WR, int NT, int SLOTS>
static bool setup_geom(ConvWinArgs& g, int H, int plo, int phi, const char* what) {
  typedef CwGeom<WR, NT, SLOTS> G;
  const int slots = cw_window_slots(H, H, plo, phi, G::R);
  if (slots > SLOTS) { printf("%s: window of %d slots does not fit %d\n", what, slots, SLOTS); return false; }
  g.nsp = (slots + 31) / 32;
  return true;
}

// gated forward of layer L over N images; `ocs`: stride of the layer that consumes the output (its rows go parity-planar for 2)
static void case_fwd(int N, Layer L, int ocs, int reps) {
  const int C = L.C, Hin = L.Hin, Co = L.Co, K = L.K, st = L.stride, pad = L.pad, H = Hin / st, taps = K * K;
  const int Min = N * Hin * Hin, M = N * H * H;
  auto hx = rnd((size_t)Min * C, 1), hwh = rnd((size_t)Co * C * taps, 2, 0.05f), hwg = rnd((size_t)Co * C * taps, 3, 0.05f), hbh = rnd(Co, 4, 0.3f), hbg = rnd(Co, 5, 0.3f);
  // input rows in the order THIS layer wants (planar when it has stride 2); hx is indexed [row_of(..)][C]
  float* dx = dev(hx); float* dwh = dev(hwh); float* dwg = dev(hwg); float* dbh = dev(hbh); float* dbg = dev(hbg);
  int plo, phi;
  const CwTaps tp = cw_taps_fwd(K, st, pad, &plo, &phi);
  const int nks_in = C / 16, ncg = C / 16, nks_w = cw_ksteps(tp, ncg), nks_o = Co / 16;
  const int bn = Co >= 64 ? 128 : 64, tiles_n = (2 * Co + bn - 1) / bn;
  const int Mi = (Min + 127) / 128 * 128;
  unsigned char* ix = devz<unsigned char>(p6_image_bytes(Min, nks_in));
  unsigned char* iw = devz<unsigned char>(p6_image_bytes(tiles_n * bn, nks_w) + 8192);
  unsigned char* io = devz<unsigned char>(p6_image_bytes(M, nks_o));
  float* ds = devz<float>((size_t)M * Co); float* dout = devz<float>((size_t)M * Co);
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Mi * nks_in * 2 + 255) / 256), 256>>>(dx, nullptr, Min, C, C, 0, Mi, nks_in, ix);
  const int wrows = (tiles_n * bn + 127) / 128 * 128;
  cw_pack_filter_kernel<<<(unsigned)(((size_t)wrows * nks_w * 2 + 255) / 256), 256>>>(dwh, dwg, Co, C, K, K, tp, ncg, 0, bn, wrows, nks_w, iw);
  CK(hipDeviceSynchronize());
  ConvWinArgs g; memset(&g, 0, sizeof(g));
  g.xin = ix; g.nks_in = nks_in; g.ncg = ncg; g.cg0 = 0; g.N = N; g.H = H; g.W = H; g.plo = plo; g.phi = phi; g.taps = tp;
  g.istride = st * st * H * H; for (int s2 = 0; s2 < st * st; ++s2) g.ioff[s2] = s2 * H * H;
  g.wimg = iw; g.nks_w = nks_w; g.Co = Co; g.tiles_n = tiles_n; g.bias0 = dbh; g.bias1 = dbg;
  g.out_planar = ocs == 2;
  g.oimg = io; g.nks_o = nks_o; g.och0 = 0; g.out_s = ds; g.out_f = dout; g.ldo = Co;
  if (getenv("CW_STAG")) g.stagger = atoi(getenv("CW_STAG"));
  char what[128]; snprintf(what, sizeof what, "cw fwd   N=%d C=%d Hin=%d Co=%d k=%d s=%d (out rows %s)", N, C, Hin, Co, K, st, ocs == 2 ? "planar" : "natural");
  bool ok;
  if (bn == 128) ok = setup_geom<2, 2, 320>(g, H, plo, phi, what); else ok = setup_geom<4, 2, 576>(g, H, plo, phi, what);
  if (!ok) return;
  auto run = [&](ConvWinArgs& a) { if (bn == 128) launch_conv_win<CW_FWD_GATED, 2, 2, 320>(a, 0, "cw fwd"); else launch_conv_win<CW_FWD_GATED, 4, 2, 576>(a, 0, "cw fwd"); };
  run(g);
  CK(hipDeviceSynchronize());
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
    if (sidx >= 8 && sidx < 24) m = (sidx - 8) * (H * H / 16) + (sidx & 1 ? H * H * (N - 1) : 0);
    if (m >= M) m = M - 1;
    const int n = m / (H * H), y = (m / H) % H, x = m % H;
    const size_t orow = row_of(n, y, x, H, ocs), nrow = row_of(n, y, x, H, 1);
    for (int co = 0; co < Co; ++co) {
      double h = hbh[co], gg = hbg[co];
      for (int kh = 0; kh < K; ++kh)
        for (int kw = 0; kw < K; ++kw) {
          const int yy = st * y + kh - pad, xx = st * x + kw - pad;
          if (yy < 0 || yy >= Hin || xx < 0 || xx >= Hin) continue;
          const float* px = &hx[row_of(n, yy, xx, Hin, st) * C];
          for (int c = 0; c < C; ++c) {
            h += (double)px[c] * hwh[((size_t)co * C + c) * taps + kh * K + kw];
            gg += (double)px[c] * hwg[((size_t)co * C + c) * taps + kh * K + kw];
          }
        }
      const double s = 1.0 / (1.0 + exp(-gg)), r = h * s;
      rmax = fmax(rmax, fabs(r));
      eo = fmax(eo, fabs(ho[nrow * Co + co] - r));
      es = fmax(es, fabs(hs[nrow * Co + co] - s));
      ei = fmax(ei, fabs(img_at(himg, (int)orow, co, nks_o) - r));
    }
  }
  const double gf = 4.0 * M * Co * C * taps * 1e-9;
  printf("%s: %.1f us %.0f TF | main loop only %.1f us | err/max: out %.2e s %.2e image %.2e\n", what, t2, gf / t2 * 1e3, tn, eo / rmax, es, ei / rmax);
  hipFree(dx); hipFree(dwh); hipFree(dwg); hipFree(ix); hipFree(iw); hipFree(io); hipFree(ds); hipFree(dout);
}

// data gradient of gated layer L (C -> Co) with the gate derivative of the layer below (C channels) in the epilogue:
// v = conv_transpose([dh | dg], [wh | wg]);  [dh' | dg'] = [v s' | v out' (1 - s')].  dy rows: `dcs` order (the stride of L's consumer);
// the result's rows (L's input pixels): planar when L has stride 2
static void case_dgrad(int N, Layer L, int dcs, int reps) {
  const int C = L.C, Hin = L.Hin, Co = L.Co, K = L.K, st = L.stride, pad = L.pad, H = Hin / st, taps = K * K, ctot = 2 * Co;
  const int Min = N * Hin * Hin, M = N * H * H;
  auto hdy = rnd((size_t)M * ctot, 11, 0.1f), hwh = rnd((size_t)Co * C * taps, 2, 0.05f), hwg = rnd((size_t)Co * C * taps, 3, 0.05f);
  auto ho_ = rnd((size_t)Min * C, 12), hs_ = rnd((size_t)Min * C, 13);      // ho_: rows in image order (planar for stride 2); hs_: natural
  for (auto& v : hs_) v = 1.f / (1.f + expf(-v));
  float* ddy = dev(hdy); float* dwh = dev(hwh); float* dwg = dev(hwg); float* dob = dev(ho_); float* dsb = dev(hs_);
  const int nks_in = ctot / 16, ncg = ctot / 16, nks_e = C / 16, nks_o = 2 * C / 16;
  const int Mi = (M + 127) / 128 * 128, Mini = (Min + 127) / 128 * 128;
  unsigned char* idy = devz<unsigned char>(p6_image_bytes(M, nks_in));
  unsigned char* ie = devz<unsigned char>(p6_image_bytes(Min, nks_e));
  unsigned char* io = devz<unsigned char>(p6_image_bytes(Min, nks_o));
  float* dout = devz<float>((size_t)Min * 2 * C);
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Mi * nks_in * 2 + 255) / 256), 256>>>(ddy, nullptr, M, ctot, ctot, 0, Mi, nks_in, idy);
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Mini * nks_e * 2 + 255) / 256), 256>>>(dob, nullptr, Min, C, C, 0, Mini, nks_e, ie);
  char what[128]; snprintf(what, sizeof what, "cw dgrad N=%d C=%d Hin=%d Co=%d k=%d s=%d (dy rows %s)", N, C, Hin, Co, K, st, dcs == 2 ? "planar" : "natural");
  // one launch per input parity class
  std::vector<ConvWinArgs> gs; std::vector<unsigned char*> iws;
  double ksum = 0;
  for (int py = 0; py < st; ++py)
    for (int px = 0; px < st; ++px) {
      int plo, phi;
      const CwTaps tp = cw_taps_dgrad(K, st, pad, py, px, &plo, &phi);
      const int nks_w = cw_ksteps(tp, ncg);
      unsigned char* iw = devz<unsigned char>(p6_image_bytes(128, nks_w) + 8192);
      cw_pack_filter_kernel<<<(unsigned)(((size_t)128 * nks_w * 2 + 255) / 256), 256>>>(dwh, dwg, Co, C, K, K, tp, ncg, 1, C, 128, nks_w, iw);
      ConvWinArgs g; memset(&g, 0, sizeof(g));
      g.xin = idy; g.nks_in = nks_in; g.ncg = ncg; g.N = N; g.H = H; g.W = H; g.plo = plo; g.phi = phi; g.taps = tp;
      g.istride = H * H; g.in_planar = dcs == 2;
      g.wimg = iw; g.nks_w = nks_w; g.Co = C; g.tiles_n = 1;
      g.ostride = st * st * H * H; g.ooff = (py * st + px) * H * H;
      g.nat_h = Hin; g.nat_w = Hin; g.nat_s = st; g.nat_y = py; g.nat_x = px;
      g.oimg = io; g.nks_o = nks_o; g.och0 = 0; g.out_f = dout; g.ldo = 2 * C;
      g.eimg = ie; g.nks_e = nks_e; g.ech0 = 0; g.e_s = dsb;
      bool ok;
      if (C == 32) ok = setup_geom<4, 1, 576>(g, H, plo, phi, what); else ok = setup_geom<4, 2, 576>(g, H, plo, phi, what);
      if (!ok) return;
      if (getenv("CW_STAG")) g.stagger = atoi(getenv("CW_STAG"));
      gs.push_back(g); iws.push_back(iw); ksum += nks_w;
    }
  CK(hipDeviceSynchronize());
  auto run = [&](bool f32, int dbg) {
    for (auto g : gs) {
      if (!f32) g.out_f = nullptr;
      g.dbg = dbg;
      if (C == 32) launch_conv_win<CW_DGRAD_GATE, 4, 1, 576>(g, 0, "cw dgrad"); else launch_conv_win<CW_DGRAD_GATE, 4, 2, 576>(g, 0, "cw dgrad");
    }
  };
  run(true, 0);
  CK(hipDeviceSynchronize());
  const float t2 = time_us([&] { run(false, 0); }, reps);
  const float tn = time_us([&] { run(false, 4); }, reps);
  run(true, 0);
  CK(hipDeviceSynchronize());
  std::vector<float> ho((size_t)Min * 2 * C);
  std::vector<unsigned char> himg(p6_image_bytes(Min, nks_o));
  CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(himg.data(), io, himg.size(), hipMemcpyDeviceToHost));
  double eo = 0, ei = 0, rmax = 0;
  for (int sidx = 0; sidx < 120; ++sidx) {
    int m = (int)(((long long)sidx * 2654435761ll) % Min);
    if (sidx < 8) m = (sidx & 1) ? Min - 1 - sidx : sidx * 13;
    if (m >= Min) m = Min - 1;
    const int n = m / (Hin * Hin), Y = (m / Hin) % Hin, X = m % Hin;
    const size_t orow = row_of(n, Y, X, Hin, st), nrow = row_of(n, Y, X, Hin, 1);
    for (int c = 0; c < C; ++c) {
      double v = 0;
      for (int kh = 0; kh < K; ++kh)
        for (int kw = 0; kw < K; ++kw) {
          const int ty = Y + pad - kh, tx = X + pad - kw;            // = stride * (output pixel whose tap (kh, kw) reads (Y, X))
          if (ty < 0 || tx < 0 || ty % st || tx % st) continue;
          const int yy = ty / st, xx = tx / st;
          if (yy >= H || xx >= H) continue;
          const float* py_ = &hdy[row_of(n, yy, xx, H, dcs) * ctot];
          for (int co = 0; co < Co; ++co) {
            v += (double)py_[co] * hwh[((size_t)co * C + c) * taps + kh * K + kw];
            v += (double)py_[Co + co] * hwg[((size_t)co * C + c) * taps + kh * K + kw];
          }
        }
      const double s = hs_[nrow * C + c], o = ho_[orow * C + c];
      const double dh = v * s, dg = v * o * (1.0 - s);
      rmax = fmax(rmax, fmax(fabs(dh), fabs(dg)));
      eo = fmax(eo, fmax(fabs(ho[nrow * 2 * C + c] - dh), fabs(ho[nrow * 2 * C + C + c] - dg)));
      ei = fmax(ei, fmax(fabs(img_at(himg, (int)orow, c, nks_o) - dh), fabs(img_at(himg, (int)orow, C + c, nks_o) - dg)));
    }
  }
  const double gf = 2.0 * M * ctot * C * taps * 1e-9;
  printf("%s: %.1f us %.0f TF | main loop only %.1f us | err/max: fp32 %.2e image %.2e\n", what, t2, gf / t2 * 1e3, tn, eo / rmax, ei / rmax);
  hipFree(ddy); hipFree(dwh); hipFree(dwg); hipFree(dob); hipFree(dsb); hipFree(idy); hipFree(ie); hipFree(io); hipFree(dout);
  for (auto p_ : iws) hipFree(p_);
}

// weight gradient of a stride-1 gated layer L: dW[cc][ci][tap], db[cc] from the merged-gradient image and the input image
static void case_wgrad(int N, Layer L, int dcs, int reps) {
  const int C = L.C, H = L.Hin, Co = L.Co, K = L.K, pad = L.pad, taps = K * K, CC = 2 * Co;
  const int M = N * H * H;
  if (L.stride != 1 || C != 32 || taps != 25) { printf("wgrad bench: 5 x 5, stride 1, 32 input channels only\n"); return; }
  auto hdy = rnd((size_t)M * CC, 21, 0.1f), hx = rnd((size_t)M * C, 22);          // hdy rows in `dcs` order, hx natural
  float* ddy = dev(hdy); float* dx = dev(hx);
  const int nks_dy = CC / 16, nks_x = C / 16, Mi = (M + 127) / 128 * 128;
  unsigned char* idy = devz<unsigned char>(p6_image_bytes(M, nks_dy));
  unsigned char* ix = devz<unsigned char>(p6_image_bytes(M, nks_x));
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Mi * nks_dy * 2 + 255) / 256), 256>>>(ddy, nullptr, M, CC, CC, 0, Mi, nks_dy, idy);
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Mi * nks_x * 2 + 255) / 256), 256>>>(dx, nullptr, M, C, C, 0, Mi, nks_x, ix);
  int plo, phi;
  const CwTaps tp = cw_taps_fwd(K, 1, pad, &plo, &phi);
  constexpr int WSL = 192;
  const int slots = cw_wgrad_window_slots(H, H, K, pad, 1, 32);
  if (slots > WSL) { printf("wgrad: window of %d slots does not fit %d\n", slots, WSL); return; }
  const int nblk = 256;
  float* part = devz<float>((size_t)nblk * CC * taps * C); float* dbp = devz<float>((size_t)nblk * CC);
  float* dw = devz<float>((size_t)CC * C * taps); float* db = devz<float>(CC);
  CwWgradArgs g; memset(&g, 0, sizeof(g));
  g.dyimg = idy; g.nks_dy = nks_dy; g.dy_planar = dcs == 2; g.ximg = ix; g.nks_x = nks_x; g.xcg0 = 0; g.nseg = 1;
  g.N = N; g.H = H; g.W = H; g.plo = plo; g.phi = phi;
  const int PW = H + plo + phi;
  g.ntap_f = taps; g.Cin = C; g.CC = CC; g.part = part; g.dbpart = dbp;
  // the 25 taps as two launches of 13 and 12 column tiles (accumulator space: 16 registers per tile)
  CwWgradArgs ga = g, gb = g;
  for (int t = 0; t < 13; ++t) { ga.tile_to[t] = (t / K) * PW + t % K; ga.tile_tap[t] = t; }
  for (int t = 13; t < 25; ++t) { gb.tile_to[t - 13] = (t / K) * PW + t % K; gb.tile_tap[t - 13] = t; }
  gb.dbpart = nullptr;
  auto run = [&]() {
    launch_conv_wgrad_win<13, WSL, 8>(ga, nblk, 0, "cw wgrad");
    launch_conv_wgrad_win<12, WSL, 8>(gb, nblk, 0, "cw wgrad");
    const int nb = (int)((size_t)ga.nchunk + ga.cper - 1) / ga.cper;
    cw_wgrad_finish_kernel<<<cw_wgrad_finish_blocks(CC, taps, C), 256>>>(part, dbp, nb, CC, taps, C, dw, db);
  };
  run();
  CK(hipDeviceSynchronize());
  const float t = time_us(run, reps);
  if (!getenv("CW_NOABL")) {
    ga.dbg = gb.dbg = 1; const float t_nocopy = time_us(run, reps);
    ga.dbg = gb.dbg = 2; const float t_copyonly = time_us(run, reps);
    ga.dbg = gb.dbg = 5; const float t_noread = time_us(run, reps);
    ga.dbg = gb.dbg = 0; run(); CK(hipDeviceSynchronize());
    printf("          (MFMA stream without the copies %.1f us; copies without the MFMA stream %.1f us; MFMAs without copies and fragment reads %.1f us)\n", t_nocopy, t_copyonly, t_noread);
  }
  std::vector<float> hw((size_t)CC * C * taps), hb(CC);
  CK(hipMemcpy(hw.data(), dw, hw.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), db, hb.size() * 4, hipMemcpyDeviceToHost));
  double e = 0, rm = 0, eb = 0, rb = 0;
  for (int sidx = 0; sidx < 40; ++sidx) {
    const int cc = (sidx * 37) % CC, ci = (sidx * 11) % C, tap = (sidx * 7) % taps, kh = tap / K, kw = tap % K;
    double r = 0, b = 0;
    for (int n = 0; n < N; ++n)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < H; ++x) {
          const double d = hdy[row_of(n, y, x, H, dcs) * CC + cc];
          b += d;
          const int yy = y + kh - pad, xx = x + kw - pad;
          if (yy < 0 || yy >= H || xx < 0 || xx >= H) continue;
          r += d * hx[row_of(n, yy, xx, H, 1) * C + ci];
        }
    rm = fmax(rm, fabs(r)); e = fmax(e, fabs(hw[((size_t)cc * C + ci) * taps + tap] - r));
    rb = fmax(rb, fabs(b)); eb = fmax(eb, fabs(hb[cc] - b));
  }
  const double gf = 2.0 * M * CC * C * taps * 1e-9;
  printf("cw wgrad N=%d C=%d H=%d Co=%d k=%d (dy rows %s, window %d slots, %d blocks): %.1f us %.0f TF | err/max: dw %.2e db %.2e\n", N, C, H, Co, K,
         dcs == 2 ? "planar" : "natural", slots, nblk, t, gf / t * 1e3, e / rm, eb / fmax(rb, 1e-30));
  hipFree(ddy); hipFree(dx); hipFree(idy); hipFree(ix); hipFree(part); hipFree(dbp); hipFree(dw); hipFree(db);
}

// timing of ONE residual block x + conv(ELU(x)) + b (forward, data gradient) at fully_conv's shapes, with ablations (no check: tests/test_gpu_conv.py)
static void case_res(int N, int C, int H, int reps) {
  const int K = 3, M = N * H * H, ncg = C / 16, tiles_n = (C + 63) / 64, wrows = (tiles_n * 64 + 127) / 128 * 128;
  auto hx = rnd((size_t)M * C, 1), hw = rnd((size_t)C * C * 9, 2, 0.05f), hb = rnd(C, 4, 0.3f);
  float* dx = dev(hx); float* dw = dev(hw); float* db = dev(hb);
  int plo, phi;
  const CwTaps tp = cw_taps_fwd(K, 1, 1, &plo, &phi);
  const int nks_w = cw_ksteps(tp, ncg);
  const int Mi = (M + 127) / 128 * 128;
  unsigned char* ix = devz<unsigned char>(p6_image_bytes(M, ncg));
  unsigned char* iw = devz<unsigned char>(p6_image_bytes(wrows, nks_w) + 8192);
  unsigned char* io = devz<unsigned char>(p6_image_bytes(M, ncg));
  float* dout = devz<float>((size_t)M * C);
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Mi * ncg * 2 + 255) / 256), 256>>>(dx, nullptr, M, C, C, 0, Mi, ncg, ix);
  cw_pack_filter_kernel<<<(unsigned)(((size_t)wrows * nks_w * 2 + 255) / 256), 256>>>(dw, nullptr, C, C, K, K, tp, ncg, 2, 64, wrows, nks_w, iw);
  CK(hipDeviceSynchronize());
  ConvWinArgs g; memset(&g, 0, sizeof(g));
  g.xin = ix; g.nks_in = ncg; g.ncg = ncg; g.N = N; g.H = H; g.W = H; g.plo = plo; g.phi = phi; g.taps = tp;
  g.wimg = iw; g.nks_w = nks_w; g.Co = C; g.tiles_n = tiles_n; g.bias0 = db;
  g.oimg = io; g.nks_o = ncg; g.out_f = dout; g.ldo = C; g.e_s = dx;
  g.eimg = ix; g.nks_e = ncg;
  if (!setup_geom<4, 2, 576>(g, H, plo, phi, "res")) return;
  auto run = [&](ConvWinArgs& a) { launch_conv_win<CW_RES_FWD, 4, 2, 576>(a, 0, "res fwd"); };
  auto runb = [&](ConvWinArgs& a) { launch_conv_win<CW_RES_BWD, 4, 2, 576>(a, 0, "res bwd"); };
  const float t_full = time_us([&] { run(g); }, reps);
  ConvWinArgs a = g; a.dbg = 4;
  const float t_loop = time_us([&] { run(a); }, reps);
  a.dbg = 4 | 8;
  const float t_loop1w = time_us([&] { run(a); }, reps);
  a = g; a.oimg = nullptr;
  const float t_noimg = time_us([&] { run(a); }, reps);
  a = g; a.out_f = nullptr;
  const float t_nof = time_us([&] { run(a); }, reps);
  const float tb_full = time_us([&] { runb(g); }, reps);
  {
    for (int n = -1; n <= 8; n += (n < 0 ? 2 : 1)) {
      a = g; a.stagger = n; const float tf = time_us([&] { run(a); }, reps);
      a = g; a.stagger = n; const float tb = time_us([&] { runb(a); }, reps);
      printf("  stagger %d: fwd %.1f us bwd %.1f us\n", n, tf, tb);
    }
  }
  const double gf = 2.0 * M * C * C * 9 * 1e-9;
  printf("res block N=%d C=%d H=%d (%d blocks): fwd %.1f us %.0f TF | loop only %.1f | loop, one window %.1f | no image out %.1f | no fp32 out %.1f | bwd %.1f us\n",
         N, C, H, (M + 255) / 256 * tiles_n, t_full, gf / t_full * 1e3, t_loop, t_loop1w, t_noimg, t_nof, tb_full);
  hipFree(dx); hipFree(dw); hipFree(db); hipFree(ix); hipFree(iw); hipFree(io); hipFree(dout);
}

// timing + ablations of ONE launch of the 3 x 3 stride-1 weight gradient at fully_conv's residual shapes (first channel-group pair)
template <int WSL, int NCG, int SS>
static void case_wgrad3(int N, int C, int H, int reps) {
  const int K = 3, taps = 9, CC = C, M = N * H * H;
  auto hdy = rnd((size_t)M * CC, 21, 0.1f), hx = rnd((size_t)M * C, 22);
  float* ddy = dev(hdy); float* dx = dev(hx);
  const int nks = C / 16, Mi = (M + 127) / 128 * 128;
  unsigned char* idy = devz<unsigned char>(p6_image_bytes(M, nks));
  unsigned char* ix = devz<unsigned char>(p6_image_bytes(M, nks));
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Mi * nks * 2 + 255) / 256), 256>>>(ddy, nullptr, M, CC, CC, 0, Mi, nks, idy);
  p6_pack_rows_kernel<<<(unsigned)(((size_t)Mi * nks * 2 + 255) / 256), 256>>>(dx, nullptr, M, C, C, 0, Mi, nks, ix);
  const int nblk = 256;
  float* part = devz<float>((size_t)nblk * CC * taps * C); float* dbp = devz<float>((size_t)nblk * CC);
  CwWgradArgs g; memset(&g, 0, sizeof(g));
  g.dyimg = idy; g.nks_dy = nks; g.ximg = ix; g.nks_x = nks; g.xcg0 = 0; g.nseg = 1;
  g.N = N; g.H = H; g.W = H; g.plo = 1; g.phi = 1; g.xs = 1;
  const int PW = H + 2;
  g.ntap_f = taps; g.Cin = C; g.CC = CC; g.part = part; g.dbpart = dbp;
  for (int t = 0; t < 9; ++t) { g.tile_to[t] = (t / K) * PW + t % K; g.tile_tap[t] = t; }
  auto run = [&]() { launch_conv_wgrad_win<9, WSL, NCG, SS>(g, nblk, 0, "cw wgrad3"); };
  run(); CK(hipDeviceSynchronize());
  const float t = time_us(run, reps);
  g.dbg = 1; const float t_nocopy = time_us(run, reps);
  g.dbg = 2; const float t_copyonly = time_us(run, reps);
  g.dbg = 5; const float t_noread = time_us(run, reps);
  g.dbg = 4; const float t_noread_copies = time_us(run, reps);
  g.dbg = 0;
  const double mf = (double)M / 16 * (NCG <= 4 ? 2 : 4) * 10 * 6;
  printf("cw wgrad3 N=%d C=%d H=%d <9,%d,%d,%d> (%d blocks x %d chunks): %.1f us (%.0f M MFMA: %.1f us at 49 M/ms) | no copies after the first %.1f | copies only %.1f | "
         "MFMAs alone %.1f | MFMAs + copies, no fragment reads %.1f\n", N, C, H, WSL, NCG, SS, (g.nchunk + g.cper - 1) / g.cper, g.cper, t, mf * 1e-6, mf / 49e3,
         t_nocopy, t_copyonly, t_noread, t_noread_copies);
  hipFree(ddy); hipFree(dx); hipFree(idy); hipFree(ix); hipFree(part); hipFree(dbp);
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 20224;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  const char* only = argc > 3 ? argv[3] : "";
  const Layer L2 = {32, 28, 32, 3, 2, 1}, L3 = {32, 14, 64, 5, 1, 2}, L4 = {64, 14, 64, 3, 2, 1}, L3k3 = {32, 14, 64, 3, 1, 1};
  if (!*only) {
    case_fwd(37, L3, 1, 2); case_fwd(37, L3, 2, 2); case_fwd(37, L2, 1, 2); case_fwd(37, L4, 1, 2);
    case_dgrad(37, L3, 1, 2); case_dgrad(37, L3, 2, 2); case_dgrad(37, L2, 1, 2); case_dgrad(37, L4, 1, 2);
    case_wgrad(37, L3, 1, 2); case_wgrad(37, L3, 2, 2);
  }
  if (!strcmp(only, "wgrad3")) {
    case_wgrad3<288, 4, 4>(100, 48, 64, reps); case_wgrad3<320, 4, 1>(100, 48, 64, reps); case_wgrad3<288, 4, 4>(100, 48, 32, reps); case_wgrad3<192, 4, 1>(100, 48, 32, reps);
    case_wgrad3<224, 8, 2>(100, 96, 32, reps); case_wgrad3<192, 8, 1>(100, 96, 32, reps); case_wgrad3<224, 8, 2>(100, 96, 16, reps);
    return 0;
  }
  if (!strcmp(only, "res")) { case_res(100, 48, 64, reps); case_res(100, 96, 32, reps); case_res(203, 48, 32, reps); case_res(203, 96, 16, reps); return 0; }
  if (!*only || !strcmp(only, "wgrad5")) case_wgrad(N, L3, 2, reps);
  if (!*only || !strcmp(only, "fwd5")) case_fwd(N, L3, 2, reps);
  if (!*only || !strcmp(only, "fwd3")) case_fwd(N, L3k3, 1, reps);
  if (!*only || !strcmp(only, "fwdL2")) case_fwd(N, L2, 1, reps);
  if (!*only || !strcmp(only, "fwdL4")) case_fwd(N, L4, 1, reps);
  if (!*only || !strcmp(only, "dgrad5")) case_dgrad(N, L3, 2, reps);
  if (!*only || !strcmp(only, "dgradL2")) case_dgrad(N, L2, 1, reps);
  if (!*only || !strcmp(only, "dgradL4")) case_dgrad(N, L4, 1, reps);
  return 0;
}
