// Stride-1 convolutions over pre-split pixel images with the input WINDOW of a block resident in LDS ("cw" kernels, r05).
//
// Arithmetic: that of evae_gemm_p6.h (every fp32 element = three round-to-nearest bf16 terms, six of the nine partial products
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, smallest first) -- the fp32 bar, not a reduced-precision path.
//
// Why a window.  gemm_p6_kernel (and the byte layer) are bound by the L2 -> LDS copy rate: a 128 x 128 tile copies 24 KB per
// 24 MFMAs per wave.  An implicit-GEMM convolution that re-gathers its A tile for every filter tap moves the same
// bytes per MFMA although the 25 taps of a 5 x 5 filter read the same few hundred input pixels.  Here a block of 128 (or 256) output
// pixels copies the (zero-padded) window of input pixels its taps touch ONCE per channel group -- <= 320 (576) pixel slots x 16
// channels x 6 bytes = 30 (54) KB -- and the A fragments of every tap's k-step are ds_read_b128 reads at a per-tap slot
// offset: no copy per tap, no im2col matrix (reference utils/nn.py:72-97 on nn.Conv2d's implicit im2col).  Only the filter image
// streams through a two-stage LDS ring: 12 KB per k-step for 4 waves x 24 MFMAs = 1/2 of gemm_p6's bytes per MFMA.
//
// Operand format.  Activations are "pixel images": the p6 rows-image (evae_p6_image.h) of X [pixels (n, y, x) x channels],
// channels a multiple of 16: per 16 pixels and 16 channels three 512-byte planes, 32 bytes per (pixel, plane).  One image per
// activation tensor serves the forward convolution of the next layer (rows = pixels), the data gradient (the merged [dh | dg]
// image, rows = pixels) and -- through the LDS transpose read -- the weight gradient (contraction over pixels).  Layers write
// their output image in their epilogue (through an LDS transpose: the matrix core's C layout holds a channel per lane, the
// image wants sixteen channels of a pixel side by side).
//
// Window.  Padded image grid PW = W + 2 pad, slots per image SP = (H + 2 pad) PW; slot of (n, py, px) = n SP + py PW + px; output
// pixel (n, y, x), tap (kh, kw) reads slot n SP + (y + kh) PW + (x + kw) = [slot of the pixel's tap (0, 0)] + kh PW + kw.  A
// block's window = the slots from its first pixel's tap (0, 0) to its last pixel's last tap, loaded with per-lane gather
// addresses by LDS-DMA copies (buffer_load ... lds: lane-linear LDS destination, per-lane source); padding slots ride an
// out-of-range offset, which the copy writes as zeros (tools/micro/dma_oob.hip checks that on the device).  LDS layout
// [plane][slot][32 B], the two 16-byte halves of a slot swapped on odd groups of eight slots (conflict-free
// ds_read_b128 for lanes on consecutive slots, whatever the tap shift).
#pragma once
#include "evae_gemm_p6.h"

namespace evae {


// Block = 4 waves (WR x WC, WR * WC = 4), wave tile 64 pixels x (32 NT) columns: block tile R = 64 WR pixels x BN = 32 NT WC columns.
// The window holds ONE channel group (16 channels: 96 bytes per slot) at a time -- the contraction runs (channel group, tap) --
// so that two blocks fit a CU (<= 80 KB each): while one block loads a window, fills its ring or stores its result, the other
// one's MFMAs run (one block of eight waves per CU was measured first: 61 % of the matrix rate in the loop with both waves of
// a SIMD in lockstep at the block's barriers, and the epilogue's stores -- 8 B/clk/CU of store issue -- added 14 % uncovered).
template <int WR, int NT, int SLOTS>
struct CwGeom {
  static constexpr int WC = 4 / WR;
  static constexpr int R = 64 * WR;
  static constexpr int BN = 32 * NT * WC;
  static constexpr int WP = SLOTS * 32;                 // bytes of one plane of the window
  static constexpr int WIN = 3 * WP;
  static constexpr int BKS = 3 * BN * 32;               // one k-step of the filter tile
  static constexpr int KST = BN >= 64 ? 2 : 4;          // k-steps per ring stage: a multiple of four 1 KB copies (one set per wave)
  static constexpr int BST = KST * BKS;                 // a ring stage
  static constexpr int LOOP_LDS = WIN + 2 * BST;
  // epilogue staging: fp32 [R][BNO + 4] (gated: two of them, BNO = BN / 2 result columns)
  static constexpr int lds_bytes(bool gated) {
    const int epi = gated ? 2 * R * (BN / 2 + 4) * 4 : R * (BN + 4) * 4;
    return LOOP_LDS > epi ? LOOP_LDS : epi;
  }
  static constexpr int NPIECE = KST * 3 * (BN / 32);    // 1 KB copies per ring stage
  static_assert(NPIECE % 4 == 0, "every wave issues the same number of copies per stage");
};

// The taps of a launch: up to four input segments (the stride-parity sub-images of a strided layer's input; one segment otherwise),
// each with a RECTANGLE of tap offsets (dy, dx) in [dy0, dy0 + ndy) x [dx0, dx0 + ndx) relative to the output pixel -- the taps of a
// stride-s filter that fall on one parity class are consecutive offsets -- and the filter tap of an offset: kh = a dy + bh,
// kw = a dx + bw (forward, stride s, input parity (py, px): a = s, b = p + pad; data gradient of a stride-s layer into the
// input pixels of parity (py, px): a = -s, b = p + pad).
struct CwSegTaps { int dy0, ndy, dx0, ndx, bh, bw; };
struct CwTaps { int nseg, a; CwSegTaps s[4]; };

struct ConvWinArgs {
  const unsigned char* xin;   // pixel image of the input, nks_in channel groups per pixel
  int nks_in;
  int ncg;                    // channel groups contracted over per segment, starting at group cg0 of the image
  int cg0;
  int N, H, W;                // images; height and width of the OUTPUT grid = of every input segment
  int plo, phi;               // tap offsets lie in [-plo, phi]: the window grid is (H + plo + phi) x (W + plo + phi)
  int nsp;                    // 32-slot pieces of the window actually needed (<= SLOTS / 32; cw_window_slots)
  int PW, SP;
  FastDiv div_w, div_hw, div_pw, div_sp;
  int istride;                // input image rows per image (segments * H * W)
  int in_planar;              // 1: the input (ONE segment) is stored parity-planar (see cw_planar)
  int ioff[4];                // input row of segment s, pixel (n, y, x): n * istride + ioff[s] + y * W + x
  CwTaps taps;
  const unsigned char* wimg;  // filter image: BN rows per column tile, k-steps ordered (segment, channel group, tap raster)
  int nks_w;
  int M;                      // N * H * W
  int Co;                     // real output columns (gated: channels per bank)
  int tiles_n;
  const float* bias0;
  const float* bias1;
  // rows of the result.  Pixel images (oimg, eimg): n * ostride + ooff + y * W + x, or (out_planar) n * H * W + cw_planar(y, x).
  // fp32 tensors (out_s, out_f, e_s) are ALWAYS channels-last in natural pixel order: row (n * nat_h + y * nat_s + nat_y) * nat_w +
  // x * nat_s + nat_x -- the identity unless the launch writes one stride-parity class of a larger grid (data gradient of a
  // strided layer: nat_s = stride, (nat_y, nat_x) = the class, nat_h x nat_w = the layer's input grid)
  int ostride, ooff;
  int out_planar;
  int nat_h, nat_w, nat_s, nat_y, nat_x;
  unsigned char* oimg;        // pixel image of the result (nks_o channel groups; result column c at image channel och0 + c;
  int nks_o, och0;            //   CW_DGRAD_GATE: dh at och0 + c, dg at och0 + Co + c)
  float* out_s;               // gated forward: the gate s, fp32 [rows][Co]
  float* out_f;               // fp32 copy of the result [rows][ldo] (optional; CW_DGRAD_GATE: [dh | dg], dg at column Co + c)
  int ldo;
  const unsigned char* eimg;  // CW_DGRAD_GATE: pixel image of the forward output of the layer below (nks_e groups, channel ech0 + c) ...
  int nks_e, ech0;
  const float* e_s;           // ... and its gate [rows][Co]: [dh | dg] = [v s | v out (1 - s)]
  int act;                    // CW_PLAIN: bit 0 = ELU on the result (out_f and the image hold ELU(conv + b)), bit 1 = the image holds ELU of that
  int stagger;                // launches of more than one round of blocks: the first 512 blocks start (id / 128) * stagger * 2048 clocks late
  int dbg;
};

// parity-planar pixel order of an H x W image (H, W even): the four stride-2 parity classes (y & 1, x & 1) as four H/2 x W/2
// sub-images one after the other -- what a layer whose consumer has stride 2 writes, so that the consumer's taps of one parity
// class are stride-1 shifts of one sub-image
__host__ __device__ __forceinline__ int cw_planar(int y, int x, int H, int W) {
  return (((y & 1) * 2 + (x & 1)) * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);
}

__host__ __device__ __forceinline__ size_t p6_off64(size_t r, int k, int nks) {
  return ((r >> 4) * nks + (k >> 4)) * P6_GROUP + (size_t)(((int)(r & 15)) * 32 + ((((k >> 3) & 1) ^ ((int)(r >> 3) & 1)) << 4) + (k & 7) * 2);
}

// ---- filter images ------------------------------------------------------------------------------------------------------------
// k-step ks <-> (segment, channel group cg, tap (i, j) of the segment's rectangle) in that nesting; contraction channel cc = cg * 16 + cl;
// filter tap (kh, kw) = (a (dy0 + i) + bh, a (dx0 + j) + bw).  Rows, per column tile of bn rows:
// mode 0 (gated forward): row (tn, c), c = wc * 64 + hg * 32 + j <-> bank hg (w0 = h, w1 = g), output channel tn * (bn / 2) + wc * 32 + j
// mode 2 (plain forward): row r <-> output channel r
// mode 1 (data gradient): row r <-> INPUT channel r of the layer (the data gradient's output channel); cc = merged gradient
//   channel (cc < Co: bank h, output channel cc; else bank g, cc - Co)
// w layout: nn.Conv2d's [Co][Ci][KH][KW]
__device__ __forceinline__ void cw_pack_filter_body(const size_t t, const float* __restrict__ w0, const float* __restrict__ w1, int Co, int Ci,
                                                    int KH, int KW, const CwTaps& tp, int ncg, int mode, int bn, int rows_img, int nks,
                                                    unsigned char* __restrict__ img) {
  const int kslots = nks * 2;
  if (t >= (size_t)rows_img * kslots) return;
  const int ri = (int)(t / kslots), ks8 = (int)(t - (size_t)ri * kslots);
  const int k0 = ks8 * 8;
  int kstep = k0 >> 4, sg = 0;
  while (sg < tp.nseg - 1 && kstep >= ncg * tp.s[sg].ndy * tp.s[sg].ndx) { kstep -= ncg * tp.s[sg].ndy * tp.s[sg].ndx; ++sg; }
  const int nt = tp.s[sg].ndy * tp.s[sg].ndx;
  const int cg = kstep / nt, ti = kstep - cg * nt, i = ti / tp.s[sg].ndx, j = ti - i * tp.s[sg].ndx;
  const int kh = tp.a * (tp.s[sg].dy0 + i) + tp.s[sg].bh, kw = tp.a * (tp.s[sg].dx0 + j) + tp.s[sg].bw;
  const bool tap_ok = cg < ncg && kh >= 0 && kh < KH && kw >= 0 && kw < KW;
  const int tap = kh * KW + kw, taps = KH * KW;
  unsigned short p0[8], p1[8], p2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int cc = cg * 16 + (k0 & 15) + e;     // contraction channel
    float v = 0.f;
    if (!tap_ok) {
    } else if (mode == 0) {
      const int tn = ri / bn, c = ri - tn * bn, wc = c >> 6, hg = (c >> 5) & 1, jj = c & 31;
      const int co = tn * (bn / 2) + wc * 32 + jj;
      const float* src = hg ? w1 : w0;
      if (co < Co && cc < Ci) v = src[((size_t)co * Ci + cc) * taps + tap];
    } else if (mode == 2) {
      if (ri < Co && cc < Ci) v = w0[((size_t)ri * Ci + cc) * taps + tap];
    } else {
      const int ctot = w1 ? 2 * Co : Co;
      if (ri < Ci && cc < ctot) {
        const float* src = cc < Co ? w0 : w1;
        const int co = cc < Co ? cc : cc - Co;
        v = src[((size_t)co * Ci + ri) * taps + tap];
      }
    }
    p6_split1(v, p0[e], p1[e], p2[e]);
  }
  unsigned char* o = img + p6_off(ri, k0, nks);
  *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(p0);
  *reinterpret_cast<uint4*>(o + P6_CHUNK) = *reinterpret_cast<const uint4*>(p1);
  *reinterpret_cast<uint4*>(o + 2 * P6_CHUNK) = *reinterpret_cast<const uint4*>(p2);
}

__global__ __launch_bounds__(256) void cw_pack_filter_kernel(const float* __restrict__ w0, const float* __restrict__ w1, int Co, int Ci,
                                                             int KH, int KW, CwTaps tp, int ncg, int mode, int bn, int rows_img, int nks,
                                                             unsigned char* __restrict__ img) {
  cw_pack_filter_body((size_t)blockIdx.x * blockDim.x + threadIdx.x, w0, w1, Co, Ci, KH, KW, tp, ncg, mode, bn, rows_img, nks, img);
}
// The filters of a RUN of same-shaped plain layers (fully_conv's residual runs), forward (mode 2, blockIdx.z = 0) and data-gradient
// (mode 1, z = 1) images, in one launch: blockIdx.y = layer; images `stride` bytes apart behind fwd / bwd (either may be NULL)
struct CwPackSet {
  const float* w[16];
  unsigned char* fwd; unsigned char* bwd;
  size_t stride;
  int C, K, ncg, rows_img, nks;
  CwTaps tpf, tpb;
};
__global__ __launch_bounds__(256) void cw_pack_filter_set_kernel(const CwPackSet s) {
  const int j = blockIdx.y, dir = blockIdx.z;
  unsigned char* const base = dir ? s.bwd : s.fwd;
  if (!base) return;
  cw_pack_filter_body((size_t)blockIdx.x * blockDim.x + threadIdx.x, s.w[j], nullptr, s.C, s.C, s.K, s.K, dir ? s.tpb : s.tpf, s.ncg, dir ? 1 : 2, 64,
                      s.rows_img, s.nks, base + (size_t)j * s.stride);
}
// k-steps of a launch's contraction
static int cw_ksteps(const CwTaps& tp, int ncg) {
  int n = 0;
  for (int s = 0; s < tp.nseg; ++s) n += ncg * tp.s[s].ndy * tp.s[s].ndx;
  return n;
}
// the taps of a forward convolution (stride 1 or 2) / of the data gradient into input parity class (py, px)
static CwTaps cw_taps_fwd(int K, int stride, int pad, int* plo, int* phi) {
  CwTaps tp = {};
  tp.a = stride;
  int lo = 0, hi = 0;
  for (int py = 0; py < stride; ++py)
    for (int px = 0; px < stride; ++px) {
      // input row stride * (y + dy) + p = stride * y + kh - pad  ->  kh = stride * dy + p + pad
      auto range = [&](int p, int& d0, int& nd) {
        int dmin = 1 << 30, dmax = -(1 << 30);
        for (int k = 0; k < K; ++k) if ((k - pad - p) % stride == 0) { const int d = (k - pad - p) / stride; dmin = std::min(dmin, d); dmax = std::max(dmax, d); }
        d0 = dmin; nd = dmax >= dmin ? dmax - dmin + 1 : 0;
      };
      CwSegTaps& s = tp.s[tp.nseg];
      range(py, s.dy0, s.ndy); range(px, s.dx0, s.ndx);
      s.bh = py + pad; s.bw = px + pad;
      if (s.ndy > 0 && s.ndx > 0) {
        lo = std::max(lo, std::max(-s.dy0, -s.dx0)); hi = std::max(hi, std::max(s.dy0 + s.ndy - 1, s.dx0 + s.ndx - 1));
      } else { s.ndy = s.ndx = 0; s.dy0 = s.dx0 = 0; }
      ++tp.nseg;
    }
  *plo = lo; *phi = hi;
  return tp;
}
static CwTaps cw_taps_dgrad(int K, int stride, int pad, int py, int px, int* plo, int* phi) {
  CwTaps tp = {};
  tp.a = -stride; tp.nseg = 1;
  // output pixel y' = y + dy of the layer reads input row stride * y + p through tap kh = p + pad - stride * dy
  auto range = [&](int p, int& d0, int& nd) {
    int dmin = 1 << 30, dmax = -(1 << 30);
    for (int k = 0; k < K; ++k) if ((p + pad - k) % stride == 0) { const int d = (p + pad - k) / stride; dmin = std::min(dmin, d); dmax = std::max(dmax, d); }
    d0 = dmin; nd = dmax >= dmin ? dmax - dmin + 1 : 0;
  };
  CwSegTaps& s = tp.s[0];
  range(py, s.dy0, s.ndy); range(px, s.dx0, s.ndx);
  s.bh = py + pad; s.bw = px + pad;
  *plo = std::max(0, std::max(-s.dy0, -s.dx0)); *phi = std::max(0, std::max(s.dy0 + s.ndy - 1, s.dx0 + s.ndx - 1));
  return tp;
}

// (the two slot counts below walk every residue of a block start modulo the image -- thousands of iterations -- and every launch asks
// several times: remembered per thread; an eager fully_conv step is ~500 launches and waits for its host)
struct CwSlotMemo { int key[7]; int val; };
static inline bool cw_memo_get(CwSlotMemo (&tab)[32], int& n, const int (&key)[7], int* val) {
  for (int i = 0; i < n; ++i) {
    bool eq = true;
    for (int j = 0; j < 7; ++j) eq = eq && tab[i].key[j] == key[j];
    if (eq) { *val = tab[i].val; return true; }
  }
  return false;
}
static inline void cw_memo_put(CwSlotMemo (&tab)[32], int& n, const int (&key)[7], int val) {
  CwSlotMemo& e = tab[n < 32 ? n++ : (key[0] + key[4]) & 31];
  for (int j = 0; j < 7; ++j) e.key[j] = key[j];
  e.val = val;
}

// window slots a block of R consecutive output pixels needs, maximised over the block starts
static int cw_window_slots_walk(int H, int W, int plo, int phi, int R);
static int cw_window_slots(int H, int W, int plo, int phi, int R) {
  static thread_local CwSlotMemo tab[32];
  static thread_local int n = 0;
  const int key[7] = {H, W, plo, phi, R, 0, 0};
  int v;
  if (cw_memo_get(tab, n, key, &v)) return v;
  v = cw_window_slots_walk(H, W, plo, phi, R);
  cw_memo_put(tab, n, key, v);
  return v;
}
static int cw_window_slots_walk(int H, int W, int plo, int phi, int R) {
  const int PW = W + plo + phi, SP = (H + plo + phi) * PW, HW = H * W;
  int best = 0;
  for (int r0 = 0; r0 < HW; ++r0) {           // first pixel of a block, modulo the image (every residue: M need not be regular)
    const int y0 = r0 / W, x0 = r0 % W;
    const int last = r0 + R - 1;
    const int n1 = last / HW, r1 = last % HW, y1 = r1 / W, x1 = r1 % W;
    const int q0 = y0 * PW + x0, q1 = n1 * SP + (y1 + plo + phi) * PW + x1 + plo + phi;
    best = std::max(best, q1 - q0 + 1);
  }
  return best;
}

enum { CW_FWD_GATED = 0, CW_DGRAD_GATE = 1, CW_PLAIN = 2, CW_RES_FWD = 3, CW_RES_BWD = 4 };

// The epilogue of a block tile (R pixels x BN columns; acc in the matrix core's C layout, 4 waves as WR x WC, wave tile 64 x 32 NT),
// shared by the window kernels and the first-layer kernel; every wave of the block must have left its main loop (the staging
// reuses the loop's LDS).
// ELU for the image of a residual block's output: y > 0 ? y : e^y - 1, branch-free (the library expm1f is a divergent call per element):
// below -0.35 the hardware exponential minus one loses no more than ~3e-7 relative, above it the series (next term < 6e-9)
__device__ __forceinline__ float cw_elu(float y) {
  const float u = __builtin_amdgcn_exp2f(1.4426950408889634f * y) - 1.0f;
  const float p = y * (1.0f + y * (0.5f + y * (0.16666667f + y * (0.041666668f + y * (0.0083333338f + y * (0.0013888889f + y * 0.0001984127f))))));
  const float e = y < -0.35f ? u : p;
  return y > 0.f ? y : e;
}

// bias[]: the bias of this lane's column in each column tile (gated: h, g), loaded by the caller before its main loop; cok0: gated --
// this lane's column is a real channel
template <int EPI, int R, int BN, int NT, int WC>
__device__ __forceinline__ void cw_epilogue(const ConvWinArgs& g, f32x16 (&acc)[2][NT], const float (&bias)[2], const bool cok0, const int m0, const int tn,
                                            const int wr, const int wc, const int lane, const int tid, float* const smem) {
  constexpr int MT = 2;
  constexpr bool GATED = EPI == CW_FWD_GATED;
  const int l31 = lane & 31, lh = lane >> 5;
  const int HW = g.H * g.W;
  // ---- epilogue ---------------------------------------------------------------------------------------------------------------
  // The result leaves pixel-major: a lane of the matrix core's C layout owns one column (channel) and sixteen rows of a 32 x 32 tile,
  // the images want the channels of a pixel side by side -> through LDS as [row][BNO channels] fp32 (row pitch BNO + 4 floats), then
  // 16-byte accesses per (row, 8 channels) piece.
  constexpr int BNO = GATED ? BN / 2 : BN;      // result columns of the block
  constexpr int RP = BNO + 4;
  float* const so = smem;                       // [R][RP]
  float* const ss = smem + R * RP;              // [R][RP]   (gated: the gate)
  // pieces (row, 8 channels): thread -> row fastest inside 16 (the 16 pixels of an image chunk), then the 8-channel group, then the
  // 16-row groups: the 32 lanes of a (chunk, channel group) pair fill whole 512-byte chunks of the image.
  // What a piece reads from memory (residual / upstream gradient e_s, the saved image eimg) is requested for a GROUP of pieces at a
  // time, one group ahead of the group being finished, the first one before the accumulators are staged: a request's latency (and
  // that of the stores in front of it -- loads and stores retire in order) is paid once per block, not once per piece.
  constexpr int C8 = BNO / 8, NPC = R * C8 / 256, GP = NPC < 4 ? NPC : 4, NG = NPC / GP;
  static_assert(NPC % GP == 0, "pieces per thread: whole groups");
  constexpr bool RD_S = EPI == CW_RES_FWD || EPI == CW_RES_BWD || EPI == CW_DGRAD_GATE;
  constexpr bool RD_I = EPI == CW_RES_BWD || EPI == CW_DGRAD_GATE;
  unsigned pm[NPC], pmn[NPC];                // rows of piece i: in the pixel images / in the fp32 tensors
  bool pok[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int pid = tid + 256 * i;
    const int r16 = pid & 15, c8 = (pid >> 4) % C8, rg = pid / (16 * C8);
    const int mm = m0 + rg * 16 + r16;
    pok[i] = mm < g.M && tn * BNO + c8 * 8 < g.Co;
    const unsigned mc = pok[i] ? (unsigned)mm : 0u;               // (an idle piece reads pixel 0: a valid address, nothing stored)
    const unsigned n = fdiv(mc, g.div_hw), rem = mc - n * (unsigned)HW;
    const unsigned y = fdiv(rem, g.div_w), x = rem - y * (unsigned)g.W;
    pm[i] = g.out_planar ? n * (unsigned)HW + (unsigned)cw_planar((int)y, (int)x, g.H, g.W) : n * (unsigned)g.ostride + (unsigned)g.ooff + rem;
    pmn[i] = (n * (unsigned)g.nat_h + y * (unsigned)g.nat_s + (unsigned)g.nat_y) * (unsigned)g.nat_w + x * (unsigned)g.nat_s + (unsigned)g.nat_x;
  }
  float4 es[2][GP][2];
  uint4 ei[2][GP][3];
  auto request = [&](auto set_, int gi) {
    constexpr int set = decltype(set_)::value;
#pragma unroll
    for (int j = 0; j < GP; ++j) {
      const int i = gi * GP + j, pid = tid + 256 * i, c8 = (pid >> 4) % C8;
      const int ch = pok[i] ? tn * BNO + c8 * 8 : 0;
      if constexpr (RD_S) {
        const float* rp = g.e_s + (size_t)pmn[i] * g.Co + ch;
        es[set][j][0] = *reinterpret_cast<const float4*>(rp); es[set][j][1] = *reinterpret_cast<const float4*>(rp + 4);
      }
      if constexpr (RD_I) {
        const unsigned char* e = g.eimg + p6_off64(pm[i], g.ech0 + ch, g.nks_e);
        ei[set][j][0] = *reinterpret_cast<const uint4*>(e); ei[set][j][1] = *reinterpret_cast<const uint4*>(e + P6_CHUNK);
        ei[set][j][2] = *reinterpret_cast<const uint4*>(e + 2 * P6_CHUNK);
      }
    }
  };
  constexpr std::integral_constant<int, 0> S0{};
  constexpr std::integral_constant<int, 1> S1{};
  if constexpr (RD_S || RD_I) request(S0, 0);

  if constexpr (GATED) {
    const int cl = wc * 32 + l31;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float h = acc[mt][0][r] + bias[0];
        const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * (acc[mt][1][r] + bias[1])));
        so[row * RP + cl] = cok0 ? h * s : 0.f;
        ss[row * RP + cl] = s;
      }
  } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int cl = wc * 32 * NT + nt * 32 + l31;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          so[row * RP + cl] = acc[mt][nt][r] + bias[nt];
        }
    }
  }
  __syncthreads();

  auto finish = [&](auto set_, int gi) {
    constexpr int set = decltype(set_)::value;
#pragma unroll
    for (int j = 0; j < GP; ++j) {
      const int i = gi * GP + j, pid = tid + 256 * i;
      const int r16 = pid & 15, c8 = (pid >> 4) % C8, rg = pid / (16 * C8);
      const int row = rg * 16 + r16;
      const int ch = tn * BNO + c8 * 8;                         // first of the eight result columns
      const size_t m = pm[i], mn = pmn[i];
      const float4 o0 = *reinterpret_cast<const float4*>(so + row * RP + c8 * 8), o1 = *reinterpret_cast<const float4*>(so + row * RP + c8 * 8 + 4);
      auto put_img = [&](int chan, const float4& a, const float4& b) {
        unsigned t0[4], t1[4], t2[4];
        p6_split2(a.x, a.y, t0[0], t1[0], t2[0]); p6_split2(a.z, a.w, t0[1], t1[1], t2[1]);
        p6_split2(b.x, b.y, t0[2], t1[2], t2[2]); p6_split2(b.z, b.w, t0[3], t1[3], t2[3]);
        unsigned char* o = g.oimg + p6_off64(m, chan, g.nks_o);
        *reinterpret_cast<uint4*>(o) = make_uint4(t0[0], t0[1], t0[2], t0[3]);
        *reinterpret_cast<uint4*>(o + P6_CHUNK) = make_uint4(t1[0], t1[1], t1[2], t1[3]);
        *reinterpret_cast<uint4*>(o + 2 * P6_CHUNK) = make_uint4(t2[0], t2[1], t2[2], t2[3]);
      };
      // element k of the saved image's piece: the sum of its three bf16 terms, smallest first (reproduces the fp32 value they were split from)
      auto img_val = [&](int k) -> float {
        const unsigned w0[4] = {ei[set][j][0].x, ei[set][j][0].y, ei[set][j][0].z, ei[set][j][0].w};
        const unsigned w1[4] = {ei[set][j][1].x, ei[set][j][1].y, ei[set][j][1].z, ei[set][j][1].w};
        const unsigned w2[4] = {ei[set][j][2].x, ei[set][j][2].y, ei[set][j][2].z, ei[set][j][2].w};
        const int sh = 16 * (k & 1);
        return (__uint_as_float(((w2[k >> 1] >> sh) & 0xFFFFu) << 16) + __uint_as_float(((w1[k >> 1] >> sh) & 0xFFFFu) << 16)) +
               __uint_as_float(((w0[k >> 1] >> sh) & 0xFFFFu) << 16);
      };
      const float v[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
      const float rv[8] = {es[set][j][0].x, es[set][j][0].y, es[set][j][0].z, es[set][j][0].w, es[set][j][1].x, es[set][j][1].y, es[set][j][1].z, es[set][j][1].w};
      if (!pok[i]) continue;
      if constexpr (GATED) {
        const float4 s0 = *reinterpret_cast<const float4*>(ss + row * RP + c8 * 8), s1 = *reinterpret_cast<const float4*>(ss + row * RP + c8 * 8 + 4);
        if (g.oimg) put_img(g.och0 + ch, o0, o1);
        if (g.out_s) {
          float* sp = g.out_s + mn * g.Co + ch;
          *reinterpret_cast<float4*>(sp) = s0; *reinterpret_cast<float4*>(sp + 4) = s1;
        }
        if (g.out_f) {
          float* op = g.out_f + mn * g.ldo + ch;
          *reinterpret_cast<float4*>(op) = o0; *reinterpret_cast<float4*>(op + 4) = o1;
        }
      } else if constexpr (EPI == CW_PLAIN) {
        // plain convolution (+ ELU: the weight-normed convolutions in front of fully_conv's residual runs, models/fully_conv.py:41-58)
        float y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = (g.act & 1) ? cw_elu(v[k]) : v[k];
        if (g.out_f) {
          float* op = g.out_f + mn * g.ldo + ch;
          *reinterpret_cast<float4*>(op) = make_float4(y[0], y[1], y[2], y[3]); *reinterpret_cast<float4*>(op + 4) = make_float4(y[4], y[5], y[6], y[7]);
        }
        if (g.oimg) {
          if (g.act & 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] = cw_elu(y[k]);
          }
          put_img(g.och0 + ch, make_float4(y[0], y[1], y[2], y[3]), make_float4(y[4], y[5], y[6], y[7]));
        }
      } else if constexpr (EPI == CW_RES_FWD) {
        // residual block of models/fully_conv.py:13-23: y = x + conv(ELU(x)) + b.  e_s = x (fp32, natural rows); out_f = y; oimg = the
        // image of ELU(y): the next block's convolution operand AND what its backward derives ELU'(y) from
        float y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = v[k] + rv[k];
        if (g.out_f) {
          float* op = g.out_f + mn * g.ldo + ch;
          *reinterpret_cast<float4*>(op) = make_float4(y[0], y[1], y[2], y[3]); *reinterpret_cast<float4*>(op + 4) = make_float4(y[4], y[5], y[6], y[7]);
        }
        if (g.oimg) {
          float a[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) a[k] = cw_elu(y[k]);
          put_img(g.och0 + ch, make_float4(a[0], a[1], a[2], a[3]), make_float4(a[4], a[5], a[6], a[7]));
        }
      } else if constexpr (EPI == CW_RES_BWD) {
        // its data gradient: dx = dy + ELU'(x) * conv_transpose(dy, w), ELU'(x) = (a > 0 ? 1 : a + 1) with a = ELU(x) summed back from
        // its image (eimg); e_s = dy (fp32, natural rows); out_f = dx, oimg = the image of dx (the next data / weight gradients' operand)
        float dx[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float av = img_val(k);
          dx[k] = rv[k] + (av > 0.f ? 1.0f : av + 1.0f) * v[k];
        }
        if (g.out_f) {
          float* op = g.out_f + mn * g.ldo + ch;
          *reinterpret_cast<float4*>(op) = make_float4(dx[0], dx[1], dx[2], dx[3]); *reinterpret_cast<float4*>(op + 4) = make_float4(dx[4], dx[5], dx[6], dx[7]);
        }
        if (g.oimg) put_img(g.och0 + ch, make_float4(dx[0], dx[1], dx[2], dx[3]), make_float4(dx[4], dx[5], dx[6], dx[7]));
      } else {
        // gate derivative of the layer below at (pixel m, channels ch .. ch + 7): out = the sum of its image's three terms (exact), s
        // fp32 (e_s); dh = v s, dg = v out (1 - s)   (reference utils/nn.py:92-97 under autograd)
        float dh[8], dg[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float ov = img_val(k);
          dh[k] = v[k] * rv[k];
          dg[k] = v[k] * ov * (1.0f - rv[k]);
        }
        if (g.oimg) {
          put_img(g.och0 + ch, make_float4(dh[0], dh[1], dh[2], dh[3]), make_float4(dh[4], dh[5], dh[6], dh[7]));
          put_img(g.och0 + g.Co + ch, make_float4(dg[0], dg[1], dg[2], dg[3]), make_float4(dg[4], dg[5], dg[6], dg[7]));
        }
        if (g.out_f) {
          float* op = g.out_f + mn * g.ldo + ch;
          *reinterpret_cast<float4*>(op) = make_float4(dh[0], dh[1], dh[2], dh[3]); *reinterpret_cast<float4*>(op + 4) = make_float4(dh[4], dh[5], dh[6], dh[7]);
          *reinterpret_cast<float4*>(op + g.Co) = make_float4(dg[0], dg[1], dg[2], dg[3]); *reinterpret_cast<float4*>(op + g.Co + 4) = make_float4(dg[4], dg[5], dg[6], dg[7]);
        }
      }
    }
  };
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    if (gi & 1) { if constexpr (RD_S || RD_I) { if (gi + 1 < NG) request(S0, gi + 1); } finish(S1, gi); }
    else { if constexpr (RD_S || RD_I) { if (gi + 1 < NG) request(S1, gi + 1); } finish(S0, gi); }
  }
}

// the bias of this lane's columns (see cw_epilogue), requested before the main loop
template <int EPI, int BN, int NT, int WC>
__device__ __forceinline__ void cw_load_bias(const ConvWinArgs& g, int tn, int wc, int lane, float (&bias)[2], bool& cok0) {
  const int l31 = lane & 31;
  bias[0] = bias[1] = 0.f; cok0 = true;
  if constexpr (EPI == CW_FWD_GATED) {
    const int c = tn * (BN / 2) + wc * 32 + l31;
    cok0 = c < g.Co;
    if (g.bias0 && cok0) bias[0] = g.bias0[c];
    if (g.bias1 && cok0) bias[1] = g.bias1[c];
  } else if constexpr (EPI == CW_PLAIN || EPI == CW_RES_FWD) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int c = tn * BN + wc * 32 * NT + nt * 32 + l31;
      if (g.bias0 && c < g.Co) bias[nt] = g.bias0[c];
    }
  }
}

// EPI: CW_FWD_GATED: [h | g] column pairs (BN / 2 gated outputs per column tile), result -> pixel image + gate (+ fp32 copy);
//      CW_PLAIN: BN plain columns, fp32 result (+ bias) and / or its pixel image;
//      CW_DGRAD_GATE: BN plain columns = the channels of the layer below, gate derivative of that layer in the epilogue,
//      [dh | dg] -> pixel image (+ fp32 copy)
template <int EPI, int WR, int NT, int SLOTS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_win_kernel(const ConvWinArgs g) {
  typedef CwGeom<WR, NT, SLOTS> G;
  constexpr int MT = 2, WC = G::WC, R = G::R, BN = G::BN, WP = G::WP, WIN = G::WIN, BKS = G::BKS, BST = G::BST, KST = G::KST;
  constexpr bool GATED = EPI == CW_FWD_GATED;
  static_assert(!GATED || NT == 2, "gated: a wave holds the h and the g column tile of its 32 channels");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < 4);
  const int wr = wave / WC, wc = wave - wr * WC, l31 = lane & 31, lh = lane >> 5;
  // XCD-aware tile order: XCD x (= block id mod 8) works on a contiguous run of tiles (neighbouring windows share halo rows)
  const int ntiles = gridDim.x;
  int tile;
  {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int qq = ntiles >> 3, rr = ntiles & 7;
    tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + slot;
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * R;
  const int HW = g.H * g.W;
  // Blocks that start together reach their epilogues together: the memory system then sees the stores of all 512 resident blocks
  // at once while every matrix pipe idles, and the two blocks of a CU cannot cover each other.  The first resident blocks therefore
  // start in four groups a fraction of a block's duration apart (measured on the residual blocks of fully_conv: -10 .. -16 %).
  if (g.stagger > 0 && blockIdx.x < 512) {
    const int nsl = (int)(blockIdx.x >> 7) * g.stagger;
    for (int i = 0; i < nsl; ++i) __builtin_amdgcn_s_sleep(32);       // 2048 clocks a unit
  }

  // first pixel of the block -> base slot; base pixel of the buffer resource (16-pixel aligned, at or before every pixel a tap reads)
  const unsigned nf = fdiv((unsigned)m0, g.div_hw), remf = (unsigned)m0 - nf * (unsigned)HW;
  const unsigned yf = fdiv(remf, g.div_w), xf = remf - yf * (unsigned)g.W;
  const int qbase = (int)(nf * (unsigned)g.SP + yf * (unsigned)g.PW + xf);
  // base row of the buffer resource over the input image: the first row of the block's first image (16-row aligned); every row a
  // tap reads lies at or behind it
  const int base_pix = __builtin_amdgcn_readfirstlane((int)((nf * (unsigned)g.istride) & ~15u));
  const rsrc_t rB = make_rsrc(g.wimg + (size_t)tn * (BN / 16) * g.nks_w * P6_GROUP, 0x7FFFFFFFu);

  float bias[2];
  bool cok0;
  cw_load_bias<EPI, BN, NT, WC>(g, tn, wc, lane, bias, cok0);
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // this lane's rows: window slot of tap (0, 0)
  int sl[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = m0 + wr * 64 + mt * 32 + l31;
    m = m < g.M ? m : g.M - 1;
    const unsigned n = fdiv((unsigned)m, g.div_hw), rem = (unsigned)m - n * (unsigned)HW;
    const unsigned y = fdiv(rem, g.div_w), x = rem - y * (unsigned)g.W;
    sl[mt] = (int)(n * (unsigned)g.SP + y * (unsigned)g.PW + x) - qbase;
  }
  // B fragment addresses: filter row c of the column tile (the image's own half swap), behind the window
  unsigned fb[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = wc * 32 * NT + nt * 32 + l31;
    fb[nt] = (unsigned)(WIN + c * 32 + ((lh ^ ((c >> 3) & 1)) << 4));
  }
  const unsigned grp = (unsigned)g.nks_w * P6_GROUP;
  const unsigned voff_b = (unsigned)lh * grp + (unsigned)l31 * 16u;
  // ring stage `buf` <- the k-steps ks0 .. ks0 + nk - 1 (nk <= KST: an integral_constant in the steady state, so that no branch
  // surrounds a copy there): 1 KB pieces id = (kk * 3 + plane) * (BN / 32) + j (j = pair of 16-row groups), id = wave + 4 q to wave `wave`
  auto issue_b = [&](int ks0, int buf, auto nk_) {
    const int nk = nk_;
#pragma unroll
    for (int q = 0; q < G::NPIECE / 4; ++q) {
      const int id = wave + 4 * q;
      const int j = id % (BN / 32), kp = id / (BN / 32), kk = kp / 3, p = kp - kk * 3;
      if (kk < nk)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (p6_lds_t)(lds + WIN + buf * BST + kk * BKS + p * (BN * 32) + j * 1024), 16, voff_b,
                                                 (unsigned)(ks0 + kk) * (unsigned)P6_GROUP + (unsigned)(2 * j) * grp + p * P6_CHUNK, 0, 0);
    }
  };

  p6_bf16x8 af[2][MT][3], bf[2][NT][3];
  auto read_a = [&](auto par_, unsigned ab, int mt, int p) {
    constexpr int par = decltype(par_)::value;
    af[par][mt][p] = *reinterpret_cast<const p6_bf16x8*>(lds + ab + p * WP);
  };
  auto read_b = [&](auto par_, int buf, int kk, int nt, int p) {
    constexpr int par = decltype(par_)::value;
    bf[par][nt][p] = *reinterpret_cast<const p6_bf16x8*>(lds + fb[nt] + buf * BST + kk * BKS + p * (BN * 32));
  };
  // window byte address of this lane's fragment of row tile mt at tap offset `to` (slots)
  auto a_addr = [&](int mt, int to) -> unsigned {
    const int a = sl[mt] + to;
    return (unsigned)(a * 32 + ((((a >> 3) & 1) ^ lh) << 4));
  };

  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};       // smallest partial products first
  constexpr std::integral_constant<int, 0> I0{};
  constexpr std::integral_constant<int, 1> I1{};
  constexpr int NMF = 6 * MT * NT, NRD = 3 * (MT + NT);
#define EVAE_CW_SB __builtin_amdgcn_sched_barrier(0)
  // the MFMAs of one k-step on fragment set `par`; READ: the reads of the next k-step's set behind the first MFMAs (A at window
  // addresses an[mt], B from ring stage nbuf, k-step nkk of it); ISSUE: the copies of the nk k-steps from ks_issue on into ring stage
  // ibuf behind MFMA 1.  READ / ISSUE (and nk in the steady state) are compile-time: no branch around a memory instruction inside the
  // MFMA sequence.
  auto mma_phase = [&](auto par_, auto read_, auto issue_, auto nk_, const unsigned (&an)[MT], int nbuf, int nkk, int ks_issue, int ibuf) {
    constexpr int par = decltype(par_)::value;
    constexpr bool READ = decltype(read_)::value, ISSUE = decltype(issue_)::value;
    constexpr std::integral_constant<int, par ^ 1> npar{};
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int j = (t * MT + mt) * NT + nt;
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[par][mt][PA[t]], bf[par][nt][PB[t]], acc[mt][nt], 0, 0, 0);
          EVAE_CW_SB;
          if constexpr (READ) {
            constexpr int per = NMF / NRD >= 2 ? 2 : 1;         // a read behind every (second) MFMA, from the first one on
            if (j % per == 0 && j / per < NRD) {
              const int k = j / per, p = k / (MT + NT), q = k - p * (MT + NT);
              if (q < MT) read_a(npar, an[q], q, p); else read_b(npar, nbuf, nkk, q - MT, p);
            }
            EVAE_CW_SB;
          }
          if constexpr (ISSUE) {
            if (j == 1) { issue_b(ks_issue, ibuf, nk_); EVAE_CW_SB; }
          }
        }
    __builtin_amdgcn_s_setprio(0);
  };

  constexpr std::true_type T{};
  constexpr std::false_type F{};
  int ksc = 0;                                   // first k-step of the current (segment, channel group)
  bool first = true;
  for (int sg = 0; sg < g.taps.nseg; ++sg) {
   const int S = g.taps.s[sg].ndy * g.taps.s[sg].ndx;          // k-steps per channel group: the segment's taps
   const int ndx = g.taps.s[sg].ndx;
   const int to0 = (g.taps.s[sg].dy0 + g.plo) * g.PW + g.taps.s[sg].dx0 + g.plo;
   if (S == 0) continue;
   for (int cgi = 0; cgi < g.ncg; ++cgi, ksc += S) {
    if (!first) __syncthreads();                 // every wave is done with the previous window and ring
    first = false;
    issue_b(ksc, 0, S < KST ? S : KST);
    if (S > KST) issue_b(ksc + KST, 1, S - KST < KST ? S - KST : KST);
    // ---- window of this channel group: slot pieces j = wave, wave + 4, ... (32 slots each), one copy per plane ----
    {
      const rsrc_t rA = make_rsrc(g.xin + ((size_t)(base_pix >> 4) * g.nks_in + (size_t)(g.cg0 + cgi)) * P6_GROUP, 0x7FFFFFFFu);
#pragma unroll
      for (int jj = 0; jj < (SLOTS / 32 + 3) / 4; ++jj) {
        const int j = wave + 4 * jj;
#ifdef EVAE_CW_ABL
        if (j < g.nsp && !((g.dbg & 8) && ksc > 0)) {          // (tools: only the first window of a block is copied)
#else
        if (j < g.nsp) {
#endif
          const int s = 32 * j + (lane >> 1), hp = lane & 1;
          const unsigned q = (unsigned)(qbase + s);
          const unsigned n = fdiv(q, g.div_sp), r = q - n * (unsigned)g.SP;
          const unsigned py = fdiv(r, g.div_pw), px = r - py * (unsigned)g.PW;
          const int y = (int)py - g.plo, x = (int)px - g.plo;
          const bool ok = (int)n < g.N && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
          const int pix = (int)n * g.istride + (g.in_planar ? cw_planar(y, x, g.H, g.W) : g.ioff[sg] + y * g.W + x);
          const int rel = pix - base_pix;
          const unsigned gh = (unsigned)(hp ^ ((s >> 3) & 1) ^ ((pix >> 3) & 1));
          unsigned vin = (unsigned)(rel >> 4) * (unsigned)(g.nks_in * P6_GROUP) + (unsigned)((rel & 15) * 32) + (gh << 4);
          EVAE_PIN(vin);                                             // (computed by every lane: a select, not a branch)
          const unsigned voff = ok ? vin : 0x80000000u;
#pragma unroll
          for (int p = 0; p < 3; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (p6_lds_t)(lds + p * WP + j * 1024), 16, voff, (unsigned)(p * P6_CHUNK), 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int kw = 0, to = to0;                        // tap column, tap offset in slots
    unsigned an[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) an[mt] = a_addr(mt, to0);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) read_a(I0, an[mt], mt, p);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) read_b(I0, 0, 0, nt, p);
    }
    auto next_tap = [&]() {                      // branch-free (scalar selects)
      const bool wrap = kw + 1 == ndx;
      kw = wrap ? 0 : kw + 1;
      to += wrap ? g.PW - ndx + 1 : 1;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) an[mt] = a_addr(mt, to);
    };
    // Ring stage st (k-steps KST st .. KST st + KST - 1) lives in buffer st & 1; k-step s computes on fragment set s & 1 while the
    // set of k-step s + 1 is read.  In front of a stage's LAST k-step: this wave's copies of stage st + 1 have landed (vmcnt(0));
    // barrier: everybody's have, and everybody holds the last k-step's fragments, so the stage's own buffer is free.  The last
    // k-step: the copies of stage st + 2 into that buffer, its MFMAs with the reads of stage st + 1's first k-step.  A stage's
    // copies are in flight for KST k-steps of MFMAs; the two blocks of a CU cover each other's waits.
    // stage_full: stage st + 2 lies entirely inside the group (KST (st + 3) <= S) -- branch-free, the steady state
    constexpr std::integral_constant<int, KST> NK{};
    auto stage_full = [&](auto buf_, int st) {
      constexpr int buf = decltype(buf_)::value;
#pragma unroll
      for (int i = 0; i < KST; ++i) {
        if (i == KST - 1) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          next_tap();
          if (i & 1) mma_phase(I1, T, T, NK, an, buf ^ 1, 0, ksc + KST * (st + 2), buf);
          else mma_phase(I0, T, T, NK, an, buf ^ 1, 0, ksc + KST * (st + 2), buf);
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          next_tap();
          if (i & 1) mma_phase(I1, T, F, NK, an, buf, i + 1, 0, 0);
          else mma_phase(I0, T, F, NK, an, buf, i + 1, 0, 0);
        }
      }
    };
    auto stage_tail = [&](auto buf_, int st) {
      constexpr int buf = decltype(buf_)::value;
      const int s0 = KST * st;
#pragma unroll
      for (int i = 0; i < KST; ++i) {
        const int sk = s0 + i;
        if (sk >= S) return;
        const bool rd = sk + 1 < S;
        if (i == KST - 1) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          next_tap();
          int nk = S - KST * (st + 2);
          nk = nk < 0 ? 0 : (nk > KST ? KST : nk);
          if (i & 1) { if (rd) mma_phase(I1, T, T, nk, an, buf ^ 1, 0, ksc + KST * (st + 2), buf); else mma_phase(I1, F, F, 0, an, 0, 0, 0, 0); }
          else { if (rd) mma_phase(I0, T, T, nk, an, buf ^ 1, 0, ksc + KST * (st + 2), buf); else mma_phase(I0, F, F, 0, an, 0, 0, 0, 0); }
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          next_tap();
          if (i & 1) { if (rd) mma_phase(I1, T, F, 0, an, buf, i + 1, 0, 0); else mma_phase(I1, F, F, 0, an, 0, 0, 0, 0); }
          else { if (rd) mma_phase(I0, T, F, 0, an, buf, i + 1, 0, 0); else mma_phase(I0, F, F, 0, an, 0, 0, 0, 0); }
        }
      }
    };
    const int nst = (S + KST - 1) / KST;
    int st = 0;
    while (KST * (st + 4) <= S) { stage_full(I0, st); stage_full(I1, st + 1); st += 2; }
    for (; st + 1 < nst; st += 2) { stage_tail(I0, st); stage_tail(I1, st + 1); }
    if (st < nst) stage_tail(I0, st);
   }
  }
#undef EVAE_CW_SB
  __syncthreads();                 // the epilogue stages through the window's LDS
  if (g.dbg & 4) return;

  cw_epilogue<EPI, R, BN, NT, WC>(g, acc, bias, cok0, m0, tn, wr, wc, lane, tid, smem);
}

template <int EPI, int WR, int NT, int SLOTS>
static int launch_conv_win(ConvWinArgs& g, hipStream_t stream, const char* what) {
  typedef CwGeom<WR, NT, SLOTS> G;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_win_kernel<EPI, WR, NT, SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, G::lds_bytes(EPI == CW_FWD_GATED));
    attr_done = true;
  }
  g.PW = g.W + g.plo + g.phi; g.SP = (g.H + g.plo + g.phi) * g.PW;
  g.div_w = make_fastdiv((unsigned)g.W); g.div_hw = make_fastdiv((unsigned)(g.H * g.W));
  g.div_pw = make_fastdiv((unsigned)g.PW); g.div_sp = make_fastdiv((unsigned)g.SP);
  if (g.istride == 0) g.istride = g.H * g.W;
  if (g.ostride == 0) g.ostride = g.H * g.W;
  if (g.nat_s == 0) { g.nat_s = 1; g.nat_h = g.H; g.nat_w = g.W; g.nat_y = g.nat_x = 0; }
  g.M = g.N * g.H * g.W;
  const int tiles_m = cdiv(g.M, G::R);
  if (g.stagger == 0 && tiles_m * g.tiles_n > 640) { const int u = g.nks_w / 8; g.stagger = u < 1 ? 1 : (u > 8 ? 8 : u); }
  if (g.stagger < 0) g.stagger = 0;
  conv_win_kernel<EPI, WR, NT, SLOTS><<<dim3(tiles_m * g.tiles_n), 256, G::lds_bytes(EPI == CW_FWD_GATED), stream>>>(g);
  return check_launch(what);
}


// =================================================================================================================================
// The FIRST layer of a stack (one input channel: the data; reference models/convHVAE_2level.py:21-27, GatedConv2d(1, 32, 7, 1, 3)): a
// contraction over <= 50 taps is no work for the bf16 pipe -- the layer is bound by the 10 bytes per output element it writes -- so
// it runs on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32: exact fp32, two taps per instruction) straight from an fp32
// window of the input pixels in LDS (the LDS-staged im2col of north_star; no patch matrix), the filters in registers, and leaves
// through the window kernels' epilogue: pixel image (parity-planar when the next layer has stride 2) + gate.
// Block = 256 output pixels x 64 columns [h 32 | g 32], four waves of 64 pixels; two blocks per CU.
// =================================================================================================================================
struct ConvFirstArgs {
  ConvWinArgs e;              // geometry (N, H, W, plo = phi = pad, PW, SP, fastdivs, M, Co, tiles_n, biases) and everything the epilogue reads
  const float* x;             // [N][H][W] fp32
  const float* w0;            // [Co][1][K][K]
  const float* w1;
  int K;
};

template <int SLOTS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_first_kernel(const ConvFirstArgs a) {
  const ConvWinArgs& g = a.e;
  constexpr int R = 256, MT = 2, NT = 2, NKS = 25;        // <= 50 taps
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int ntiles = gridDim.x;
  int tile;
  {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int qq = ntiles >> 3, rr = ntiles & 7;
    tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + slot;
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * R, HW = g.H * g.W, K = a.K, taps = K * K;
  const unsigned nf = fdiv((unsigned)m0, g.div_hw), remf = (unsigned)m0 - nf * (unsigned)HW;
  const unsigned yf = fdiv(remf, g.div_w), xf = remf - yf * (unsigned)g.W;
  const int qbase = (int)(nf * (unsigned)g.SP + yf * (unsigned)g.PW + xf);
  // window: fp32 slots qbase .. qbase + SLOTS - 1 (zero outside the images)
  for (int s = tid; s < SLOTS; s += 256) {
    const unsigned q = (unsigned)(qbase + s);
    const unsigned n = fdiv(q, g.div_sp), r = q - n * (unsigned)g.SP;
    const unsigned py = fdiv(r, g.div_pw), px = r - py * (unsigned)g.PW;
    const int y = (int)py - g.plo, x = (int)px - g.plo;
    const bool ok = (int)n < g.N && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
    smem[s] = ok ? a.x[((size_t)n * g.H + y) * g.W + x] : 0.f;
  }
  // filters: k-step ks = taps 2 ks, 2 ks + 1 (this lane: tap 2 ks + lh), column l31 of bank nt
  float wv[NT][NKS];
  {
    const int co = tn * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int t = 2 * ks + lh;
      const bool ok = t < taps && co < g.Co;
      wv[0][ks] = ok ? a.w0[(size_t)co * taps + t] : 0.f;
      wv[1][ks] = ok ? a.w1[(size_t)co * taps + t] : 0.f;
    }
  }
  int sl[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = m0 + wave * 64 + mt * 32 + l31;
    m = m < g.M ? m : g.M - 1;
    const unsigned n = fdiv((unsigned)m, g.div_hw), rem = (unsigned)m - n * (unsigned)HW;
    const unsigned y = fdiv(rem, g.div_w), x = rem - y * (unsigned)g.W;
    sl[mt] = (int)(n * (unsigned)g.SP + y * (unsigned)g.PW + x) - qbase;
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  __syncthreads();
  // this lane's tap (kh, kw): starts at tap lh, advances by two per k-step; taps beyond the filter read slot offset 0 (weight 0)
  int kh = 0, kw = lh;
  if (kw >= K) { kw -= K; ++kh; }
  float av[2][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) av[0][mt] = smem[sl[mt] + kh * g.PW + kw];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int par = ks & 1;
    if (ks + 1 < NKS) {
      kw += 2;
      if (kw >= K) { kw -= K; ++kh; }
      if (kw >= K) { kw -= K; ++kh; }            // (K = 1)
      const int off = kh < K ? kh * g.PW + kw : 0;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[par ^ 1][mt] = smem[sl[mt] + off];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[par][mt], wv[nt][ks], acc[mt][nt], 0, 0, 0);
  }
  __syncthreads();
  float bias[2];
  bool cok0;
  cw_load_bias<CW_FWD_GATED, 64, NT, 1>(g, tn, 0, lane, bias, cok0);
  cw_epilogue<CW_FWD_GATED, R, 64, NT, 1>(g, acc, bias, cok0, m0, tn, wave, 0, lane, tid, smem);
}

template <int SLOTS>
static int launch_conv_first(ConvFirstArgs& a, hipStream_t stream, const char* what) {
  ConvWinArgs& g = a.e;
  constexpr int LDS = 2 * 256 * (32 + 4) * 4 > SLOTS * 4 ? 2 * 256 * (32 + 4) * 4 : SLOTS * 4;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_first_kernel<SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  g.PW = g.W + g.plo + g.phi; g.SP = (g.H + g.plo + g.phi) * g.PW;
  g.div_w = make_fastdiv((unsigned)g.W); g.div_hw = make_fastdiv((unsigned)(g.H * g.W));
  g.div_pw = make_fastdiv((unsigned)g.PW); g.div_sp = make_fastdiv((unsigned)g.SP);
  g.ostride = g.H * g.W;
  g.nat_s = 1; g.nat_h = g.H; g.nat_w = g.W; g.nat_y = g.nat_x = 0;
  g.M = g.N * g.H * g.W;
  conv_first_kernel<SLOTS><<<dim3(cdiv(g.M, 256) * g.tiles_n), 256, LDS, stream>>>(a);
  return check_launch(what);
}

// Weight gradient of the first layer: dW[cc][tap] = sum over pixels of dy[pixel][cc] * x[pixel + tap], db[cc] = sum of dy -- dy the
// merged fp32 gradient [N H W][CC] (CC = 2 Co <= 64) the data gradient of the layer above wrote.  fp32 matrix instruction again:
// rows = the 64 merged channels (A = dy^T straight from global memory: 32 lanes x 4 bytes of one pixel's channels per load), columns
// = the taps (B from the fp32 window in LDS; column `taps` reads a constant 1: the bias gradient), two pixels per instruction.
// A block walks a contiguous run of 256-pixel stages (split contraction), its four waves 64 pixels each; the waves' sums meet in
// LDS, the blocks' partial [64][64] planes in cw_first_wgrad_finish_kernel (block order: deterministic).
struct ConvFirstWgradArgs {
  ConvWinArgs e;              // geometry
  const float* x;             // [N][H][W]
  const float* dy;            // [N H W][CC]
  int CC, K;
  int cper, nstage;
  float* part;                // [blocks][64][64]
};

template <int SLOTS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_first_wgrad_kernel(const ConvFirstWgradArgs a) {
  const ConvWinArgs& g = a.e;
  constexpr int R = 256, MT = 2, NT = 2;
  __shared__ float win[2][SLOTS];
  __shared__ int stab[2][R];
  __shared__ float red[4][64][17];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int HW = g.H * g.W, K = a.K, taps = K * K;
  const int c0 = blockIdx.x * a.cper, c1 = min(c0 + a.cper, a.nstage);
  // this lane's B columns: taps l31 and 32 + l31 -> slot offsets; column == taps: the ones column (bias gradient)
  int toff[NT]; bool tone[NT], tok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int t = nt * 32 + l31;
    tok[nt] = t < taps; tone[nt] = t == taps;
    toff[nt] = tok[nt] ? (t / K) * g.PW + t % K : 0;
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto load_stage = [&](int c, int buf) {
    const int p0 = c * R;
    const unsigned nf = fdiv((unsigned)p0, g.div_hw), remf = (unsigned)p0 - nf * (unsigned)HW;
    const unsigned yf = fdiv(remf, g.div_w), xf = remf - yf * (unsigned)g.W;
    const int qbase = (int)(nf * (unsigned)g.SP + yf * (unsigned)g.PW + xf);
    for (int s = tid; s < SLOTS; s += 256) {
      const unsigned q = (unsigned)(qbase + s);
      const unsigned n = fdiv(q, g.div_sp), r = q - n * (unsigned)g.SP;
      const unsigned py = fdiv(r, g.div_pw), px = r - py * (unsigned)g.PW;
      const int y = (int)py - g.plo, x = (int)px - g.plo;
      const bool ok = (int)n < g.N && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
      win[buf][s] = ok ? a.x[((size_t)n * g.H + y) * g.W + x] : 0.f;
    }
    {
      int m = p0 + tid;
      m = m < g.M ? m : g.M - 1;
      const unsigned n = fdiv((unsigned)m, g.div_hw), rem = (unsigned)m - n * (unsigned)HW;
      const unsigned y = fdiv(rem, g.div_w), x = rem - y * (unsigned)g.W;
      stab[buf][tid] = (int)(n * (unsigned)g.SP + y * (unsigned)g.PW + x) - qbase;
    }
  };
  if (c0 < c1) load_stage(c0, 0);
  for (int c = c0; c < c1; ++c) {
    const int buf = (c - c0) & 1;
    __syncthreads();                                   // stage c is in LDS; everybody is done with the other buffer
    if (c + 1 < c1) load_stage(c + 1, buf ^ 1);
    const int p0 = c * R + wave * 64;                  // this wave's 64 pixels = 32 k-steps of two
    constexpr int UN = 8;
#pragma unroll 1
    for (int j0 = 0; j0 < 32; j0 += UN) {
      float av[UN][MT], bv[UN][NT];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int pl = wave * 64 + 2 * (j0 + u) + lh;  // pixel of this lane's k row, local to the stage
        const size_t pg = (size_t)c * R + pl;
        const bool ok = pg < (size_t)g.M;
        const float* d = a.dy + (ok ? pg : 0) * a.CC;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[u][mt] = (ok && mt * 32 + l31 < a.CC) ? d[mt * 32 + l31] : 0.f;
        const int sb = stab[buf][pl];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { const float v = win[buf][sb + toff[nt]]; bv[u][nt] = tone[nt] ? 1.f : (tok[nt] ? v : 0.f); }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][mt], bv[u][nt], acc[mt][nt], 0, 0, 0);
    }
    (void)p0;
  }
  // the four waves' sums -> one [64][64] plane per block (fixed order), through LDS one 16-row slab at a time
  float* const pb = a.part + (size_t)blockIdx.x * 64 * 64;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave][lane][r] = acc[mt][nt][r];
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = ((red[0][lane][r] + red[1][lane][r]) + red[2][lane][r]) + red[3][lane][r];
          const int cc = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = nt * 32 + l31;
          pb[cc * 64 + col] = v;
        }
      }
    }
}

// dw[cc][tap] = sum over blocks of part[b][cc][tap], db[cc] = ... part[b][cc][taps]
__global__ __launch_bounds__(256) void cw_first_wgrad_finish_kernel(const float* __restrict__ part, int nblk, int CC, int taps, float* __restrict__ dw,
                                                                    float* __restrict__ db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= CC * 64) return;
  const int cc = i >> 6, col = i & 63;
  if (col > taps) return;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += part[(size_t)b * 4096 + i];
  if (col < taps) dw[cc * taps + col] = s;
  else if (db) db[cc] = s;
}

// =================================================================================================================================
// Weight gradient over pixel images: dW[cc][ci][tap] = sum over output pixels of dy[pixel][cc] * x[pixel + tap][ci]  (cc: the merged
// [dh | dg] channels; reference utils/nn.py:92-97 under autograd).  The contraction runs over PIXELS, so both operands are read through
// the LDS transpose read (ds_read_b64_tr_b16) from their pixel-major images: dy's rows of a 32-pixel chunk, gathered (its row order
// may be parity-planar), and the window of input pixels the chunk's taps touch.
//
// Shape: M = 2 Co <= 128 merged channels, N = taps x 32 input channels (a channel-group pair), K = all pixels.  The result is small
// and the operands are streamed once, so the lever is accumulator space, not tiles: a block is four waves, ONE per SIMD with the
// whole 512-register file, each wave holds one 32-channel row tile against EVERY tap's column tile (25 x 16 = 400 accumulator
// registers for a 5 x 5 filter), and one block per CU streams 32-pixel stages (dy rows 24 KB + window 37 KB, double-buffered) at
// 150 MFMAs per wave per 16 pixels: ~26 bytes copied per MFMA (gemm_p6: 256).  Every block owns a contiguous run of pixel chunks
// (split contraction) and writes its partial [cc][tap][ci] plane; evae::cw_wgrad_finish_kernel adds the planes in block order
// (deterministic) into nn.Conv2d's [cc][ci][kh][kw] layout.  The bias gradient is one more column tile against a constant ones
// operand (three of the six products: the ones have a single non-zero term).
// =================================================================================================================================
struct CwWgradArgs {
  const unsigned char* dyimg; int nks_dy;     // merged-gradient image, channel groups per pixel (2 Co / 16)
  int dy_planar;                              // its rows are parity-planar (the layer's consumer has stride 2)
  const unsigned char* ximg; int nks_x, xcg0; // input image; the channel-group pair contracted here starts at group xcg0
  int nseg, istride, ioff[4];                 // input rows: n * istride + ioff[0] + y * Win + x, or (x_planar) n * istride + cw_planar(y, x)
  int xs, x_planar;                           // stride of the layer: the window lies on the INPUT grid (H xs) x (W xs), tap (kh, kw) of output
                                              // pixel (y, x) reads input (y xs + kh - plo, x xs + kw - plo); x_planar: the image is parity-planar
  int N, H, W, plo, phi, PW, SP;              // H x W: the OUTPUT grid; PW, SP: the padded input grid
  FastDiv div_w, div_hw, div_pw, div_sp;
  int M;                                      // N * H * W output pixels
  int tile_seg[32], tile_to[32], tile_tap[32];   // per column tile: input segment, tap offset in window slots, filter tap kh * KW + kw
  int ntap_f, Cin, CC;                        // filter taps (KH * KW), input channels of the layer, merged channels 2 Co
  int cper, nchunk;                           // chunks per block, chunks in all
  float* part;                                // [blocks][CC][ntap_f][Cin] partial planes (only this launch's taps / channel pair are written)
  float* dbpart;                              // [blocks][CC]
  int dbg;                                    // tools: 1 = no copies after the first stage, 2 = copies only (no MFMA stream), 4 = no fragment reads
};

// window slots (on the padded INPUT grid) the taps of R consecutive output pixels touch
// (align: the runs start at multiples of `align` pixels -- their residues modulo the image are the multiples of gcd(align, H W))
static int cw_wgrad_window_slots_walk(int H, int W, int K, int pad, int xs, int R, int align);
static int cw_wgrad_window_slots(int H, int W, int K, int pad, int xs, int R, int align = 1) {
  static thread_local CwSlotMemo tab[32];
  static thread_local int n = 0;
  const int key[7] = {H, W, K, pad, xs, R, align};
  int v;
  if (cw_memo_get(tab, n, key, &v)) return v;
  v = cw_wgrad_window_slots_walk(H, W, K, pad, xs, R, align);
  cw_memo_put(tab, n, key, v);
  return v;
}
static int cw_wgrad_window_slots_walk(int H, int W, int K, int pad, int xs, int R, int align) {
  const int PW = W * xs + 2 * pad, SP = (H * xs + 2 * pad) * PW, HW = H * W;
  int best = 0, step = align, t = HW;
  while (t) { const int r = step % t; step = t; t = r; }          // gcd(align, HW)
  for (int r0 = 0; r0 < HW; r0 += step) {
    const int y0 = r0 / W, x0 = r0 % W, last = r0 + R - 1;
    const int n1 = last / HW, r1 = last % HW, y1 = r1 / W, x1 = r1 % W;
    best = std::max(best, n1 * SP + (xs * y1 + K - 1) * PW + xs * x1 + K - 1 - ((xs * y0) * PW + xs * x0) + 1);
  }
  return best;
}

template <int NTW, int WSL, int NCGDY, int SS = 1, bool PAIR = false>
struct CwWgGeom {
  static constexpr int RK = 32;                                  // pixels per stage (two k-steps)
  static constexpr int DYCG = 3 * RK * 32 + 128;                 // one channel group of the dy chunk (three planes of [32 rows][32 B]); + 128:
  static constexpr int XPL = WSL * 32;                           //   the two 16-lane groups of a transpose read (even / odd channel group)
  static constexpr int XCG = 3 * XPL + 128;                      //   then fall on different bank halves
  static constexpr int DY = NCGDY * DYCG;                        // NCGDY channel groups of merged gradient (128 channels: 8)
  static constexpr int WINB = PAIR ? XCG : 2 * XCG;              // a window: both channel groups of the pair (PAIR: the one group)
  static constexpr int DR = SS > 1 ? 3 : 2;                      // dy ring depth: with a shared window the dy rows are requested TWO stages ahead
  static constexpr int LDS = DR * DY + 2 * WINB;                 // dy ring [DR][DY], then the window ring [2][WINB]
  static constexpr int NXP = WSL / 32;                           // 32-slot pieces of a window plane
};

// NTW column tiles (taps) per launch, window of WSL slots on the input grid, NCGDY channel groups of dy.  Eight waves, two per SIMD:
// wave = (row tile wm = wave % MW: merged channels 32 wm .., column part ch = wave / MW); MW = 4 row tiles (<= 128 merged channels)
// x 2 column halves, or MW = 2 (<= 64 merged channels: NCGDY <= 4) x 4 column quarters -- so that every wave has a row tile that
// exists.  The NTW + 1 columns (the taps' tiles, then the bias column) are dealt to the column parts, NCW = ceil((NTW + 1) / parts)
// each (dummy columns at the end).  One wave per SIMD with all columns was measured first: issue-bound (a transpose read and its
// address add per MFMA in ONE instruction stream: 42 % of the matrix rate); with two waves a SIMD issues one wave's reads under the
// other's MFMAs.
// SS: a window serves SS consecutive 32-pixel stages (a block's run of chunks starts at a multiple of SS): the taps of 32 pixels touch
// ~3 rows of the padded grid -- six to nine times the pixels -- and consecutive stages nearly the same rows; copied once per SS stages
// (its pieces dealt to those stages), the window costs a fraction of the dy rows instead of three to five times them.
// PAIR: the launch contracts ONE channel group (the odd last one of the input): a column tile is then that group at TWO taps --
// columns 0 .. 15 tap tile_to[2 c], columns 16 .. 31 tap tile_to[2 c + 1] -- instead of a channel-group pair at one tap with the
// second group all zeros: five column tiles for nine taps, not nine (48-channel layers: a fifth of their weight gradient).
template <int NTW, int WSL, int NCGDY, int SS, bool PAIR>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wgrad_win_kernel(const CwWgradArgs g) {
  typedef CwWgGeom<NTW, WSL, NCGDY, SS, PAIR> G;
  constexpr int RK = G::RK, NKS = RK / 16, NW = 8;
  constexpr int MW = NCGDY <= 4 ? 2 : 4, CQ = NW / MW;
  constexpr int NCW = (NTW + 1 + CQ - 1) / CQ;     // columns per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);
  const unsigned lds_base = (unsigned)(uintptr_t)(p6_lds_t)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < NW);
  const int wm = wave % MW, ch = wave / MW;
  const int l31 = lane & 31, lh = lane >> 5, ib = (lane >> 4) & 1, t16 = lane & 15;
  const int HW = g.H * g.W;
  const int c0 = blockIdx.x * g.cper, c1 = min(c0 + g.cper, g.nchunk);
  const int ncgdy = g.nks_dy;
  const bool active = 2 * wm < ncgdy;              // this wave's row tile exists (merged channels 32 wm .. 32 wm + 31)
  const int cb = ch * NCW;                         // this wave's first column

  f32x16 acc[NCW];
#pragma unroll
  for (int j = 0; j < NCW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // Stage of chunk c -> ring buffer buf, in units per wave so that they can be placed between the MFMA groups of the stage before
  // (their address arithmetic is VALU work: ~20 instructions per unit).  Unit 0: the dy rows -- 3 NCGDY (cg, plane) pieces dealt
  // round-robin; unit k >= 1: window slot piece wave + 8 (k - 1) (if it exists), its six (cg, plane) parts.
  constexpr int NUNIT = 1 + (G::NXP + NW - 1) / NW;
  constexpr int NDYQ = (3 * NCGDY + NW - 1) / NW;
  const int Hin = g.H * g.xs, Win = g.W * g.xs;
  // unit 0: chunk c's dy rows -> dy buffer `buf`; unit k >= 1: chunk c is the first of a group of SS, its window -> window buffer `buf`.
  // Returns the number of copy instructions this wave issued (the wait in front of a later stage leaves the newest ones in flight).
  auto issue_unit = [&](int c, int buf, int unit) -> int {
    char* const st = lds + (unit == 0 ? buf * G::DY : G::DR * G::DY + buf * G::WINB - G::DY);      // (window parts are addressed st + DY + ...)
    int issued = 0;
    const int p0 = c * RK;
    const unsigned nf = fdiv((unsigned)p0, g.div_hw), remf = (unsigned)p0 - nf * (unsigned)HW;
    const unsigned yf = fdiv(remf, g.div_w), xf = remf - yf * (unsigned)g.W;
    if (unit == 0) {
      // ---- dy rows p0 .. p0 + 31 (image rows: natural or parity-planar), relative to the first image of the chunk ----
      const int dbase = (int)((nf * (unsigned)HW) & ~15u);
      const rsrc_t rD = make_rsrc(g.dyimg + (size_t)(dbase >> 4) * g.nks_dy * P6_GROUP, 0x7FFFFFFFu);
      const int r = lane >> 1, hp = lane & 1;
      const int mm = p0 + r;
      const bool ok = mm < g.M;
      const unsigned mc = ok ? (unsigned)mm : 0u;
      int row = (int)mc;                                              // natural rows: the pixel index itself
      if (g.dy_planar) {
        const unsigned n = fdiv(mc, g.div_hw), rem = mc - n * (unsigned)HW;
        const unsigned y = fdiv(rem, g.div_w), x = rem - y * (unsigned)g.W;
        row = (int)n * HW + cw_planar((int)y, (int)x, g.H, g.W);
      }
      const int rel = row - dbase;
      const unsigned gh = (unsigned)(hp ^ ((row >> 3) & 1));          // the image's own half swap; LDS keeps logical halves in place
      unsigned vin = (unsigned)(rel >> 4) * (unsigned)(g.nks_dy * P6_GROUP) + (unsigned)((rel & 15) * 32) + (gh << 4);
      EVAE_PIN(vin);                                                 // (computed by every lane: a select, not a branch)
      const unsigned voff = ok ? vin : 0x80000000u;
#pragma unroll
      for (int q = 0; q < NDYQ; ++q) {
        const int id = wave + NW * q, cg = id / 3, p = id - cg * 3;
        if (id < 3 * NCGDY && cg < ncgdy) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rD, (p6_lds_t)(st + cg * G::DYCG + p * (RK * 32)), 16, voff, (unsigned)(cg * P6_GROUP + p * P6_CHUNK), 0, 0);
          ++issued;
        }
      }
    } else {
      // ---- window slot piece jj of the input grid ----
      const int jj = wave + NW * (unit - 1);
      if (jj >= G::NXP) return 0;
      issued = 6;
      const int qbase = (int)(nf * (unsigned)g.SP + yf * (unsigned)(g.xs * g.PW) + xf * (unsigned)g.xs);
      const int xbase = (int)((nf * (unsigned)g.istride) & ~15u);
      const rsrc_t rX = make_rsrc(g.ximg + ((size_t)(xbase >> 4) * g.nks_x + (size_t)g.xcg0) * P6_GROUP, 0x7FFFFFFFu);
      const int s = 32 * jj + (lane >> 1), hp = lane & 1;
      const unsigned q = (unsigned)(qbase + s);
      const unsigned n = fdiv(q, g.div_sp), r = q - n * (unsigned)g.SP;
      const unsigned py = fdiv(r, g.div_pw), px = r - py * (unsigned)g.PW;
      const int y = (int)py - g.plo, x = (int)px - g.plo;
      const bool ok = (int)n < g.N && (unsigned)y < (unsigned)Hin && (unsigned)x < (unsigned)Win;
      const int pix = (int)n * g.istride + (g.x_planar ? cw_planar(y, x, Hin, Win) : g.ioff[0] + y * Win + x);
      const int rel = pix - xbase;
      const unsigned gh = (unsigned)(hp ^ ((pix >> 3) & 1));
      unsigned vin = (unsigned)(rel >> 4) * (unsigned)(g.nks_x * P6_GROUP) + (unsigned)((rel & 15) * 32) + (gh << 4);
      EVAE_PIN(vin);
      const unsigned voff = ok ? vin : 0x80000000u;
      const bool cg1 = g.xcg0 + 1 < g.nks_x;               // (an odd number of channel groups: the last pair's second group reads as zeros)
      if (PAIR) issued = 3;
#pragma unroll
      for (int k = 0; k < (PAIR ? 3 : 6); ++k) {
        const int cg = k / 3, p = k - cg * 3;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (p6_lds_t)(st + G::DY + cg * G::XCG + p * G::XPL + jj * 1024), 16,
                                                 (cg == 0 || cg1) ? voff : 0x80000000u, (unsigned)(cg * P6_GROUP + p * P6_CHUNK), 0, 0);
      }
    }
    return issued;
  };
  // s_waitcnt vmcnt(n) for a wave-uniform n (copies retire in order: the newest n may stay in flight)
  auto wait_copies = [&](int n) {
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    }
  };

  // window slot (relative to the stage's base slot) of chunk row r: this lane's k rows are 16 ks + 8 lh + (t16 >> 2) (+ 4)
  auto slot_of = [&](int p0, int qbase, int r) -> int {
    int mm = p0 + r;
    mm = mm < g.M ? mm : g.M - 1;
    const unsigned n = fdiv((unsigned)mm, g.div_hw), rem = (unsigned)mm - n * (unsigned)HW;
    const unsigned y = fdiv(rem, g.div_w), x = rem - y * (unsigned)g.W;
    return (int)(n * (unsigned)g.SP + y * (unsigned)(g.xs * g.PW) + x * (unsigned)g.xs) - qbase;
  };

  // the ones operand of the bias column: column 0 = 1.0 (plane 0 only)
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  const unsigned one1 = l31 == 0 ? 0x3F803F80u : 0u;
  // a fragment = two transpose reads (k rows c and c + 4 -> the 8 k of this lane's row); the halves stay separate until the wait
  // for them has passed (inline-asm results are invisible to the compiler's own lgkmcnt bookkeeping)
  struct Raw { p6_u32x2 lo, hi; };
  auto tr2 = [&](Raw& f, unsigned a0, unsigned a1) {
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a1));
  };
  auto cook = [&](const Raw& f) -> p6_bf16x8 {
    const u32x4_ v = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
    return __builtin_bit_cast(p6_bf16x8, v);
  };
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};       // smallest partial products first
  // this wave's columns: tile offsets (bytes) of the real tiles; a column at or beyond NTW reads tile NTW - 1's address (and, for
  // the bias column, has its fragments replaced by the ones operand)
  unsigned toff[NCW];
  bool isb[NCW];
#pragma unroll
  for (int u = 0; u < NCW; ++u) {
    const int c = cb + u;
    const int cq = c < NTW ? c : NTW - 1;
    toff[u] = PAIR ? (unsigned)((ib ? g.tile_to[2 * cq + 1] : g.tile_to[2 * cq]) * 32) : (unsigned)(g.tile_to[cq] * 32);
    isb[u] = c == NTW;
  }

  // lane addresses of stage i's k-steps (dy buffer bufi): A rows kr, kr + 4 of this wave's channel groups; B window slots of those rows.
  // ~100 VALU instructions -- computed for stage i + 1 between the MFMAs of stage i's last step: in front of the stage, behind its
  // barrier, both waves of a SIMD would do them at the same time with the matrix pipe idle (measured: 1600 of a stage's 3900 clocks)
  auto stage_addrs = [&](int i, int bufi, unsigned (&A)[NKS], unsigned (&B0)[NKS], unsigned (&B1)[NKS]) {
    const int sup = i / SS, wbuf = sup & 1;
    const unsigned st = lds_base + (unsigned)(bufi * G::DY);                                  // dy rows of the stage
    const unsigned sw = lds_base + (unsigned)(G::DR * G::DY + wbuf * G::WINB);                // window of its group
    const int p0 = (c0 + i) * RK, p0w = (c0 + sup * SS) * RK;
    const unsigned nf = fdiv((unsigned)p0w, g.div_hw), remf = (unsigned)p0w - nf * (unsigned)HW;
    const unsigned yf = fdiv(remf, g.div_w), xf = remf - yf * (unsigned)g.W;
    const int qbase = (int)(nf * (unsigned)g.SP + yf * (unsigned)(g.xs * g.PW) + xf * (unsigned)g.xs);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int kr = 16 * ks + 8 * lh + (t16 >> 2);
      A[ks] = st + (unsigned)((2 * wm + ib) * G::DYCG + kr * 32 + (t16 & 3) * 8);
      const unsigned xb = sw + (unsigned)((PAIR ? 0 : ib * G::XCG) + (t16 & 3) * 8);
      B0[ks] = xb + (unsigned)(slot_of(p0, qbase, kr) * 32);
      B1[ks] = xb + (unsigned)(slot_of(p0, qbase, kr + 4) * 32);
    }
  };

#ifdef EVAE_CW_ABL
  const int dbg = g.dbg;                             // (tools/micro/cw_bench.hip: ablations; the library build carries no such branches)
#else
  constexpr int dbg = 0;
#endif
  const int nst = c1 - c0;
  constexpr int LA = G::DR - 1;                     // stages the dy rows are requested ahead
  static_assert(SS == 1 || NUNIT <= SS, "a shared window's units: one per stage, the last two stages before the group it serves");
  if (nst > 0) {
    int flying = 0;                                 // copies this wave issued in the latest slot (they may stay in flight over the next wait)
#pragma unroll
    for (int u = NUNIT - 1; u >= 0; --u) (void)issue_unit(c0, 0, u);          // window first, then the dy rows
    if (LA == 2 && nst > 1 && !(dbg & 1)) flying = issue_unit(c0 + 1, 1, 0);
    int buf = 0;
    unsigned aA[NKS], bB0[NKS], bB1[NKS], nA[NKS], nB0[NKS], nB1[NKS];
    stage_addrs(0, 0, aA, bB0, bB1);
    for (int i = 0; i < nst; ++i, buf = buf + 1 == G::DR ? 0 : buf + 1) {
      const int c = c0 + i;
      const int sup = i / SS, sq = i - sup * SS, wbuf = sup & 1;       // window group, stage inside it, its buffer
      const int nbuf = LA == 2 ? (buf == 0 ? 2 : buf - 1) : buf ^ 1;   // dy buffer of stage i + LA (= that of stage i - 1)
      // this stage's copies have landed (LA == 2: those of the slot before the latest one); barrier: everybody's have, and everybody
      // is done with stage i - 1: its dy buffer is overwritten by the copies of stage i + LA and -- when it closed a group -- its
      // window buffer by the next group's window, both issued between this stage's MFMA groups
      wait_copies(LA == 2 ? flying : 0);
      __builtin_amdgcn_s_barrier();
      const bool more = i + LA < nst && !(dbg & 1);
      const bool morew = (sup + 1) * SS < nst && !(dbg & 1);         // a next group exists: its window's units are dealt to this group's stages
      flying = 0;
      if (dbg & 2) {
#pragma unroll
        for (int u = 1; u < NUNIT; ++u) if (morew && (SS == 1 || u - 1 == sq)) flying += issue_unit(c0 + (sup + 1) * SS, wbuf ^ 1, u);
        if (more) flying += issue_unit(c + LA, nbuf, 0);
        continue;
      }
      // The stage as NKS * NGRP steps (k-step ks, column group jg): a step's MFMAs run on fragments read during the step BEFORE --
      // two transpose reads behind each of its first MFMAs, so that they have half a step to land before the wait in front of
      // the next one; only a stage's first step waits for its own reads.
      constexpr int GS = NCW >= 4 ? 4 : NCW, NGRP = (NCW + GS - 1) / GS, NSTEP = NKS * NGRP;
      Raw ar[2][3], br[2][GS][3];
      // read #e of step t into fragment buffers: e < 6 (only when the step opens a k-step): A (plane e / 2, half e & 1); then the
      // group's B fragments: column u = e' / 6, plane (e' % 6) / 2, half e' & 1
      auto read_step = [&](int t, int e) {
        if (dbg & 4) return;                         // (tools: the MFMA stream without its fragment reads)
        const int ks = t / NGRP, jg = t - ks * NGRP, j = GS * jg;
        const int na = jg == 0 ? 6 : 0;
        // (plane / half offsets as the instruction's offset field: t and e are constants once the step loop is unrolled)
        if (e < na) {
          const int p = e >> 1, hf = e & 1;
          if (hf) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ar[ks & 1][p].hi) : "v"(aA[ks]), "n"(p * (RK * 32) + hf * 128));
          else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ar[ks & 1][p].lo) : "v"(aA[ks]), "n"(p * (RK * 32) + hf * 128));
        } else {
          const int eb = e - na, u = eb / 6, p = (eb % 6) >> 1, hf = eb & 1;
          if (u < GS && j + u < NCW) {
            const unsigned a = (hf ? bB1[ks] : bB0[ks]) + toff[j + u];
            if (hf) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(br[t & 1][u][p].hi) : "v"(a), "n"(p * G::XPL));
            else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(br[t & 1][u][p].lo) : "v"(a), "n"(p * G::XPL));
          }
        }
      };
#pragma unroll
      for (int e = 0; e < 6 + 6 * GS; ++e) read_step(0, e);
      p6_bf16x8 af[3];
#pragma unroll
      for (int t = 0; t < NSTEP; ++t) {
        const int ks = t / NGRP, jg = t - ks * NGRP, j = GS * jg;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (jg == 0) {
#pragma unroll
          for (int p = 0; p < 3; ++p) af[p] = cook(ar[ks & 1][p]);
        }
        p6_bf16x8 bf_[GS][3];
#pragma unroll
        for (int u = 0; u < GS; ++u)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            if (j + u < NCW) {
              u32x4_ v = {br[t & 1][u][p].lo[0], br[t & 1][u][p].lo[1], br[t & 1][u][p].hi[0], br[t & 1][u][p].hi[1]};
              if ((NTW - (j + u)) % NCW == 0 && (NTW - (j + u)) / NCW < CQ && NTW >= j + u) {  // this column is some column part's bias column: the ones operand instead
                const unsigned o = p == 0 ? one1 : 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = isb[j + u] ? o : v[q];
              }
              bf_[u][p] = __builtin_bit_cast(p6_bf16x8, v);
            }
        __builtin_amdgcn_sched_barrier(0);
        // 6 GS MFMAs (a wave whose row tile does not exist -- fewer than 128 merged channels -- runs the same stream on whatever its
        // LDS rows hold and stores nothing: no branch around the MFMAs, whose accumulators would otherwise be copied at every join)
        constexpr int NRD = 6 + 6 * GS;
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int u = 0; u < GS; ++u) {
            if (j + u < NCW) acc[j + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[q]], bf_[u][PB[q]], acc[j + u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < NSTEP) {
              const int m = q * GS + u;
              if (2 * m < NRD) read_step(t + 1, 2 * m);
              if (2 * m + 1 < NRD) read_step(t + 1, 2 * m + 1);
            } else if (q == 1 && u == 0) {
              stage_addrs(i + 1, buf + 1 == G::DR ? 0 : buf + 1, nA, nB0, nB1);      // (the last step reads nothing ahead: room for the next stage's addresses)
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        // a unit of the coming copies behind the first steps of the stage (the matrix pipe works them off meanwhile): the next
        // stage's dy rows, this stage's share of the next group's window
        // (window units first: the wait in front of the next stage leaves only this slot's copies in flight, in issue order)
        if (t + 1 < NUNIT) { if (morew && (SS == 1 || t == sq)) flying += issue_unit(c0 + (sup + 1) * SS, wbuf ^ 1, t + 1); }
        else if (t + 1 == NUNIT) { if (more) flying += issue_unit(c + LA, nbuf, 0); }
        __builtin_amdgcn_sched_barrier(0);
      }
      {                                              // fewer steps than units: the remaining units
#pragma unroll
        for (int u = NSTEP; u < NUNIT; ++u) {
          if (u + 1 < NUNIT) { if (morew && (SS == 1 || u == sq)) flying += issue_unit(c0 + (sup + 1) * SS, wbuf ^ 1, u + 1); }
          else if (more) flying += issue_unit(c + LA, nbuf, 0);
        }
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) { aA[ks] = nA[ks]; bB0[ks] = nB0[ks]; bB1[ks] = nB1[ks]; }
    }
  }
  // ---- partial plane of this block: part[block][cc][tap][ci], dbpart[block][cc] ----
  if (active) {
    float* const pb = g.part + (size_t)blockIdx.x * g.CC * g.ntap_f * g.Cin;
#pragma unroll
    for (int u = 0; u < NCW; ++u) {
      const int c = cb + u;
      if (c < NTW) {
        const int tap = PAIR ? g.tile_tap[2 * c + ib] : g.tile_tap[c];           // (PAIR: tap < 0 = the empty half of the last tile)
        const int ci = g.xcg0 * 16 + (PAIR ? (l31 & 15) : l31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cc = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (cc < g.CC && ci < g.Cin && tap >= 0) pb[((size_t)cc * g.ntap_f + tap) * g.Cin + ci] = acc[u][r];
        }
      } else if (c == NTW && g.dbpart && l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cc = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (cc < g.CC) g.dbpart[(size_t)blockIdx.x * g.CC + cc] = acc[u][r];
        }
      }
    }
  }
}

// dw[cc][ci][tap] = sum over blocks of part[b][cc][tap][ci]; db[cc] = sum of dbpart[b][cc].  Deterministic: a thread block = 32 outputs
// x 8 slices of the planes (slice q sums planes q, q + 8, ... in order), the eight slice sums meet in LDS in a fixed order.  (One
// thread per output walking all 256 planes was a 256-deep dependent chain on ~300 blocks: 109 us per call in the c5 step.)
__global__ __launch_bounds__(256) void cw_wgrad_finish_kernel(const float* __restrict__ part, const float* __restrict__ dbpart, int nblk, int CC,
                                                              int ntap, int Cin, float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float red[8][33];
  const int lane = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int n = CC * ntap * Cin;
  const int i = blockIdx.x * 32 + lane;
  const int nw = (n + 31) / 32;                 // blocks that reduce dw; the rest reduce db
  float s = 0.f;
  if (blockIdx.x < nw) {
    if (i < n)
      for (int b = q; b < nblk; b += 8) s += part[(size_t)b * n + i];
  } else {
    const int j = (blockIdx.x - nw) * 32 + lane;
    if (db && j < CC)
      for (int b = q; b < nblk; b += 8) s += dbpart[(size_t)b * CC + j];
  }
  red[q][lane] = s;
  __syncthreads();
  if (q == 0) {
    float t = red[0][lane];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][lane];
    if (blockIdx.x < nw) {
      if (i < n) {
        const int ci = i % Cin, tap = (i / Cin) % ntap, cc = i / (Cin * ntap);
        dw[((size_t)cc * Cin + ci) * ntap + tap] = t;
      }
    } else {
      const int j = (blockIdx.x - nw) * 32 + lane;
      if (db && j < CC) db[j] = t;
    }
  }
}
static inline int cw_wgrad_finish_blocks(int CC, int ntap, int Cin) { return (CC * ntap * Cin + 31) / 32 + (CC + 31) / 32; }

template <int NTW, int WSL, int NCGDY, int SS = 1, bool PAIR = false>
static int launch_conv_wgrad_win(CwWgradArgs& g, int nblk, hipStream_t stream, const char* what) {
  typedef CwWgGeom<NTW, WSL, NCGDY, SS, PAIR> G;
  static_assert(G::LDS <= 160 * 1024, "stage ring beyond a CU's LDS");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_win_kernel<NTW, WSL, NCGDY, SS, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    attr_done = true;
  }
  if (g.xs == 0) g.xs = 1;
  g.PW = g.W * g.xs + g.plo + g.phi; g.SP = (g.H * g.xs + g.plo + g.phi) * g.PW;
  g.div_w = make_fastdiv((unsigned)g.W); g.div_hw = make_fastdiv((unsigned)(g.H * g.W));
  g.div_pw = make_fastdiv((unsigned)g.PW); g.div_sp = make_fastdiv((unsigned)g.SP);
  g.M = g.N * g.H * g.W;
  if (g.istride == 0) g.istride = g.H * g.W * g.xs * g.xs;
  g.nchunk = cdiv(g.M, G::RK);
  g.cper = cdiv(g.nchunk, nblk);
  if (cw_wgrad_window_slots(g.H, g.W, g.plo + g.phi + 1, g.plo, g.xs, G::RK * SS) > WSL) {
    // the window holds SS stages only from aligned starts: every block's run starts at a multiple of SS chunks
    g.cper = (g.cper + SS - 1) / SS * SS;
    if (cw_wgrad_window_slots(g.H, g.W, g.plo + g.phi + 1, g.plo, g.xs, G::RK * SS, G::RK * SS) > WSL) {
      set_error("%s: window of %d slots too small for %d stages", what, WSL, SS);
      return EVAE_EINVAL;
    }
  }
  conv_wgrad_win_kernel<NTW, WSL, NCGDY, SS, PAIR><<<dim3(cdiv(g.nchunk, g.cper)), 512, G::LDS, stream>>>(g);
  return check_launch(what);
}

}  // namespace evae
