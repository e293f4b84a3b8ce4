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
  static constexpr int EPI_LDS = 2 * R * 68 * 4;        // epilogue staging: two fp32 arrays [R][64 + 4]
  static constexpr int LDS = LOOP_LDS > EPI_LDS ? LOOP_LDS : EPI_LDS;
  static constexpr int NPIECE = KST * 3 * (BN / 32);    // 1 KB copies per ring stage
  static_assert(NPIECE % 4 == 0, "every wave issues the same number of copies per stage");
};

struct ConvWinArgs {
  const unsigned char* xin;   // pixel image of the input, nks_in channel groups per pixel
  int nks_in;
  int ncg;                    // channel groups contracted over (k-steps = ncg * ntaps), starting at group cg0 of the image
  int cg0;
  int N, H, W, KH, KW, pad;   // stride 1, 2 pad + 1 == KH == KW: output H x W
  int PW, SP;
  FastDiv div_w, div_hw, div_pw, div_sp;
  const unsigned char* wimg;  // filter image: BN rows per column tile, k = (cg * ntaps + tap) * 16 + c
  int nks_w;
  int M;                      // N * H * W
  int Co;                     // real output columns (gated: channels per bank)
  int tiles_n;
  const float* bias0;
  const float* bias1;
  unsigned char* oimg;        // pixel image of the result (nks_o channel groups; result column c at image channel och0 + c;
  int nks_o, och0;            //   CW_DGRAD_GATE: dh at och0 + c, dg at och0 + Co + c)
  float* out_s;               // gated forward: the gate s, fp32 [M][Co]
  float* out_f;               // fp32 copy of the result [M][ldo] (optional; CW_DGRAD_GATE: [dh | dg], dg at column Co + c)
  int ldo;
  const unsigned char* eimg;  // CW_DGRAD_GATE: pixel image of the forward output of the layer below (nks_e groups, channel ech0 + c) ...
  int nks_e, ech0;
  const float* e_s;           // ... and its gate [M][Co]: [dh | dg] = [v s | v out (1 - s)]
  int dbg;
};

// ---- filter images ------------------------------------------------------------------------------------------------------------
// k = (cg * taps + t) * 16 + cl <-> contraction channel cc = cg * 16 + cl, tap t.  Rows, per column tile of bn rows:
// mode 0 (gated forward): row (tn, c), c = wc * 64 + hg * 32 + j <-> bank hg (w0 = h, w1 = g), output channel tn * (bn / 2) + wc * 32 + j
// mode 2 (plain forward): row r <-> output channel r
// mode 1 (data gradient): row r <-> INPUT channel r of the layer (the data gradient's output channel); cc = merged gradient
//   channel (cc < Co: bank h, output channel cc; else bank g, cc - Co), filter tap (KH - 1 - th, KW - 1 - tw) of t = (th, tw):
//   the flipped filter
// w layout: nn.Conv2d's [Co][Ci][KH][KW]
__global__ __launch_bounds__(256) void cw_pack_filter_kernel(const float* __restrict__ w0, const float* __restrict__ w1, int Co, int Ci,
                                                             int taps, int mode, int bn, int rows_img, int nks, unsigned char* __restrict__ img) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int kslots = nks * 2;
  if (t >= (size_t)rows_img * kslots) return;
  const int ri = (int)(t / kslots), ks8 = (int)(t - (size_t)ri * kslots);
  const int k0 = ks8 * 8;
  const int kstep = k0 >> 4, tap = kstep % taps, cg = kstep / taps;
  unsigned short p0[8], p1[8], p2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int cc = cg * 16 + (k0 & 15) + i;     // contraction channel
    float v = 0.f;
    if (mode == 0) {
      const int tn = ri / bn, c = ri - tn * bn, wc = c >> 6, hg = (c >> 5) & 1, j = c & 31;
      const int co = tn * (bn / 2) + wc * 32 + j;
      const float* src = hg ? w1 : w0;
      if (co < Co && cc < Ci) v = src[((size_t)co * Ci + cc) * taps + tap];
    } else if (mode == 2) {
      if (ri < Co && cc < Ci) v = w0[((size_t)ri * Ci + cc) * taps + tap];
    } else {
      const int ctot = w1 ? 2 * Co : Co;
      if (ri < Ci && cc < ctot) {
        const float* src = cc < Co ? w0 : w1;
        const int co = cc < Co ? cc : cc - Co;
        v = src[((size_t)co * Ci + ri) * taps + (taps - 1 - tap)];
      }
    }
    p6_split1(v, p0[i], p1[i], p2[i]);
  }
  unsigned char* o = img + p6_off(ri, k0, nks);
  *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(p0);
  *reinterpret_cast<uint4*>(o + P6_CHUNK) = *reinterpret_cast<const uint4*>(p1);
  *reinterpret_cast<uint4*>(o + 2 * P6_CHUNK) = *reinterpret_cast<const uint4*>(p2);
}

// window slots a block of R consecutive output pixels needs, maximised over the block starts that occur
static int cw_window_slots(int H, int W, int KH, int KW, int pad, int R) {
  const int PW = W + 2 * pad, SP = (H + 2 * pad) * PW, HW = H * W;
  int best = 0;
  for (int r0 = 0; r0 < HW; ++r0) {           // first pixel of a block, modulo the image (every residue: M need not be regular)
    const int y0 = r0 / W, x0 = r0 % W;
    const int last = r0 + R - 1;
    const int n1 = last / HW, r1 = last % HW, y1 = r1 / W, x1 = r1 % W;
    const int q0 = y0 * PW + x0, q1 = n1 * SP + (y1 + KH - 1) * PW + x1 + KW - 1;
    best = std::max(best, q1 - q0 + 1);
  }
  return best;
}

enum { CW_FWD_GATED = 0, CW_DGRAD_GATE = 1, CW_PLAIN = 2 };

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
  const int HW = g.H * g.W, ntaps = g.KH * g.KW;

  // first pixel of the block -> base slot; base pixel of the buffer resource (16-pixel aligned, at or before every pixel a tap reads)
  const unsigned nf = fdiv((unsigned)m0, g.div_hw), remf = (unsigned)m0 - nf * (unsigned)HW;
  const unsigned yf = fdiv(remf, g.div_w), xf = remf - yf * (unsigned)g.W;
  const int qbase = (int)(nf * (unsigned)g.SP + yf * (unsigned)g.PW + xf);
  int bp = m0 - g.pad * g.W - g.pad;
  bp = bp < 0 ? 0 : (bp & ~15);
  const int base_pix = __builtin_amdgcn_readfirstlane(bp);
  const rsrc_t rB = make_rsrc(g.wimg + (size_t)tn * (BN / 16) * g.nks_w * P6_GROUP, 0x7FFFFFFFu);

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

  const int S = ntaps;                           // k-steps per channel group
  constexpr std::true_type T{};
  constexpr std::false_type F{};
  for (int cgi = 0; cgi < g.ncg; ++cgi) {
    if (cgi > 0) __syncthreads();                // every wave is done with the previous group's window and ring
    const int ksc = cgi * S;                     // first k-step of the group
    issue_b(ksc, 0, S < KST ? S : KST);
    if (S > KST) issue_b(ksc + KST, 1, S - KST < KST ? S - KST : KST);
    // ---- window of this channel group: slot pieces j = wave, wave + 4, ... (32 slots each), one copy per plane ----
    {
      const rsrc_t rA = make_rsrc(g.xin + ((size_t)(base_pix >> 4) * g.nks_in + (size_t)(g.cg0 + cgi)) * P6_GROUP, 0x7FFFFFFFu);
#pragma unroll
      for (int jj = 0; jj < (SLOTS / 32 + 3) / 4; ++jj) {
        const int j = wave + 4 * jj;
        if (j < SLOTS / 32) {
          const int s = 32 * j + (lane >> 1), hp = lane & 1;
          const unsigned q = (unsigned)(qbase + s);
          const unsigned n = fdiv(q, g.div_sp), r = q - n * (unsigned)g.SP;
          const unsigned py = fdiv(r, g.div_pw), px = r - py * (unsigned)g.PW;
          const int y = (int)py - g.pad, x = (int)px - g.pad;
          const bool ok = (int)n < g.N && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
          const int pix = ((int)n * g.H + y) * g.W + x;
          const int rel = pix - base_pix;
          const unsigned gh = (unsigned)(hp ^ ((s >> 3) & 1) ^ ((pix >> 3) & 1));
          const unsigned voff = ok ? (unsigned)(rel >> 4) * (unsigned)(g.nks_in * P6_GROUP) + (unsigned)((rel & 15) * 32) + (gh << 4) : 0x80000000u;
#pragma unroll
          for (int p = 0; p < 3; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (p6_lds_t)(lds + p * WP + j * 1024), 16, voff, (unsigned)(p * P6_CHUNK), 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int kw = 0, to = 0;                          // tap column, tap offset in slots (kh PW + kw)
    unsigned an[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) an[mt] = a_addr(mt, 0);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) read_a(I0, an[mt], mt, p);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) read_b(I0, 0, 0, nt, p);
    }
    auto next_tap = [&]() {                      // branch-free (scalar selects)
      const bool wrap = kw + 1 == g.KW;
      kw = wrap ? 0 : kw + 1;
      to += wrap ? g.PW - g.KW + 1 : 1;
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
#undef EVAE_CW_SB
  __syncthreads();                 // the epilogue stages through the window's LDS
  if (g.dbg == 4) return;

  // ---- epilogue ---------------------------------------------------------------------------------------------------------------
  // The result leaves pixel-major: a lane of the matrix core's C layout owns one column (channel) and sixteen rows of a 32 x 32 tile,
  // the images want the channels of a pixel side by side -> through LDS as [row][BNO channels] fp32 (row pitch BNO + 4 floats), then
  // 16-byte accesses per (row, 8 channels) piece.
  constexpr int BNO = GATED ? BN / 2 : BN;      // result columns of the block
  constexpr int RP = BNO + 4;
  float* const so = smem;                       // [R][RP]
  float* const ss = smem + R * RP;              // [R][RP]   (gated: the gate)
  if constexpr (GATED) {
    const int cl = wc * 32 + l31, c = tn * BNO + cl;
    const bool cok = c < g.Co;
    const float bh = (g.bias0 && cok) ? g.bias0[c] : 0.f, bg = (g.bias1 && cok) ? g.bias1[c] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float h = acc[mt][0][r] + bh;
        const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * (acc[mt][1][r] + bg)));
        so[row * RP + cl] = cok ? h * s : 0.f;
        ss[row * RP + cl] = s;
      }
  } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int cl = wc * 32 * NT + nt * 32 + l31, c = tn * BNO + cl;
      const float b = (EPI == CW_PLAIN && g.bias0 && c < g.Co) ? g.bias0[c] : 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          so[row * RP + cl] = acc[mt][nt][r] + b;
        }
    }
  }
  __syncthreads();
  // pieces (row, 8 channels): thread -> row fastest inside 16 (the 16 pixels of an image chunk), then the 8-channel group, then the
  // 16-row groups: the 32 lanes of a (chunk, channel group) pair fill whole 512-byte chunks of the image
  constexpr int C8 = BNO / 8, NPC = R * C8 / 256;
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int pid = tid + 256 * i;
    const int r16 = pid & 15, c8 = (pid >> 4) % C8, rg = pid / (16 * C8);
    const int row = rg * 16 + r16, m = m0 + row;
    const int ch = tn * BNO + c8 * 8;                         // first of the eight result columns
    if (m >= g.M || ch >= g.Co) continue;
    const float4 o0 = *reinterpret_cast<const float4*>(so + row * RP + c8 * 8), o1 = *reinterpret_cast<const float4*>(so + row * RP + c8 * 8 + 4);
    auto put_img = [&](int chan, const float4& a, const float4& b) {
      unsigned t0[4], t1[4], t2[4];
      p6_split2(a.x, a.y, t0[0], t1[0], t2[0]); p6_split2(a.z, a.w, t0[1], t1[1], t2[1]);
      p6_split2(b.x, b.y, t0[2], t1[2], t2[2]); p6_split2(b.z, b.w, t0[3], t1[3], t2[3]);
      unsigned char* o = g.oimg + p6_off(m, chan, g.nks_o);
      *reinterpret_cast<uint4*>(o) = make_uint4(t0[0], t0[1], t0[2], t0[3]);
      *reinterpret_cast<uint4*>(o + P6_CHUNK) = make_uint4(t1[0], t1[1], t1[2], t1[3]);
      *reinterpret_cast<uint4*>(o + 2 * P6_CHUNK) = make_uint4(t2[0], t2[1], t2[2], t2[3]);
    };
    if constexpr (GATED) {
      const float4 s0 = *reinterpret_cast<const float4*>(ss + row * RP + c8 * 8), s1 = *reinterpret_cast<const float4*>(ss + row * RP + c8 * 8 + 4);
      if (g.oimg) put_img(g.och0 + ch, o0, o1);
      if (g.out_s) {
        float* sp = g.out_s + (size_t)m * g.Co + ch;
        *reinterpret_cast<float4*>(sp) = s0; *reinterpret_cast<float4*>(sp + 4) = s1;
      }
      if (g.out_f) {
        float* op = g.out_f + (size_t)m * g.ldo + ch;
        *reinterpret_cast<float4*>(op) = o0; *reinterpret_cast<float4*>(op + 4) = o1;
      }
    } else if constexpr (EPI == CW_PLAIN) {
      if (g.oimg) put_img(g.och0 + ch, o0, o1);
      if (g.out_f) {
        float* op = g.out_f + (size_t)m * g.ldo + ch;
        *reinterpret_cast<float4*>(op) = o0; *reinterpret_cast<float4*>(op + 4) = o1;
      }
    } else {
      // gate derivative of the layer below at (pixel m, channels ch .. ch + 7): out = the sum of its image's three terms (exact), s
      // fp32; dh = v s, dg = v out (1 - s)   (reference utils/nn.py:92-97 under autograd)
      const unsigned char* e = g.eimg + p6_off(m, g.ech0 + ch, g.nks_e);
      const uint4 e0 = *reinterpret_cast<const uint4*>(e), e1 = *reinterpret_cast<const uint4*>(e + P6_CHUNK), e2 = *reinterpret_cast<const uint4*>(e + 2 * P6_CHUNK);
      const float* sp = g.e_s + (size_t)m * g.Co + ch;
      const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
      const unsigned w0[4] = {e0.x, e0.y, e0.z, e0.w}, w1[4] = {e1.x, e1.y, e1.z, e1.w}, w2[4] = {e2.x, e2.y, e2.z, e2.w};
      const float v[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      float dh[8], dg[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int sh = 16 * (k & 1);
        // smallest terms first: the sum of the three bf16 terms reproduces the fp32 value they were split from
        const float ov = (__uint_as_float(((w2[k >> 1] >> sh) & 0xFFFFu) << 16) + __uint_as_float(((w1[k >> 1] >> sh) & 0xFFFFu) << 16)) +
                         __uint_as_float(((w0[k >> 1] >> sh) & 0xFFFFu) << 16);
        dh[k] = v[k] * sv[k];
        dg[k] = v[k] * ov * (1.0f - sv[k]);
      }
      if (g.oimg) {
        put_img(g.och0 + ch, make_float4(dh[0], dh[1], dh[2], dh[3]), make_float4(dh[4], dh[5], dh[6], dh[7]));
        put_img(g.och0 + g.Co + ch, make_float4(dg[0], dg[1], dg[2], dg[3]), make_float4(dg[4], dg[5], dg[6], dg[7]));
      }
      if (g.out_f) {
        float* op = g.out_f + (size_t)m * g.ldo + ch;
        *reinterpret_cast<float4*>(op) = make_float4(dh[0], dh[1], dh[2], dh[3]); *reinterpret_cast<float4*>(op + 4) = make_float4(dh[4], dh[5], dh[6], dh[7]);
        *reinterpret_cast<float4*>(op + g.Co) = make_float4(dg[0], dg[1], dg[2], dg[3]); *reinterpret_cast<float4*>(op + g.Co + 4) = make_float4(dg[4], dg[5], dg[6], dg[7]);
      }
    }
  }
}

template <int EPI, int WR, int NT, int SLOTS>
static int launch_conv_win(ConvWinArgs& g, hipStream_t stream, const char* what) {
  typedef CwGeom<WR, NT, SLOTS> G;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_win_kernel<EPI, WR, NT, SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    attr_done = true;
  }
  g.PW = g.W + 2 * g.pad; g.SP = (g.H + 2 * g.pad) * g.PW;
  g.div_w = make_fastdiv((unsigned)g.W); g.div_hw = make_fastdiv((unsigned)(g.H * g.W));
  g.div_pw = make_fastdiv((unsigned)g.PW); g.div_sp = make_fastdiv((unsigned)g.SP);
  g.M = g.N * g.H * g.W;
  const int tiles_m = cdiv(g.M, G::R);
  conv_win_kernel<EPI, WR, NT, SLOTS><<<dim3(tiles_m * g.tiles_n), 256, G::LDS, stream>>>(g);
  return check_launch(what);
}

}  // namespace evae
