// Convolutions of the conv encoders/decoders as implicit GEMMs on the fp32 matrix cores of gfx950.
// Replaces utils/nn.py:72-114 (GatedConv2d, Conv2d) and the nn.Conv2d layers of models/convHVAE_2level.py
// and models/fully_conv.py, forward and backward.
//
// No im2col matrix ever exists in memory: the A (or B) tile of each K-slab is gathered straight from the NCHW
// tensor into LDS ("LDS-staged im2col"), then the same MFMA slab loop as the dense layers runs on it
// (evae_gemm_core.h: 128 x BN x 32 block tile, 8 waves, v_mfma_f32_32x32x2_f32, register prefetch one slab
// ahead, zero-fill applied at the LDS store so the gather loads stay in flight under the MFMAs).
//
//   forward   out[(n,oh,ow), co] = sum_k A[(n,oh,ow), k] W[co, k]      A = im2col(x),  k = (ci, kh, kw)
//             epilogue: + bias, optional gate  h * sigmoid(g)  (both filter banks in one pass over x), NCHW store
//   data grad dx[(n,ih,iw), ci]  = sum_k A'[(n,ih,iw), k] W'[k, ci]    A' = col2im gather of dy (stride-aware),
//             k = (pair, co, kh, kw); W' = filters permuted to [k][ci] by a small pre-pass; the two halves of a
//             gated layer are one GEMM over the concatenated k range
//   weight grad dW[co, k] = sum_m dy[m, co] A[m, k]   (+ db through a virtual ones column, as in the dense layers)
//             split over m (= N*OH*OW, tens of millions) into planner-chosen slices, finished deterministically
//
// Gather cost: each thread fetches 8 consecutive k (or 8 consecutive columns) of one row per slab with scalar
// loads; row decomposition (n, y, x) is done once per tile (or once per slab when the row index is the
// contraction index), the (c, kh, kw) decomposition comes from a small table built per call.
#include "evae_gemm_core.h"
#include <type_traits>

namespace evae {

struct ConvGeom {
  int N, C, H, W;      // input  [N, C, H, W]
  int Co, KH, KW, stride, pad;
  int OH, OW;          // output [N, Co, OH, OW]
};

enum { CONV_FWD = 0, CONV_DGRAD = 1, CONV_WGRAD = 2 };
enum { CEPI_LINEAR = 0, CEPI_GATED = 1, CEPI_DX = 2, CEPI_RAW = 3 };

struct ConvArgs {
  ConvGeom g;
  const float* src0;    // FWD/WGRAD: x ; DGRAD: dy_h
  const float* src1;    // DGRAD: dy_g (or NULL) ; WGRAD: dy_h
  const float* src2;    // WGRAD: dy_g (or NULL)
  const float* w0;      // FWD: Wh [Co x K] ; DGRAD: W' [Ktot x C]
  const float* w1;      // FWD gated: Wg
  const float* bias0;
  const float* bias1;
  const int* tab;       // k -> (c << 16 | kh << 8 | kw)   (FWD, WGRAD: c = ci ; DGRAD: c = pair*Co + co)
  float* out0;
  float* out1;
  float* out2;
  int M, Ncols, K;      // GEMM extents: rows, columns, contraction
  int ksplit;           // slabs per blockIdx.z (WGRAD)
  int act;
  float lo, hi;
  int tiles_m, tiles_n;
  int rows_per_bank;    // WGRAD: Co (row r >= Co reads dy_g)
  // DGRAD: one GEMM per stride-parity class (blockIdx.z): an input pixel (ih, iw) only meets the taps with
  // kh = (ih + pad) mod s (mod s), so each class has its own (shorter) tap list and permuted filter block
  int cls_K[9], cls_tab[9], cls_w[9];
};

__device__ __forceinline__ void decomp_row(int m, int P, int OWW, int& n, int& y, int& x) {
  n = m / P;
  const int pix = m - n * P;
  y = pix / OWW;
  x = pix - y * OWW;
}

// 8 gathered elements of one tile row (KC operand) or one tile k-row (RC operand)
struct Gathered {
  float v[8];
  unsigned mask;   // bit j: element j is in range
};

template <int MODE, int EPI, int BN_, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void conv_gemm_kernel(const ConvArgs a) {
  constexpr bool GATED = (EPI == CEPI_GATED);
  constexpr int GNT = 64 * NW;
  static_assert(NW == 8, "conv kernel is written for 8-wave blocks");
  constexpr int MT = 8 / NW, NT = BN_ / 64;
  static_assert(!GATED || BN_ == 128, "gated epilogue needs the h and g column tiles in one wave");
  constexpr bool B_KC = (MODE == CONV_FWD);     // FWD: filters [Co x K] k-contiguous; DGRAD/WGRAD: [k][cols]
  constexpr int BRS = BN_ + 4;
  constexpr int STAGE = A_TILE_FLOATS + b_tile_floats(BN_);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  auto As = [&](int b) -> float* { return smem + b * STAGE; };
  auto Bs = [&](int b) -> float* { return smem + b * STAGE + A_TILE_FLOATS; };
  const ConvGeom& g = a.g;

  const int ntiles = a.tiles_m * a.tiles_n;
  int tile;
  {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int qq = ntiles >> 3, rr = ntiles & 7;
    tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + slot;
  }
  const int tm = tile / a.tiles_n, tn = tile - tm * a.tiles_n;
  const int m0 = tm * BM;
  const int n0 = GATED ? tn * 64 : tn * BN_;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tid = threadIdx.x;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int KK = a.K;                         // contraction length (per class in DGRAD)
  const int* tabp = a.tab;
  const float* w0p = a.w0;
  int cls_py = 0, cls_px = 0, Hc = g.H, Wc = g.W, Mrows = a.M;
  if (MODE == CONV_DGRAD) {
    const int cls = blockIdx.z;
    cls_py = cls / g.stride; cls_px = cls - cls_py * g.stride;
    Hc = (g.H - cls_py + g.stride - 1) / g.stride;
    Wc = (g.W - cls_px + g.stride - 1) / g.stride;
    Mrows = g.N * Hc * Wc;
    if (m0 >= Mrows) return;
    KK = a.cls_K[cls]; tabp = a.tab + a.cls_tab[cls]; w0p = a.w0 + a.cls_w[cls];
  }
  const int nslab = (KK + BK - 1) / BK;
  int s_begin = 0, s_end = nslab;
  if (a.ksplit > 0 && MODE != CONV_DGRAD) {
    s_begin = blockIdx.z * a.ksplit;
    int e = s_begin + a.ksplit;
    if (e < s_end) s_end = e;
  }

  const int HW = g.H * g.W, P = g.OH * g.OW;

  // ---- A operand: row = tid >> 2 (0..127), 8 consecutive k starting at (tid & 3) * 8 ----------------------
  const int arow = tid >> 2, akq = (tid & 3) * 8;
  int ar_n = 0, ar_y = 0, ar_x = 0;
  bool ar_ok = false;
  {
    const int m = m0 + arow;
    if (MODE == CONV_FWD) {            // row = output pixel (n, oh, ow)
      ar_ok = m < a.M;
      if (ar_ok) { decomp_row(m, P, g.OW, ar_n, ar_y, ar_x); ar_y = ar_y * g.stride - g.pad; ar_x = ar_x * g.stride - g.pad; }
    } else if (MODE == CONV_DGRAD) {   // row = input pixel (n, ih, iw) of this parity class
      ar_ok = m < Mrows;
      if (ar_ok) {
        decomp_row(m, Hc * Wc, Wc, ar_n, ar_y, ar_x);
        ar_y = ar_y * g.stride + cls_py + g.pad; ar_x = ar_x * g.stride + cls_px + g.pad;
      }
    } else {                           // WGRAD: row = output channel of bank 0/1
      ar_ok = m < a.M;
    }
  }
  auto gather_a = [&](int k0) -> Gathered {
    Gathered r;
    r.mask = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + akq + j;
      bool ok = ar_ok && k < KK;
      size_t addr = 0;
      const float* src = a.src0;
      if (MODE == CONV_FWD) {
        const int t = ok ? a.tab[k] : 0;
        const int c = t >> 16, kh = (t >> 8) & 255, kw = t & 255;
        const int y = ar_y + kh, x = ar_x + kw;
        ok = ok && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
        addr = ok ? ((size_t)(ar_n * g.C + c) * HW + y * g.W + x) : 0;
      } else if (MODE == CONV_DGRAD) {
        const int t = ok ? tabp[k] : 0;
        int c = t >> 16;
        const int kh = (t >> 8) & 255, kw = t & 255;
        if (c >= g.Co) { c -= g.Co; src = a.src1; }
        int yy = ar_y - kh, xx = ar_x - kw;      // multiples of the stride by construction of the class
        ok = ok && yy >= 0 && xx >= 0;
        if (g.stride != 1) { yy /= g.stride; xx /= g.stride; }
        ok = ok && yy < g.OH && xx < g.OW;
        addr = ok ? ((size_t)(ar_n * g.Co + c) * P + yy * g.OW + xx) : 0;
      } else {                          // WGRAD: A[row = co'][k = m]: dy in NCHW, m = (n, pix)
        int co = m0 + arow;
        src = a.src1;
        if (co >= a.rows_per_bank) { co -= a.rows_per_bank; src = a.src2; }
        const int n = k / P, pix = k - n * P;
        addr = ok ? ((size_t)(n * g.Co + co) * P + pix) : 0;
        if (!ok) src = a.src1;
      }
      r.v[j] = src[addr];
      r.mask |= (ok ? 1u : 0u) << j;
    }
    return r;
  };
  auto store_a = [&](float* tile, const Gathered& r) {
    float4 lo4, hi4;
    lo4.x = (r.mask & 1u) ? r.v[0] : 0.f;   lo4.y = (r.mask & 2u) ? r.v[1] : 0.f;
    lo4.z = (r.mask & 4u) ? r.v[2] : 0.f;   lo4.w = (r.mask & 8u) ? r.v[3] : 0.f;
    hi4.x = (r.mask & 16u) ? r.v[4] : 0.f;  hi4.y = (r.mask & 32u) ? r.v[5] : 0.f;
    hi4.z = (r.mask & 64u) ? r.v[6] : 0.f;  hi4.w = (r.mask & 128u) ? r.v[7] : 0.f;
    *reinterpret_cast<float4*>(tile + arow * KS + akq) = lo4;
    *reinterpret_cast<float4*>(tile + arow * KS + akq + 4) = hi4;
  };

  // ---- B operand -----------------------------------------------------------------------------------------
  // FWD: filters, KC vector loads (2 float4 per thread for BN=128, 1 for 64), gated rows = [wc][h|g][32]
  // DGRAD: W' [Ktot x C], RC vector loads
  // WGRAD: im2col tile [32 m][BN k-columns] gathered: k-row = tid >> 4, BN/16 consecutive columns
  constexpr int BNV = BN_ * BK / 4 / GNT;       // float4 per thread for the vector forms
  const float* bbase[BNV];
  bool bok[BNV];
  if (MODE == CONV_FWD) {
#pragma unroll
    for (int i = 0; i < BNV; ++i) {
      const int f = tid + GNT * i;
      const int r = f >> 3;
      int n;
      const float* w = a.w0;
      if (GATED) { n = n0 + (r >> 6) * 32 + (r & 31); if (r & 32) w = a.w1; }
      else n = n0 + r;
      bok[i] = n < a.Ncols;
      bbase[i] = w + (size_t)(bok[i] ? n : 0) * a.K + 4 * (f & 7);
    }
  }
  constexpr int WCOLS = BN_ / 16;                // WGRAD columns per thread
  const int brow = tid >> 4, bcq = (tid & 15) * WCOLS;
  int wtab[MODE == CONV_WGRAD ? WCOLS : 1];
  if (MODE == CONV_WGRAD) {
#pragma unroll
    for (int j = 0; j < WCOLS; ++j) {
      const int kc = n0 + bcq + j;                // column = k index of the filter, or the ones column
      wtab[j] = kc < a.Ncols - 1 ? a.tab[kc] : (kc == a.Ncols - 1 ? -1 : -2);
    }
  }
  struct BRegs { float4 v[BNV > 0 ? BNV : 1]; float s[MODE == CONV_WGRAD ? WCOLS : 1]; unsigned mask; };
  const bool kvec = (a.K & 3) == 0;
  auto load_b = [&](int k0) -> BRegs {
    BRegs r;
    r.mask = 0;
    if (MODE == CONV_FWD) {
#pragma unroll
      for (int i = 0; i < BNV; ++i) {
        const int f = tid + GNT * i;
        const int k = k0 + 4 * (f & 7);
        if (kvec) {
          const bool ok = bok[i] && (k + 4 <= a.K);
          r.v[i] = *reinterpret_cast<const float4*>(bbase[i] + (ok ? k0 : -4 * (f & 7)));
          r.mask |= (ok ? 15u : 0u) << (4 * i);
        } else {                                   // K not a multiple of 4 (7x7x1 = 49): element-wise
          const float* p = bbase[i] + k0;
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
          const int valid = bok[i] ? (a.K - k) : 0;
          if (valid > 0) t.x = p[0];
          if (valid > 1) t.y = p[1];
          if (valid > 2) t.z = p[2];
          if (valid > 3) t.w = p[3];
          r.v[i] = t;
          r.mask |= 15u << (4 * i);
        }
      }
    } else if (MODE == CONV_DGRAD) {
      constexpr int RQ = BN_ / 4;
#pragma unroll
      for (int i = 0; i < BNV; ++i) {
        const int f = tid + GNT * i;
        const int k = k0 + f / RQ;
        const int c = n0 + 4 * (f % RQ);
        const bool ok = k < KK && (c + 4 <= a.Ncols);
        if ((a.Ncols & 3) == 0) {
          r.v[i] = *reinterpret_cast<const float4*>(w0p + (size_t)(ok ? k : 0) * a.Ncols + (ok ? c : 0));
          r.mask |= (ok ? 15u : 0u) << (4 * i);
        } else {                                   // C = 1 or 3: element-wise
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
          if (k < KK) {
            const float* p = w0p + (size_t)k * a.Ncols;
            if (c + 0 < a.Ncols) t.x = p[c + 0];
            if (c + 1 < a.Ncols) t.y = p[c + 1];
            if (c + 2 < a.Ncols) t.z = p[c + 2];
            if (c + 3 < a.Ncols) t.w = p[c + 3];
          }
          r.v[i] = t;
          r.mask |= 15u << (4 * i);
        }
      }
    } else {                                       // WGRAD: gather im2col(x)[m = k0 + brow][columns]
      const int m = k0 + brow;
      const bool mok = m < a.K;
      int n = 0, y = 0, x = 0;
      if (mok) { decomp_row(m, P, g.OW, n, y, x); y = y * g.stride - g.pad; x = x * g.stride - g.pad; }
#pragma unroll
      for (int j = 0; j < WCOLS; ++j) {
        const int t = wtab[j];
        const int c = t >> 16, kh = (t >> 8) & 255, kw = t & 255;
        const int yy = y + kh, xx = x + kw;
        const bool ok = mok && t >= 0 && (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W;
        r.s[j] = a.src0[ok ? ((size_t)(n * g.C + c) * HW + yy * g.W + xx) : 0];
        r.mask |= (ok ? 1u : ((mok && t == -1) ? 2u : 0u)) << (2 * j);   // 2 = the ones column (bias gradient)
      }
    }
    return r;
  };
  auto store_b = [&](float* tile, const BRegs& r) {
    if (MODE == CONV_WGRAD) {
      float o[WCOLS];
#pragma unroll
      for (int j = 0; j < WCOLS; ++j) {
        const unsigned sel = (r.mask >> (2 * j)) & 3u;
        o[j] = sel == 1u ? r.s[j] : (sel == 2u ? 1.f : 0.f);
      }
#pragma unroll
      for (int j = 0; j < WCOLS; j += 4)
        *reinterpret_cast<float4*>(tile + brow * BRS + bcq + j) = make_float4(o[j], o[j + 1], o[j + 2], o[j + 3]);
    } else {
#pragma unroll
      for (int i = 0; i < BNV; ++i) {
        const int f = tid + GNT * i;
        const float4 w = ((r.mask >> (4 * i)) & 15u) ? r.v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (B_KC) *reinterpret_cast<float4*>(tile + (f >> 3) * KS + 4 * (f & 7)) = w;
        else      *reinterpret_cast<float4*>(tile + (f / (BN_ / 4)) * BRS + 4 * (f % (BN_ / 4))) = w;
      }
    }
  };

  if (s_begin < s_end) {
    Gathered ra = gather_a(s_begin * BK);
    BRegs rb = load_b(s_begin * BK);
    store_a(As(0), ra);
    store_b(Bs(0), rb);
    if (s_begin + 1 < s_end) { ra = gather_a((s_begin + 1) * BK); rb = load_b((s_begin + 1) * BK); }
    __syncthreads();
    // same slab schedule as the dense kernel (evae_gemm_kernel.h): memory work slotted between the MFMA steps,
    // no branches around memory instructions in the steady state (last two slabs peeled), barrier before
    // the last k-group with the next slab's first fragments requested right behind it
    Frag<MT, NT> f0, f1;
    load_frag<true, B_KC, MT, NT, BN_>(f0, As(0), Bs(0), wr, wc, lane, 0);
#define EVAE_SB __builtin_amdgcn_sched_barrier(0)
    auto slab = [&](int s, auto ST_, auto LD_, auto NX_) {
      constexpr bool ST = decltype(ST_)::value, LD = decltype(LD_)::value, NX = decltype(NX_)::value;
      const int cur = (s - s_begin) & 1;
      EVAE_SB; mma_step<MT, NT>(acc, f0, 0); EVAE_SB;
      load_frag<true, B_KC, MT, NT, BN_>(f1, As(cur), Bs(cur), wr, wc, lane, 1);
      if constexpr (ST) store_a(As(cur ^ 1), ra);
      EVAE_SB; mma_step<MT, NT>(acc, f0, 1); EVAE_SB;
      if constexpr (ST) store_b(Bs(cur ^ 1), rb);
      EVAE_SB; mma_step<MT, NT>(acc, f0, 2); EVAE_SB;
      if constexpr (LD) ra = gather_a((s + 2) * BK);
      EVAE_SB; mma_step<MT, NT>(acc, f0, 3); EVAE_SB;
      if constexpr (LD) rb = load_b((s + 2) * BK);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        EVAE_SB; mma_step<MT, NT>(acc, f1, q); EVAE_SB;
        if (q < MT + NT) load_frag_part<true, B_KC, MT, NT, BN_>(f0, As(cur), Bs(cur), wr, wc, lane, 2, q);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        EVAE_SB; mma_step<MT, NT>(acc, f0, q); EVAE_SB;
        if (q < MT + NT) load_frag_part<true, B_KC, MT, NT, BN_>(f1, As(cur), Bs(cur), wr, wc, lane, 3, q);
      }
      EVAE_SB;
      __syncthreads();
      EVAE_SB;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mma_step<MT, NT>(acc, f1, q); EVAE_SB;
        if constexpr (NX) {
          if (q < MT + NT) load_frag_part<true, B_KC, MT, NT, BN_>(f0, As(cur ^ 1), Bs(cur ^ 1), wr, wc, lane, 0, q);
        }
        EVAE_SB;
      }
    };
    constexpr std::true_type T{};
    constexpr std::false_type F{};
    int s = s_begin;
    for (; s + 2 < s_end; ++s) slab(s, T, T, T);
    if (s + 1 < s_end) { slab(s, T, F, T); ++s; }
    slab(s, F, F, F);
#undef EVAE_SB
  }

  // ---- epilogue -------------------------------------------------------------------------------------------
  const int l31 = lane & 31, lh = lane >> 5;
  if (EPI == CEPI_RAW) {                 // weight gradient partials [z][rows][Ncols]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wc * 32 * NT + nt * 32 + l31;
      if (n >= a.Ncols) continue;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wr * 32 * MT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < a.M) a.out0[(size_t)blockIdx.z * a.M * a.Ncols + (size_t)m * a.Ncols + n] = acc[mt][nt][r];
        }
    }
    return;
  }
  // NCHW stores: row m = (n, pix) over PO pixels per image, column = channel
  const int PO = (MODE == CONV_FWD) ? P : HW;
  const int CO = a.Ncols;
  const int mbase = m0 + wr * 32 * MT;
  const int PC = (MODE == CONV_DGRAD) ? Hc * Wc : PO;   // rows per image in this launch's row space
  int nb = mbase / PC;
  int pb = mbase - nb * PC;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int m = mbase + d;
      if (m >= Mrows) continue;
      int n = nb, pix = pb + d;
      while (pix >= PC) { pix -= PC; ++n; }
      if (MODE == CONV_DGRAD && g.stride != 1) {           // class-local (y', x') -> input pixel
        const int yc = pix / Wc, xc = pix - yc * Wc;
        pix = (yc * g.stride + cls_py) * g.W + xc * g.stride + cls_px;
      }
      if (GATED) {
        const int co = n0 + wc * 32 + l31;
        if (co < CO) {
          const float h = apply_act(acc[mt][0][r] + (a.bias0 ? a.bias0[co] : 0.f), a.act, a.lo, a.hi);
          const float s = 1.0f / (1.0f + expf(-(acc[mt][NT - 1][r] + (a.bias1 ? a.bias1[co] : 0.f))));
          const size_t o = ((size_t)n * CO + co) * PO + pix;
          a.out0[o] = h * s;
          if (a.out1) a.out1[o] = h;
          if (a.out2) a.out2[o] = s;
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int co = n0 + wc * 32 * NT + nt * 32 + l31;
          if (co >= CO) continue;
          const size_t o = ((size_t)n * CO + co) * PO + pix;
          if (EPI == CEPI_LINEAR) {
            const float pre = acc[mt][nt][r] + (a.bias0 ? a.bias0[co] : 0.f);
            if (a.out1) a.out1[o] = pre;
            a.out0[o] = apply_act(pre, a.act, a.lo, a.hi);
          } else {
            a.out0[o] = acc[mt][nt][r];
          }
        }
      }
    }
}

// k -> (c << 16 | kh << 8 | kw) for k = (c, kh, kw) row-major
__global__ void conv_tab_kernel(int* __restrict__ tab, int K, int KHKW, int KW) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const int c = k / KHKW, r = k - c * KHKW, kh = r / KW, kw = r - kh * KW;
  tab[k] = (c << 16) | (kh << 8) | kw;
}

// data-gradient tap table of one stride-parity class: k = (c, i, j) -> (c << 16 | kh << 8 | kw) with
// kh = rh + i*s, kw = rw + j*s; c = pair*Co + co
__global__ void conv_tab_class_kernel(int* __restrict__ tab, int K, int nkh, int nkw, int rh, int rw, int s) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const int per = nkh * nkw;
  const int c = k / per, r = k - c * per, i = r / nkw, j = r - i * nkw;
  tab[k] = (c << 16) | ((rh + i * s) << 8) | (rw + j * s);
}

// W'[k][ci] = W_pair[co][ci][kh][kw] for the taps listed in tab (c = pair*Co + co)
__global__ void conv_permute_w_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                      const int* __restrict__ tab, int K, int Co, int C, int KH, int KW,
                                      float* __restrict__ wp) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * C) return;
  const int k = i / C, ci = i - k * C;
  const int t = tab[k];
  int c = t >> 16;
  const int kh = (t >> 8) & 255, kw = t & 255;
  const float* w = w0;
  if (c >= Co) { c -= Co; w = w1; }
  wp[i] = w[(((size_t)c * C + ci) * KH + kh) * KW + kw];
}

static bool geom_ok(const evae_conv_desc_t* d, ConvGeom* g) {
  if (!d || d->N < 0 || d->C <= 0 || d->H <= 0 || d->W <= 0 || d->Co <= 0 || d->KH <= 0 || d->KW <= 0 ||
      d->stride <= 0 || d->pad < 0 || d->KH > 255 || d->KW > 255)
    return false;
  g->N = d->N; g->C = d->C; g->H = d->H; g->W = d->W; g->Co = d->Co; g->KH = d->KH; g->KW = d->KW;
  g->stride = d->stride; g->pad = d->pad;
  g->OH = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  g->OW = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  return g->OH > 0 && g->OW > 0 && (long)d->C * d->KH * d->KW < 32768 && 2L * d->Co < 32768;
}

template <int MODE, int EPI, int BN_>
static int launch_conv(ConvArgs& a, int nz, hipStream_t stream, const char* what) {
  static bool attr = false;
  constexpr size_t lds = gemm_lds_bytes(BN_);
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_gemm_kernel<MODE, EPI, BN_, 8>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  a.tiles_m = cdiv(a.M, BM);
  a.tiles_n = cdiv(a.Ncols, EPI == CEPI_GATED ? 64 : BN_);
  conv_gemm_kernel<MODE, EPI, BN_, 8><<<dim3(a.tiles_m * a.tiles_n, 1, nz), 512, lds, stream>>>(a);
  return check_launch(what);
}

}  // namespace evae

using namespace evae;

// workspace: [table ints | permuted filters | split-K partials]
extern "C" size_t evae_conv2d_workspace_bytes(const evae_conv_desc_t* d, int what, int gated) {
  ConvGeom g;
  if (!geom_ok(d, &g)) return 256;
  const size_t K = (size_t)g.C * g.KH * g.KW, Kd = (size_t)g.Co * g.KH * g.KW * (gated ? 2 : 1);
  if (what == 0) return align_up(K * sizeof(int), 256) + 256;
  if (what == 1) return align_up(Kd * sizeof(int), 256) + align_up(Kd * g.C * sizeof(float), 256) + 256;
  const long M = (long)g.N * g.OH * g.OW;
  const int rows = g.Co * (gated ? 2 : 1);
  Plan pl = make_plan(rows, (int)K + 1, cdiv((int)(M > 0 ? M : 1), BK), false, true, 1);
  return align_up(K * sizeof(int), 256) + align_up((size_t)pl.nz * rows * (K + 1) * sizeof(float), 256) + 256;
}

extern "C" int evae_conv2d_fwd(const float* x, const evae_conv_desc_t* d, const float* wh, const float* bh,
                               const float* wg, const float* bg, int act, float act_lo, float act_hi,
                               float* out, float* save_h, float* save_s, void* ws, size_t ws_bytes,
                               evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom g;
  EVAE_REQUIRE(geom_ok(d, &g), "conv2d_fwd: bad geometry");
  EVAE_REQUIRE(act >= 0 && act <= 2, "conv2d_fwd: bad activation %d", act);
  if (g.N == 0) return EVAE_OK;
  EVAE_REQUIRE(x && wh && out, "conv2d_fwd: null pointer");
  const bool gated = wg != nullptr;
  if (ws == nullptr || ws_bytes < evae_conv2d_workspace_bytes(d, 0, gated)) {
    set_error("conv2d_fwd: workspace too small (%zu)", ws_bytes);
    return EVAE_EWORKSPACE;
  }
  const int K = g.C * g.KH * g.KW;
  int* tab = (int*)ws;
  conv_tab_kernel<<<cdiv(K, 256), 256, 0, stream>>>(tab, K, g.KH * g.KW, g.KW);
  int rc = check_launch("conv_tab");
  if (rc) return rc;
  ConvArgs a = {};
  a.g = g; a.src0 = x; a.w0 = wh; a.w1 = wg; a.bias0 = bh; a.bias1 = bg; a.tab = tab;
  a.out0 = out; a.out1 = save_h; a.out2 = save_s;
  a.M = g.N * g.OH * g.OW; a.Ncols = g.Co; a.K = K; a.act = act; a.lo = act_lo; a.hi = act_hi;
  if (gated) return launch_conv<CONV_FWD, CEPI_GATED, 128>(a, 1, stream, "conv2d_fwd(gated)");
  if (g.Co > 64) return launch_conv<CONV_FWD, CEPI_LINEAR, 128>(a, 1, stream, "conv2d_fwd");
  return launch_conv<CONV_FWD, CEPI_LINEAR, 64>(a, 1, stream, "conv2d_fwd");
}

extern "C" int evae_conv2d_bwd_data(const float* dyh, const float* wh, const float* dyg, const float* wg,
                                    const evae_conv_desc_t* d, float* dx, void* ws, size_t ws_bytes,
                                    evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom g;
  EVAE_REQUIRE(geom_ok(d, &g), "conv2d_bwd_data: bad geometry");
  if (g.N == 0) return EVAE_OK;
  EVAE_REQUIRE(dyh && wh && dx, "conv2d_bwd_data: null pointer");
  EVAE_REQUIRE((dyg == nullptr) == (wg == nullptr), "conv2d_bwd_data: dyg/wg must come together");
  const bool gated = wg != nullptr;
  if (ws == nullptr || ws_bytes < evae_conv2d_workspace_bytes(d, 1, gated)) {
    set_error("conv2d_bwd_data: workspace too small (%zu)", ws_bytes);
    return EVAE_EWORKSPACE;
  }
  const int khw = g.KH * g.KW;
  const int Kd = g.Co * khw * (gated ? 2 : 1);
  const int ctot = g.Co * (gated ? 2 : 1);
  int* tab = (int*)ws;
  float* wp = (float*)((char*)ws + align_up((size_t)Kd * sizeof(int), 256));
  EVAE_REQUIRE(g.stride <= 3, "conv2d_bwd_data: stride %d > 3 unsupported", g.stride);
  ConvArgs a = {};
  a.g = g; a.src0 = dyh; a.src1 = dyg ? dyg : dyh; a.w0 = wp; a.tab = tab; a.out0 = dx;
  a.M = g.N * cdiv(g.H, g.stride) * cdiv(g.W, g.stride);   // rows of the largest class (0, 0)
  a.Ncols = g.C; a.K = Kd;
  int off = 0, rc = 0;
  const int ncls = g.stride * g.stride;
  for (int cls = 0; cls < ncls; ++cls) {
    const int py = cls / g.stride, px = cls % g.stride;
    const int rh = (py + g.pad) % g.stride, rw = (px + g.pad) % g.stride;
    const int nkh = rh < g.KH ? (g.KH - rh + g.stride - 1) / g.stride : 0;
    const int nkw = rw < g.KW ? (g.KW - rw + g.stride - 1) / g.stride : 0;
    const int Kc = ctot * nkh * nkw;
    a.cls_K[cls] = Kc; a.cls_tab[cls] = off; a.cls_w[cls] = off * g.C;
    if (Kc > 0) {
      conv_tab_class_kernel<<<cdiv(Kc, 256), 256, 0, stream>>>(tab + off, Kc, nkh, nkw, rh, rw, g.stride);
      rc = check_launch("conv_tab_class");
      if (rc) return rc;
      conv_permute_w_kernel<<<cdiv(Kc * g.C, 256), 256, 0, stream>>>(wh, wg, tab + off, Kc, g.Co, g.C, g.KH, g.KW,
                                                                   wp + (size_t)off * g.C);
      rc = check_launch("conv_permute_w");
      if (rc) return rc;
    }
    off += Kc;
  }
  if (g.C > 64) return launch_conv<CONV_DGRAD, CEPI_DX, 128>(a, ncls, stream, "conv2d_bwd_data");
  return launch_conv<CONV_DGRAD, CEPI_DX, 64>(a, ncls, stream, "conv2d_bwd_data");
}

extern "C" int evae_conv2d_bwd_weight(const float* dyh, const float* dyg, const float* x,
                                      const evae_conv_desc_t* d, float* dw, float* db, void* ws,
                                      size_t ws_bytes, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom g;
  EVAE_REQUIRE(geom_ok(d, &g), "conv2d_bwd_weight: bad geometry");
  EVAE_REQUIRE(dw != nullptr, "conv2d_bwd_weight: null dw");
  const bool gated = dyg != nullptr;
  if (ws == nullptr || ws_bytes < evae_conv2d_workspace_bytes(d, 2, gated)) {
    set_error("conv2d_bwd_weight: workspace too small (%zu)", ws_bytes);
    return EVAE_EWORKSPACE;
  }
  const int K = g.C * g.KH * g.KW, rows = g.Co * (gated ? 2 : 1);
  const long M = (long)g.N * g.OH * g.OW;
  if (M == 0) {
    (void)hipMemsetAsync(dw, 0, (size_t)rows * K * sizeof(float), stream);
    if (db) (void)hipMemsetAsync(db, 0, (size_t)rows * sizeof(float), stream);
    return check_launch("conv2d_bwd_weight(empty)");
  }
  EVAE_REQUIRE(dyh && x, "conv2d_bwd_weight: null pointer");
  EVAE_REQUIRE(M < (1L << 31), "conv2d_bwd_weight: too many output pixels");
  int* tab = (int*)ws;
  float* part = (float*)((char*)ws + align_up((size_t)K * sizeof(int), 256));
  conv_tab_kernel<<<cdiv(K, 256), 256, 0, stream>>>(tab, K, g.KH * g.KW, g.KW);
  int rc = check_launch("conv_tab");
  if (rc) return rc;
  Plan pl = make_plan(rows, K + 1, cdiv((int)M, BK), false, true, 1);
  ConvArgs a = {};
  a.g = g; a.src0 = x; a.src1 = dyh; a.src2 = dyg ? dyg : dyh; a.tab = tab; a.out0 = part;
  a.M = rows; a.Ncols = K + 1; a.K = (int)M; a.ksplit = pl.nz > 1 ? pl.ksplit : 0; a.rows_per_bank = g.Co;
  if (pl.bn == 128) rc = launch_conv<CONV_WGRAD, CEPI_RAW, 128>(a, pl.nz, stream, "conv2d_bwd_weight");
  else rc = launch_conv<CONV_WGRAD, CEPI_RAW, 64>(a, pl.nz, stream, "conv2d_bwd_weight");
  if (rc) return rc;
  FinishArgs f = {};
  f.part = part; f.nz = pl.nz; f.M = rows; f.N = K + 1; f.ldo = K + 1; f.epi = EPI_RAW; f.out0 = dw;
  f.ones_col = K; f.out_db = db;
  return launch_finish(f, stream);
}
