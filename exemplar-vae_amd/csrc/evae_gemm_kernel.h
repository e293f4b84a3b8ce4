// The fp32-MFMA GEMM kernel shared by the dense layers (evae_dense.hip) and the channels-last convolutions
// (evae_conv_cl.hip): argument block, tile loaders, the kernel and its launch helpers.  The header comment of
// evae_dense.hip describes the tiling; the slab schedule is explained where it is built, in the kernel below.
#pragma once
#include "evae_gemm_core.h"
#include "evae_p6_image.h"
#include <type_traits>
#include <algorithm>

namespace evae {

// Convolution as a GEMM over channels-last tensors (template parameter CV of gemm_kernel):
//   CV = 1 (forward, data gradient): A rows are the pixels (n, ry, rx) of a grid, the contraction runs over
//          (tap, channel) with the channel fastest and Cg % 32 == 0, so a 32-wide K-slab is 32 consecutive
//          channels of ONE tap for the whole block: the A tile is the dense KC tile with a per-slab byte
//          offset (tap) in an SGPR and one validity bit per (row, tap) -- no im2col arithmetic in the loop;
//          out-of-image taps ride the same out-of-range buffer offset as every other zero fill;
//   CV = 2 (weight gradient): the contraction runs over the pixels, B is the im2col matrix
//          [pixel][(tap, channel)] gathered as float4 (4 channels of one tap), A is dy, row-contiguous.
// Source element of row (n, ry, rx), tap t, channel c:
//   src[((n*IH + ry*rs + roy + tdy[t]) * IW + rx*rsx + rox + tdx[t]) * ps + c]
// Output row (CV = 1): ((n*OH2 + ry*os + ooy) * OW2 + rx*osx + oox)  (identity for the forward; the strided
// pixels of one stride-parity class -- or pairs of x-adjacent pixels, see evae_conv2d_cl_bwd_data -- for the data
// gradient).
struct FastDiv { unsigned mul, sh, one; };   // q = one ? n : (t = umulhi(mul, n), (t + ((n - t) >> 1)) >> sh)
__host__ __device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv d) {
#ifdef __HIP_DEVICE_COMPILE__
  const unsigned t = __umulhi(d.mul, n);
#else
  const unsigned t = (unsigned)(((unsigned long long)d.mul * n) >> 32);
#endif
  return d.one ? n : (t + ((n - t) >> 1)) >> d.sh;
}
static FastDiv make_fastdiv(unsigned d) {
  FastDiv f = {0u, 0u, 0u};
  if (d <= 1) { f.one = 1; return f; }
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  f.sh = l - 1;
  return f;
}
struct ConvMap {
  int Cg, ntaps;
  int creal;                   // CV = 1: channels of a tap that exist in the source (Cg rounded up to a multiple of 32 is the
                               // K-extent of a tap; chunks at or beyond creal read as zero)
  int ps;                      // floats per source pixel (Cg, or 2 Cg when the h and g gradients share one buffer)
  int RH, RW, IH, IW;
  int rs, rsx, roy, rox;       // anchor of row (ry, rx) in the source: (ry*rs + roy, rx*rsx + rox)
  int OH2, OW2, os, osx, ooy, oox;   // output row of (n, ry, rx): (n*OH2 + ry*os + ooy)*OW2 + rx*osx + oox
  int remap;                   // CV = 1: output rows are not the identity
  unsigned bias;               // bytes added to every per-row offset (the buffer base is moved back by bias + tbias)
  FastDiv div_rw, div_rhw;
  signed char tdy[64], tdx[64];
  int tsoff[64];               // CV = 1: byte offset of tap t, >= 0 after adding tbias
};

struct GemmArgs {
  const float* A[2];
  const float* B[2];
  int lda[2], ldb[2];
  int Kc[2];               // contraction length of each (A,B) pair
  int npairs;
  const int64_t* a_rows;   // KC A: gather of output rows
  const int64_t* b_krows;  // RC B: gather along the contraction index (weight gradient x rows)
  const float* Bg;         // gated: second weight matrix (g), same layout as B[0]
  int M, N;                // output rows / columns
  int ksplit;              // K-slabs per blockIdx.z (split-K); 0 = whole contraction
  const float* bias0;
  const float* bias1;
  float* out0;
  float* out1;
  float* out2;
  int ldo;
  const float* e0;         // EPI_GATE_BWD: gated output h*s of the layer below
  const float* e1;         //               gate s of the layer below
  int act;
  float lo, hi;
  int tiles_m, tiles_n;
  int ones_col;            // RC B: virtual all-ones column index (bias gradient folded into the weight GEMM); -1 = none
  int dbg;                 // EVAE_GEMM_DBG (tools/gemm_ablate.py): 4 = skip the epilogue, 512 = clock probe; 0 in production
  ConvMap cv;              // used by the CV != 0 instances only
  int* aux_cnt;            // EPI_DIST_COLLECT: per-column candidate counters
  int* aux_cand;           //                   candidate rows, [column][ldo]
  // EPI_PRIOR_*: dataset indices of the rows (exemplars) / columns (queries) for the leave-one-out mask (both or
  // neither), the constant -1/2 sum (log_var + log 2 pi) (one device float), and a device flag that, when non-zero,
  // cancels the launch
  const int64_t* pr_ridx;
  const int64_t* pr_cidx;
  const float* pr_cst_dev;
  const unsigned* skip_flag;
  // EPI_GATE_BWD_IMG: image base, K-slabs (of 32 rows) per column tile, row index of this launch's row 0 in the merged buffer
  unsigned short* img;
  int img_nslab, img_mbase;
  int ldo2;                // EPI_DIST_TILEMIN: > 0 = tile minima stored query-major, out0[n * ldo2 + tile] (else out0[tile * ldo + n])
  int ldq;                 // EPI_DIST_TILEMIN: > 0 = the kept distances query-major, out1[n * ldq + m], ldq a multiple of the row
                           // tile (whole tiles are written, rows beyond M as +inf); e1 may then be NULL: |b_n|^2 is left out of
                           // every distance of column n (a per-query constant: the two-launch top-K adds it to its threshold)
  float* tile_max;         // x6 kernel, e0 == NULL: [2 * tiles_m] the largest |a_m|^2 of each half row tile, plain stores (instead
                           // of the atomic maximum into aux_cnt, which wants a cleared word)
  int sk_local;            // > 0: XCD-local split-K mapping on a 1-D grid, value = number of slices (see gemm_kernel)
  P6Sink tsink;            // EPI_GATED / EPI_GATE_BWD, img != NULL: the result also leaves as the pre-split bf16 image of its transpose
                           // (evae_p6_image.h; EPI_GATE_BWD: [dh | dg]^T, and out0 may then be NULL -- no fp32 copy)
  int direct;              // EPI_RAW without split-K: out0[m * ldo + n] for n != ones_col, out1[m] for n == ones_col (a weight
                           // gradient over a few rows writes dw / db itself: no partial plane, no finish launch)
};

// "This loaded value is needed HERE": an empty asm that takes the register in and out, so the compiler places the wait for
// its load at this point, in code every lane's path runs through.  Used in front of epilogue store loops whose rows are
// guarded (m < M): left alone, the compiler sinks the first use of a loaded bias into the guarded blocks and, not knowing
// whether an earlier guarded block already waited, puts a full s_waitcnt vmcnt(0) in front of every row's stores -- loads
// and stores share vmcnt, so each row then waits for all stores in flight (measured: 10-13 us of a 17 us launch).
#define EVAE_PIN(x) asm volatile("" : "+v"(x))

// Out-of-range chunks are loaded from a clamped, always-mapped address and zeroed LATER, when the
// prefetched registers are written to LDS: masking right after the load would make the compiler wait
// for the prefetch before the MFMAs it is meant to overlap.
__device__ __forceinline__ float4 ld4v(const float* p) { return *reinterpret_cast<const float4*>(p); }
// Scalar path for odd extents (e.g. the 294-wide head of convhvae_2level): correct, not fast.
__device__ __forceinline__ float4 ld4s(const float* p, int valid) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid > 0) v.x = p[0];
  if (valid > 1) v.y = p[1];
  if (valid > 2) v.z = p[2];
  if (valid > 3) v.w = p[3];
  return v;
}

typedef __amdgpu_buffer_rsrc_t rsrc_t;
// Raw buffer resource over a wave-uniform pointer (stride 0, DATA_FORMAT = 32-bit): offsets at or beyond
// num_records read as zero, which is how tails and out-of-matrix rows are blanked without any VALU work.
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned num_records) {
  const uintptr_t u = (uintptr_t)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, num_records, 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(rsrc_t r, unsigned voff, unsigned soff) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  const f32x4 v = __builtin_bit_cast(f32x4, u);   // (component access on the integer vector degrades to a dword load)
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ unsigned buf_ld1(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
}

// ---- tile loader: ROWS x BK floats per K-slab, NV float4 per thread -------------------------------
// KC: tile[row][k], f = tid + 256 i -> row = f >> 3, k = 4 (f & 7)
// RC: tile[k][row], f -> k = f / (ROWS/4), row = 4 (f % (ROWS/4))
template <int ROWS, bool KC, int GNT>
struct TileLoader {
  static constexpr int NV = ROWS * BK / 4 / GNT;
  static constexpr int RS = ROWS + 4;
  static constexpr int RQ = ROWS / 4;
  const float* base[NV];
  bool rowok[NV];

  __device__ __forceinline__ void init(const float* src, int ld, int r0, int nrows, const int64_t* gather) {
    if (KC) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        int f = threadIdx.x + GNT * i;
        int r = r0 + (f >> 3);
        rowok[i] = r < nrows;
        int64_t gr = rowok[i] ? (gather ? gather[r] : (int64_t)r) : 0;
        base[i] = src + gr * ld + 4 * (f & 7);
      }
    }
  }
  template <bool VEC>
  __device__ __forceinline__ unsigned load_kc(float4 (&v)[NV], int k0, int kend) const {
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int f = threadIdx.x + GNT * i;
      int k = k0 + 4 * (f & 7);
      if (VEC) {
        const bool ok = rowok[i] && (k + 4 <= kend);
        v[i] = ld4v(base[i] + (ok ? k0 : -4 * (f & 7)));   // invalid -> start of a mapped row
        mask |= (ok ? 1u : 0u) << (2 * i);
      } else {
        v[i] = ld4s(base[i] + k0, rowok[i] ? (kend - k) : 0);
        mask |= 1u << (2 * i);
      }
    }
    return mask;
  }
  // mask: 2 bits per chunk -- 0 = zero fill, 1 = loaded data, 2 = the virtual ones column (1,0,0,0)
  template <bool VEC>
  __device__ __forceinline__ unsigned load_rc(float4 (&v)[NV], const float* src, int ld, int r0, int nrows,
                                              int k0, int kend, const int64_t* kgather, int ones_col) const {
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int f = threadIdx.x + GNT * i;
      int k = k0 + f / RQ;
      int r = r0 + 4 * (f % RQ);
      const bool kok = k < kend;
      int64_t gk = kok ? (kgather ? kgather[k] : (int64_t)k) : 0;
      if (VEC) {
        const bool ok = kok && (r + 4 <= nrows);
        v[i] = ld4v(src + gk * ld + (ok ? r : 0));
        mask |= (ok ? 1u : ((kok && r == ones_col) ? 2u : 0u)) << (2 * i);
      } else {
        v[i] = ld4s(src + gk * ld + r, kok ? (nrows - r) : 0);
        if (kok && ones_col >= r && ones_col < r + 4) (&v[i].x)[ones_col - r] = 1.0f;
        mask |= 1u << (2 * i);
      }
    }
    return mask;
  }
  __device__ __forceinline__ void store(float* tile, const float4 (&v)[NV], unsigned mask) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int f = threadIdx.x + GNT * i;
      const unsigned sel = (mask >> (2 * i)) & 3u;
      const float4 w = sel == 1u ? v[i] : make_float4(sel == 2u ? 1.f : 0.f, 0.f, 0.f, 0.f);
      if (KC) *reinterpret_cast<float4*>(tile + (f >> 3) * KS + 4 * (f & 7)) = w;
      else    *reinterpret_cast<float4*>(tile + (f / RQ) * RS + 4 * (f % RQ)) = w;
    }
  }
};

// The epilogue of a block tile, shared by the fp32-MFMA kernel below and the split-bf16 kernel (evae_gemm_x6.h): both keep
// acc[mt][nt] in the 32x32 C/D layout with NW/2 wave rows x 2 wave columns.
template <int EPI, int BN_, int NW, int CV>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[8 / NW][BN_ / 64], const int m0, const int n0,
                                              const int tm, const int wr, const int wc, const int lane, float* smem,
                                              const int zslice) {
  constexpr bool GATED = (EPI == EPI_GATED || EPI == EPI_RAW_GATED);
  constexpr int MT = 8 / NW, NT = BN_ / 64;
  (void)GATED; (void)tm; (void)zslice;
  // ---- epilogue.  acc[mt][nt][r] <-> row m0 + wr*32*MT + mt*32 + (r&3) + 8*(r>>2) + 4*(lane>>5),
  //                                    col (within the wave tile) nt*32 + (lane&31)
  const int l31 = lane & 31, lh = lane >> 5;
  // output row of GEMM row m (identity unless this is the data gradient of a strided convolution)
  auto orow = [&](int m) -> size_t {
    if (CV == 1 && g.cv.remap) {
      const unsigned nn = fdiv((unsigned)m, g.cv.div_rhw), rem = (unsigned)m - nn * (unsigned)(g.cv.RH * g.cv.RW);
      const unsigned ry = fdiv(rem, g.cv.div_rw), rx = rem - ry * (unsigned)g.cv.RW;
      return ((size_t)nn * g.cv.OH2 + ry * g.cv.os + g.cv.ooy) * g.cv.OW2 + rx * g.cv.osx + g.cv.oox;
    }
    return (size_t)m;
  };
  if constexpr (EPI == EPI_DIST_TILEMIN || EPI == EPI_DIST_COLLECT) {
    // e0 = squared norms of the A rows, e1 = of the B rows; acc = dot products
    // The row norms go into registers BEFORE the first store: a load behind a store waits for every store in flight
    // (loads and stores share vmcnt and return in order), which serialised the 64 stores of a lane at one round trip
    // each (measured: 10-13 us of a 17 us launch).  e0 == nullptr: the x6 kernel left the norms of its tile in LDS
    // (evae_gemm_x6.h); two separate loops so that neither becomes a flat load.
    float e0v[MT][16];
    if (g.e0 != nullptr) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wr * 32 * MT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          e0v[mt][r] = m < g.M ? g.e0[m] : 0.f;
        }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) e0v[mt][r] = smem[512 + wr * 32 * MT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
    }
    float bnv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wc * 32 * NT + nt * 32 + l31;
      bnv[nt] = (g.e1 != nullptr && n < g.N) ? g.e1[n] : 0.f;
    }
    float tmin[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wc * 32 * NT + nt * 32 + l31;
      const bool nok = n < g.N;
      const float bn = bnv[nt];
      const float thr = (EPI == EPI_DIST_COLLECT && nok) ? g.bias0[n] : -INFINITY;
      tmin[nt] = INFINITY;
      float dq[4];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wr * 32 * MT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          // d is formed outside the row guard: the (single) wait for the preloaded norms then sits in code every path runs
          // through -- inside the guard the compiler repeats a full vmcnt(0) in every guarded block, stores included
          const bool mok = m < g.M;
          const float d = e0v[mt][r] + bn - 2.0f * acc[mt][nt][r];
          if (EPI == EPI_DIST_TILEMIN) {
            if (g.ldq > 0) {
              // query-major: a lane's four consecutive rows are one 16-byte store; the two lanes of a column fill a 128-byte line
              dq[r & 3] = mok ? d : INFINITY;
              if ((r & 3) == 3 && nok) *reinterpret_cast<float4*>(g.out1 + (size_t)n * g.ldq + (m - 3)) = make_float4(dq[0], dq[1], dq[2], dq[3]);
            } else if (g.out1 && nok && mok) g.out1[(size_t)m * g.ldo + n] = d;    // kept for the collect scan
            tmin[nt] = fminf(tmin[nt], mok ? d : INFINITY);
          } else if (mok && nok && d <= thr) {
            g.aux_cand[(size_t)n * g.ldo + atomicAdd(&g.aux_cnt[n], 1)] = m;
          }
        }
    }
    if (EPI == EPI_DIST_TILEMIN) {
      // lanes l / l+32 hold the same column; then the NW/2 wave rows through LDS (free after the last barrier)
      float* red = smem;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        tmin[nt] = fminf(tmin[nt], __shfl_xor(tmin[nt], 32, 64));
        if (lh == 0) red[wr * BN_ + wc * 32 * NT + nt * 32 + l31] = tmin[nt];
      }
      __syncthreads();
      if ((int)threadIdx.x < BN_ && n0 + (int)threadIdx.x < g.N) {
        float v = red[threadIdx.x];
#pragma unroll
        for (int w = 1; w < NW / 2; ++w) v = fminf(v, red[w * BN_ + threadIdx.x]);
        if (g.ldo2 > 0) g.out0[(size_t)(n0 + threadIdx.x) * g.ldo2 + tm] = v;
        else g.out0[(size_t)tm * g.ldo + n0 + threadIdx.x] = v;
      }
    }
  } else if constexpr (EPI == EPI_PRIOR_LSE) {
    // e0 = |c'_m|^2 (rows), e1 = |z'_n|^2 (columns), acc = c'_m . z'_n.  Per column the block's rows are reduced on
    // t = acc - |c'|^2/2 (= -d2/2 + |z'|^2/2: the column's own norm drops out of every difference) to
    // (max, sum exp(t - max), #masked); lanes l / l+32 hold the same column, the NW/2 wave rows meet in LDS.
    const bool masked = g.pr_ridx != nullptr && g.pr_cidx != nullptr;
    float hc[MT][16];
    long long ri[MT][16];
    unsigned live[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      live[mt] = 0u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 32 * MT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool ok = m < g.M;
        hc[mt][r] = ok ? 0.5f * g.e0[m] : 0.f;
        ri[mt][r] = (masked && ok) ? (long long)g.pr_ridx[m] : -2;
        if (ok) live[mt] |= 1u << r;
      }
    }
    float* red = smem;          // [NW/2][BN][3], free after the last barrier of the slab loop
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wc * 32 * NT + nt * 32 + l31;
      const long long zi = (masked && n < g.N) ? (long long)g.pr_cidx[n] : -1;
      float tmax = -INFINITY, ssum = 0.f, nm = 0.f;
      unsigned use[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        use[mt] = live[mt];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[mt][nt][r] -= hc[mt][r];
          if (masked && ((use[mt] >> r) & 1u) && (ri[mt][r] == zi || ri[mt][r] == (long long)EVAE_PRIOR_MASK_ALL)) { nm += 1.f; use[mt] &= ~(1u << r); }
          if ((use[mt] >> r) & 1u) tmax = fmaxf(tmax, acc[mt][nt][r]);
        }
      }
      const float mk = -tmax * kLog2e;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if ((use[mt] >> r) & 1u) ssum += fast_exp2(fmaf(acc[mt][nt][r], kLog2e, mk));
      const float ot = __shfl_xor(tmax, 32, 64), os = __shfl_xor(ssum, 32, 64), on = __shfl_xor(nm, 32, 64);
      const float mx = fmaxf(tmax, ot);
      const float fa = (tmax == mx) ? 1.f : fast_exp2((tmax - mx) * kLog2e);      // -inf - (-inf) never evaluated
      const float fb = (ot == mx) ? 1.f : fast_exp2((ot - mx) * kLog2e);
      if (lh == 0) {
        float* cb = red + (wr * BN_ + wc * 32 * NT + nt * 32 + l31) * 3;
        cb[0] = mx; cb[1] = ssum * fa + os * fb; cb[2] = nm + on;
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < BN_ && n0 + (int)threadIdx.x < g.N) {
      float mx = -INFINITY, sacc = 0.f, nacc = 0.f;
#pragma unroll
      for (int w = 0; w < NW / 2; ++w) mx = fmaxf(mx, red[(w * BN_ + threadIdx.x) * 3]);
#pragma unroll
      for (int w = 0; w < NW / 2; ++w) {
        const float tw = red[(w * BN_ + threadIdx.x) * 3];
        if (tw != -INFINITY) sacc += red[(w * BN_ + threadIdx.x) * 3 + 1] * fast_exp2((tw - mx) * kLog2e);
        nacc += red[(w * BN_ + threadIdx.x) * 3 + 2];
      }
      const int n = n0 + threadIdx.x;
      const size_t o = (size_t)tm * g.ldo + n;
      g.out0[o] = (mx == -INFINITY) ? -INFINITY : *g.pr_cst_dev + (mx - 0.5f * g.e1[n]);
      g.out1[o] = sacc;
      g.out2[o] = nacc;
    }
  } else if constexpr (EPI == EPI_PRIOR_P) {
    // P[m][n] = g_n exp(cst - d2_mn / 2 - lse_n), d2 = |c'|^2 + |z'|^2 - 2 acc; bias0 = lse, bias1 = upstream gradient.
    // Columns N <= n < ldo (padding up to a multiple of 4) are written as zeros: P is the operand of two more GEMMs.
    const bool masked = g.pr_ridx != nullptr && g.pr_cidx != nullptr;
    const float cst = *g.pr_cst_dev;
    // Everything the rows need is loaded (and pinned) before the first store: a load between the stores waits for every
    // store in flight (shared vmcnt), which ran the stores of a lane one round trip at a time.
    float e0v[MT][16];
    long long ri[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 32 * MT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool mok = m < g.M;
        e0v[mt][r] = mok ? g.e0[m] : 0.f;
        ri[mt][r] = (masked && mok) ? (long long)g.pr_ridx[m] : -2;
      }
    float zn[NT], gq[NT], kq[NT], kq2[NT];
    long long zi[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wc * 32 * NT + nt * 32 + l31;
      const bool nok = n < g.N;
      zn[nt] = nok ? g.e1[n] : 0.f;
      gq[nt] = nok ? g.bias1[n] : 0.f;
      kq[nt] = nok ? g.bias0[n] : 0.f;                         // token of the merge: row max ...
      kq2[nt] = nok ? g.bias0[g.N + n] * kLog2e : 0.f;         // ... and log of the normalised sum (evae_prior.hip::prior_merge_kernel)
      zi[nt] = (masked && nok) ? (long long)g.pr_cidx[n] : -1;
      EVAE_PIN(zn[nt]); EVAE_PIN(gq[nt]); EVAE_PIN(kq[nt]); EVAE_PIN(kq2[nt]);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) EVAE_PIN(e0v[mt][r]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wc * 32 * NT + nt * 32 + l31;
      if (n >= g.ldo) continue;
      const bool nok = n < g.N;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wr * 32 * MT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const float d = fmaxf(e0v[mt][r] + zn[nt] - 2.0f * acc[mt][nt][r], 0.f);
          bool ok = nok;
          if (masked) ok = ok && (ri[mt][r] != zi[nt]) && (ri[mt][r] != (long long)EVAE_PRIOR_MASK_ALL);
          const float pv = ok ? gq[nt] * fast_exp2(fmaf(fmaf(-0.5f, d, cst) - kq[nt], kLog2e, -kq2[nt])) : 0.f;
          if (m < g.M) g.out0[(size_t)m * g.ldo + n] = pv;
        }
    }
  } else if (GATED) {
    const int n = n0 + wc * 32 + l31;
    const bool nok = n < g.N;
    const bool timg = EPI == EPI_GATED && g.tsink.img != nullptr;        // (wave-uniform)
    if (nok || timg) {
      float bh = (EPI == EPI_GATED && g.bias0 && nok) ? g.bias0[n] : 0.f;
      float bg = (EPI == EPI_GATED && g.bias1 && nok) ? g.bias1[n] : 0.f;
      EVAE_PIN(bh); EVAE_PIN(bg);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float ov[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wr * 32 * MT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (EPI == EPI_GATED) {
            // The values are formed OUTSIDE the row guard: the wait for the bias loads then sits once, in code every path
            // runs through.  Inside the guard the compiler cannot know whether an earlier guarded block already waited and
            // puts a full s_waitcnt vmcnt(0) -- which also waits for every store in flight -- in front of each row's
            // stores: 32 store round trips in a row per lane (measured on the top-K epilogue: 10-13 us of a block's life).
            const float h = acc[mt][0][r] + bh;
            // sigmoid on the hardware exp2 / rcp (about 1e-7 relative): the precise expf costs ~15 VALU instructions
            // per element, and VALU issue is what the co-resident block's MFMAs wait on
            const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * (acc[mt][NT - 1][r] + bg)));
            ov[r] = h * s;
            if (m < g.M && nok) {
              const size_t o = orow(m) * g.ldo + n;
              g.out0[o] = ov[r];
              if (g.out1) g.out1[o] = h;
              if (g.out2) g.out2[o] = s;
            }
          } else if (m < g.M && nok) {
            {          // EPI_RAW_GATED: partial planes [z][2][M][N]
              const size_t plane = (size_t)g.M * g.N;
              const size_t o = (size_t)zslice * 2 * plane + (size_t)m * g.N + n;
              g.out0[o] = acc[mt][0][r];
              g.out0[o + plane] = acc[mt][NT - 1][r];
            }
          }
        }
        if (EPI == EPI_GATED && timg)       // the layer's output as the operand of the next GEMMs
          p6_emit_tile(g.tsink, g.tsink.row0 + n, nok, g.tsink.kbase + m0 + wr * 32 * MT + mt * 32, ov, lh);
      }
    }
  } else if constexpr (EPI == EPI_GATE_BWD_IMG) {
    // dh = v s, dg = v (h s)(1 - s) as in EPI_GATE_BWD; column cc of the merged [dh | dg] buffer is n (dh) / N + n (dg).  The
    // values leave as the bf16 tile images the byte layer's weight gradient reads (u8_gemm_kernel<false>'s B operand):
    //   img[((cc >> 7) * nslab + (row >> 5)) * 3 + p][cc & 127][slot (row & 31) >> 3, XOR-swizzled][row & 7]
    // (truncation split: w0 = top 8 significant bits, w1 of w - w0, w2 the rest: exact).  A lane holds rows 8 j + 4 lh .. + 3 of
    // its column, its partner lane ^ 32 the other four of the same eight: the two swap halves (one cross-lane move per
    // dword), so every lane stores whole 16-byte slots -- lanes lh = 0 the even j, lanes lh = 1 the odd j.  (r02 stored 8-byte
    // halves: twice the store instructions, each touching 64 lines.)  The image row of this launch's row 0 (img_mbase) must
    // be a multiple of 8; rows between M and the next multiple of 8 are written as zeros (they must not belong to
    // another launch: the host places the shorter launch last).
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wc * 32 * NT + nt * 32 + l31;
      if (n >= g.N) continue;                      // (the partner lane has the same n: the pair leaves together)
      float go[MT][16], sv[MT][16];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wr * 32 * MT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const size_t oe = (size_t)(m < g.M ? m : 0) * g.N + n;
          go[mt][r] = g.e0[oe];
          sv[mt][r] = g.e1[oe];
        }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          const int cc = which ? g.N + n : n;
          const int c = cc & 127;
          unsigned t[4][3][2];                      // [r-group j][term][dword]: four rows of one term = 8 bytes
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int q = 0; q < 3; ++q) t[j][q][0] = t[j][q][1] = 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int r = 4 * j + i;
              const int m = m0 + wr * 32 * MT + mt * 32 + 8 * j + 4 * lh + i;
              const float v = acc[mt][nt][r], s_ = sv[mt][r];
              float w = which ? v * go[mt][r] * (1.0f - s_) : v * s_;
              if (m >= g.M) w = 0.f;
              const unsigned u0 = __float_as_uint(w) & 0xFFFF0000u;
              const float r1 = w - __uint_as_float(u0);
              const unsigned u1 = __float_as_uint(r1) & 0xFFFF0000u;
              const float r2 = r1 - __uint_as_float(u1);
              const unsigned u2 = __float_as_uint(r2);
              const int sh = 16 * (i & 1);
              t[j][0][i >> 1] |= (u0 >> 16) << sh; t[j][1][i >> 1] |= (u1 >> 16) << sh; t[j][2][i >> 1] |= (u2 >> 16) << sh;
            }
          }
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {          // the pair of r-groups (2 q2, 2 q2 + 1): lh = 0 keeps the even one
            const int jk = 2 * q2 + lh;             // the group this lane stores
            const int ml = m0 + wr * 32 * MT + mt * 32 + 8 * jk;       // first of its eight rows (local)
            const int gm = ml + g.img_mbase, slab = gm >> 5, mi = gm & 31;
            char* base = reinterpret_cast<char*>(g.img) + ((size_t)((cc >> 7) * g.img_nslab + slab) * 3 * 128 + c) * 64 +
                         ((((mi >> 3) ^ ((c >> 2) & 3))) << 4);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              // send the group the partner keeps, receive the partner's half of mine
              const unsigned s0 = lh ? t[2 * q2][q][0] : t[2 * q2 + 1][q][0];
              const unsigned s1 = lh ? t[2 * q2][q][1] : t[2 * q2 + 1][q][1];
              const unsigned r0 = (unsigned)__shfl_xor((int)s0, 32), r1_ = (unsigned)__shfl_xor((int)s1, 32);
              const unsigned k0 = lh ? t[2 * q2 + 1][q][0] : t[2 * q2][q][0];
              const unsigned k1 = lh ? t[2 * q2 + 1][q][1] : t[2 * q2][q][1];
              // rows 0..3 of the eight come from the lh = 0 lane, rows 4..7 from the lh = 1 lane
              const uint4 val = lh ? make_uint4(r0, r1_, k0, k1) : make_uint4(k0, k1, r0, r1_);
              if (ml < g.M) *reinterpret_cast<uint4*>(base + (size_t)q * 128 * 64) = val;
            }
          }
        }
    }
  } else if constexpr (EPI == EPI_GATE_BWD) {
    // dh = v * s, dg = v * (h s) * (1 - s) with h s and s of the layer below read from dense [M x N] arrays.  All the
    // reads of a fragment are issued before the first store: the stores may alias them as far as the compiler knows,
    // and a read -> store -> read chain would pay one memory latency per element (that is the whole run time of a
    // launch with few blocks, and of a block's tail in any launch).
    // tsink: (dh, dg) also (or only: out0 == NULL) as the pre-split image of [dh | dg]^T -- column n -> image rows n and N + n
    const bool timg = g.tsink.img != nullptr;                             // (wave-uniform)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wc * 32 * NT + nt * 32 + l31;
      const bool nok = n < g.N;
      if (!nok && !timg) continue;
      float go[MT][16], sv[MT][16];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wr * 32 * MT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const size_t oe = (size_t)(m < g.M ? m : 0) * g.N + (nok ? n : 0);
          go[mt][r] = g.e0[oe];
          sv[mt][r] = g.e1[oe];
        }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float dhv[16], dgv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wr * 32 * MT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const float v = acc[mt][nt][r], s_ = sv[mt][r];
          dhv[r] = v * s_;                              // dh
          dgv[r] = v * go[mt][r] * (1.0f - s_);         // dg = v * h * s * (1 - s)
          if (m < g.M && nok && g.out0) {
            const size_t o = orow(m) * g.ldo + n;
            g.out0[o] = dhv[r];
            g.out1[o] = dgv[r];
          }
        }
        if (timg) {
          const int k0 = g.tsink.kbase + m0 + wr * 32 * MT + mt * 32;
          p6_emit_tile(g.tsink, g.tsink.row0 + n, nok, k0, dhv, lh);
          p6_emit_tile(g.tsink, g.tsink.row0 + g.N + n, nok, k0, dgv, lh);
        }
      }
    }
  } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wc * 32 * NT + nt * 32 + l31;
      if (n >= g.N) continue;
      float bias = (EPI == EPI_LINEAR && g.bias0) ? g.bias0[n] : 0.f;
      EVAE_PIN(bias);
      const bool extras = EPI == EPI_LINEAR && (g.e0 || g.e1);
      auto put = [&](int mt, int r, float a0, float a1) {      // one element: row guard around the stores only
        const int m = m0 + wr * 32 * MT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.M) return;
        const float v = acc[mt][nt][r];
        if (EPI == EPI_LINEAR) {
          const size_t o = orow(m) * g.ldo + n;
          const float pre = v + bias;
          if (g.out1) g.out1[o] = pre;
          float res = apply_act(pre, g.act, g.lo, g.hi);
          if (extras) res = res * (a1 > 0.f ? 1.0f : a1 + 1.0f) + a0;
          g.out0[o] = res;
        } else if (g.direct) {                  // EPI_RAW, one slice: the result itself (bias gradient = the ones column)
          // direct == 2: added to what is there (the second application of a shared layer accumulates into the first one's
          // gradient: old + v, the very bits of the sum autograd would form)
          if (n == g.ones_col) { if (g.out1) g.out1[m] = (g.direct == 2) ? g.out1[m] + v : v; }
          else { float* o = g.out0 + (size_t)m * g.ldo + n; *o = (g.direct == 2) ? *o + v : v; }
        } else {                                // EPI_RAW: partial plane [z][M][N]
          g.out0[(size_t)zslice * g.M * g.N + (size_t)m * g.N + n] = v;
        }
      };
      // residual blocks of fully_conv (models/fully_conv.py:13-23: x + conv(ELU(x))): e1 = ELU(x) of the block whose data
      // gradient this is -> times ELU'(x) = (a > 0 ? 1 : a + 1); e0 = the tensor added to the result (x forward, dy
      // backward).  Eight rows at a time: their sixteen values are fetched and PINNED before the first of their stores
      // (stores to out0 may alias them as far as the compiler knows, and a wait placed inside a row guard is a full
      // vmcnt(0), stores included).  One code path: without extras the pins have nothing to wait for but the stores of the
      // eight rows before -- two waits per sixteen rows instead of sixteen.
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          float ev0[8], ev1[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) { ev0[i] = 0.f; ev1[i] = 1.f; }
          if (extras) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int r = 8 * half + i;
              const int m = m0 + wr * 32 * MT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
              const bool ok = m < g.M;
              const size_t o = orow(ok ? m : m0) * g.ldo + n;
              if (ok && g.e0) ev0[i] = g.e0[o];
              if (ok && g.e1) ev1[i] = g.e1[o];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { EVAE_PIN(ev0[i]); EVAE_PIN(ev1[i]); }
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) put(mt, 8 * half + i, ev0[i], ev1[i]);
        }
    }
  }
}

// GATHER_A: the rows of a KC A operand are gathered through g.a_rows (the exemplar gather of the first encoder layer).  A
// template parameter rather than a run-time test so that the gathered launch -- the dominant one of a training step -- is
// a kernel symbol of its own in per-kernel profiles.
// The kernel's body is a device function of (arguments, block id, slice id): gemm_kernel runs it on blockIdx, the grouped
// launch of several thin weight gradients (gemm_group_wgrad_kernel below) on a per-problem block id.
template <bool A_KC, bool B_KC, int EPI, bool VEC, int BN_, int NW, int CV = 0, bool GATHER_A = false>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int blk_x, const int blk_z, float* const smem) {
  constexpr bool GATED = (EPI == EPI_GATED || EPI == EPI_RAW_GATED);
  constexpr int GNT = 64 * NW;
  constexpr int MT = 8 / NW;        // NW/2 wave rows x 2 wave columns; wave tile (32 MT) x (32 NT)
  constexpr int NT = BN_ / 64;
  static_assert(!GATED || BN_ == 128, "gated epilogue needs the h and g column tiles in one wave");
  constexpr int STAGE = A_TILE_FLOATS + b_tile_floats(BN_);
  if constexpr (EPI == EPI_PRIOR_LSE || EPI == EPI_PRIOR_P) {
    if (g.skip_flag != nullptr && *g.skip_flag != 0u) return;     // block-uniform: the caller's guard chose the other path
  }
  auto As = [&](int b) -> float* { return smem + b * STAGE; };
  auto Bs = [&](int b) -> float* { return smem + b * STAGE + A_TILE_FLOATS; };

  int tm, tn, zslice = blk_z;
  if (g.sk_local) {
    // XCD-local split-K (1-D grid): a UNIT = the tiles_n column tiles of one (contraction slice, row tile) -- they read the
    // same A tile -- sits on one XCD, dispatched back to back (ids 8 j + x -> XCD x), so its blocks walk the slice in step
    // and share every A slab through that XCD's L2; units are ordered slice-major and dealt to the XCDs in contiguous runs,
    // so the row tiles of a slice (sharing B) mostly meet on one XCD as well.  (r02: a split-K launch's grid (tiles, 1, nz)
    // scattered the blocks of a slice over all eight L2s: 346 MB read against 91 MB of operands for the layer-2 weight
    // gradient.)
    const int nunits = g.sk_local * g.tiles_m;                // sk_local = number of slices
    const int id = blk_x, xcd = id & 7, slot = id >> 3;
    const int qq = nunits >> 3, rr = nunits & 7;
    const int ul = slot / g.tiles_n;
    if (ul >= qq + (xcd < rr ? 1 : 0)) return;
    const int u = xcd * qq + (xcd < rr ? xcd : rr) + ul;
    tn = slot - ul * g.tiles_n;
    zslice = u / g.tiles_m; tm = u - zslice * g.tiles_m;
  } else {
    // XCD-aware bijective remap: XCD x (= id % 8) works on a contiguous run of tiles
    const int ntiles = g.tiles_m * g.tiles_n;
    const int id = blk_x, xcd = id & 7, slot = id >> 3;
    const int qq = ntiles >> 3, rr = ntiles & 7;
    const int tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + slot;
    tm = tile / g.tiles_n; tn = tile - tm * g.tiles_n;
  }
  const int m0 = tm * BM;
  // gated: a block covers 64 gated output columns; B tile rows = [wc][h|g][32]
  const int n0 = GATED ? tn * 64 : tn * BN_;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  const long long dbg_c0 = __builtin_readcyclecounter(), dbg_w0 = wall_clock64();
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // flattened list of K-slabs over the (A,B) pairs
  int nslab[2];
  nslab[0] = (g.Kc[0] + BK - 1) / BK;
  nslab[1] = g.npairs > 1 ? (g.Kc[1] + BK - 1) / BK : 0;
  int s_begin = 0, s_end = nslab[0] + nslab[1];
  if (g.ksplit > 0) {
    s_begin = zslice * g.ksplit;
    int e = s_begin + g.ksplit;
    if (e < s_end) s_end = e;
  }

  if constexpr (VEC) {
    // ---- fast path: every extent a multiple of 4, non-gathered operands below 2 GiB.  The slab loop
    // holds almost no VALU work (measured: each VALU instruction costs the matrix pipe its 4 issue
    // cycles): non-gathered tiles come through buffer loads -- per-thread byte offset fixed at kernel
    // start, the slab offset in an SGPR, rows/columns outside the matrix and the K tail parked on an
    // out-of-range offset that the hardware returns as zero -- so nothing is masked afterwards.
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NVA = BM * BK / 4 / GNT, NVB = BN_ * BK / 4 / GNT;
    constexpr int RQA = BM / 4, RQB = BN_ / 4, RSA = BM + 4, RSB = BN_ + 4;
    constexpr bool PAIRS = A_KC && !B_KC;   // only the data gradient chains two (A,B) pairs
    constexpr bool gatherA = A_KC && GATHER_A;
    const bool gatherB = !B_KC && g.b_krows != nullptr;
    const bool has_ones = !B_KC && g.ones_col >= n0 && g.ones_col < n0 + BN_;
    const rsrc_t rA0 = make_rsrc(g.A[0], 0x7FFFFFFFu);
    const rsrc_t rA1 = make_rsrc(PAIRS && g.npairs > 1 ? g.A[1] : g.A[0], 0x7FFFFFFFu);
    const rsrc_t rB0 = make_rsrc(g.B[0], 0x7FFFFFFFu);
    const rsrc_t rB1 = make_rsrc(GATED ? g.Bg : (PAIRS && g.npairs > 1 ? g.B[1] : g.B[0]), 0x7FFFFFFFu);
    const rsrc_t rIdx = make_rsrc(gatherB ? (const void*)g.b_krows : (const void*)g.B[0],
                                  gatherB ? (unsigned)g.Kc[0] * 8u : 0u);
    const bool wave_upper = (__builtin_amdgcn_readfirstlane(threadIdx.x) & 256) != 0;

    unsigned voA[PAIRS ? 2 : 1][NVA], voB[PAIRS ? 2 : 1][NVB];
    unsigned long long tapmask[NVA];       // CV = 1: bit t = tap t of this chunk's row lies inside the source image
    int colch[NVB], coltap[NVB];           // CV = 2: channel offset / tap of this chunk's four im2col columns
    const float* gpA[NVA];
    const float* gpB[NVB];
    unsigned kidx[NVB];
    bool onesB[NVB];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int f = threadIdx.x + GNT * i;
      gpA[i] = g.A[0];
      tapmask[i] = 0ull;
      if (A_KC && CV == 1) {
        // row = pixel (n, ry, rx): anchor offset in the channels-last source + which taps fall inside the image
        const int r = m0 + (f >> 3);
        const bool ok = r < g.M;
        const unsigned rr = ok ? (unsigned)r : 0u;
        const unsigned n = fdiv(rr, g.cv.div_rhw), rem = rr - n * (unsigned)(g.cv.RH * g.cv.RW);
        const unsigned ry = fdiv(rem, g.cv.div_rw), rx = rem - ry * (unsigned)g.cv.RW;
        const int ay = (int)ry * g.cv.rs + g.cv.roy, ax = (int)rx * g.cv.rsx + g.cv.rox;
        const long long off = (((long long)n * g.cv.IH + ay) * g.cv.IW + ax) * g.cv.ps * 4 + 16 * (f & 7) + (long long)g.cv.bias;
#pragma unroll
        for (int p = 0; p < (PAIRS ? 2 : 1); ++p) voA[p][i] = ok ? (unsigned)off : OOB;
        if (ok)
          for (int t = 0; t < g.cv.ntaps; ++t) {
            const int y = ay + g.cv.tdy[t], x = ax + g.cv.tdx[t];
            if ((unsigned)y < (unsigned)g.cv.IH && (unsigned)x < (unsigned)g.cv.IW) tapmask[i] |= 1ull << t;
          }
      } else if (A_KC) {
        const int r = m0 + (f >> 3);
        const bool ok = r < g.M;
        if (gatherA) {
          gpA[i] = g.A[0] + (size_t)g.a_rows[ok ? r : m0] * g.lda[0] + 4 * (f & 7);
        }
#pragma unroll
        for (int p = 0; p < (PAIRS ? 2 : 1); ++p)
          voA[p][i] = ok ? (unsigned)(r * g.lda[p] + 4 * (f & 7)) * 4u : OOB;
      } else {
        const int c = m0 + 4 * (f % RQA);
        voA[0][i] = (c + 4 <= g.M) ? (unsigned)((f / RQA) * g.lda[0] + c) * 4u : OOB;
      }
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int f = threadIdx.x + GNT * i;
      gpB[i] = g.B[0];
      onesB[i] = false;
      kidx[i] = 0;
      if (GATED) {
        const int r = f >> 3;
        const int n = n0 + (r >> 6) * 32 + (r & 31);
        voB[0][i] = (n < g.N) ? (unsigned)(n * g.ldb[0] + 4 * (f & 7)) * 4u : OOB;
      } else if (B_KC) {
        const int n = n0 + (f >> 3);
        voB[0][i] = (n < g.N) ? (unsigned)(n * g.ldb[0] + 4 * (f & 7)) * 4u : OOB;
      } else {
        const int c = n0 + 4 * (f % RQB);
        const int nlim = g.ones_col >= 0 ? g.ones_col : g.N;
        const bool ok = c + 4 <= nlim;
        onesB[i] = (c == g.ones_col);
        coltap[i] = -1; colch[i] = 0;
        if (CV == 2 && ok) { coltap[i] = c / g.cv.Cg; colch[i] = c - coltap[i] * g.cv.Cg; }
        if (gatherB) gpB[i] = g.B[0] + (ok ? c : 0);
#pragma unroll
        for (int p = 0; p < (PAIRS ? 2 : 1); ++p)
          voB[p][i] = ok ? (unsigned)((f / RQB) * g.ldb[p] + c) * 4u : OOB;
      }
    }
    if (gatherB) {
#pragma unroll
      for (int i = 0; i < NVB; ++i)
        kidx[i] = buf_ld1(rIdx, (unsigned)(s_begin * BK + (threadIdx.x + GNT * i) / RQB) * 8u, 0u);
    }

    // The slab loop must not branch around memory instructions: the compiler's s_waitcnt bookkeeping
    // merges pessimistically at every join, and a "vmcnt(0)" in front of the B loads then waits for the
    // A loads issued a few cycles earlier -- a full memory latency per slab with the matrix pipe idle
    // (measured: 15 % of the kernel).  So gather / plain operands are compile-time variants of the
    // loop (GA, GB), the two (A,B) pairs of the data gradient are selected with scalar selects, and
    // the last two slabs (nothing left to store / load) are peeled off as compile-time variants too.
    // kv = valid contraction rows of a slab (>= BK: full slab).
    auto run = [&](auto GA_, auto GB_) {
      constexpr bool GA = decltype(GA_)::value, GB = decltype(GB_)::value;
      auto load_a = [&](int s, float4 (&ra)[NVA], int& kv) {
        const bool p1 = PAIRS && s >= nslab[0];
        const int k0 = (p1 ? s - nslab[0] : s) * BK;
        kv = (p1 ? g.Kc[1] : g.Kc[0]) - k0;
        const bool tail = kv < BK;
        if constexpr (GA) {
#pragma unroll
          for (int i = 0; i < NVA; ++i) {
            const int kc = 4 * ((threadIdx.x + GNT * i) & 7);
            ra[i] = ld4v(gpA[i] + ((tail && kc + 4 > kv) ? -kc : k0));
          }
        } else if constexpr (CV == 1 && A_KC) {
          // the slab is 32 channels of one tap: tap offset in an SGPR, validity = one bit per row
          const rsrc_t rA = p1 ? rA1 : rA0;
          const int tap = k0 / g.cv.Cg;                    // uniform; Cg is a multiple of 32
          const int kin = k0 - tap * g.cv.Cg;              // first channel of this slab within the tap
          const unsigned so = (unsigned)(g.cv.tsoff[tap] + kin * 4);
          const int cleft = g.cv.creal - kin;              // real channels from here on (>= 32: the whole slab exists)
#pragma unroll
          for (int i = 0; i < NVA; ++i) {
            const bool live = ((tapmask[i] >> tap) & 1ull) && (4 * ((threadIdx.x + GNT * i) & 7) + 4 <= cleft);
            ra[i] = buf_ld4(rA, live ? voA[0][i] : OOB, so);
          }
        } else {
          const rsrc_t rA = p1 ? rA1 : rA0;
          const unsigned so = A_KC ? (unsigned)k0 * 4u : (unsigned)(k0 * (p1 ? g.lda[1] : g.lda[0])) * 4u;
#pragma unroll
          for (int i = 0; i < NVA; ++i) {
            const int f = threadIdx.x + GNT * i;
            const bool dead = tail && (A_KC ? 4 * (f & 7) + 4 > kv : f / RQA >= kv);
            const unsigned vo = (PAIRS && p1) ? voA[PAIRS ? 1 : 0][i] : voA[0][i];
            ra[i] = buf_ld4(rA, dead ? OOB : vo, so);
          }
        }
      };
      auto load_b = [&](int s, float4 (&rb)[NVB], int& kv) {
        const bool p1 = PAIRS && s >= nslab[0];
        const int k0 = (p1 ? s - nslab[0] : s) * BK;
        kv = (p1 ? g.Kc[1] : g.Kc[0]) - k0;
        const bool tail = kv < BK;
        if constexpr (GB) {
#pragma unroll
          for (int i = 0; i < NVB; ++i) rb[i] = ld4v(gpB[i] + (size_t)kidx[i] * g.ldb[0]);
#pragma unroll
          for (int i = 0; i < NVB; ++i)   // indices of the slab after this one (past the end -> 0)
            kidx[i] = buf_ld1(rIdx, (unsigned)(k0 + BK + (threadIdx.x + GNT * i) / RQB) * 8u, 0u);
        } else if constexpr (CV == 2 && !B_KC) {
          // im2col(x)[pixel m = k0 + kk][4 channels of one tap]: pixel decomposed per slab, tap fixed per thread
#pragma unroll
          for (int i = 0; i < NVB; ++i) {
            const int kk = (threadIdx.x + GNT * i) / RQB;
            const unsigned m = (unsigned)(k0 + kk);
            const unsigned n = fdiv(m, g.cv.div_rhw), rem = m - n * (unsigned)(g.cv.RH * g.cv.RW);
            const unsigned ry = fdiv(rem, g.cv.div_rw), rx = rem - ry * (unsigned)g.cv.RW;
            const int t = coltap[i] < 0 ? 0 : coltap[i];
            const int y = (int)ry * g.cv.rs + g.cv.roy + g.cv.tdy[t], x = (int)rx * g.cv.rsx + g.cv.rox + g.cv.tdx[t];
            const bool live = coltap[i] >= 0 && kk < kv && (unsigned)y < (unsigned)g.cv.IH && (unsigned)x < (unsigned)g.cv.IW;
            const unsigned off = (unsigned)((((int)n * g.cv.IH + y) * g.cv.IW + x) * g.cv.ps + colch[i]) * 4u;
            rb[i] = buf_ld4(rB0, live ? off : OOB, 0u);
          }
        } else {
          const unsigned so = B_KC ? (unsigned)k0 * 4u : (unsigned)(k0 * (p1 ? g.ldb[1] : g.ldb[0])) * 4u;
#pragma unroll
          for (int i = 0; i < NVB; ++i) {
            const int f = threadIdx.x + GNT * i;
            const bool dead = tail && (B_KC ? 4 * (f & 7) + 4 > kv : f / RQB >= kv);
            const bool up = GATED ? ((GNT == 512) ? wave_upper : ((i & 1) != 0)) : p1;
            const unsigned vo = (PAIRS && p1) ? voB[PAIRS ? 1 : 0][i] : voB[0][i];
            rb[i] = buf_ld4(up ? rB1 : rB0, dead ? OOB : vo, so);
          }
        }
      };
      auto store_a = [&](int buf, float4 (&ra)[NVA], int kv) {
        float* at = As(buf);
        if (GA && kv < BK) {
#pragma unroll
          for (int i = 0; i < NVA; ++i)
            if (4 * ((threadIdx.x + GNT * i) & 7) + 4 > kv) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
          const int f = threadIdx.x + GNT * i;
          if (A_KC) *reinterpret_cast<float4*>(at + (f >> 3) * KS + 4 * (f & 7)) = ra[i];
          else      *reinterpret_cast<float4*>(at + (f / RQA) * RSA + 4 * (f % RQA)) = ra[i];
        }
      };
      auto store_b = [&](int buf, float4 (&rb)[NVB], int kv) {
        float* bt = Bs(buf);
        if (GB && kv < BK) {
#pragma unroll
          for (int i = 0; i < NVB; ++i)
            if ((threadIdx.x + GNT * i) / RQB >= kv) rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (has_ones) {
#pragma unroll
          for (int i = 0; i < NVB; ++i)
            if (onesB[i]) rb[i] = make_float4((threadIdx.x + GNT * i) / RQB < kv ? 1.f : 0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
          const int f = threadIdx.x + GNT * i;
          if (B_KC) *reinterpret_cast<float4*>(bt + (f >> 3) * KS + 4 * (f & 7)) = rb[i];
          else      *reinterpret_cast<float4*>(bt + (f / RQB) * RSB + 4 * (f % RQB)) = rb[i];
        }
      };

      float4 ra[NVA], rb[NVB];
      int kva = 0, kvb = 0;
      load_a(s_begin, ra, kva);
      load_b(s_begin, rb, kvb);
      store_a(0, ra, kva);
      store_b(0, rb, kvb);
      if (s_begin + 1 < s_end) { load_a(s_begin + 1, ra, kva); load_b(s_begin + 1, rb, kvb); }
      __syncthreads();
      // Schedule of one slab (per wave).  Memory instructions are slotted one by one between the MFMAs
      // of the same wave: a wave has only MT*NT independent accumulator chains, so after MT*NT MFMAs
      // it stalls on the dependency anyway and whatever issues in that shadow is free.  The barrier sits
      // before the last k-group, whose fragments are already in registers, and the first fragments of
      // the next slab are requested right behind it.
      Frag<MT, NT> f0, f1;
      load_frag<A_KC, B_KC, MT, NT, BN_>(f0, As(0), Bs(0), wr, wc, lane, 0);
#define EVAE_SB __builtin_amdgcn_sched_barrier(0)
      // ST: the registers hold slab s+1 -> write it to the idle LDS buffer; LD: fetch slab s+2; NX: slab s+1 exists
      auto slab = [&](int s, auto ST_, auto LD_, auto NX_) {
        constexpr bool ST = decltype(ST_)::value, LD = decltype(LD_)::value, NX = decltype(NX_)::value;
        const int cur = (s - s_begin) & 1;
        EVAE_SB; mma_step<MT, NT>(acc, f0, 0); EVAE_SB;
        load_frag<A_KC, B_KC, MT, NT, BN_>(f1, As(cur), Bs(cur), wr, wc, lane, 1);
        if constexpr (ST) store_a(cur ^ 1, ra, kva);
        EVAE_SB; mma_step<MT, NT>(acc, f0, 1); EVAE_SB;
        if constexpr (ST) store_b(cur ^ 1, rb, kvb);
        EVAE_SB; mma_step<MT, NT>(acc, f0, 2); EVAE_SB;
        if constexpr (LD) load_a(s + 2, ra, kva);
        EVAE_SB; mma_step<MT, NT>(acc, f0, 3); EVAE_SB;
        if constexpr (LD) load_b(s + 2, rb, kvb);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          EVAE_SB; mma_step<MT, NT>(acc, f1, q); EVAE_SB;
          if (q < MT + NT) load_frag_part<A_KC, B_KC, MT, NT, BN_>(f0, As(cur), Bs(cur), wr, wc, lane, 2, q);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          EVAE_SB; mma_step<MT, NT>(acc, f0, q); EVAE_SB;
          if (q < MT + NT) load_frag_part<A_KC, B_KC, MT, NT, BN_>(f1, As(cur), Bs(cur), wr, wc, lane, 3, q);
        }
        EVAE_SB;
        __syncthreads();
        EVAE_SB;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          mma_step<MT, NT>(acc, f1, q); EVAE_SB;
          if constexpr (NX) {
            if (q < MT + NT) load_frag_part<A_KC, B_KC, MT, NT, BN_>(f0, As(cur ^ 1), Bs(cur ^ 1), wr, wc, lane, 0, q);
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
    };
    if (s_begin < s_end) {
      if constexpr (gatherA) run(std::true_type{}, std::false_type{});
      else {
        if (gatherB) run(std::false_type{}, std::true_type{});
        else run(std::false_type{}, std::false_type{});
      }
    }
  } else {
    typedef TileLoader<BM, A_KC, GNT> LA;
    typedef TileLoader<BN_, B_KC, GNT> LB;
    LA la[2];
    LB lb[2];
  #pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (p < g.npairs) {
        la[p].init(g.A[p], g.lda[p], m0, g.M, GATHER_A ? g.a_rows : nullptr);
        if (!GATED) lb[p].init(g.B[p], g.ldb[p], n0, g.N, nullptr);
      }
    }
    // gated B tile: LDS row r -> weight row n0 + (r>>6)*32 + (r&31) of (r&32 ? Bg : B[0])
    const float* gb_base[LB::NV];
    bool gb_ok[LB::NV];
    if (GATED) {
  #pragma unroll
      for (int i = 0; i < LB::NV; ++i) {
        int f = threadIdx.x + GNT * i;
        int r = f >> 3;
        int n = n0 + (r >> 6) * 32 + (r & 31);
        gb_ok[i] = n < g.N;
        const float* w = (r & 32) ? g.Bg : g.B[0];
        gb_base[i] = w + (size_t)(gb_ok[i] ? n : 0) * g.ldb[0] + 4 * (f & 7);
      }
    }

    auto load_slab = [&](int s, float4 (&ra)[LA::NV], float4 (&rb)[LB::NV], unsigned& ma, unsigned& mb) {
      const int p = (s < nslab[0]) ? 0 : 1;
      const int k0 = (p == 0 ? s : s - nslab[0]) * BK;
      const int kend = g.Kc[p];
      if (A_KC) ma = la[p].template load_kc<VEC>(ra, k0, kend);
      else ma = la[p].template load_rc<VEC>(ra, g.A[p], g.lda[p], m0, g.M, k0, kend, nullptr, -1);
      if (GATED) {
        mb = 0;
  #pragma unroll
        for (int i = 0; i < LB::NV; ++i) {
          int f = threadIdx.x + GNT * i;
          int k = k0 + 4 * (f & 7);
          if (VEC) {
            const bool ok = gb_ok[i] && (k + 4 <= kend);
            rb[i] = ld4v(gb_base[i] + (ok ? k0 : -4 * (f & 7)));
            mb |= (ok ? 1u : 0u) << (2 * i);
          } else {
            rb[i] = ld4s(gb_base[i] + k0, gb_ok[i] ? (kend - k) : 0);
            mb |= 1u << (2 * i);
          }
        }
      } else if (B_KC) {
        mb = lb[p].template load_kc<VEC>(rb, k0, kend);
      } else {
        mb = lb[p].template load_rc<VEC>(rb, g.B[p], g.ldb[p], n0, g.ones_col >= 0 ? g.ones_col : g.N, k0, kend, g.b_krows, g.ones_col);
      }
    };

    // Pipeline: registers always hold the slab AFTER the one being multiplied.  Its global loads were
    // issued a whole slab earlier; they are written to the idle LDS buffer after the first k-group of
    // MFMAs and the loads of the slab after that are issued right behind, so VMEM latency, the LDS
    // stores and their address arithmetic all sit in the shadow of the 64-cycle MFMAs.
    if (s_begin < s_end) {
      float4 ra[LA::NV], rb[LB::NV];
      unsigned ma, mb;
      load_slab(s_begin, ra, rb, ma, mb);
      la[0].store(As(0), ra, ma);
      lb[0].store(Bs(0), rb, mb);
      if (s_begin + 1 < s_end) load_slab(s_begin + 1, ra, rb, ma, mb);
      __syncthreads();
      for (int s = s_begin; s < s_end; ++s) {
        const int cur = (s - s_begin) & 1;
        Frag<MT, NT> f0, f1;
        load_frag<A_KC, B_KC, MT, NT, BN_>(f0, As(cur), Bs(cur), wr, wc, lane, 0);
        load_frag<A_KC, B_KC, MT, NT, BN_>(f1, As(cur), Bs(cur), wr, wc, lane, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_frag<MT, NT>(acc, f0);
        __builtin_amdgcn_sched_barrier(0);
        load_frag<A_KC, B_KC, MT, NT, BN_>(f0, As(cur), Bs(cur), wr, wc, lane, 2);
        if (s + 1 < s_end) {
          la[0].store(As(cur ^ 1), ra, ma);
          lb[0].store(Bs(cur ^ 1), rb, mb);
        }
        if (s + 2 < s_end) load_slab(s + 2, ra, rb, ma, mb);
        __builtin_amdgcn_sched_barrier(0);
        mma_frag<MT, NT>(acc, f1);
        __builtin_amdgcn_sched_barrier(0);
        load_frag<A_KC, B_KC, MT, NT, BN_>(f1, As(cur), Bs(cur), wr, wc, lane, 3);
        __builtin_amdgcn_sched_barrier(0);
        mma_frag<MT, NT>(acc, f0);
        mma_frag<MT, NT>(acc, f1);
        __syncthreads();
      }
    }

  }

  if ((g.dbg & 512) && g.out2) {   // clock probe: shader-clock ticks vs the constant 100 MHz counter over this block's main loop
    const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0 && (blk_x & 63) == 5) {
      float* d = g.out2 + (size_t)(g.M - 1) * g.ldo;   // last row of the save_s output (overwritten, debug only)
      atomicAdd(d + 0, (float)(c1 - dbg_c0));
      atomicAdd(d + 1, (float)(w1 - dbg_w0));
      atomicAdd(d + 2, 1.0f);
    }
    return;
  }
  if (g.dbg & 4) {           // ablation: no epilogue (the accumulators stay live through an impossible store)
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 1.2345e-30f) g.out0[0] = t;
    return;
  }
  gemm_epilogue<EPI, BN_, NW, CV>(g, acc, m0, n0, tm, wr, wc, lane, smem, zslice);
}

template <bool A_KC, bool B_KC, int EPI, bool VEC, int BN_, int NW, int CV = 0, bool GATHER_A = false>
__global__ __launch_bounds__(64 * NW, NW / 2) void gemm_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  gemm_body<A_KC, B_KC, EPI, VEC, BN_, NW, CV, GATHER_A>(g, (int)blockIdx.x, (int)blockIdx.z, smem);
}

// ---- several thin weight gradients in ONE launch (the batch rows' leaf layers of a training step: decoder layers, the
// log-variance head).  Every job is dw [N x K] = dy^T x over at most four K-slabs of contraction rows, written by the GEMM
// itself (GemmArgs::direct: no partial plane, no finish) with db in the ones column; a block finds its job by the running
// tile counts and runs gemm_body on its own copy of the arguments.  The jobs travel by value in the kernel arguments, so a
// captured step needs no table upload.
struct WgradJob { const float* dy; const float* x; float* dw; float* db; int M, N, K, ldy, ldx, accumulate; };
constexpr int kWgradGroupMax = 6;
struct WgradGroup { WgradJob job[kWgradGroupMax]; int start[kWgradGroupMax + 1]; int n; };

__device__ __forceinline__ const float* uni_ptr(const float* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return (const float*)(((unsigned long long)hi << 32) | lo);
}

template <int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void gemm_group_wgrad_kernel(const WgradGroup grp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  int j = 0;
#pragma unroll
  for (int i = 1; i < kWgradGroupMax; ++i)
    if (i < grp.n && bid >= grp.start[i]) j = i;
  j = __builtin_amdgcn_readfirstlane(j);
  const WgradJob& w = grp.job[j];
  GemmArgs g = {};
  g.A[0] = uni_ptr(w.dy); g.B[0] = uni_ptr(w.x);
  g.lda[0] = __builtin_amdgcn_readfirstlane(w.ldy); g.ldb[0] = __builtin_amdgcn_readfirstlane(w.ldx);
  g.Kc[0] = __builtin_amdgcn_readfirstlane(w.M); g.npairs = 1;
  g.M = __builtin_amdgcn_readfirstlane(w.N); g.N = __builtin_amdgcn_readfirstlane(w.K) + 1;
  g.out0 = (float*)uni_ptr(w.dw); g.out1 = (float*)uni_ptr(w.db); g.ldo = g.N - 1; g.ones_col = g.N - 1;
  g.direct = __builtin_amdgcn_readfirstlane(w.accumulate) ? 2 : 1;
  g.tiles_m = (g.M + BM - 1) / BM; g.tiles_n = (g.N + 63) / 64;
  gemm_body<false, false, EPI_RAW, true, 64, NW>(g, bid - __builtin_amdgcn_readfirstlane(grp.start[j]), 0, smem);
}

// ---- host side: plan, launch --------------------------------------------------------------------------
static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <bool A_KC, bool B_KC>
static bool gemm_vec_ok(const GemmArgs& g) {
  bool ok = true;
  for (int p = 0; p < g.npairs; ++p) {
    ok = ok && al16(g.A[p]) && al16(g.B[p]) && (g.lda[p] % 4 == 0) && (g.ldb[p] % 4 == 0);
    if (A_KC || B_KC) ok = ok && (g.Kc[p] % 4 == 0);
  }
  if (!A_KC) ok = ok && (g.M % 4 == 0);
  if (!B_KC) ok = ok && ((g.ones_col >= 0 ? g.ones_col : g.N) % 4 == 0);
  if (g.Bg) ok = ok && al16(g.Bg);
  // buffer-load offsets of the fast path are 31-bit byte offsets (gathered operands use 64-bit pointers)
  const int64_t lim = (int64_t)1 << 29;   // floats
  const int Bn = (!B_KC && g.ones_col >= 0) ? g.ones_col : g.N;
  for (int p = 0; p < g.npairs; ++p) {
    const int64_t ea = A_KC ? (int64_t)g.M * g.lda[p] + g.Kc[p] : (int64_t)g.Kc[p] * g.lda[p] + g.M;
    const int64_t eb = B_KC ? (int64_t)g.N * g.ldb[p] + g.Kc[p] : (int64_t)g.Kc[p] * g.ldb[p] + Bn;
    if (!(A_KC && g.a_rows)) ok = ok && ea + BK * (int64_t)g.lda[p] < lim;
    if (!(!B_KC && g.b_krows)) ok = ok && eb + BK * (int64_t)g.ldb[p] < lim;
  }
  return ok;
}

template <bool A_KC, bool B_KC, int EPI, bool VEC, int BN_, int NW, int CV = 0, bool GA = false>
static int launch_gemm_w(GemmArgs& g, int nz, hipStream_t stream, const char* what) {
  if ((g.a_rows != nullptr) != GA) { set_error("%s: row gather does not match the kernel variant", what); return EVAE_EINVAL; }
  static bool attr = false;
  constexpr size_t lds = gemm_lds_bytes(BN_);
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_kernel<A_KC, B_KC, EPI, VEC, BN_, NW, CV, GA>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  constexpr bool GATED = (EPI == EPI_GATED || EPI == EPI_RAW_GATED);
  g.tiles_m = cdiv(g.M, BM);
  g.tiles_n = cdiv(g.N, GATED ? 64 : BN_);
  dim3 grid(g.tiles_m * g.tiles_n, 1, nz);
  if (g.sk_local > 0) {
    if (g.sk_local != nz || g.ksplit <= 0) { set_error("%s: XCD-local split-K needs sk_local == nz and a split contraction", what); return EVAE_EINVAL; }
    grid = dim3(8 * g.tiles_n * cdiv(nz * g.tiles_m, 8), 1, 1);
  }
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("EVAE_GEMM_DBG"); dbg = e ? atoi(e) : 0; }
    g.dbg = dbg;
  }
  gemm_kernel<A_KC, B_KC, EPI, VEC, BN_, NW, CV, GA><<<grid, 64 * NW, lds, stream>>>(g);
  return check_launch(what);
}

// waves per block: 8 (4 waves/SIMD at 2 blocks/CU, the default) or 4; EVAE_GEMM_NW=4 selects the latter
static int gemm_nw() {
  static int nw = 0;
  if (nw == 0) {
    const char* e = getenv("EVAE_GEMM_NW");
    nw = (e && atoi(e) == 4) ? 4 : 8;
  }
  return nw;
}

template <bool A_KC, bool B_KC, int EPI, bool VEC, int BN_, bool GA = false>
static int launch_gemm_v(GemmArgs& g, int nz, hipStream_t stream, const char* what) {
  if (gemm_nw() == 4) return launch_gemm_w<A_KC, B_KC, EPI, VEC, BN_, 4, 0, GA>(g, nz, stream, what);
  return launch_gemm_w<A_KC, B_KC, EPI, VEC, BN_, 8, 0, GA>(g, nz, stream, what);
}

template <bool A_KC, bool B_KC, int EPI, bool GA>
static int launch_gemm_ga(GemmArgs& g, const Plan& pl, hipStream_t stream, const char* what) {
  constexpr bool GATED = (EPI == EPI_GATED || EPI == EPI_RAW_GATED);
  const bool vec = gemm_vec_ok<A_KC, B_KC>(g);
  g.ksplit = pl.nz > 1 ? pl.ksplit : 0;
  if (GATED || pl.bn == 128) {
    if (vec) return launch_gemm_v<A_KC, B_KC, EPI, true, 128, GA>(g, pl.nz, stream, what);
    return launch_gemm_v<A_KC, B_KC, EPI, false, 128, GA>(g, pl.nz, stream, what);
  }
  if constexpr (!GATED) {
    if (vec) return launch_gemm_v<A_KC, B_KC, EPI, true, 64, GA>(g, pl.nz, stream, what);
    return launch_gemm_v<A_KC, B_KC, EPI, false, 64, GA>(g, pl.nz, stream, what);
  }
  return EVAE_EINVAL;
}

template <bool A_KC, bool B_KC, int EPI>
static int launch_gemm(GemmArgs& g, const Plan& pl, hipStream_t stream, const char* what) {
  if constexpr (A_KC && B_KC) {       // only the forward launches take a row-gather list
    if (g.a_rows != nullptr) return launch_gemm_ga<A_KC, B_KC, EPI, true>(g, pl, stream, what);
  }
  return launch_gemm_ga<A_KC, B_KC, EPI, false>(g, pl, stream, what);
}

static int total_slabs(int k0, int k1) { return cdiv(k0, BK) + (k1 > 0 ? cdiv(k1, BK) : 0); }

}  // namespace evae
