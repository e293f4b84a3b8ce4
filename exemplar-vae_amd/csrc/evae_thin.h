// Thin dense layers: the B batch rows of a training step (reference utils/training.py:27-46 at batch_size 100; utils/nn.py:44-69
// GatedDense / NonLinear forward, their data gradients) through ONE launch each.
//
// The tiled GEMM kernels give a [100 x 300] layer one row tile and five column tiles: to fill the machine they split the
// contraction over blockIdx.z and a second launch sums the partial planes -- two graph nodes per layer, and a node of the
// replayed step costs >= 4.5 us whatever it computes (the batch-row chain was ~55 nodes = 0.25 ms).  Here a
// block owns a 16 x 16 output tile (x 2 banks for a gated layer), its four waves take interleaved quarters of the contraction
// straight from L2 into v_mfma_f32_16x16x4_f32 operands (no LDS staging: 100 rows of activations and one layer's weights are
// L2-resident), the four partial tiles meet in LDS in a fixed order, and the block applies the layer's epilogue itself.
// 133 blocks for a 300-wide gated layer, 343 for the 784-wide output layer: one node per layer, fp32 arithmetic, deterministic.
#pragma once
#include "evae_gemm_core.h"

namespace evae {

enum { THIN_GATED = 0, THIN_LINEAR = 1, THIN_GATE_BWD = 2 };     // (THIN_LINEAR with no bias / activation = the plain product)

struct ThinArgs {
  // forward (BWD = false): out [M x N] = A [M x K] W^T, W0 / W1 [N x K] (W1: the gate bank of THIN_GATED)
  // data gradient (BWD = true): out [M x K] = dy0 W0 (+ dy1 W1), dy [M x N] (row stride lda), W [N x K]: contraction over N
  const float* A;
  const float* A1;         // BWD: dy of the second bank (NULL: one bank)
  int lda;
  const float* W0;
  const float* W1;
  int M, N, K;
  const float* b0;
  const float* b1;
  float* out0;             // GATED: h * s; LINEAR: act(pre); GATE_BWD: dh (may be NULL when only the sinks are wanted)
  float* out1;             // GATED: h (or NULL); LINEAR: pre (or NULL); GATE_BWD: dg
  float* out2;             // GATED: s (or NULL)
  int ldo;
  const float* e0;         // GATE_BWD: gated output h s and gate s of the layer below, dense [M x K]
  const float* e1;
  int act;
  float lo, hi;
  P6Sink tsink;            // GATED / GATE_BWD: the result also into the pre-split image of its transpose (evae_p6_image.h)
  unsigned short* u8img;   // GATE_BWD: (dh, dg) also as the tile images of evae_dense_bwd_weight_u8 (EPI_GATE_BWD_IMG's layout)
  int u8_nslab, u8_mbase;
};

typedef float thin_f32x4 __attribute__((ext_vector_type(4)));

template <int EPI, bool BWD>
__global__ __launch_bounds__(256) void thin_layer_kernel(const ThinArgs t) {
  constexpr bool TWO = (EPI == THIN_GATED);            // two accumulators: the h and g banks of a gated forward
  __shared__ float part[4][TWO ? 2 : 1][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int nout = BWD ? t.K : t.N;                    // output columns
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  thin_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int mrow = (m0 + i < t.M) ? m0 + i : t.M - 1;                   // (rows beyond M: a valid row, discarded later)
  if constexpr (!BWD) {
    const int ncol = (n0 + i < t.N) ? n0 + i : t.N - 1;
    const float* pa = t.A + (size_t)mrow * t.lda + 4 * kq;
    const float* pw0 = t.W0 + (size_t)ncol * t.K + 4 * kq;
    const float* pw1 = TWO ? t.W1 + (size_t)ncol * t.K + 4 * kq : nullptr;
    const int nchunk = (t.K + 15) >> 4;
#pragma unroll 4
    for (int c = wave; c < nchunk; c += 4) {
      const int k0 = c * 16;
      const bool ok = k0 + 4 * kq + 4 <= t.K;            // (K % 4 == 0: a float4 is whole or absent)
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), w0 = a, w1 = a;
      if (ok) {
        a = *reinterpret_cast<const float4*>(pa + k0);
        w0 = *reinterpret_cast<const float4*>(pw0 + k0);
        if constexpr (TWO) w1 = *reinterpret_cast<const float4*>(pw1 + k0);
      }
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w0.x, acc0, 0, 0, 0);
      if constexpr (TWO) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w0.y, acc0, 0, 0, 0);
      if constexpr (TWO) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w0.z, acc0, 0, 0, 0);
      if constexpr (TWO) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w0.w, acc0, 0, 0, 0);
      if constexpr (TWO) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w1.w, acc1, 0, 0, 0);
    }
  } else {
    // contraction over the N rows of W, bank by bank; lane (output column i, k quarter kq) reads W[n][n0 + i] for its four n
    const int kcol = (n0 + i < t.K) ? n0 + i : t.K - 1;
    const int nchunk = (t.N + 15) >> 4, nbank = t.A1 ? 2 : 1;
    for (int bank = 0; bank < nbank; ++bank) {
      const float* pa = (bank ? t.A1 : t.A) + (size_t)mrow * t.lda + 4 * kq;
      const float* pw = (bank ? t.W1 : t.W0) + (size_t)(4 * kq) * t.K + kcol;
#pragma unroll 4
      for (int c = wave; c < nchunk; c += 4) {
        const int k0 = c * 16;
        const bool ok = k0 + 4 * kq + 4 <= t.N;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
          a = *reinterpret_cast<const float4*>(pa + k0);
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = pw[(size_t)(k0 + j) * t.K];
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[0], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[1], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[2], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[3], acc0, 0, 0, 0);
      }
    }
  }
  // C layout of the 16 x 16 tile: register r <-> row 4 (lane >> 4) + r, column lane & 15
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    part[wave][0][4 * kq + r][i] = acc0[r];
    if constexpr (TWO) part[wave][1][4 * kq + r][i] = acc1[r];
  }
  __syncthreads();
  const int row = tid >> 4, col = tid & 15;
  const int m = m0 + row, n = n0 + col;
  float v0 = ((part[0][0][row][col] + part[1][0][row][col]) + part[2][0][row][col]) + part[3][0][row][col];
  float v1 = 0.f;
  if constexpr (TWO) v1 = ((part[0][1][row][col] + part[1][1][row][col]) + part[2][1][row][col]) + part[3][1][row][col];
  if (n >= nout) return;
  if constexpr (EPI == THIN_GATED) {
    if (m >= t.M) return;
    const float h = v0 + (t.b0 ? t.b0[n] : 0.f);
    const float s = 1.0f / (1.0f + expf(-(v1 + (t.b1 ? t.b1[n] : 0.f))));
    const size_t o = (size_t)m * t.ldo + n;
    t.out0[o] = h * s;
    if (t.out1) t.out1[o] = h;
    if (t.out2) t.out2[o] = s;
    if (t.tsink.img) p6_sink_element(t.tsink, t.tsink.row0 + n, t.tsink.kbase + m, h * s);
  } else if constexpr (EPI == THIN_LINEAR) {
    if (m >= t.M) return;
    const float pre = v0 + (t.b0 ? t.b0[n] : 0.f);
    const size_t o = (size_t)m * t.ldo + n;
    if (t.out1) t.out1[o] = pre;
    t.out0[o] = apply_act(pre, t.act, t.lo, t.hi);
  } else {
    // dh = v s, dg = v (h s)(1 - s); column cc of the merged [dh | dg] buffer is n (dh) / K + n (dg)
    const bool live = m < t.M;
    if (!live && !(t.u8img && m < ((t.M + 7) & ~7))) return;       // (the byte-layer images want zeros up to the next multiple of 8)
    float dh = 0.f, dg = 0.f;
    if (live) {
      const size_t oe = (size_t)m * t.K + n;
      const float go = t.e0[oe], s = t.e1[oe];
      dh = v0 * s; dg = v0 * go * (1.0f - s);
      if (t.out0) { const size_t o = (size_t)m * t.ldo + n; t.out0[o] = dh; t.out1[o] = dg; }
      if (t.tsink.img) {
        p6_sink_element(t.tsink, t.tsink.row0 + n, t.tsink.kbase + m, dh);
        p6_sink_element(t.tsink, t.tsink.row0 + t.K + n, t.tsink.kbase + m, dg);
      }
    }
    if (t.u8img) {
      // img[((cc >> 7) * nslab + (row >> 5)) * 3 + p][cc & 127][slot (row & 31) >> 3, XOR-swizzled by (c >> 2) & 3][row & 7],
      // truncation split (csrc/evae_gemm_kernel.h, EPI_GATE_BWD_IMG)
      const int gm = m + t.u8_mbase;
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        const float w = which ? dg : dh;
        const int cc = which ? t.K + n : n, c = cc & 127;
        const unsigned u0 = __float_as_uint(w) & 0xFFFF0000u;
        const float r1 = w - __uint_as_float(u0);
        const unsigned u1 = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(u1);
        const unsigned u2 = __float_as_uint(r2);
        unsigned short* o = t.u8img + ((size_t)((cc >> 7) * t.u8_nslab + (gm >> 5)) * 3 * 128 + c) * 32 +
                            (((((gm & 31) >> 3) ^ ((c >> 2) & 3))) << 3) + (gm & 7);
        o[0] = (unsigned short)(u0 >> 16); o[128 * 32] = (unsigned short)(u1 >> 16); o[2 * 128 * 32] = (unsigned short)(u2 >> 16);
      }
    }
  }
}

// A gated layer's backward of the modular autograd path as ONE launch (r04): the gate derivative (dh, dg) = (dout s, dout (h s)(1 - s))
// of reference utils/nn.py:62-68 formed in the operand load of the data gradient dx = dh Wh + dg Wg, instead of an element-wise
// launch that writes [dh | dg] and a data-gradient launch that reads it back (two nodes of ~5 us on the batch rows' chain of the
// 2-level models, twelve times per step).  Same tiling, contraction split and summation order as thin_layer_kernel<THIN_LINEAR,
// true> over two banks -- dx is bit-identical to the two-launch form -- and the blocks of the first column tile also store
// [dh | dg] (row stride ldp) for the layer's weight gradient.
struct ThinGateBwdArgs {
  const float* dout; int ldd;      // [M x N] gradient of the layer's output
  const float* gout;               // [M x N] the layer's output h s (dense)
  const float* s;                  // [M x N] its gate
  const float* Wh; const float* Wg;   // [N x K]
  int M, N, K;
  float* dpre; int ldp;            // [M x 2N]: dh | dg
  float* dx; int ldo;              // [M x K]
};

__global__ __launch_bounds__(256) void thin_gated_bwd_kernel(const ThinGateBwdArgs t) {
  __shared__ float part[4][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  thin_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int mrow = (m0 + i < t.M) ? m0 + i : t.M - 1;
  const int kcol = (n0 + i < t.K) ? n0 + i : t.K - 1;
  const int nchunk = (t.N + 15) >> 4;
  const bool writer = blockIdx.x == 0 && m0 + i < t.M;
  const float* pd = t.dout + (size_t)mrow * t.ldd + 4 * kq;
  const float* pg = t.gout + (size_t)mrow * t.N + 4 * kq;
  const float* ps = t.s + (size_t)mrow * t.N + 4 * kq;
  float* pp = t.dpre + (size_t)mrow * t.ldp + 4 * kq;
  for (int bank = 0; bank < 2; ++bank) {
    const float* pw = (bank ? t.Wg : t.Wh) + (size_t)(4 * kq) * t.K + kcol;
#pragma unroll 4
    for (int c = wave; c < nchunk; c += 4) {
      const int k0 = c * 16;
      const bool ok = k0 + 4 * kq + 4 <= t.N;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      float w[4] = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        const float4 d = *reinterpret_cast<const float4*>(pd + k0);
        const float4 sv = *reinterpret_cast<const float4*>(ps + k0);
        if (bank == 0) {
          a = make_float4(d.x * sv.x, d.y * sv.y, d.z * sv.z, d.w * sv.w);
        } else {
          const float4 ov = *reinterpret_cast<const float4*>(pg + k0);
          a = make_float4(d.x * ov.x * (1.0f - sv.x), d.y * ov.y * (1.0f - sv.y), d.z * ov.z * (1.0f - sv.z), d.w * ov.w * (1.0f - sv.w));
        }
        if (writer) *reinterpret_cast<float4*>(pp + (bank ? t.N : 0) + k0) = a;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = pw[(size_t)(k0 + j) * t.K];
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[3], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[wave][4 * kq + r][i] = acc[r];
  __syncthreads();
  const int row = tid >> 4, col = tid & 15;
  const int m = m0 + row, n = n0 + col;
  if (m >= t.M || n >= t.K) return;
  t.dx[(size_t)m * t.ldo + n] = ((part[0][row][col] + part[1][row][col]) + part[2][row][col]) + part[3][row][col];
}

// which launches take the thin kernel: a batch-sized row count, aligned float4 rows (EVAE_THIN=0: the tiled kernels)
static bool thin_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EVAE_THIN"); on = (e && atoi(e) == 0) ? 0 : 1; }
  return on == 1;
}
// Batch-sized row counts always; up to EVAE_THIN_ROWS (default 4096) rows when the layer is a hidden-width one (contraction
// <= 640): measured on the captured step, C = 400 exemplars 0.219 -> 0.210 ms, C = 1000 (c1) 0.234 -> 0.224, C = 3125 (one
// rank's shard of c2 over 8 GPUs) 0.293 -> 0.281 -- a tile there re-reads 2 x 16 rows of operands from L2 per 16 x 16 outputs,
// which a 784-wide contraction over thousands of rows would not repay.
static int g_thin_rows = -1;          // (this header belongs to one translation unit, evae_dense.hip: evae_thin_configure sets it)
static int thin_max_rows() {
  if (g_thin_rows < 0) { const char* e = getenv("EVAE_THIN_ROWS"); g_thin_rows = e ? atoi(e) : 4096; }
  return g_thin_rows;
}
static bool thin_ok(int M, int contraction, int lda, const void* a, const void* a1) {
  return thin_enabled() && M > 0 && (M <= 128 || (M <= thin_max_rows() && contraction <= 640)) && contraction % 4 == 0 &&
         lda % 4 == 0 && contraction >= 16 && ((((uintptr_t)a | (uintptr_t)a1) & 15) == 0);
}
template <int EPI, bool BWD>
static int launch_thin(const ThinArgs& t, hipStream_t stream, const char* what) {
  const int nout = BWD ? t.K : t.N;
  const int mrows = (EPI == THIN_GATE_BWD && t.u8img) ? ((t.M + 7) & ~7) : t.M;
  thin_layer_kernel<EPI, BWD><<<dim3(cdiv(nout, 16), cdiv(mrows, 16)), 256, 0, stream>>>(t);
  return check_launch(what);
}


// ---- the two heads on one trunk + the sample (or the density of a given sample), batch-sized: ONE launch ---------------------
// mean = x Wm^T + bm, logvar = clamp(x Wl^T + bl), z = mean + eps exp(logvar / 2) (or a given z), log q = sum_k log N(z_k | ...)
// (reference models/VAE.py:24-26, models/BaseModel.py:79-82, utils/distributions.py:28-33; what evae_heads_reparam_fwd computes
// with a split-K GEMM and a finish launch).  A block owns 16 rows and ALL Z <= 64 columns -- wave w the columns 16 w .. -- so the
// row reduction of log q stays inside the block: lanes of a row add their columns in a fixed order.
struct ThinHeadsArgs {
  const float* x; int ldx, M, K, Z;
  const float* wm; const float* bm; const float* wl; const float* bl;
  float lo, hi;
  const float* eps;        // fresh sample: noise [M x Z]
  const float* z_given;    // or: the sample whose density is wanted (eps unused)
  float* z_mean; float* lv_pre; float* logvar; float* z; float* logq;
  const float* bc_src; float* bc_dst; int bc_n;     // block 0 also writes bc_dst[0 .. bc_n) = bc_src[0] (evae_broadcast_scalar's work)
};

template <int NTILE>       // column tiles of 16: Z <= 16 NTILE
__global__ __launch_bounds__(256) void thin_heads_kernel(const ThinHeadsArgs t) {
  __shared__ float part[4][2][16][16 * NTILE + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * 16;
  if (t.bc_dst && blockIdx.x == 0) {
    const float v = t.bc_src[0];
    for (int j = tid; j < t.bc_n; j += 256) t.bc_dst[j] = v;
  }
  thin_f32x4 am[NTILE], al[NTILE];
#pragma unroll
  for (int j = 0; j < NTILE; ++j) { am[j] = thin_f32x4{0.f, 0.f, 0.f, 0.f}; al[j] = thin_f32x4{0.f, 0.f, 0.f, 0.f}; }
  {
    // wave w takes the chunks w, w + 4, ... of the contraction for every column tile (the four partial sums meet in LDS in a
    // fixed order: the same four-way split as thin_layer_kernel, and a quarter of the accumulation chain)
    const int mrow = (m0 + i < t.M) ? m0 + i : t.M - 1;
    const float* pa = t.x + (size_t)mrow * t.ldx + 4 * kq;
    const float* pm[NTILE]; const float* pl[NTILE];
#pragma unroll
    for (int j = 0; j < NTILE; ++j) {
      const int ncol = (16 * j + i < t.Z) ? 16 * j + i : t.Z - 1;
      pm[j] = t.wm + (size_t)ncol * t.K + 4 * kq;
      pl[j] = t.wl + (size_t)ncol * t.K + 4 * kq;
    }
    const int nchunk = (t.K + 15) >> 4;
#pragma unroll 2
    for (int c = wave; c < nchunk; c += 4) {
      const int k0 = c * 16;
      const bool ok = k0 + 4 * kq + 4 <= t.K;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 wm4[NTILE], wl4[NTILE];
      if (ok) a = *reinterpret_cast<const float4*>(pa + k0);
#pragma unroll
      for (int j = 0; j < NTILE; ++j) {
        wm4[j] = ok ? *reinterpret_cast<const float4*>(pm[j] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
        wl4[j] = ok ? *reinterpret_cast<const float4*>(pl[j] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < NTILE; ++j) {
        am[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wm4[j].x, am[j], 0, 0, 0);
        al[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wl4[j].x, al[j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NTILE; ++j) {
        am[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wm4[j].y, am[j], 0, 0, 0);
        al[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wl4[j].y, al[j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NTILE; ++j) {
        am[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wm4[j].z, am[j], 0, 0, 0);
        al[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wl4[j].z, al[j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NTILE; ++j) {
        am[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wm4[j].w, am[j], 0, 0, 0);
        al[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wl4[j].w, al[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < NTILE; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        part[wave][0][4 * kq + r][16 * j + i] = am[j][r];
        part[wave][1][4 * kq + r][16 * j + i] = al[j][r];
      }
  }
  __syncthreads();
  // 16 lanes per row, lane c of them the columns c, c + 16, ...
  const int row = tid >> 4, c0 = tid & 15, m = m0 + row;
  float acc = 0.f;
  if (m < t.M) {
    for (int k = c0; k < t.Z; k += 16) {
      const size_t o = (size_t)m * t.Z + k;
      const float mu = (((part[0][0][row][k] + part[1][0][row][k]) + part[2][0][row][k]) + part[3][0][row][k]) + (t.bm ? t.bm[k] : 0.f);
      const float p = (((part[0][1][row][k] + part[1][1][row][k]) + part[2][1][row][k]) + part[3][1][row][k]) + (t.bl ? t.bl[k] : 0.f);
      const float lv = fminf(fmaxf(p, t.lo), t.hi);
      const float zz = t.z_given ? t.z_given[o] : t.eps[o] * expf(0.5f * lv) + mu;
      t.z_mean[o] = mu;
      if (t.lv_pre) t.lv_pre[o] = p;
      t.logvar[o] = lv;
      if (!t.z_given) t.z[o] = zz;
      const float d = zz - mu;
      acc += -0.5f * (lv + kLog2Pi + d * d / expf(lv));
    }
  }
  // the 16 lanes of a row are consecutive lanes of one wave: fixed butterfly
  acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 8, 64);
  if (c0 == 0 && m < t.M && t.logq) t.logq[m] = acc;
}

static bool thin_heads_ok(int M, int K, int Z, int ldx, const void* x, const void* wm, const void* wl) {
  return thin_ok(M, K, ldx, x, (const void*)((uintptr_t)wm | (uintptr_t)wl)) && Z <= 64;
}
static int launch_thin_heads(const ThinHeadsArgs& t, hipStream_t stream, const char* what) {
  const int nt = cdiv(t.Z, 16), nb = cdiv(t.M, 16);
  if (nt <= 1) thin_heads_kernel<1><<<nb, 256, 0, stream>>>(t);
  else if (nt == 2) thin_heads_kernel<2><<<nb, 256, 0, stream>>>(t);
  else if (nt == 3) thin_heads_kernel<3><<<nb, 256, 0, stream>>>(t);
  else thin_heads_kernel<4><<<nb, 256, 0, stream>>>(t);
  return check_launch(what);
}

}  // namespace evae
