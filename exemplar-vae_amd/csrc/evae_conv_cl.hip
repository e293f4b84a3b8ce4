#include "evae_gemm_x6.h"

// ========================================================================================================
// Convolutions over channels-last tensors as instances of the GEMM of evae_gemm_kernel.h (CV = 1 / 2, see ConvMap).
// Activations are [N][H][W][C] ("NHWC": torch.channels_last storage of a logical NCHW tensor); filters keep the
// nn.Conv2d layout [Co][C][KH][KW] at the boundary and are re-ordered to (kh, kw, c) by a small pre-pass.
// ========================================================================================================
namespace evae {

// wp[co][t][c] = w[co][c][t] for c < C, 0 for C <= c < Cv   (forward B operand, k-contiguous rows of K = taps * Cv; Cv = C
// rounded up to a multiple of 32: the K-extent of one tap, so that a 32-wide K-slab never straddles two taps)
__global__ void cl_permute_fwd_kernel(const float* __restrict__ w, int Co, int C, int Cv, int taps, float* __restrict__ wp) {
  const size_t n = (size_t)Co * Cv * taps;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cv);
    const int t = (int)((i / Cv) % taps);
    const int co = (int)(i / ((size_t)Cv * taps));
    wp[i] = c < C ? w[((size_t)co * C + c) * taps + t] : 0.f;
  }
}
// wp[j][cc][c], cc < ld: rows of the data-gradient B operand of one stride-parity class, k' = (j, cc) with cc running
// over the channels of the merged gradient buffer: cc < Co -> wh[cc][c][tap_j], Co <= cc < 2Co -> wg[cc - Co][c][tap_j]
// (gated), anything beyond -> 0 (the buffer's zero padding up to a multiple of 32)
struct TapList { int n; int t[64]; };
// kc: the transposed arrangement wp[c][j][cc] (contraction-contiguous rows, the B operand of the split-bf16 kernel)
__global__ void cl_permute_dgrad_kernel(const float* __restrict__ wh, const float* __restrict__ wg, int Co, int C,
                                        int taps, int ld, TapList tl, float* __restrict__ wp, int kc) {
  const size_t n = (size_t)tl.n * ld * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int cc = (int)((i / C) % ld);
    const int j = (int)(i / ((size_t)C * ld));
    float v = 0.f;
    if (cc < Co) v = wh[((size_t)cc * C + c) * taps + tl.t[j]];
    else if (wg != nullptr && cc < 2 * Co) v = wg[((size_t)(cc - Co) * C + c) * taps + tl.t[j]];
    wp[kc ? ((size_t)c * tl.n + j) * ld + cc : i] = v;
  }
}

// Patch matrix of a thin first layer (C*KH*KW <= 64, e.g. 1 x 7 x 7): P[m = (n, oy, ox)][k = (tap, c)], zero-padded to
// Kp columns, so that the layer is ONE dense GEMM with a 32- or 64-wide contraction and a channels-last output.
// One thread per (pixel, four consecutive columns of its patch row): a wave writes 1 KB of the matrix with 16-byte stores (r04; it
// was one thread per (pixel, filter row) writing KW * C scalars at a 28-byte pitch: 0.8 TB/s, 7.2 % of the c3 step); the four
// source elements of a thread come from the image's few KB, which the pixel's neighbours keep in L1.  Kp % 4 == 0.
__global__ void cl_patches_kernel(const float* __restrict__ x, int C, int H, int W, int OH, int OW, int KH, int KW,
                                  int stride, int pad, int Kp, size_t npix, float* __restrict__ P) {
  const int q = Kp >> 2, run = KW * C, kreal = KH * run;
  const size_t total = npix * (size_t)q;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / q;
    const int k0 = (int)(i - m * q) << 2;
    const int ox = (int)(m % OW), oy = (int)((m / OW) % OH);
    const size_t n = m / ((size_t)OW * OH);
    const int y0 = oy * stride - pad, x0 = ox * stride - pad;
    const float* img = x + n * (size_t)H * W * C;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + j;
      const int kh = k / run, r = k - kh * run, kw = r / C, c = r - kw * C;       // column k = (tap (kh, kw), channel c)
      const int y = y0 + kh, xx = x0 + kw;
      const bool ok = k < kreal && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
      v[j] = ok ? img[((size_t)y * W + xx) * C + c] : 0.f;
    }
    *reinterpret_cast<float4*>(P + m * Kp + k0) = make_float4(v[0], v[1], v[2], v[3]);
  }
}
// wp[co][k] = w[co][c][t] for k = t*C + c < C*taps, 0 for the padding columns
__global__ void cl_permute_patch_w_kernel(const float* __restrict__ w, int Co, int C, int taps, int Kp,
                                          float* __restrict__ wp) {
  const size_t n = (size_t)Co * Kp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kp);
    const int co = (int)(i / Kp);
    const int t = k / C, c = k - t * C;
    wp[i] = t < taps ? w[((size_t)co * C + c) * taps + t] : 0.f;
  }
}
// Data gradient into a 32-channel input, two x-adjacent output pixels per GEMM row (64 output columns = a full tile):
// wp[u][cc][b*32 + c] = w_merged[cc][c][tb[b][u]] when tap u of the union belongs to pixel b (tb >= 0), else 0
struct PairTaps { int n; int tb[2][64]; };
__global__ void cl_permute_dgrad_pair_kernel(const float* __restrict__ wh, const float* __restrict__ wg, int Co, int taps,
                                             int ld, PairTaps pt, float* __restrict__ wp, int kc) {
  const int C = 32;
  const size_t n = (size_t)pt.n * ld * 64;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % 64), b = col >> 5, c = col & 31;
    const int cc = (int)((i / 64) % ld);
    const int u = (int)(i / ((size_t)64 * ld));
    const int t = pt.tb[b][u];
    float v = 0.f;
    if (t >= 0) {
      if (cc < Co) v = wh[((size_t)cc * C + c) * taps + t];
      else if (wg != nullptr && cc < 2 * Co) v = wg[((size_t)(cc - Co) * C + c) * taps + t];
    }
    wp[kc ? ((size_t)col * pt.n + u) * ld + cc : i] = v;
  }
}

static bool cl_patch_mode(const evae_conv_desc_t* d) { return d->C % 32 != 0 && d->C * d->KH * d->KW <= 64; }
// columns of the patch matrix: the taps, padded to fours (the GEMM kernels mask a K tail; every padded column costs 4 bytes
// of HBM per output pixel twice, written and read)
static int cl_patch_kp(const evae_conv_desc_t* d) { return (d->C * d->KH * d->KW + 3) / 4 * 4; }
// images per pass of the patch path: the patch matrix of a pass stays below 1 GiB
static int cl_patch_images(const evae_conv_desc_t* d, int OH, int OW, int chan_out) {
  const int64_t lim = (int64_t)1 << 28;     // floats
  const int64_t per = std::max((int64_t)OH * OW * cl_patch_kp(d), (int64_t)OH * OW * chan_out);
  int64_t n = lim / (per > 0 ? per : 1);
  if (const char* e = getenv("EVAE_CL_IMAGES_PER_PASS")) {
    const int64_t f = atoll(e);
    if (f > 0 && f < n) n = f;
  }
  return (int)std::min<int64_t>(n, d->N);
}

// The buffer-load offsets are 31-bit: a pass handles at most this many images (0: one image alone is too big)
static int cl_images_per_pass(const evae_conv_desc_t* d, int OH, int OW, int chan_in, int chan_out) {
  const int64_t lim = ((int64_t)1 << 29) - ((int64_t)1 << 22);   // floats; room for the bias terms
  const int64_t per = std::max((int64_t)d->H * d->W * chan_in, (int64_t)OH * OW * chan_out);
  int64_t n = lim / (per > 0 ? per : 1);
  if (const char* e = getenv("EVAE_CL_IMAGES_PER_PASS")) {       // tests: exercise the multi-pass logic on small tensors
    const int64_t f = atoll(e);
    if (f > 0 && f < n) n = f;
  }
  return (int)std::min<int64_t>(n, d->N);
}
static void cl_out_dims(const evae_conv_desc_t* d, int* OH, int* OW) {
  *OH = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  *OW = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
}

}  // namespace evae

using namespace evae;

extern "C" int evae_conv2d_cl_dy_stride(int ctot);

extern "C" int evae_conv2d_cl_supported(const evae_conv_desc_t* d, int what, int gated) {
  if (!d || d->N <= 0 || d->stride < 1 || d->KH * d->KW > 64) return 0;
  int OH, OW;
  cl_out_dims(d, &OH, &OW);
  if (OH <= 0 || OW <= 0) return 0;
  const int ctot = d->Co * (gated ? 2 : 1);
  if (cl_images_per_pass(d, OH, OW, d->C, evae_conv2d_cl_dy_stride(ctot)) < 1) return 0;
  if (cl_patch_mode(d)) {                       // thin first layer: patch matrix + dense GEMM (no data gradient)
    if (cl_patch_images(d, OH, OW, evae_conv2d_cl_dy_stride(ctot)) < 1) return 0;
    return what == 0 || (what == 2 && ctot % 4 == 0);
  }
  if (what == 0) return d->C % 4 == 0 && d->C >= 16;      // a tap's channels are padded to a multiple of 32 in the K index only
  if (what == 1) return d->C % 4 == 0;
  return d->C % 4 == 0 && ctot % 4 == 0;
}

// channels per pixel of the merged gradient buffer: ctot itself when rows stay 16-byte aligned (the contraction pads a
// pixel to a multiple of 32 channels in the K index only, reading the missing ones as zero), else ctot rounded up to a
// multiple of 32 with real zero padding
extern "C" int evae_conv2d_cl_dy_stride(int ctot) { return ctot % 4 == 0 ? ctot : (ctot + 31) / 32 * 32; }
static int cl_ldv(int ldy) { return (ldy + 31) / 32 * 32; }      // K-extent of one pixel of the gradient buffer

extern "C" size_t evae_conv2d_cl_workspace_bytes(const evae_conv_desc_t* d, int what, int gated) {
  if (!d) return 256;
  const size_t K = (size_t)d->C * d->KH * d->KW;
  const size_t Kv = (size_t)((d->C + 31) / 32 * 32) * d->KH * d->KW;      // forward: taps padded to a multiple of 32 channels
  const size_t wbytes = align_up((size_t)d->Co * Kv * sizeof(float), 256);
  if (cl_patch_mode(d)) {
    int OH, OW;
    cl_out_dims(d, &OH, &OW);
    const int ctot = d->Co * (gated ? 2 : 1), Kp = cl_patch_kp(d);
    const int per = cl_patch_images(d, OH, OW, evae_conv2d_cl_dy_stride(ctot));
    const size_t pbytes = align_up((size_t)per * OH * OW * Kp * sizeof(float), 256);
    if (what == 0) return pbytes + 2 * align_up((size_t)d->Co * Kp * sizeof(float), 256) + 256;
    Plan pl = make_plan(ctot, Kp + 1, cdiv(per * OH * OW, BK), false, true, 1);
    return pbytes + align_up((size_t)pl.nz * ctot * (Kp + 1) * sizeof(float), 256) + 256;
  }
  if (what == 0) return (gated ? 2 : 1) * wbytes + 256;
  if (what == 1) {   // one permuted copy [taps][ldy][C], class slices are disjoint parts of it
    const int ctot1 = d->Co * (gated ? 2 : 1);
    const size_t plain = (size_t)d->KH * d->KW * cl_ldv(ctot1) * d->C;
    const size_t paired = (size_t)d->KH * (d->KW + 1) * cl_ldv(ctot1) * 64;   // pixel-pair form (C = 32)
    return align_up(std::max(plain, paired) * sizeof(float), 256) + 256;
  }
  // weight gradient: split-K partial planes [nz][ctot][K + 1]
  int OH, OW;
  cl_out_dims(d, &OH, &OW);
  const int ctot = d->Co * (gated ? 2 : 1);
  const int per = cl_images_per_pass(d, OH, OW, d->C, evae_conv2d_cl_dy_stride(ctot));
  size_t best = 0;     // the first (full) pass and the last (remainder) pass may plan different splits
  for (int nn : {std::min(per, d->N), per > 0 ? (d->N % per) : 0}) {
    if (nn <= 0) continue;
    Plan pl = make_plan(ctot, (int)K + 1, cdiv(nn * OH * OW, BK), false, true, 1);
    const int nz6 = x6t_split(nn * OH * OW, ctot, (int)K + 1).nz;         // the split-bf16 kernel's own split
    best = std::max(best, (size_t)std::max(pl.nz, nz6) * ctot * (K + 1) * sizeof(float));
  }
  return align_up(best, 256) + 256;
}

static int cl_fwd_impl(const float* x, const evae_conv_desc_t* d, const float* wh, const float* bh,
                                  const float* wg, const float* bg, int act, float act_lo, float act_hi,
                                  float* out, float* save_h, float* save_s, void* ws, size_t ws_bytes,
                                  evae_stream_t stream_, const float* residual) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(d && x && wh && out, "conv2d_cl_fwd: null pointer");
  const bool gated = wg != nullptr;
  EVAE_REQUIRE(evae_conv2d_cl_supported(d, 0, gated), "conv2d_cl_fwd: unsupported geometry (C %% 32, size)");
  EVAE_REQUIRE(ws && ws_bytes >= evae_conv2d_cl_workspace_bytes(d, 0, gated), "conv2d_cl_fwd: workspace too small");
  EVAE_REQUIRE(act >= 0 && act <= 2, "conv2d_cl_fwd: bad activation %d", act);
  int OH, OW;
  cl_out_dims(d, &OH, &OW);
  if (cl_patch_mode(d)) {
    const int taps = d->KH * d->KW, Kp = cl_patch_kp(d);
    const int per = cl_patch_images(d, OH, OW, evae_conv2d_cl_dy_stride(d->Co * (gated ? 2 : 1)));   // as sized in the workspace query
    float* P = (float*)ws;
    const size_t pbytes = align_up((size_t)per * OH * OW * Kp * sizeof(float), 256);
    float* wph = (float*)((char*)ws + pbytes);
    float* wpg = (float*)((char*)wph + align_up((size_t)d->Co * Kp * sizeof(float), 256));
    cl_permute_patch_w_kernel<<<elt_grid((size_t)d->Co * Kp), 256, 0, stream>>>(wh, d->Co, d->C, taps, Kp, wph);
    if (gated) cl_permute_patch_w_kernel<<<elt_grid((size_t)d->Co * Kp), 256, 0, stream>>>(wg, d->Co, d->C, taps, Kp, wpg);
    for (int n0 = 0; n0 < d->N; n0 += per) {
      const int nn = std::min(per, d->N - n0);
      const size_t npix = (size_t)nn * OH * OW, oo = (size_t)n0 * OH * OW * d->Co;
      cl_patches_kernel<<<elt_grid(npix * (size_t)(Kp / 4)), 256, 0, stream>>>(x + (size_t)n0 * d->H * d->W * d->C, d->C, d->H, d->W,
                                                                     OH, OW, d->KH, d->KW, d->stride, d->pad, Kp, npix, P);
      int rc = check_launch("cl_patches_kernel");
      if (rc) return rc;
      GemmArgs g = {};
      g.ones_col = -1;
      g.A[0] = P; g.B[0] = wph; g.Bg = gated ? wpg : nullptr;
      g.lda[0] = Kp; g.ldb[0] = Kp; g.Kc[0] = Kp; g.npairs = 1;
      g.M = (int)npix; g.N = d->Co; g.bias0 = bh; g.bias1 = bg; g.ldo = d->Co;
      g.act = act; g.lo = act_lo; g.hi = act_hi; g.ksplit = 0;
      g.out0 = out + oo; g.out1 = save_h ? save_h + oo : nullptr; g.out2 = (gated && save_s) ? save_s + oo : nullptr;
      if (gated && gemm_x6_use(g, true)) rc = launch_gemm_x6<EPI_GATED, 0, 128>(g, 1, stream, "conv2d_cl_fwd(patches, gated, x6)");
      else if (gated) rc = launch_gemm_w<true, true, EPI_GATED, true, 128, 8>(g, 1, stream, "conv2d_cl_fwd(patches, gated)");
      else if (gemm_x6_use(g) && d->Co <= 64) rc = launch_gemm_x6<EPI_LINEAR, 0, 64>(g, 1, stream, "conv2d_cl_fwd(patches, x6)");
      else if (gemm_x6_use(g)) rc = launch_gemm_x6<EPI_LINEAR, 0, 128>(g, 1, stream, "conv2d_cl_fwd(patches, x6)");
      else if (d->Co <= 64) rc = launch_gemm_w<true, true, EPI_LINEAR, true, 64, 8>(g, 1, stream, "conv2d_cl_fwd(patches)");
      else rc = launch_gemm_w<true, true, EPI_LINEAR, true, 128, 8>(g, 1, stream, "conv2d_cl_fwd(patches)");
      if (rc) return rc;
    }
    return EVAE_OK;
  }
  const int Cv = (d->C + 31) / 32 * 32;
  const int taps = d->KH * d->KW, K = taps * Cv, M = d->N * OH * OW;
  float* wph = (float*)ws;
  float* wpg = (float*)((char*)ws + align_up((size_t)d->Co * K * sizeof(float), 256));
  cl_permute_fwd_kernel<<<elt_grid((size_t)d->Co * K), 256, 0, stream>>>(wh, d->Co, d->C, Cv, taps, wph);
  if (gated) cl_permute_fwd_kernel<<<elt_grid((size_t)d->Co * K), 256, 0, stream>>>(wg, d->Co, d->C, Cv, taps, wpg);
  int rc = check_launch("cl_permute_fwd_kernel");
  if (rc) return rc;
  GemmArgs g = {};
  g.ones_col = -1;
  ConvMap& cv = g.cv;
  cv.Cg = Cv; cv.creal = d->C; cv.ps = d->C; cv.ntaps = taps;
  cv.RH = OH; cv.RW = OW; cv.IH = d->H; cv.IW = d->W;
  cv.rs = d->stride; cv.rsx = d->stride; cv.roy = -d->pad; cv.rox = -d->pad;
  cv.remap = 0;
  cv.div_rw = make_fastdiv((unsigned)OW); cv.div_rhw = make_fastdiv((unsigned)(OH * OW));
  cv.bias = (unsigned)((d->pad * d->W + d->pad) * d->C * 4);
  for (int kh = 0; kh < d->KH; ++kh)
    for (int kw = 0; kw < d->KW; ++kw) {
      const int t = kh * d->KW + kw;
      cv.tdy[t] = (signed char)kh; cv.tdx[t] = (signed char)kw;
      cv.tsoff[t] = (kh * d->W + kw) * d->C * 4;
    }
  g.B[0] = wph; g.Bg = gated ? wpg : nullptr;
  g.lda[0] = d->C; g.ldb[0] = K; g.Kc[0] = K; g.npairs = 1;
  g.N = d->Co; g.bias0 = bh; g.bias1 = bg; g.ldo = d->Co;
  g.act = act; g.lo = act_lo; g.hi = act_hi;
  g.ksplit = 0;
  (void)M;
  const int per = cl_images_per_pass(d, OH, OW, d->C, d->Co);
  for (int n0 = 0; n0 < d->N; n0 += per) {           // independent images: passes of at most 2 GiB each
    const int nn = std::min(per, d->N - n0);
    const size_t xo = (size_t)n0 * d->H * d->W * d->C, oo = (size_t)n0 * OH * OW * d->Co;
    g.A[0] = x + xo - cv.bias / 4;     // the per-row offsets carry +bias (they are never negative)
    g.M = nn * OH * OW;
    int rc;
    const bool x6 = gemm_x6_enabled() && gemm_x6_fills(g.M, d->Co, gated);
    if (gated) {
      g.out0 = out + oo; g.out1 = save_h ? save_h + oo : nullptr; g.out2 = save_s ? save_s + oo : nullptr;
      if (x6) rc = launch_gemm_x6<EPI_GATED, 1, 128>(g, 1, stream, "conv2d_cl_fwd(gated, x6)");
      else rc = launch_gemm_w<true, true, EPI_GATED, true, 128, 8, 1>(g, 1, stream, "conv2d_cl_fwd(gated)");
    } else {
      g.out0 = out + oo; g.out1 = save_h ? save_h + oo : nullptr;    // pre-activation when requested
      g.e0 = residual ? residual + oo : nullptr;                     // residual block: out = conv(x) + residual
      if (x6 && d->Co <= 64) rc = launch_gemm_x6<EPI_LINEAR, 1, 64>(g, 1, stream, "conv2d_cl_fwd(x6)");
      else if (x6) rc = launch_gemm_x6<EPI_LINEAR, 1, 128>(g, 1, stream, "conv2d_cl_fwd(x6)");
      else if (d->Co <= 64) rc = launch_gemm_w<true, true, EPI_LINEAR, true, 64, 8, 1>(g, 1, stream, "conv2d_cl_fwd");
      else rc = launch_gemm_w<true, true, EPI_LINEAR, true, 128, 8, 1>(g, 1, stream, "conv2d_cl_fwd");
    }
    if (rc) return rc;
  }
  return EVAE_OK;
}

extern "C" int evae_conv2d_cl_fwd(const float* x, const evae_conv_desc_t* d, const float* wh, const float* bh,
                                  const float* wg, const float* bg, int act, float act_lo, float act_hi,
                                  float* out, float* save_h, float* save_s, void* ws, size_t ws_bytes,
                                  evae_stream_t stream_) {
  return cl_fwd_impl(x, d, wh, bh, wg, bg, act, act_lo, act_hi, out, save_h, save_s, ws, ws_bytes, stream_, nullptr);
}

// One predicate for the residual-block entry points below (Python asks it instead of re-stating the geometry rules)
extern "C" int evae_conv2d_cl_res_supported(const evae_conv_desc_t* d) {
  if (!d || d->C != d->Co || d->stride != 1 || d->KH != d->KW || 2 * d->pad + 1 != d->KH || d->C == 32) return 0;
  if (cl_patch_mode(d)) return 0;
  return evae_conv2d_cl_supported(d, 0, 0) && evae_conv2d_cl_supported(d, 1, 0) && evae_conv2d_cl_supported(d, 2, 0);
}

// out = conv(a, w) + b + residual: the forward of a fully_conv residual block (a = ELU(x), residual = x; models/fully_conv.py:13-23)
extern "C" int evae_conv2d_cl_fwd_res(const float* a, const evae_conv_desc_t* d, const float* w, const float* b,
                                      const float* residual, float* out, void* ws, size_t ws_bytes, evae_stream_t stream_) {
  EVAE_REQUIRE(d && residual && d->C == d->Co && d->stride == 1 && 2 * d->pad + 1 == d->KH && d->KH == d->KW,
               "conv2d_cl_fwd_res: a residual block keeps the tensor shape");
  EVAE_REQUIRE(!cl_patch_mode(d), "conv2d_cl_fwd_res: thin layers are not residual blocks");
  return cl_fwd_impl(a, d, w, b, nullptr, nullptr, EVAE_ACT_NONE, 0.f, 0.f, out, nullptr, nullptr, ws, ws_bytes, stream_, residual);
}

// dy: [N][OH][OW][ldy], ldy = evae_conv2d_cl_dy_stride(ctot): channels [0, Co) = dh, [Co, 2Co) = dg (gated), the rest
// zero; dx: [N][H][W][C].  One GEMM per stride-parity class, contraction over (tap of the class, buffer channel).
static int cl_bwd_data_impl(const float* dy, const float* wh, const float* wg, const evae_conv_desc_t* d,
                            float* dx, void* ws, size_t ws_bytes, evae_stream_t stream_, const float* residual,
                            const float* elu_out) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(d && dy && wh && dx, "conv2d_cl_bwd_data: null pointer");
  const bool gated = wg != nullptr;
  EVAE_REQUIRE(evae_conv2d_cl_supported(d, 1, gated), "conv2d_cl_bwd_data: unsupported geometry");
  EVAE_REQUIRE(ws && ws_bytes >= evae_conv2d_cl_workspace_bytes(d, 1, gated), "conv2d_cl_bwd_data: workspace too small");
  int OH, OW;
  cl_out_dims(d, &OH, &OW);
  const int taps = d->KH * d->KW, s = d->stride, Co = d->Co, C = d->C;
  const int ldy = evae_conv2d_cl_dy_stride(Co * (gated ? 2 : 1)), ldv = cl_ldv(ldy);
  float* wp = (float*)ws;
  if (C == 32 && (s == 1 || s == 2) && d->W % 2 == 0 && d->KH * (d->KW + 1) <= 64) {
    // A 32-column result would leave half of the 64-wide tile idle.  Two x-adjacent output pixels (x = 2 rx + b) share
    // one GEMM row instead: their taps are the same source pixels shifted by one, so the contraction runs over the
    // union of the two tap sets with zero filter blocks where a tap belongs to only one of them, and the 64 output
    // columns are the two pixels' channels -- contiguous in the channels-last dx.
    const int W2 = d->W / 2;
    size_t usedp = 0;
    for (int py = 0; py < s && py < d->H; ++py) {
      PairTaps pt; pt.n = 0;
      GemmArgs g = {};
      g.ones_col = -1;
      ConvMap& cv = g.cv;
      int tmin = 0;
      for (int kh = 0; kh < d->KH; ++kh) {
        if ((py + d->pad - kh) % s != 0) continue;
        const int dy_ = (py + d->pad - kh) / s;
        for (int b = 0; b < 2; ++b)
          for (int kw = 0; kw < d->KW; ++kw) {
            // source column of tap kw for pixel b, relative to the anchor (rx*rsx): s = 1: x = 2rx + b -> b + pad - kw;
            // s = 2: x = 2rx + b is the pixel of class px = b -> (b + pad - kw) / 2 when divisible
            if (s == 2 && (b + d->pad - kw) % 2 != 0) continue;
            const int dx_ = s == 1 ? b + d->pad - kw : (b + d->pad - kw) / 2;
            int u = -1;
            for (int q = 0; q < pt.n; ++q)
              if (cv.tdy[q] == dy_ && cv.tdx[q] == dx_) u = q;
            if (u < 0) {
              u = pt.n++;
              cv.tdy[u] = (signed char)dy_; cv.tdx[u] = (signed char)dx_;
              pt.tb[0][u] = pt.tb[1][u] = -1;
              cv.tsoff[u] = (dy_ * OW + dx_) * ldy * 4;
              if (cv.tsoff[u] < tmin) tmin = cv.tsoff[u];
            }
            pt.tb[b][u] = kh * d->KW + kw;
          }
      }
      const int RH = (d->H - py + s - 1) / s;
      if (pt.n == 0) {     // no tap reaches these rows (stride larger than the filter): their gradient is zero
        for (int n = 0; n < d->N; ++n)
          for (int ry = 0; ry < RH; ++ry) {
            hipError_t e = hipMemsetAsync(dx + (((size_t)n * d->H + ry * s + py) * d->W) * C, 0, (size_t)d->W * C * sizeof(float), stream);
            EVAE_REQUIRE(e == hipSuccess, "conv2d_cl_bwd_data: memset failed");
          }
        continue;
      }
      const unsigned tbias = (unsigned)(-tmin);
      for (int u = 0; u < pt.n; ++u) cv.tsoff[u] += (int)tbias;
      const size_t cls = (size_t)pt.n * ldv * 64;
      float* wc = wp + usedp;
      usedp += cls;
      const bool x6 = gemm_x6_enabled() && !residual && gemm_x6_fills(std::min(cl_images_per_pass(d, OH, OW, C, ldy), d->N) * RH * W2, 64, false);
      cl_permute_dgrad_pair_kernel<<<elt_grid(cls), 256, 0, stream>>>(wh, wg, Co, taps, ldv, pt, wc, x6 ? 1 : 0);
      int rc = check_launch("cl_permute_dgrad_pair_kernel");
      if (rc) return rc;
      cv.Cg = ldv; cv.creal = ldy; cv.ps = ldy; cv.ntaps = pt.n;
      cv.RH = RH; cv.RW = W2; cv.IH = OH; cv.IW = OW;
      cv.rs = 1; cv.rsx = (s == 1) ? 2 : 1; cv.roy = 0; cv.rox = 0;
      cv.OH2 = d->H; cv.OW2 = W2; cv.os = s; cv.osx = 1; cv.ooy = py; cv.oox = 0;   // output rows in units of pixel pairs
      cv.remap = 1;
      cv.div_rw = make_fastdiv((unsigned)W2); cv.div_rhw = make_fastdiv((unsigned)(RH * W2));
      cv.bias = 0;
      g.B[0] = wc;
      g.lda[0] = ldy; g.ldb[0] = x6 ? pt.n * ldv : 64;
      g.Kc[0] = pt.n * ldv; g.npairs = 1;
      g.N = 64; g.ldo = 64;
      g.ksplit = 0;
      const int per = cl_images_per_pass(d, OH, OW, C, ldy);
      for (int n0 = 0; n0 < d->N; n0 += per) {
        const int nn = std::min(per, d->N - n0);
        g.A[0] = dy + (size_t)n0 * OH * OW * ldy - tbias / 4;
        g.out0 = dx + (size_t)n0 * d->H * d->W * C;
        g.M = nn * RH * W2;
        if (x6) rc = launch_gemm_x6<EPI_LINEAR, 1, 64>(g, 1, stream, "conv2d_cl_bwd_data(pixel pairs, x6)");
        else rc = launch_gemm_w<true, false, EPI_LINEAR, true, 64, 8, 1>(g, 1, stream, "conv2d_cl_bwd_data(pixel pairs)");
        if (rc) return rc;
      }
    }
    return EVAE_OK;
  }
  size_t used = 0;      // floats of the permuted copy consumed by the classes so far
  bool any_empty = false;
  for (int py = 0; py < s && py < d->H; ++py)
    for (int px = 0; px < s && px < d->W; ++px) {
      int n = 0;
      for (int kh = 0; kh < d->KH; ++kh)
        for (int kw = 0; kw < d->KW; ++kw)
          if ((py + d->pad - kh) % s == 0 && (px + d->pad - kw) % s == 0) ++n;
      if (n == 0) any_empty = true;
    }
  if (any_empty) {
    hipError_t e = hipMemsetAsync(dx, 0, (size_t)d->N * d->H * d->W * C * sizeof(float), stream);
    EVAE_REQUIRE(e == hipSuccess, "conv2d_cl_bwd_data: memset failed");
  }
  for (int py = 0; py < s && py < d->H; ++py)
    for (int px = 0; px < s && px < d->W; ++px) {
      TapList tl; tl.n = 0;
      GemmArgs g = {};
      g.ones_col = -1;
      ConvMap& cv = g.cv;
      int tmin = 0;
      for (int kh = 0; kh < d->KH; ++kh)
        for (int kw = 0; kw < d->KW; ++kw)
          if ((py + d->pad - kh) % s == 0 && (px + d->pad - kw) % s == 0) {
            const int j = tl.n++;
            tl.t[j] = kh * d->KW + kw;
            const int dy_ = (py + d->pad - kh) / s, dx_ = (px + d->pad - kw) / s;   // exact; may be negative
            cv.tdy[j] = (signed char)dy_; cv.tdx[j] = (signed char)dx_;
            cv.tsoff[j] = (dy_ * OW + dx_) * ldy * 4;
            if (cv.tsoff[j] < tmin) tmin = cv.tsoff[j];
          }
      if (tl.n == 0) continue;
      const unsigned tbias = (unsigned)(-tmin);
      for (int j = 0; j < tl.n; ++j) cv.tsoff[j] += (int)tbias;
      const int RH = (d->H - py + s - 1) / s, RW = (d->W - px + s - 1) / s;
      const size_t cls = (size_t)tl.n * ldv * C;
      float* wc = wp + used;
      used += cls;
      const bool x6 = gemm_x6_enabled() && gemm_x6_fills(std::min(cl_images_per_pass(d, OH, OW, C, ldy), d->N) * RH * RW, C, false);
      cl_permute_dgrad_kernel<<<elt_grid(cls), 256, 0, stream>>>(wh, wg, Co, C, taps, ldv, tl, wc, x6 ? 1 : 0);
      int rc = check_launch("cl_permute_dgrad_kernel");
      if (rc) return rc;
      cv.Cg = ldv; cv.creal = ldy; cv.ps = ldy; cv.ntaps = tl.n;
      cv.RH = RH; cv.RW = RW; cv.IH = OH; cv.IW = OW;
      cv.rs = 1; cv.rsx = 1; cv.roy = 0; cv.rox = 0;
      cv.OH2 = d->H; cv.OW2 = d->W; cv.os = s; cv.osx = s; cv.ooy = py; cv.oox = px;
      cv.remap = 1;
      cv.div_rw = make_fastdiv((unsigned)RW); cv.div_rhw = make_fastdiv((unsigned)(RH * RW));
      cv.bias = 0;
      g.B[0] = wc;
      g.lda[0] = ldy; g.ldb[0] = x6 ? tl.n * ldv : C;
      g.Kc[0] = tl.n * ldv; g.npairs = 1;
      g.N = C; g.ldo = C;
      g.ksplit = 0;
      const int per = cl_images_per_pass(d, OH, OW, C, ldy);
      for (int n0 = 0; n0 < d->N; n0 += per) {
        const int nn = std::min(per, d->N - n0);
        g.A[0] = dy + (size_t)n0 * OH * OW * ldy - tbias / 4;
        g.out0 = dx + (size_t)n0 * d->H * d->W * C;
        g.e0 = residual ? residual + (size_t)n0 * d->H * d->W * C : nullptr;      // residual block: dx = dy + ELU'(x) * acc
        g.e1 = elu_out ? elu_out + (size_t)n0 * d->H * d->W * C : nullptr;
        g.M = nn * RH * RW;
        if (x6 && C <= 64) rc = launch_gemm_x6<EPI_LINEAR, 1, 64>(g, 1, stream, "conv2d_cl_bwd_data(x6)");
        else if (x6) rc = launch_gemm_x6<EPI_LINEAR, 1, 128>(g, 1, stream, "conv2d_cl_bwd_data(x6)");
        else if (C <= 64) rc = launch_gemm_w<true, false, EPI_LINEAR, true, 64, 8, 1>(g, 1, stream, "conv2d_cl_bwd_data");
        else rc = launch_gemm_w<true, false, EPI_LINEAR, true, 128, 8, 1>(g, 1, stream, "conv2d_cl_bwd_data");
        if (rc) return rc;
      }
    }
  return EVAE_OK;
}

extern "C" int evae_conv2d_cl_bwd_data(const float* dy, const float* wh, const float* wg, const evae_conv_desc_t* d,
                                       float* dx, void* ws, size_t ws_bytes, evae_stream_t stream_) {
  return cl_bwd_data_impl(dy, wh, wg, d, dx, ws, ws_bytes, stream_, nullptr, nullptr);
}

// dx = residual + ELU'(x) * conv_transpose(dy, w) with ELU'(x) = (a > 0 ? 1 : a + 1), a = ELU(x): the data gradient of a
// fully_conv residual block in one launch (residual = the upstream gradient itself, un-padded [N][H][W][C])
extern "C" int evae_conv2d_cl_bwd_data_res(const float* dy, const float* w, const evae_conv_desc_t* d, const float* residual,
                                           const float* elu_out, float* dx, void* ws, size_t ws_bytes, evae_stream_t stream_) {
  EVAE_REQUIRE(d && residual && elu_out && d->C == d->Co && d->stride == 1 && d->C != 32,
               "conv2d_cl_bwd_data_res: residual block geometry (C == Co, stride 1, C != 32)");
  return cl_bwd_data_impl(dy, w, nullptr, d, dx, ws, ws_bytes, stream_, residual, elu_out);
}

// dy as above ([..][ldy] per pixel); x: [N][H][W][C]; dw: [ctot][C][KH][KW] (nn.Conv2d layout, h rows then g rows), db: [ctot]
extern "C" int evae_conv2d_cl_bwd_weight(const float* dy, const float* x, const evae_conv_desc_t* d, int gated,
                                         float* dw, float* db, void* ws, size_t ws_bytes, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(d && dy && x && dw, "conv2d_cl_bwd_weight: null pointer");
  EVAE_REQUIRE(evae_conv2d_cl_supported(d, 2, gated), "conv2d_cl_bwd_weight: unsupported geometry");
  EVAE_REQUIRE(ws && ws_bytes >= evae_conv2d_cl_workspace_bytes(d, 2, gated), "conv2d_cl_bwd_weight: workspace too small");
  int OH, OW;
  cl_out_dims(d, &OH, &OW);
  const int taps = d->KH * d->KW, K = taps * d->C, Mpix = d->N * OH * OW;
  const int ctot = d->Co * (gated ? 2 : 1);
  const int ldy = evae_conv2d_cl_dy_stride(ctot);
  if (cl_patch_mode(d)) {
    const int Kp = cl_patch_kp(d);
    const int per = cl_patch_images(d, OH, OW, ldy);
    float* P = (float*)ws;
    float* part = (float*)((char*)ws + align_up((size_t)per * OH * OW * Kp * sizeof(float), 256));
    for (int n0 = 0; n0 < d->N; n0 += per) {
      const int nn = std::min(per, d->N - n0);
      const size_t npix = (size_t)nn * OH * OW;
      cl_patches_kernel<<<elt_grid(npix * (size_t)(Kp / 4)), 256, 0, stream>>>(x + (size_t)n0 * d->H * d->W * d->C, d->C, d->H, d->W,
                                                                     OH, OW, d->KH, d->KW, d->stride, d->pad, Kp, npix, P);
      int rc = check_launch("cl_patches_kernel");
      if (rc) return rc;
      Plan pl = make_plan(ctot, Kp + 1, cdiv((int)npix, BK), false, true, 1);
      GemmArgs g = {};
      g.A[0] = dy + (size_t)n0 * OH * OW * ldy; g.B[0] = P; g.lda[0] = ldy; g.ldb[0] = Kp; g.Kc[0] = (int)npix; g.npairs = 1;
      g.M = ctot; g.N = Kp + 1; g.ones_col = Kp; g.ldo = Kp + 1; g.out0 = part;
      g.ksplit = pl.nz > 1 ? pl.ksplit : 0;
      if (pl.bn == 128) rc = launch_gemm_w<false, false, EPI_RAW, true, 128, 8>(g, pl.nz, stream, "conv2d_cl_bwd_weight(patches)");
      else rc = launch_gemm_w<false, false, EPI_RAW, true, 64, 8>(g, pl.nz, stream, "conv2d_cl_bwd_weight(patches)");
      if (rc) return rc;
      FinishArgs f = {};
      f.part = part; f.nz = pl.nz; f.M = ctot; f.N = Kp + 1; f.ldo = Kp + 1; f.epi = EPI_RAW; f.out0 = dw;
      f.ones_col = Kp; f.out_db = db; f.perm_c = d->C; f.perm_taps = taps; f.perm_k = K; f.accumulate = n0 > 0;
      rc = launch_finish(f, stream);
      if (rc) return rc;
    }
    return EVAE_OK;
  }
  const int per = cl_images_per_pass(d, OH, OW, d->C, ldy);
  (void)Mpix;
  GemmArgs g = {};
  ConvMap& cv = g.cv;
  cv.Cg = d->C; cv.creal = d->C; cv.ps = d->C; cv.ntaps = taps;
  cv.RH = OH; cv.RW = OW; cv.IH = d->H; cv.IW = d->W;
  cv.rs = d->stride; cv.rsx = d->stride; cv.roy = -d->pad; cv.rox = -d->pad;
  cv.div_rw = make_fastdiv((unsigned)OW); cv.div_rhw = make_fastdiv((unsigned)(OH * OW));
  for (int kh = 0; kh < d->KH; ++kh)
    for (int kw = 0; kw < d->KW; ++kw) { cv.tdy[kh * d->KW + kw] = (signed char)kh; cv.tdx[kh * d->KW + kw] = (signed char)kw; }
  g.lda[0] = ldy; g.ldb[0] = K + 1; g.npairs = 1;
  g.M = ctot; g.N = K + 1; g.ones_col = K; g.ldo = K + 1;
  g.out0 = (float*)ws;
  for (int n0 = 0; n0 < d->N; n0 += per) {            // passes over at most 2 GiB of images, accumulated by the finish
    const int nn = std::min(per, d->N - n0);
    const int mp = nn * OH * OW;
    Plan pl = make_plan(ctot, K + 1, cdiv(mp, BK), false, true, 1);
    g.A[0] = dy + (size_t)n0 * OH * OW * ldy;
    g.B[0] = x + (size_t)n0 * d->H * d->W * d->C;
    g.Kc[0] = mp;
    g.ksplit = pl.nz > 1 ? pl.ksplit : 0;
    int rc, nz_used = pl.nz;
    // both operands k-major (dy rows x im2col rows): the transposing split-bf16 kernel when its 128 x 128 tiles are well filled
    if (gemm_x6_enabled() && gemm_x6t_ok(g, true) && d->C % 4 == 0 &&
        (gemm_x6_min_rows() == 0 || ((double)mp * ctot * (K + 1) >= 2e9 && gemm_x6t_fill(ctot, K + 1) >= 0.85))) {
      const X6tSplit sp6 = x6t_split(mp, ctot, K + 1);
      g.ksplit = sp6.nz > 1 ? sp6.ksplit : 0;
      nz_used = sp6.nz;
      rc = launch_gemm_x6t<EPI_RAW, 2>(g, sp6.nz, stream, "conv2d_cl_bwd_weight(x6)");
    } else if (pl.bn == 128) rc = launch_gemm_w<false, false, EPI_RAW, true, 128, 8, 2>(g, pl.nz, stream, "conv2d_cl_bwd_weight");
    else rc = launch_gemm_w<false, false, EPI_RAW, true, 64, 8, 2>(g, pl.nz, stream, "conv2d_cl_bwd_weight");
    if (rc) return rc;
    FinishArgs f = {};
    f.part = (const float*)ws; f.nz = nz_used; f.M = ctot; f.N = K + 1; f.ldo = K + 1; f.epi = EPI_RAW; f.out0 = dw;
    f.ones_col = K; f.out_db = db; f.perm_c = d->C; f.perm_taps = taps; f.perm_k = K; f.accumulate = n0 > 0;
    rc = launch_finish(f, stream);
    if (rc) return rc;
  }
  return EVAE_OK;
}
