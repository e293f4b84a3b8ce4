// Dense layers of the encoder/decoder on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact fp32 products/accumulation, 157 TFLOP/s peak).
// Replaces utils/nn.py:29-69 (NonLinear, GatedDense) + torch.nn.Linear and their autograd.
//
// One templated LDS-tiled GEMM serves every call:
//   block tile 128 x BN x 32 (BN = 128 or 64), 256 threads = 4 waves in a 2 x 2 grid, wave tile
//   64 x BN/2 = 2 x (1|2) MFMA tiles of 32 x 32, LDS double-buffered; the next K-slab's global loads are
//   issued before the current slab's MFMAs and land in LDS after them (register prefetch, one barrier
//   per slab).
//   Operands come in two layouts, chosen per call so that global reads are always contiguous float4
//   streams and no transposition pass is ever needed:
//     KC ("k-contiguous", [rows][k]):  x and nn.Linear weights in the forward; fragments are read as
//        ds_read_b128 along k -- lanes 0-31 take k..k+3 and lanes 32-63 take k+4..k+7 of an 8-wide
//        k-group, feeding four consecutive MFMAs; row stride 36 floats (stride/4 odd) makes the
//        16-lane b128 groups conflict-free;
//     RC ("row-contiguous", [k][rows]): dy^T and x in the weight gradient, W in the data gradient;
//        fragments are ds_read_b32 of 32 consecutive floats (conflict-free by construction).
//   Row gathers (the exemplar gather of models/BaseModel.py:247) are folded into the tile loads.
//   Epilogues: bias + activation, the GatedDense gate h*sigmoid(g) (h and g column tiles live in the
//   same wave, so the product is register-local), the gate derivative for the layer below, or raw
//   split-K partials finished by a second kernel (deterministic reduction order).
//   A host-side planner picks BN and the split-K factor per call from a wave-quantisation model
//   (512 resident blocks per launch round), so that thin problems (the 100-row batch path, the
//   [600 x 784] weight gradient over 25 000 rows) still fill the 256 CUs.
//   Workgroup ids are remapped so that each XCD owns a contiguous run of tiles (shared A row-panels
//   stay in one L2).

#include "evae_gemm_x6.h"
#include "evae_gemm_p6.h"
#include "evae_thin.h"
#include "evae_u8_prepare.h"

namespace evae {

__global__ void gated_bwd_input_kernel(const float* __restrict__ dout, const float* __restrict__ gout,
                                       const float* __restrict__ s, int M, int N, int ldo,
                                       float* __restrict__ dh, float* __restrict__ dg, int ldd) {
  const size_t n = (size_t)M * N;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const size_t row = i / N, col = i % N;
    const float d = dout[row * (size_t)ldd + col], ov = gout[i], sv = s[i];
    const size_t o = row * (size_t)ldo + col;
    dh[o] = d * sv;
    dg[o] = d * ov * (1.0f - sv);
  }
}

// the same, four columns per thread (16-byte accesses, 32-bit index arithmetic): N, ldo, ldd multiples of 4, aligned pointers,
// fewer than 2^31 quads -- the channels-last convolutions' gate derivative over millions of pixels was bound by the scalar
// kernel's 64-bit division per element (c3: 7.5 % of the step)
__global__ __launch_bounds__(256) void gated_bwd_input_v4_kernel(const float4* __restrict__ dout, const float4* __restrict__ gout,
                                                                 const float4* __restrict__ s, unsigned nquads, unsigned nq,
                                                                 unsigned ldo4, float4* __restrict__ dh, float4* __restrict__ dg,
                                                                 unsigned ldd4) {
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < nquads; q += stride) {
    const unsigned row = q / nq, c = q - row * nq;
    const float4 d = dout[(size_t)row * ldd4 + c], ov = gout[q], sv = s[q];
    const size_t o = (size_t)row * ldo4 + c;
    dh[o] = make_float4(d.x * sv.x, d.y * sv.y, d.z * sv.z, d.w * sv.w);
    dg[o] = make_float4(d.x * ov.x * (1.0f - sv.x), d.y * ov.y * (1.0f - sv.y), d.z * ov.z * (1.0f - sv.z),
                        d.w * ov.w * (1.0f - sv.w));
  }
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ yp, size_t n,
                               int act, float lo, float hi, float* __restrict__ dpre) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float d = dy[i], v = yp[i];
    float r = d;
    if (act == EVAE_ACT_SIGMOID) r = d * v * (1.0f - v);                 // v = y
    else if (act == EVAE_ACT_HARDTANH) r = (v > lo && v < hi) ? d : 0.f;  // v = pre-activation
    dpre[i] = r;
  }
}

}  // namespace evae

namespace evae {
int g_x6_enabled = -1, g_x6_min_rows = -1;
void gemm_x6_init_policy() {
  const char* e = getenv("EVAE_X6");
  if (g_x6_enabled < 0) g_x6_enabled = (e && atoi(e) == 0) ? 0 : 1;
  const char* r = getenv("EVAE_X6_MIN_ROWS");
  if (g_x6_min_rows < 0) g_x6_min_rows = r ? atoi(r) : 2048;
}
}  // namespace evae

using namespace evae;

// enabled: 0 = fp32 matrix pipe only, 1 = split-bf16 kernel where it applies, < 0 = keep; min_rows < 0 = keep
extern "C" int evae_gemm_x6_configure(int enabled, int min_rows) {
  gemm_x6_init_policy();
  if (enabled >= 0) g_x6_enabled = enabled ? 1 : 0;
  if (min_rows >= 0) g_x6_min_rows = min_rows;
  return EVAE_OK;
}

// largest row count whose hidden-width layers take the one-launch thin kernels (csrc/evae_thin.h); < 0: leave as is.  Returns
// the value in force.  (Tests of the tiled kernels at a few thousand rows set 128.)
extern "C" int evae_thin_configure(int max_rows) {
  (void)thin_max_rows();
  if (max_rows >= 0) g_thin_rows = max_rows;
  return g_thin_rows;
}

// would a launch with M output rows and N output columns (gated: N gated outputs) run on the split-bf16 kernel?
extern "C" int evae_gemm_x6_applies(int M, int N, int gated) {
  return (gemm_x6_enabled() && gemm_x6_fills(M, N, gated != 0)) ? 1 : 0;
}

// ---- forward -------------------------------------------------------------------------------------------
extern "C" size_t evae_dense_fwd_workspace_bytes(int M, int K, int N, int gated) {
  if (M <= 0 || K <= 0 || N <= 0) return 256;
  Plan pl = make_plan(M, N, cdiv(K, BK), gated != 0, false, gated ? 2 : 1);
  if (pl.nz <= 1) return 256;
  return align_up((size_t)pl.nz * (gated ? 2 : 1) * M * N * sizeof(float), 256) + 256;
}

static int gated_dense_fwd_core(const float* x, const int64_t* rows, int M, int K, int ldx,
                                const float* wh, const float* bh, const float* wg, const float* bg,
                                int N, float* out, float* save_h, float* save_s, const P6Sink& tsink, void* ws,
                                size_t ws_bytes, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(M >= 0 && K > 0 && N > 0 && ldx >= K, "gated_dense_fwd: bad sizes M=%d K=%d N=%d ldx=%d", M, K, N, ldx);
  if (M == 0) return EVAE_OK;
  EVAE_REQUIRE(x && wh && wg && out, "gated_dense_fwd: null pointer");
  if (rows == nullptr && thin_ok(M, K, ldx, x, (const void*)((uintptr_t)wh | (uintptr_t)wg))) {
    // batch-sized row counts: one launch, no split-K planes (csrc/evae_thin.h)
    ThinArgs t = {};
    t.A = x; t.lda = ldx; t.W0 = wh; t.W1 = wg; t.M = M; t.N = N; t.K = K; t.b0 = bh; t.b1 = bg;
    t.out0 = out; t.out1 = save_h; t.out2 = save_s; t.ldo = N; t.tsink = tsink;
    return launch_thin<THIN_GATED, false>(t, stream, "gated_dense_fwd(thin)");
  }
  Plan pl = make_plan(M, N, cdiv(K, BK), true, false, 2);
  GemmArgs g = {};
  g.ones_col = -1;
  g.A[0] = x; g.B[0] = wh; g.Bg = wg; g.lda[0] = ldx; g.ldb[0] = K; g.Kc[0] = K; g.npairs = 1;
  g.a_rows = rows; g.M = M; g.N = N; g.bias0 = bh; g.bias1 = bg;
  g.out0 = out; g.out1 = save_h; g.out2 = save_s; g.ldo = N; g.tsink = tsink;
  if (pl.nz <= 1 && gemm_x6_use(g, true)) return launch_gemm_x6<EPI_GATED>(g, 1, stream, "gated_dense_fwd(x6)");
  if (pl.nz <= 1) return launch_gemm<true, true, EPI_GATED>(g, pl, stream, "gated_dense_fwd");
  if (ws == nullptr || ws_bytes < evae_dense_fwd_workspace_bytes(M, K, N, 1)) {
    set_error("gated_dense_fwd: workspace too small (%zu)", ws_bytes);
    return EVAE_EWORKSPACE;
  }
  g.out0 = (float*)ws; g.out1 = g.out2 = nullptr;
  int rc;
  if (gemm_x6_use(g, true)) {
    g.ksplit = pl.ksplit;
    rc = launch_gemm_x6<EPI_RAW_GATED>(g, pl.nz, stream, "gated_dense_fwd(split-K, x6)");
  } else {
    rc = launch_gemm<true, true, EPI_RAW_GATED>(g, pl, stream, "gated_dense_fwd(split-K)");
  }
  if (rc) return rc;
  FinishArgs f = {};
  f.ones_col = -1;
  f.part = (const float*)ws; f.nz = pl.nz; f.M = M; f.N = N; f.ldo = N; f.epi = EPI_GATED;
  f.bias0 = bh; f.bias1 = bg; f.out0 = out; f.out1 = save_h; f.out2 = save_s; f.tsink = tsink;
  return launch_finish(f, stream);
}

extern "C" int evae_gated_dense_fwd(const float* x, const int64_t* rows, int M, int K, int ldx,
                                    const float* wh, const float* bh, const float* wg, const float* bg,
                                    int N, float* out, float* save_h, float* save_s, void* ws,
                                    size_t ws_bytes, evae_stream_t stream_) {
  const P6Sink none = {nullptr, 0, 0, 0, 0};
  return gated_dense_fwd_core(x, rows, M, K, ldx, wh, bh, wg, bg, N, out, save_h, save_s, none, ws, ws_bytes, stream_);
}

// ... whose output also leaves as the pre-split bf16 image of out^T (evae_p6_image.h): output column n = image row t_row0 + n,
// output row m = k index t_kbase + m (a multiple of 8) of an image with t_nks k-steps -- the operand the next layer's forward
// (evae_gated_dense_fwd_p6t) and this layer's successor's weight gradient (evae_dense_bwd_weight_p6) read
extern "C" int evae_gated_dense_fwd_timg(const float* x, const int64_t* rows, int M, int K, int ldx,
                                         const float* wh, const float* bh, const float* wg, const float* bg,
                                         int N, float* out, float* save_h, float* save_s, void* timg, int t_nks, int t_row0,
                                         int t_kbase, void* ws, size_t ws_bytes, evae_stream_t stream_) {
  EVAE_REQUIRE(timg && t_nks > 0 && t_row0 >= 0 && t_kbase >= 0 && (t_kbase % 8) == 0 && (t_kbase + M + 15) / 16 <= t_nks,
               "gated_dense_fwd_timg: bad image placement (k-steps %d, first k %d, rows %d)", t_nks, t_kbase, M);
  const P6Sink sink = {(unsigned char*)timg, t_nks, t_row0, t_kbase, t_kbase + M};
  return gated_dense_fwd_core(x, rows, M, K, ldx, wh, bh, wg, bg, N, out, save_h, save_s, sink, ws, ws_bytes, stream_);
}

extern "C" int evae_linear_fwd(const float* x, const int64_t* rows, int M, int K, int ldx,
                               const float* w, const float* b, int N, int act, float act_lo,
                               float act_hi, float* y, float* pre, void* ws, size_t ws_bytes,
                               evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(M >= 0 && K > 0 && N > 0 && ldx >= K, "linear_fwd: bad sizes M=%d K=%d N=%d ldx=%d", M, K, N, ldx);
  EVAE_REQUIRE(act >= 0 && act <= 2, "linear_fwd: bad activation %d", act);
  if (M == 0) return EVAE_OK;
  EVAE_REQUIRE(x && w && y, "linear_fwd: null pointer");
  if (rows == nullptr && thin_ok(M, K, ldx, x, w)) {
    ThinArgs t = {};
    t.A = x; t.lda = ldx; t.W0 = w; t.M = M; t.N = N; t.K = K; t.b0 = b; t.out0 = y; t.out1 = pre; t.ldo = N;
    t.act = act; t.lo = act_lo; t.hi = act_hi;
    return launch_thin<THIN_LINEAR, false>(t, stream, "linear_fwd(thin)");
  }
  Plan pl = make_plan(M, N, cdiv(K, BK), false, false, 1);
  GemmArgs g = {};
  g.ones_col = -1;
  g.A[0] = x; g.B[0] = w; g.lda[0] = ldx; g.ldb[0] = K; g.Kc[0] = K; g.npairs = 1;
  g.a_rows = rows; g.M = M; g.N = N; g.bias0 = b; g.out0 = y; g.out1 = pre; g.ldo = N;
  g.act = act; g.lo = act_lo; g.hi = act_hi;
  if (pl.nz <= 1 && gemm_x6_use(g) && gemm_x6_pick_bn(M, N) == 64) return launch_gemm_x6<EPI_LINEAR, 0, 64>(g, 1, stream, "linear_fwd(x6)");
  if (pl.nz <= 1 && gemm_x6_use(g)) return launch_gemm_x6<EPI_LINEAR>(g, 1, stream, "linear_fwd(x6)");
  if (pl.nz <= 1) return launch_gemm<true, true, EPI_LINEAR>(g, pl, stream, "linear_fwd");
  if (ws == nullptr || ws_bytes < evae_dense_fwd_workspace_bytes(M, K, N, 0)) {
    set_error("linear_fwd: workspace too small (%zu)", ws_bytes);
    return EVAE_EWORKSPACE;
  }
  g.out0 = (float*)ws; g.out1 = nullptr;
  int rc;
  if (gemm_x6_use(g)) {
    g.ksplit = pl.ksplit;
    rc = launch_gemm_x6<EPI_RAW>(g, pl.nz, stream, "linear_fwd(split-K, x6)");
  } else {
    rc = launch_gemm<true, true, EPI_RAW>(g, pl, stream, "linear_fwd(split-K)");
  }
  if (rc) return rc;
  FinishArgs f = {};
  f.ones_col = -1;
  f.part = (const float*)ws; f.nz = pl.nz; f.M = M; f.N = N; f.ldo = N; f.epi = EPI_LINEAR;
  f.bias0 = b; f.out0 = y; f.out1 = pre; f.act = act; f.lo = act_lo; f.hi = act_hi;
  return launch_finish(f, stream);
}

// ---- the two heads of the encoder + the sample, for thin launches -------------------------------------------------------
// q_z_mean = Linear, q_z_logvar = Hardtanh(Linear) on the same input (models/VAE.py:24-26), then z = mean + eps exp(logvar / 2)
// and log q(z | x) (models/BaseModel.py:79-82, utils/distributions.py:28-33).  As separate calls that is two split-K GEMMs, two
// finish launches and the sampling kernel -- five dependent launches of the batch-row chain.  The two products share their
// input, which is what the gated GEMM computes (bank h = mean weights, bank g = log-variance weights): one split-K launch
// into [z][2][M][N] planes, and ONE finish launch (a wave per row) that sums the planes in fixed order, adds the biases, clamps
// the log-variance, samples and reduces log q.
__global__ __launch_bounds__(256) void heads_reparam_finish_kernel(const float* __restrict__ part, int nz, int M, int Z,
                                                                   const float* __restrict__ bm, const float* __restrict__ bl,
                                                                   float lo, float hi, const float* __restrict__ eps,
                                                                   float* __restrict__ z_mean, float* __restrict__ lv_pre,
                                                                   float* __restrict__ logvar, float* __restrict__ z,
                                                                   float* __restrict__ logq, const float* __restrict__ z_given) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const size_t plane = (size_t)M * Z;
  float acc = 0.f;
  for (int k = lane; k < Z; k += 64) {
    const size_t o = (size_t)row * Z + k;
    float m = 0.f, p = 0.f;
    for (int s = 0; s < nz; ++s) { m += part[(size_t)s * 2 * plane + o]; p += part[(size_t)s * 2 * plane + plane + o]; }
    m += bm ? bm[k] : 0.f;
    p += bl ? bl[k] : 0.f;
    const float lv = fminf(fmaxf(p, lo), hi);
    // z_given: the density of a sample drawn elsewhere (p(z1 | z2) of the 2-level models) instead of a fresh one
    const float zz = z_given ? z_given[o] : eps[o] * expf(0.5f * lv) + m;
    z_mean[o] = m;
    if (lv_pre) lv_pre[o] = p;
    logvar[o] = lv;
    if (!z_given) z[o] = zz;
    const float d = zz - m;
    acc += -0.5f * (lv + kLog2Pi + d * d / expf(lv));
  }
  acc = wave_sum(acc);
  if (lane == 0 && logq) logq[row] = acc;
}

static Plan heads_plan(int M, int K, int Z) { return make_plan(M, Z, cdiv(K, BK), true, true, 2); }

extern "C" size_t evae_heads_reparam_fwd_workspace_bytes(int M, int K, int Z) {
  if (M <= 0 || K <= 0 || Z <= 0) return 256;
  return align_up((size_t)heads_plan(M, K, Z).nz * 2 * M * Z * sizeof(float), 256) + 256;
}

static int heads_reparam_fwd_core(const float* x, int M, int K, int ldx, const float* wm, const float* bm, const float* wl,
                                  const float* bl, int Z, float lv_lo, float lv_hi, const float* eps, float* z_mean,
                                  float* lv_pre, float* logvar, float* z, float* logq, void* ws, size_t ws_bytes,
                                  const float* bc_src, float* bc_dst, int bc_n, evae_stream_t stream_);

extern "C" int evae_heads_reparam_fwd(const float* x, int M, int K, int ldx, const float* wm, const float* bm, const float* wl,
                                      const float* bl, int Z, float lv_lo, float lv_hi, const float* eps, float* z_mean,
                                      float* lv_pre, float* logvar, float* z, float* logq, void* ws, size_t ws_bytes,
                                      evae_stream_t stream_) {
  return heads_reparam_fwd_core(x, M, K, ldx, wm, bm, wl, bl, Z, lv_lo, lv_hi, eps, z_mean, lv_pre, logvar, z, logq, ws, ws_bytes,
                                nullptr, nullptr, 0, stream_);
}

// ... and dst[0 .. n) = src[0] in the same launch when the batch-sized kernel serves (evae_broadcast_scalar's work: the exemplar
// prior's log-variance row, which the step's prior launch reads behind this one); returns EVAE_EINVAL without launching anything
// when it does not (evae_heads_reparam_fwd_bcast_applies says so up front).
extern "C" int evae_heads_reparam_fwd_bcast_applies(int M, int K, int Z, int ldx) {
  return M > 0 && thin_heads_ok(M, K, Z, ldx, nullptr, nullptr, nullptr) ? 1 : 0;
}
extern "C" int evae_heads_reparam_fwd_bcast(const float* x, int M, int K, int ldx, const float* wm, const float* bm, const float* wl,
                                            const float* bl, int Z, float lv_lo, float lv_hi, const float* eps, float* z_mean,
                                            float* lv_pre, float* logvar, float* z, float* logq, const float* src, float* dst,
                                            int n, evae_stream_t stream_) {
  EVAE_REQUIRE(src && dst && n > 0 && M > 0 && thin_heads_ok(M, K, Z, ldx, x, wm, wl), "heads_reparam_fwd_bcast: not the batch-sized case");
  return heads_reparam_fwd_core(x, M, K, ldx, wm, bm, wl, bl, Z, lv_lo, lv_hi, eps, z_mean, lv_pre, logvar, z, logq, nullptr, 0,
                                src, dst, n, stream_);
}

static int heads_reparam_fwd_core(const float* x, int M, int K, int ldx, const float* wm, const float* bm, const float* wl,
                                  const float* bl, int Z, float lv_lo, float lv_hi, const float* eps, float* z_mean,
                                  float* lv_pre, float* logvar, float* z, float* logq, void* ws, size_t ws_bytes,
                                  const float* bc_src, float* bc_dst, int bc_n, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(M >= 0 && K > 0 && Z > 0 && ldx >= K, "heads_reparam_fwd: bad sizes M=%d K=%d Z=%d ldx=%d", M, K, Z, ldx);
  if (M == 0) return EVAE_OK;
  EVAE_REQUIRE(x && wm && wl && eps && z_mean && logvar && z, "heads_reparam_fwd: null pointer");
  if (thin_heads_ok(M, K, Z, ldx, x, wm, wl)) {       // batch-sized: one launch (csrc/evae_thin.h)
    const ThinHeadsArgs t = {x, ldx, M, K, Z, wm, bm, wl, bl, lv_lo, lv_hi, eps, nullptr, z_mean, lv_pre, logvar, z, logq, bc_src, bc_dst, bc_n};
    return launch_thin_heads(t, stream, "heads_reparam_fwd(thin)");
  }
  EVAE_REQUIRE(ws && ws_bytes >= evae_heads_reparam_fwd_workspace_bytes(M, K, Z), "heads_reparam_fwd: workspace too small (%zu)", ws_bytes);
  const Plan pl = heads_plan(M, K, Z);
  GemmArgs g = {};
  g.ones_col = -1;
  g.A[0] = x; g.B[0] = wm; g.Bg = wl; g.lda[0] = ldx; g.ldb[0] = K; g.Kc[0] = K; g.npairs = 1;
  g.M = M; g.N = Z; g.ldo = Z; g.out0 = (float*)ws;
  int rc = launch_gemm<true, true, EPI_RAW_GATED>(g, pl, stream, "heads_reparam_fwd(split-K)");
  if (rc) return rc;
  heads_reparam_finish_kernel<<<cdiv(M, 4), 256, 0, stream>>>((const float*)ws, pl.nz, M, Z, bm, bl, lv_lo, lv_hi, eps, z_mean,
                                                             lv_pre, logvar, z, logq, nullptr);
  return check_launch("heads_reparam_finish_kernel");
}

// The same two heads with the log-density of a GIVEN sample: logp[m] = log N(zq[m] | z_mean[m], exp(logvar[m])) -- p(z1 | z2) of
// the 2-level models (models/AbsHModel.py:17-20,99-100); workspace of evae_heads_reparam_fwd_workspace_bytes.
extern "C" int evae_heads_density_fwd(const float* x, int M, int K, int ldx, const float* wm, const float* bm, const float* wl,
                                      const float* bl, int Z, float lv_lo, float lv_hi, const float* zq, float* z_mean, float* lv_pre,
                                      float* logvar, float* logp, void* ws, size_t ws_bytes, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(M >= 0 && K > 0 && Z > 0 && ldx >= K, "heads_density_fwd: bad sizes M=%d K=%d Z=%d ldx=%d", M, K, Z, ldx);
  if (M == 0) return EVAE_OK;
  EVAE_REQUIRE(x && wm && wl && zq && z_mean && logvar && logp, "heads_density_fwd: null pointer");
  if (thin_heads_ok(M, K, Z, ldx, x, wm, wl)) {
    const ThinHeadsArgs t = {x, ldx, M, K, Z, wm, bm, wl, bl, lv_lo, lv_hi, nullptr, zq, z_mean, lv_pre, logvar, nullptr, logp};
    return launch_thin_heads(t, stream, "heads_density_fwd(thin)");
  }
  EVAE_REQUIRE(ws && ws_bytes >= evae_heads_reparam_fwd_workspace_bytes(M, K, Z), "heads_density_fwd: workspace too small (%zu)", ws_bytes);
  const Plan pl = heads_plan(M, K, Z);
  GemmArgs g = {};
  g.ones_col = -1;
  g.A[0] = x; g.B[0] = wm; g.Bg = wl; g.lda[0] = ldx; g.ldb[0] = K; g.Kc[0] = K; g.npairs = 1;
  g.M = M; g.N = Z; g.ldo = Z; g.out0 = (float*)ws;
  int rc = launch_gemm<true, true, EPI_RAW_GATED>(g, pl, stream, "heads_density_fwd(split-K)");
  if (rc) return rc;
  heads_reparam_finish_kernel<<<cdiv(M, 4), 256, 0, stream>>>((const float*)ws, pl.nz, M, Z, bm, bl, lv_lo, lv_hi, nullptr, z_mean,
                                                             lv_pre, logvar, nullptr, logp, zq);
  return check_launch("heads_density_finish");
}

// ---- data gradient ---------------------------------------------------------------------------------------
// transposed weights for the split-bf16 kernel (contraction-contiguous B): wT[p][k][n] = w_p[n][k], row stride ldt
static int x6_wt_ld(int N) { return (N + 3) / 4 * 4; }
static size_t x6_wt_bytes(int N, int K, int npairs) { return align_up((size_t)npairs * K * x6_wt_ld(N) * sizeof(float), 256); }

__global__ __launch_bounds__(256) void transpose_pairs_kernel(const float* __restrict__ w1, const float* __restrict__ w2, int N,
                                                              int K, int ldt, float* __restrict__ wT) {
  __shared__ float tile[32][33];
  const float* w = blockIdx.z ? w2 : w1;
  float* o = wT + (size_t)blockIdx.z * K * ldt;
  const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8)
    tile[r][tx] = (n0 + r < N && k0 + tx < K) ? w[(size_t)(n0 + r) * K + k0 + tx] : 0.f;
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (k0 + r < K && n0 + tx < ldt) o[(size_t)(k0 + r) * ldt + n0 + tx] = tile[tx][r];
}

extern "C" size_t evae_dense_bwd_data_workspace_bytes(int M, int N, int K, int npairs) {
  if (M <= 0 || K <= 0 || N <= 0) return 256;
  Plan pl = make_plan(M, K, total_slabs(N, npairs > 1 ? N : 0), false, false, 1);
  const size_t wt = x6_wt_bytes(N, K, npairs > 1 ? 2 : 1);      // transposed weights of the split-bf16 path, behind the planes
  if (pl.nz <= 1) return 256 + wt;
  return align_up((size_t)pl.nz * M * K * sizeof(float), 256) + wt + 256;
}

// wT_ext: the transposed weights of the split-bf16 path ([npairs][K][x6_wt_ld(N)], evae_transpose_pairs / the step-head
// launch) when the caller already has them -- they depend on the weights alone; NULL: transposed here, into the workspace.
// img: (dh, dg) leave as the bf16 tile images of evae_dense_bwd_weight_u8 (EPI_GATE_BWD_IMG) instead of fp32 rows
struct ImgSink { unsigned short* img; int nslab, mbase; };

static int dense_bwd_data_core(const float* dy1, const float* w1, const float* dy2, const float* w2,
                               int M, int N, int ldy, int K, const float* out_prev, const float* s_prev,
                               float* dx_or_dh, float* dg, int ldo, const float* wT_ext, void* ws, size_t ws_bytes,
                               evae_stream_t stream_, const ImgSink* sink = nullptr, const P6Sink* tsink = nullptr) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(M >= 0 && N > 0 && K > 0 && ldy >= N && (sink || (tsink && !dx_or_dh) || ldo >= K), "dense_bwd_data: bad sizes M=%d N=%d K=%d", M, N, K);
  if (M == 0) return EVAE_OK;
  EVAE_REQUIRE(dy1 && w1 && (dx_or_dh || sink || tsink), "dense_bwd_data: null pointer");
  EVAE_REQUIRE((dy2 == nullptr) == (w2 == nullptr), "dense_bwd_data: dy2/w2 must come together");
  const bool gate = out_prev != nullptr;
  EVAE_REQUIRE(!gate || (s_prev && (dg || sink || (tsink && !dx_or_dh))), "dense_bwd_data: gate fusion needs out_prev, s_prev and dg");
  EVAE_REQUIRE(!tsink || gate, "dense_bwd_data: the transposed-image output comes with the gate epilogue only");
  const int np = dy2 ? 2 : 1;
  if (thin_ok(M, N, ldy, dy1, dy2) && (!sink || (sink->mbase % 8) == 0)) {
    // batch-sized row counts: one launch (csrc/evae_thin.h), with whichever sinks the caller asked for
    ThinArgs t = {};
    t.A = dy1; t.A1 = dy2; t.lda = ldy; t.W0 = w1; t.W1 = w2; t.M = M; t.N = N; t.K = K;
    t.out0 = dx_or_dh; t.out1 = gate ? dg : nullptr; t.ldo = ldo; t.e0 = out_prev; t.e1 = s_prev;
    if (tsink) t.tsink = *tsink;
    if (sink) { t.u8img = sink->img; t.u8_nslab = sink->nslab; t.u8_mbase = sink->mbase; }
    if (gate) return launch_thin<THIN_GATE_BWD, true>(t, (hipStream_t)stream_, "dense_bwd_data(thin, gate)");
    return launch_thin<THIN_LINEAR, true>(t, (hipStream_t)stream_, "dense_bwd_data(thin)");
  }
  Plan pl = make_plan(M, K, total_slabs(N, np > 1 ? N : 0), false, false, 1);
  if (sink) { pl.nz = 1; pl.ksplit = 0; }          // the image epilogue lives in the GEMM itself: no split-K
  GemmArgs g = {};
  if (sink) { g.img = sink->img; g.img_nslab = sink->nslab; g.img_mbase = sink->mbase; }
  if (tsink) g.tsink = *tsink;
  g.ones_col = -1;
  g.A[0] = dy1; g.B[0] = w1; g.lda[0] = ldy; g.ldb[0] = K; g.Kc[0] = N; g.npairs = np;
  if (dy2) { g.A[1] = dy2; g.B[1] = w2; g.lda[1] = ldy; g.ldb[1] = K; g.Kc[1] = N; }
  g.M = M; g.N = K; g.out0 = dx_or_dh; g.out1 = gate ? dg : nullptr; g.ldo = ldo; g.e0 = out_prev; g.e1 = s_prev;
  const size_t part_bytes = pl.nz > 1 ? align_up((size_t)pl.nz * M * K * sizeof(float), 256) : 0;
  const bool x6 = gemm_x6_enabled() && gemm_x6_fills(M, K, false) && ws && ws_bytes >= part_bytes + x6_wt_bytes(N, K, np) &&
                  N % 4 == 0 && ldy % 4 == 0 && (((uintptr_t)dy1 | (uintptr_t)dy2) & 15) == 0 && (long long)M * ldy < (1ll << 29);
  if (x6) {
    // contraction-contiguous weights: wT[p] = w_p^T behind the split-K planes, then the split-bf16 kernel (A = dy, B = wT)
    const int ldt = x6_wt_ld(N);
    const float* wT = wT_ext;
    if (wT == nullptr) {
      float* wTw = (float*)((char*)ws + part_bytes);
      transpose_pairs_kernel<<<dim3(cdiv(K, 32), cdiv(ldt, 32), np), 256, 0, stream>>>(w1, w2, N, K, ldt, wTw);
      int rc = check_launch("transpose_pairs_kernel");
      if (rc) return rc;
      wT = wTw;
    }
    g.B[0] = wT; g.ldb[0] = ldt;
    if (dy2) { g.B[1] = wT + (size_t)K * ldt; g.ldb[1] = ldt; }
  }
  if (pl.nz <= 1) {
    const bool narrow = x6 && gemm_x6_pick_bn(M, K) == 64;
    if (sink) {
      if (x6 && narrow) return launch_gemm_x6<EPI_GATE_BWD_IMG, 0, 64>(g, 1, stream, "dense_bwd_data(gate, bf16 tile images, x6)");
      if (x6) return launch_gemm_x6<EPI_GATE_BWD_IMG>(g, 1, stream, "dense_bwd_data(gate, bf16 tile images, x6)");
      return launch_gemm<true, false, EPI_GATE_BWD_IMG>(g, pl, stream, "dense_bwd_data(gate, bf16 tile images)");
    }
    if (x6 && gate && narrow) return launch_gemm_x6<EPI_GATE_BWD, 0, 64>(g, 1, stream, "dense_bwd_data(gate, x6)");
    if (x6 && gate) return launch_gemm_x6<EPI_GATE_BWD>(g, 1, stream, "dense_bwd_data(gate, x6)");
    if (x6 && narrow) return launch_gemm_x6<EPI_LINEAR, 0, 64>(g, 1, stream, "dense_bwd_data(x6)");
    if (x6) return launch_gemm_x6<EPI_LINEAR>(g, 1, stream, "dense_bwd_data(x6)");
    if (gate) return launch_gemm<true, false, EPI_GATE_BWD>(g, pl, stream, "dense_bwd_data(gate)");
    return launch_gemm<true, false, EPI_LINEAR>(g, pl, stream, "dense_bwd_data");
  }
  if (ws == nullptr || ws_bytes < part_bytes + 256) {
    set_error("dense_bwd_data: workspace too small (%zu)", ws_bytes);
    return EVAE_EWORKSPACE;
  }
  g.out0 = (float*)ws; g.out1 = nullptr;
  int rc;
  if (x6) {
    g.ksplit = pl.ksplit;
    rc = launch_gemm_x6<EPI_RAW>(g, pl.nz, stream, "dense_bwd_data(split-K, x6)");
  } else {
    rc = launch_gemm<true, false, EPI_RAW>(g, pl, stream, "dense_bwd_data(split-K)");
  }
  if (rc) return rc;
  FinishArgs f = {};
  f.ones_col = -1;
  f.part = (const float*)ws; f.nz = pl.nz; f.M = M; f.N = K; f.ldo = ldo;
  f.epi = gate ? EPI_GATE_BWD : EPI_LINEAR; f.out0 = dx_or_dh; f.out1 = gate ? dg : nullptr;
  f.e0 = out_prev; f.e1 = s_prev;
  if (tsink) f.tsink = *tsink;
  return launch_finish(f, stream);
}

extern "C" int evae_dense_bwd_data(const float* dy1, const float* w1, const float* dy2, const float* w2,
                                   int M, int N, int ldy, int K, const float* out_prev, const float* s_prev,
                                   float* dx_or_dh, float* dg, int ldo, void* ws, size_t ws_bytes,
                                   evae_stream_t stream_) {
  return dense_bwd_data_core(dy1, w1, dy2, w2, M, N, ldy, K, out_prev, s_prev, dx_or_dh, dg, ldo, nullptr, ws, ws_bytes, stream_);
}

extern "C" size_t evae_dense_bwd_data_wt_bytes(int N, int K, int npairs) {
  return (N > 0 && K > 0) ? x6_wt_bytes(N, K, npairs > 1 ? 2 : 1) : 0;
}
extern "C" int evae_dense_bwd_data_wt_ld(int N) { return x6_wt_ld(N); }

extern "C" int evae_dense_bwd_data_wt(const float* dy1, const float* w1, const float* dy2, const float* w2,
                                      int M, int N, int ldy, int K, const float* out_prev, const float* s_prev,
                                      float* dx_or_dh, float* dg, int ldo, const float* wT, void* ws, size_t ws_bytes,
                                      evae_stream_t stream_) {
  return dense_bwd_data_core(dy1, w1, dy2, w2, M, N, ldy, K, out_prev, s_prev, dx_or_dh, dg, ldo, wT, ws, ws_bytes, stream_);
}

// The gate-fused data gradient whose (dh, dg) also -- or, with dx_or_dh == NULL, only -- leave as the pre-split bf16 image of
// [dh | dg]^T (evae_p6_image.h): column n of dh = image row t_row0 + n, of dg = t_row0 + K + n; row m = k index t_kbase + m
extern "C" int evae_dense_bwd_data_timg(const float* dy1, const float* w1, const float* dy2, const float* w2,
                                        int M, int N, int ldy, int K, const float* out_prev, const float* s_prev,
                                        float* dx_or_dh, float* dg, int ldo, const float* wT, void* timg, int t_nks, int t_row0,
                                        int t_kbase, void* ws, size_t ws_bytes, evae_stream_t stream_) {
  EVAE_REQUIRE(timg && t_nks > 0 && t_row0 >= 0 && t_kbase >= 0 && (t_kbase % 8) == 0 && (t_kbase + M + 15) / 16 <= t_nks,
               "dense_bwd_data_timg: bad image placement (k-steps %d, first k %d, rows %d)", t_nks, t_kbase, M);
  const P6Sink sink = {(unsigned char*)timg, t_nks, t_row0, t_kbase, t_kbase + M};
  return dense_bwd_data_core(dy1, w1, dy2, w2, M, N, ldy, K, out_prev, s_prev, dx_or_dh, dg, ldo, wT, ws, ws_bytes, stream_, nullptr,
                             &sink);
}

// The data gradient of the layer ABOVE the byte-store layer: (dh, dg) of that layer are written as the bf16 tile images the
// weight gradient of evae_dense_bwd_weight_u8 reads (see EPI_GATE_BWD_IMG), not as an fp32 [M x 2K] buffer.
extern "C" int evae_dense_bwd_data_img(const float* dy1, const float* w1, const float* dy2, const float* w2, int M, int N,
                                       int ldy, int K, const float* out_prev, const float* s_prev, void* img, int img_nslab,
                                       int m_base, const float* wT, void* ws, size_t ws_bytes, evae_stream_t stream_) {
  EVAE_REQUIRE(M >= 0 && N > 0 && K > 0 && ldy >= N, "dense_bwd_data_img: bad sizes M=%d N=%d K=%d", M, N, K);
  if (M == 0) return EVAE_OK;
  EVAE_REQUIRE(dy1 && w1 && out_prev && s_prev && img, "dense_bwd_data_img: null pointer");
  EVAE_REQUIRE(m_base % 8 == 0 && m_base >= 0 && img_nslab > 0 && (long long)m_base + M <= (long long)img_nslab * 32,
               "dense_bwd_data_img: the launch's first image row must be a multiple of 8 inside the image (M=%d m_base=%d nslab=%d)",
               M, m_base, img_nslab);
  const ImgSink sink = {(unsigned short*)img, img_nslab, m_base};
  return dense_bwd_data_core(dy1, w1, dy2, w2, M, N, ldy, K, out_prev, s_prev, nullptr, nullptr, K, wT, ws, ws_bytes, stream_, &sink);
}


// ---- weight gradient of a NARROW layer (N <= 64 outputs: the encoder's mean / log-variance heads, [40 x 300] over all C + B rows).
// On the GEMM kernel its one row tile is 31 % live and 500 blocks each pay a prologue and a 128 x 64 partial tile for eight
// K-slabs: 25 us + finish for 34 MB of operands.  Here it is what it is -- a reduction over the rows bound by reading them:
// grid (64-column tiles of [x | 1], row slices); a block stages 32 rows of dy (all N columns) and of its x columns in LDS,
// a thread keeps a 4 (outputs) x 4 (columns) tile in registers (its two LDS reads per row are shared by 16 lanes each:
// broadcasts); partial planes [slice][N][K + 1] in the layout of gemm_finish_body (EPI_RAW, ones column = db).
__global__ __launch_bounds__(256) void narrow_wgrad_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x,
                                                           int ldx, int M, int N, int K, int rows_per_slice,
                                                           float* __restrict__ part) {
  __shared__ float4 sdy[2][32][16];
  __shared__ float4 sx[2][32][16];
  const int Kp = K + 1;
  const int k0 = blockIdx.x * 64, z = blockIdx.y;
  const int r_begin = z * rows_per_slice, r_end = min(M, r_begin + rows_per_slice);
  const int tid = threadIdx.x, kg = tid & 15, ng = tid >> 4;
  const bool live = 4 * ng < N;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  // a thread stages rows (tid >> 4) and (tid >> 4) + 16 of a 32-row chunk, float4 (tid & 15) of dy and of its x columns
  const int c4 = tid & 15, rr0 = tid >> 4;
  const bool dy_col = 4 * c4 < N;
  const int kx = k0 + 4 * c4;
  const bool x_vec = kx + 3 < K;
  float4 vd[2], vx[2];
  auto fetch = [&](int r0) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int r = r0 + rr0 + 16 * it;
      vd[it] = make_float4(0.f, 0.f, 0.f, 0.f); vx[it] = vd[it];
      if (r < r_end) {
        if (dy_col) vd[it] = *reinterpret_cast<const float4*>(dy + (size_t)r * ldy + 4 * c4);
        if (x_vec) {
          vx[it] = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + kx);
        } else {                                    // the tile that holds the end of x and the virtual ones column K
          float t[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] = (kx + j < K) ? x[(size_t)r * ldx + kx + j] : (kx + j == K ? 1.f : 0.f);
          vx[it] = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 2; ++it) { sdy[buf][rr0 + 16 * it][c4] = vd[it]; sx[buf][rr0 + 16 * it][c4] = vx[it]; }
  };
  fetch(r_begin);
  stage(0);
  __syncthreads();
  int buf = 0;
  for (int r0 = r_begin; r0 < r_end; r0 += 32, buf ^= 1) {
    const bool more = r0 + 32 < r_end;
    if (more) fetch(r0 + 32);                       // the next chunk's loads fly while this one is multiplied
    if (live) {
#pragma unroll 8
      for (int rr = 0; rr < 32; ++rr) {
        const float4 a = sdy[buf][rr][ng], b = sx[buf][rr][kg];
        acc[0][0] = fmaf(a.x, b.x, acc[0][0]); acc[0][1] = fmaf(a.x, b.y, acc[0][1]); acc[0][2] = fmaf(a.x, b.z, acc[0][2]); acc[0][3] = fmaf(a.x, b.w, acc[0][3]);
        acc[1][0] = fmaf(a.y, b.x, acc[1][0]); acc[1][1] = fmaf(a.y, b.y, acc[1][1]); acc[1][2] = fmaf(a.y, b.z, acc[1][2]); acc[1][3] = fmaf(a.y, b.w, acc[1][3]);
        acc[2][0] = fmaf(a.z, b.x, acc[2][0]); acc[2][1] = fmaf(a.z, b.y, acc[2][1]); acc[2][2] = fmaf(a.z, b.z, acc[2][2]); acc[2][3] = fmaf(a.z, b.w, acc[2][3]);
        acc[3][0] = fmaf(a.w, b.x, acc[3][0]); acc[3][1] = fmaf(a.w, b.y, acc[3][1]); acc[3][2] = fmaf(a.w, b.z, acc[3][2]); acc[3][3] = fmaf(a.w, b.w, acc[3][3]);
      }
    }
    if (more) stage(buf ^ 1);                       // (the other buffer: its last readers passed the barrier below one chunk ago)
    __syncthreads();
  }
  if (!live) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float* row = part + ((size_t)z * N + 4 * ng + i) * Kp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 4 * kg + j;
      if (k < Kp) row[k] = acc[i][j];
    }
  }
}

// r06: the same reduction on the fp32 matrix instruction, each operand row read ONCE.  The VALU form above re-reads dy for every
// 64-column tile, spends 64 x 320 FMAs per row on the vector pipe (10 us chip-wide) and leaves 154 planes for a finish that gathers
// them in 32-byte pieces: 57 + 31 us inside the c2 step, BESIDE layer 2's data gradient, which they stretch from 60 to 98 us.
// Here: one block per row slice, one wave per 64 columns of [x | 1]; a k-step is FOUR rows -- lane l reads the float4 of columns
// 64 w + 4 (l & 15) .. + 3 of row r0 + (l >> 4) (a wave: four rows x 256 contiguous bytes) and one dy element per 16-output tile,
// and v_mfma_f32_16x16x4_f32 (exact fp32, A = dy^T, B = one component of the float4: MFMA j of a tile owns columns 4 n + j) adds
// the products into 16 NT x 4 accumulators.  Planes [slice][N][K + 1] as before; summed by narrow_finish_kernel below.
template <int NT>
__global__ __launch_bounds__(1024) void narrow_wgrad_mfma_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x,
                                                                 int ldx, int M, int N, int K, int rows_per_slice,
                                                                 float* __restrict__ part) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int U = 4;                                        // k-steps (of four rows) whose loads are in flight together
  const int Kp = K + 1;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, kr = l >> 4, cq = l & 15;
  const int z = blockIdx.x;
  const int r_begin = z * rows_per_slice, r_end = min(M, r_begin + rows_per_slice);
  const int c = w * 64 + 4 * cq;
  const bool x_vec = c + 3 < K;
  f32x4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int r0 = r_begin; r0 < r_end; r0 += 4 * U) {
    float4 xv[U];
    float a[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + 4 * u + kr;
      const bool live = r < r_end;
      xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) {
        if (x_vec) {
          xv[u] = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + c);
        } else if (c <= K) {                                  // the piece that holds the end of x and the virtual ones column K
          float t4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) t4[j] = (c + j < K) ? x[(size_t)r * ldx + c + j] : (c + j == K ? 1.f : 0.f);
          xv[u] = make_float4(t4[0], t4[1], t4[2], t4[3]);
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) a[u][t] = (live && 16 * t + cq < N) ? dy[(size_t)r * ldy + 16 * t + cq] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], xv[u].x, acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], xv[u].y, acc[t][1], 0, 0, 0);
        acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], xv[u].z, acc[t][2], 0, 0, 0);
        acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], xv[u].w, acc[t][3], 0, 0, 0);
      }
  }
  // D of MFMA j, tile t: lane l holds outputs 16 t + 4 (l >> 4) + reg of column 64 w + 4 (l & 15) + j
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = 16 * t + 4 * kr + reg;
      if (i < N) {
        float* row = part + ((size_t)z * N + i) * Kp + c;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < Kp) row[j] = acc[t][j][reg];
      }
    }
}

// the planes of the kernel above -> dw [N x K], db [N] (column K): 64 consecutive outputs per block, wave g of 16 sums planes
// g, g + 16, ... (a wave reads 256 contiguous bytes of a plane), the 16 sums meet in LDS and are added in a fixed order
__global__ __launch_bounds__(1024) void narrow_finish_kernel(const float* __restrict__ part, int nz, int N, int K, float* __restrict__ dw,
                                                             float* __restrict__ db, int accumulate) {
  __shared__ float red[16][64];
  const int Kp = K + 1;
  const size_t plane = (size_t)N * Kp;
  const int g = threadIdx.x >> 6, l = threadIdx.x & 63;
  const size_t i = (size_t)blockIdx.x * 64 + l;
  float v = 0.f;
  if (i < plane)
    for (int z = g; z < nz; z += 16) v += part[(size_t)z * plane + i];
  red[g][l] = v;
  __syncthreads();
  if (g != 0 || i >= plane) return;
  float s = red[0][l];
#pragma unroll
  for (int q = 1; q < 16; ++q) s += red[q][l];
  const int m = (int)(i / Kp), n = (int)(i - (size_t)m * Kp);
  if (n == K) { if (db) db[m] = (accumulate ? db[m] : 0.f) + s; }
  else { const size_t o = (size_t)m * K + n; dw[o] = (accumulate ? dw[o] : 0.f) + s; }
}

static int narrow_wgrad_form() {       // EVAE_WGRAD_NARROW: 0 = the GEMM kernel, 1 = the VALU reduction (r03), 2 (default) = fp32 MFMA (r06)
  static int on = -1;
  if (on < 0) { const char* e = getenv("EVAE_WGRAD_NARROW"); on = e ? atoi(e) : 2; }
  return on;
}
static bool narrow_mfma_shape(int N, int K) { return narrow_wgrad_form() >= 2 && N <= 64 && K + 1 <= 1024; }
// rows of a slice of the MFMA form: whole 16-row groups of loads, at most 256 slices
static int narrow_mfma_rows(int M) { return std::max(64, (int)align_up((size_t)cdiv(M, 256), 16)); }

// row slices of the narrow kernels.  VALU form: ~768 blocks over the column tiles, at least 128 rows (four chunks) a slice
static int narrow_wgrad_slices(int M, int K) {
  const int tiles = cdiv(K + 1, 64);
  return std::max(1, std::min(cdiv(768, tiles), cdiv(M, 128)));
}
static int narrow_planes(int M, int N, int K) {
  return narrow_mfma_shape(N, K) ? cdiv(M, narrow_mfma_rows(M)) : cdiv(M, cdiv(M, narrow_wgrad_slices(M, K)));
}
static bool narrow_wgrad_enabled() { return narrow_wgrad_form() != 0; }
static bool narrow_wgrad_shape(int M, int N, int K) { return narrow_wgrad_enabled() && N <= 64 && N % 4 == 0 && M >= 2048 && K >= 64; }

// ---- weight gradient -------------------------------------------------------------------------------------
// The bias gradient db = column sums of dy is folded into the same GEMM: x gets a virtual all-ones
// column K (never read from memory), so column K of dy^T [x | 1] is db.
// partial planes the workspace must hold: whichever of the three split-K plans the launch ends up taking
static int wgrad_max_planes(int M, int N, int K) {
  const Plan pl = make_plan(N, K + 1, cdiv(M, BK), false, true, 1);
  const Plan pll = make_plan_local(N, K + 1, cdiv(M, BK));
  int nz = std::max(std::max(pl.nz, pll.nz), x6t_split(M, N, K + 1).nz);
  if (narrow_wgrad_shape(M, N, K)) nz = std::max(nz, std::max(narrow_wgrad_slices(M, K), narrow_planes(M, N, K)));
  return nz;
}

extern "C" size_t evae_dense_bwd_weight_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 256;
  return align_up((size_t)wgrad_max_planes(M, N, K) * N * (K + 1) * sizeof(float), 256) + 256;
}

// phase 0: both launches; 1: the split-K GEMM into the workspace partials; 2: the finish (sum of the partial planes in a fixed
// order -> dw, db) -- so that a caller can put the finish on another stream than the GEMM that follows
static bool thin_wgrad_direct() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EVAE_WGRAD_DIRECT"); on = e ? atoi(e) : 1; }
  return on != 0;
}

static int dense_bwd_weight_core(const float* dy, int M, int N, int ldy, const float* x,
                                 const int64_t* rows, int K, int ldx, float* dw, float* db,
                                 int accumulate, void* ws, size_t ws_bytes, int phase, hipStream_t stream,
                                 FinishArgs* finish_out = nullptr) {
  EVAE_REQUIRE(M >= 0 && N > 0 && K > 0 && ldx >= K && ldy >= N, "dense_bwd_weight: bad sizes M=%d N=%d K=%d", M, N, K);
  EVAE_REQUIRE(dw != nullptr, "dense_bwd_weight: null dw");
  if (ws == nullptr || ws_bytes < evae_dense_bwd_weight_workspace_bytes(M, N, K)) {
    set_error("dense_bwd_weight: workspace too small (%zu)", ws_bytes);
    return EVAE_EWORKSPACE;
  }
  if (M == 0) {
    if (phase == 1) return EVAE_OK;
    if (!accumulate) {
      (void)hipMemsetAsync(dw, 0, (size_t)N * K * sizeof(float), stream);
      if (db) (void)hipMemsetAsync(db, 0, (size_t)N * sizeof(float), stream);
    }
    return check_launch("dense_bwd_weight(empty)");
  }
  EVAE_REQUIRE(dy && x, "dense_bwd_weight: null pointer");
  const int Kp = K + 1;
  Plan pl = make_plan(N, Kp, cdiv(M, BK), false, true, 1);
  float* part = (float*)ws;
  GemmArgs g = {};
  g.A[0] = dy; g.B[0] = x; g.lda[0] = ldy; g.ldb[0] = ldx; g.Kc[0] = M; g.npairs = 1;
  g.b_krows = rows; g.M = N; g.N = Kp; g.out0 = part; g.ldo = Kp; g.ones_col = K;
  // a long contraction (the exemplar rows): slices laid out XCD by XCD, one round of blocks (see gemm_kernel, sk_local)
  if (sk_local_enabled() && cdiv(M, BK) >= 256) {
    const Plan pll = make_plan_local(N, Kp, cdiv(M, BK));
    if (pll.nz >= 2 && pll.nz <= wgrad_max_planes(M, N, K)) { pl = pll; g.sk_local = pll.nz; }
  }
  // both operands k-major: the transposing variant of the split-bf16 kernel when the product is large AND its 128-wide column
  // tiles are well filled -- measured: K = 784 (785 of 896 columns) 240 vs 260 us, K = 300 (301 of 384) 116 vs 119 us: at 78 %
  // fill the fp32 kernel's 64-wide tiles are as fast
  const bool x6 = gemm_x6_enabled() && gemm_x6t_ok(g) &&
                  (gemm_x6_min_rows() == 0 || ((double)M * N * Kp >= 2e9 && gemm_x6t_fill(128, Kp) >= 0.85));
  const X6tSplit sp6 = x6t_split(M, N, Kp);
  if (phase == 0 && !accumulate && !x6 && cdiv(M, BK) <= 4 && thin_wgrad_direct()) {
    // a contraction of at most four K-slabs (the batch rows' leaf layers): nothing to split -- the GEMM writes dw / db itself
    Plan p1 = make_plan(N, Kp, cdiv(M, BK), false, false, 1);
    p1.nz = 1; p1.ksplit = cdiv(M, BK);
    g.out0 = dw; g.ldo = K; g.out1 = db; g.direct = 1;
    return launch_gemm<false, false, EPI_RAW>(g, p1, stream, "dense_bwd_weight(direct)");
  }
  // a narrow output over many rows: the streaming reduction above instead of a GEMM with one mostly empty row tile
  const bool aligned = ((((uintptr_t)dy | (uintptr_t)x) & 15) == 0);
  // (a phased call's second half -- and the grouped finish, which re-derives the plan from the workspace pointer -- must arrive
  // at the SAME plane count and layout as the first: the pointer-alignment term may not differ between them, so a phased
  // narrow-shaped call with unaligned operands is refused instead of summed wrongly later)
  EVAE_REQUIRE(phase == 0 || x6 || rows != nullptr || !narrow_wgrad_shape(M, N, K) || ldy % 4 || ldx % 4 || aligned,
               "dense_bwd_weight_phased: operands of a narrow weight gradient must be 16-byte aligned");
  const bool narrow = !x6 && rows == nullptr && narrow_wgrad_shape(M, N, K) && ldy % 4 == 0 && ldx % 4 == 0 && aligned;
  const int nslice = narrow ? narrow_wgrad_slices(M, K) : 0;
  const bool narrow_mfma = narrow && narrow_mfma_shape(N, K);
  if (narrow && phase != 2) {
    if (narrow_mfma) {
      const int rps = narrow_mfma_rows(M), nb = cdiv(M, rps), nt = 64 * cdiv(Kp, 64);
      switch (cdiv(N, 16)) {
        case 1: narrow_wgrad_mfma_kernel<1><<<nb, nt, 0, stream>>>(dy, ldy, x, ldx, M, N, K, rps, part); break;
        case 2: narrow_wgrad_mfma_kernel<2><<<nb, nt, 0, stream>>>(dy, ldy, x, ldx, M, N, K, rps, part); break;
        case 3: narrow_wgrad_mfma_kernel<3><<<nb, nt, 0, stream>>>(dy, ldy, x, ldx, M, N, K, rps, part); break;
        default: narrow_wgrad_mfma_kernel<4><<<nb, nt, 0, stream>>>(dy, ldy, x, ldx, M, N, K, rps, part); break;
      }
    } else {
      const int rps = cdiv(M, nslice);
      narrow_wgrad_kernel<<<dim3(cdiv(Kp, 64), cdiv(M, rps)), 256, 0, stream>>>(dy, ldy, x, ldx, M, N, K, rps, part);
    }
    const int rc = check_launch("narrow_wgrad_kernel");
    if (rc || phase == 1) return rc;
  }
  if (phase != 2 && !narrow) {
    int rc;
    if (x6) {
      g.ksplit = sp6.nz > 1 ? sp6.ksplit : 0;
      rc = launch_gemm_x6t<EPI_RAW>(g, sp6.nz, stream, "dense_bwd_weight(x6)");
    } else {
      rc = launch_gemm<false, false, EPI_RAW>(g, pl, stream, "dense_bwd_weight");
    }
    if (rc) return rc;
    if (phase == 1) return EVAE_OK;
  }
  FinishArgs f = {};
  f.part = part; f.nz = narrow ? narrow_planes(M, N, K) : (x6 ? sp6.nz : pl.nz); f.M = N; f.N = Kp; f.ldo = Kp; f.epi = EPI_RAW; f.out0 = dw; f.accumulate = accumulate;
  f.ones_col = K; f.out_db = db;
  if (finish_out) { *finish_out = f; return EVAE_OK; }      // (the caller only wants to know what the finish would be)
  if (narrow_mfma) {
    narrow_finish_kernel<<<cdiv(N * Kp, 64), 1024, 0, stream>>>(part, f.nz, N, K, dw, db, accumulate);
    return check_launch("narrow_finish_kernel");
  }
  return launch_finish(f, stream);
}

extern "C" int evae_dense_bwd_weight(const float* dy, int M, int N, int ldy, const float* x,
                                     const int64_t* rows, int K, int ldx, float* dw, float* db,
                                     int accumulate, void* ws, size_t ws_bytes, evae_stream_t stream_) {
  return dense_bwd_weight_core(dy, M, N, ldy, x, rows, K, ldx, dw, db, accumulate, ws, ws_bytes, 0, (hipStream_t)stream_);
}

extern "C" int evae_dense_bwd_weight_phased(const float* dy, int M, int N, int ldy, const float* x,
                                            const int64_t* rows, int K, int ldx, float* dw, float* db,
                                            int accumulate, void* ws, size_t ws_bytes, int phase, evae_stream_t stream_) {
  EVAE_REQUIRE(phase == 1 || phase == 2, "dense_bwd_weight_phased: phase must be 1 or 2");
  return dense_bwd_weight_core(dy, M, N, ldy, x, rows, K, ldx, dw, db, accumulate, ws, ws_bytes, phase, (hipStream_t)stream_);
}

// ---- ONE finish launch for the weight gradients of a training step: the byte layer's (u8_wgrad_finish_body) and up to three
// fp32 layers' (gemm_finish_body), whose GEMMs were issued without their own finish (evae_dense_bwd_weight_phased phase 1,
// evae_dense_bwd_weight_u8_phased phase | 16).  Blocks of 256 threads; a block belongs to one job.
struct FinishGroup { U8FinishArgs u8; int u8_tx, u8_blocks; FinishArgs f[3]; int fstart[4]; int flanes[3]; int nf; };

__global__ __launch_bounds__(256) void wgrad_finish_group_kernel(const FinishGroup grp) {
  __shared__ float tile[32][33];
  const int b = blockIdx.x;
  if (b < grp.u8_blocks) { u8_wgrad_finish_body(grp.u8, b % grp.u8_tx, b / grp.u8_tx, tile); return; }
  const int fb = b - grp.u8_blocks;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j < grp.nf && fb >= grp.fstart[j] && fb < grp.fstart[j + 1]) {
      const size_t t = (size_t)(fb - grp.fstart[j]) * 256 + threadIdx.x;
      if (grp.flanes[j] == 8) gemm_finish_body<8>(grp.f[j], t);
      else gemm_finish_body<1>(grp.f[j], t);
      return;
    }
  }
}

extern "C" int evae_dense_bwd_weight_finish_group(const evae_wgrad_finish_job_t* jobs, int njobs, evae_stream_t stream_) {
  EVAE_REQUIRE(jobs && njobs >= 1 && njobs <= 4, "dense_bwd_weight_finish_group: 1 .. 4 jobs");
  FinishGroup grp = {};
  for (int j = 0; j < njobs; ++j) {
    const evae_wgrad_finish_job_t& w = jobs[j];
    if (w.byte_rows) {
      EVAE_REQUIRE(grp.u8_blocks == 0, "dense_bwd_weight_finish_group: one byte-layer job at most");
      int tx = 0, ty = 0;
      int rc = u8_wgrad_finish_job(w.M, w.N, w.K, w.x_scale, w.dw, w.db, w.ws, w.ws_bytes, &grp.u8, &tx, &ty);
      if (rc) return rc;
      grp.u8_tx = tx; grp.u8_blocks = tx * ty;
    } else {
      EVAE_REQUIRE(grp.nf < 3, "dense_bwd_weight_finish_group: three fp32 jobs at most");
      EVAE_REQUIRE(w.M > 0 && w.N > 0 && w.K > 0 && w.dw && w.ws, "dense_bwd_weight_finish_group: bad job %d", j);
      FinishArgs f = {};
      // phase 2 of the same call, stopped before its launch: the plan (and with it the plane count) is a function of the sizes
      int rc = dense_bwd_weight_core((const float*)w.ws, w.M, w.N, w.ldy, (const float*)w.ws, nullptr, w.K, w.ldx, w.dw, w.db, 0,
                                     w.ws, w.ws_bytes, 2, (hipStream_t)stream_, &f);
      if (rc) return rc;
      grp.f[grp.nf] = f; grp.flanes[grp.nf] = finish_lanes(f);
      grp.fstart[grp.nf + 1] = grp.fstart[grp.nf] + (int)finish_blocks(f);
      ++grp.nf;
    }
  }
  for (int j = grp.nf + 1; j < 4; ++j) grp.fstart[j] = grp.fstart[grp.nf];
  const int total = grp.u8_blocks + grp.fstart[grp.nf];
  wgrad_finish_group_kernel<<<total, 256, 0, (hipStream_t)stream_>>>(grp);
  return check_launch("wgrad_finish_group_kernel");
}

// Several thin weight gradients in one launch (gemm_group_wgrad_kernel): each job as evae_dense_bwd_weight with a
// contraction of at most four K-slabs (M <= 128 rows), no row gather; job.accumulate: dw += dy^T x, db += column sums.  Returns EVAE_EINVAL (nothing
// launched) when a job does not qualify: the caller then issues them one by one.
extern "C" int evae_dense_bwd_weight_group(const evae_wgrad_job_t* jobs, int njobs, evae_stream_t stream_) {
  EVAE_REQUIRE(jobs && njobs >= 1 && njobs <= kWgradGroupMax, "dense_bwd_weight_group: 1 .. %d jobs", kWgradGroupMax);
  WgradGroup grp = {};
  int total = 0;
  for (int j = 0; j < njobs; ++j) {
    const evae_wgrad_job_t& w = jobs[j];
    EVAE_REQUIRE(w.dy && w.x && w.dw && w.M > 0 && w.N > 0 && w.K > 0 && w.ldy >= w.N && w.ldx >= w.K,
                 "dense_bwd_weight_group: bad job %d", j);
    EVAE_REQUIRE(cdiv(w.M, BK) <= 4, "dense_bwd_weight_group: job %d contracts over %d rows (> 128)", j, w.M);
    GemmArgs g = {};
    g.A[0] = w.dy; g.B[0] = w.x; g.lda[0] = w.ldy; g.ldb[0] = w.ldx; g.Kc[0] = w.M; g.npairs = 1;
    g.M = w.N; g.N = w.K + 1; g.ones_col = w.K;
    EVAE_REQUIRE((gemm_vec_ok<false, false>(g)), "dense_bwd_weight_group: job %d is not 16-byte aligned / a multiple of 4", j);
    grp.job[j] = {w.dy, w.x, w.dw, w.db, w.M, w.N, w.K, w.ldy, w.ldx, w.accumulate ? 1 : 0};
    grp.start[j] = total;
    total += cdiv(w.N, BM) * cdiv(w.K + 1, 64);
  }
  grp.start[njobs] = total; grp.n = njobs;
  for (int j = njobs + 1; j <= kWgradGroupMax; ++j) grp.start[j] = total;
  static bool attr = false;
  constexpr size_t lds = gemm_lds_bytes(64);
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_group_wgrad_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  gemm_group_wgrad_kernel<8><<<total, 512, lds, (hipStream_t)stream_>>>(grp);
  return check_launch("gemm_group_wgrad_kernel");
}

static int launch_gated_bwd_input(const float* dout, int ldd, const float* out, const float* s, int M, int N, float* dh, float* dg,
                                  int ldo, hipStream_t stream) {
  const size_t quads = (size_t)M * (N / 4);
  const bool v4 = N % 4 == 0 && ldo % 4 == 0 && ldd % 4 == 0 && quads < ((size_t)1 << 31) &&
                  ((((uintptr_t)dout | (uintptr_t)out | (uintptr_t)s | (uintptr_t)dh | (uintptr_t)dg) & 15) == 0);
  if (v4) {
    gated_bwd_input_v4_kernel<<<elt_grid(quads), 256, 0, stream>>>((const float4*)dout, (const float4*)out, (const float4*)s,
                                                                  (unsigned)quads, (unsigned)(N / 4), (unsigned)(ldo / 4), (float4*)dh,
                                                                  (float4*)dg, (unsigned)(ldd / 4));
  } else {
    gated_bwd_input_kernel<<<elt_grid((size_t)M * N), 256, 0, stream>>>(dout, out, s, M, N, ldo, dh, dg, ldd);
  }
  return check_launch("gated_dense_bwd_input");
}

extern "C" int evae_gated_dense_bwd_input(const float* dout, const float* out, const float* s, int M, int N,
                                          float* dh, float* dg, int ldo, evae_stream_t stream_) {
  if (M <= 0 || N <= 0) return EVAE_OK;
  EVAE_REQUIRE(dout && out && s && dh && dg && ldo >= N, "gated_dense_bwd_input: bad arguments");
  return launch_gated_bwd_input(dout, N, out, s, M, N, dh, dg, ldo, (hipStream_t)stream_);
}

extern "C" int evae_gated_dense_bwd_input_ld(const float* dout, int ldd, const float* out, const float* s, int M, int N,
                                             float* dh, float* dg, int ldo, evae_stream_t stream_) {
  if (M <= 0 || N <= 0) return EVAE_OK;
  EVAE_REQUIRE(dout && out && s && dh && dg && ldo >= N && ldd >= N, "gated_dense_bwd_input_ld: bad arguments");
  return launch_gated_bwd_input(dout, ldd, out, s, M, N, dh, dg, ldo, (hipStream_t)stream_);
}

// A gated layer's whole backward with respect to its input: [dh | dg] (row stride ldp, for the weight gradient) and
// dx = dh Wh + dg Wg.  Batch-sized row counts take ONE launch (thin_gated_bwd_kernel: the gate derivative in the operand load);
// anything else the element-wise launch + evae_dense_bwd_data, with the same bits.  EVAE_THIN_GATE_BWD=0: always two launches.
extern "C" int evae_gated_dense_bwd(const float* dout, int ldd, const float* out, const float* s, int M, int N, const float* wh,
                                    const float* wg, int K, float* dpre, int ldp, float* dx, int ldo, void* ws, size_t ws_bytes,
                                    evae_stream_t stream_) {
  if (M <= 0) return EVAE_OK;
  EVAE_REQUIRE(dout && out && s && wh && wg && dpre && dx && N > 0 && K > 0 && ldd >= N && ldp >= 2 * N && ldo >= K,
               "gated_dense_bwd: bad arguments (M=%d N=%d K=%d)", M, N, K);
  static int fused = -1;
  if (fused < 0) { const char* e = getenv("EVAE_THIN_GATE_BWD"); fused = (e && atoi(e) == 0) ? 0 : 1; }
  const uintptr_t al = (uintptr_t)dout | (uintptr_t)out | (uintptr_t)s | (uintptr_t)dpre;
  if (fused && thin_ok(M, N, ldd, (const void*)al, nullptr) && N % 4 == 0 && ldp % 4 == 0) {
    ThinGateBwdArgs t = {dout, ldd, out, s, wh, wg, M, N, K, dpre, ldp, dx, ldo};
    thin_gated_bwd_kernel<<<dim3(cdiv(K, 16), cdiv(M, 16)), 256, 0, (hipStream_t)stream_>>>(t);
    return check_launch("gated_dense_bwd(thin)");
  }
  int rc = launch_gated_bwd_input(dout, ldd, out, s, M, N, dpre, dpre + N, ldp, (hipStream_t)stream_);
  if (rc != EVAE_OK) return rc;
  return dense_bwd_data_core(dpre, wh, dpre + N, wg, M, N, ldp, K, nullptr, nullptr, dx, nullptr, ldo, nullptr, ws, ws_bytes, stream_);
}

// dst[r][:] = scale[r] * src[idx[r]][:] (scale NULL: 1): the captured step's bridge between the distinct exemplar rows it encodes
// and the draws the prior sees (centres of all draws from the distinct rows' encodings; a distinct row's head gradient from one
// of its draws x its multiplicity -- evae/fused_vae.py::DEDUP).  z % 4 == 0, 16-byte aligned rows.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float4* __restrict__ src, const long long* __restrict__ idx,
                                                          const float* __restrict__ scale, unsigned n, unsigned z4,
                                                          float4* __restrict__ dst) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * z4) return;
  const unsigned r = i / z4, c = i - r * z4;
  float4 v = src[(size_t)idx[r] * z4 + c];
  if (scale) { const float w = scale[r]; v.x *= w; v.y *= w; v.z *= w; v.w *= w; }
  dst[i] = v;
}
extern "C" int evae_gather_rows(const float* src, const int64_t* idx, const float* scale, int n, int z, float* dst,
                                evae_stream_t stream_) {
  if (n <= 0) return EVAE_OK;
  EVAE_REQUIRE(src && idx && dst && z > 0 && z % 4 == 0 && ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) &&
               (long long)n * (z / 4) < (1ll << 31), "gather_rows: bad arguments (n=%d z=%d)", n, z);
  const unsigned total = (unsigned)n * (unsigned)(z / 4);
  gather_rows_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream_>>>((const float4*)src, (const long long*)idx, scale, (unsigned)n,
                                                                          (unsigned)(z / 4), (float4*)dst);
  return check_launch("gather_rows");
}

extern "C" int evae_act_bwd(const float* dy, const float* y_or_pre, size_t n, int act, float act_lo,
                            float act_hi, float* dpre, evae_stream_t stream_) {
  if (n == 0) return EVAE_OK;
  EVAE_REQUIRE(dy && dpre && (act == EVAE_ACT_NONE || y_or_pre), "act_bwd: null pointer");
  EVAE_REQUIRE(act >= 0 && act <= 2, "act_bwd: bad activation %d", act);
  act_bwd_kernel<<<elt_grid(n), 256, 0, (hipStream_t)stream_>>>(dy, y_or_pre ? y_or_pre : dy, n, act, act_lo, act_hi, dpre);
  return check_launch("act_bwd");
}


// ---- GEMMs over pre-split bf16 operand images (evae_gemm_p6.h, evae_p6_image.h) ---------------------------------------------
// The exemplar rows' chain of a training step: every tensor that is the operand of a big GEMM leaves its producer's epilogue
// as ONE image (of its transpose: evae_gated_dense_fwd_timg / _u8_timg, evae_dense_bwd_data_timg), the weights are split once
// per step (evae_p6_pack_rows / _cols), and the three GEMMs of a hidden layer -- forward, data gradient, weight gradient -- run
// on the bf16 matrix pipe with no splitting in their main loops.
extern "C" int evae_p6_nks(int K) { return p6_nks(K); }
extern "C" int evae_p6_nks_rows(int M) { return p6_nks_rows(M); }
extern "C" size_t evae_p6_image_bytes(int rows, int nks) {
  if (rows <= 0 || nks <= 0) return 256;
  return align_up(p6_image_bytes(rows, nks), 256);
}

// X [R x K] (row stride ld) -> image(rows = R, k = K); gated != 0: x / x2 = the h / g banks of a gated layer, image rows in the
// order of the forward kernel's [h | g] column pairs (64 outputs = 128 image rows)
extern "C" int evae_p6_pack_rows(const float* x, const float* x2, int R, int K, long long ld, int gated, void* img,
                                 size_t img_bytes, evae_stream_t stream_) {
  EVAE_REQUIRE(x && img && R > 0 && K > 0 && ld >= K && (!gated || x2), "p6_pack_rows: bad arguments");
  const int rows_img = gated ? cdiv(R, 64) * 128 : cdiv(R, 128) * 128, nks = p6_nks(K);
  EVAE_REQUIRE(img_bytes >= p6_image_bytes(rows_img, nks), "p6_pack_rows: image buffer too small (%zu)", img_bytes);
  const size_t total = (size_t)rows_img * nks * 2;
  p6_pack_rows_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream_>>>(x, x2, R, K, ld, gated, rows_img, nks,
                                                                                        (unsigned char*)img);
  return check_launch("p6_pack_rows_kernel");
}

// X^T: X [Kd x R] with the contraction along its ROWS (rows Kd .. 2 Kd - 1 from x2 when given) -> image(rows = R, k) with nks
// k-steps (>= the k-steps of the contraction; p6_nks_rows(M) for a tensor with M batch rows); ones_row >= 0: that image row = 1
extern "C" int evae_p6_pack_cols(const float* x, const float* x2, int Kd, int R, long long ld, int ones_row, int nks, void* img,
                                 size_t img_bytes, evae_stream_t stream_) {
  EVAE_REQUIRE(x && img && R > 0 && Kd > 0 && ld >= R && nks >= p6_nks(x2 ? 2 * Kd : Kd), "p6_pack_cols: bad arguments");
  const int rows_img = cdiv(std::max(R, ones_row + 1), 128) * 128;
  EVAE_REQUIRE(img_bytes >= p6_image_bytes(rows_img, nks), "p6_pack_cols: image buffer too small (%zu)", img_bytes);
  const size_t total = (size_t)rows_img * nks * 2;
  p6_pack_cols_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream_>>>(x, x2, Kd, R, ld, ones_row, rows_img, nks,
                                                                                        (unsigned char*)img);
  return check_launch("p6_pack_cols_kernel");
}

// image row `row` := value for k in [k_begin, k_end) (the all-ones row that carries a bias gradient through a weight gradient)
__global__ void p6_fill_row_kernel(unsigned char* img, int nks, int row, float value, int k_begin, int k_end) {
  const int k = k_begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= k_end) return;
  unsigned short t0, t1, t2;
  p6_split1(value, t0, t1, t2);
  unsigned short* o = reinterpret_cast<unsigned short*>(img + p6_off(row, k, nks));
  o[0] = t0; o[P6_CHUNK / 2] = t1; o[P6_CHUNK] = t2;
}
extern "C" int evae_p6_fill_row(void* img, int nks, int row, float value, int k_begin, int k_end, evae_stream_t stream_) {
  EVAE_REQUIRE(img && nks > 0 && row >= 0 && k_begin >= 0 && k_end <= nks * P6_KS, "p6_fill_row: bad arguments");
  if (k_end <= k_begin) return EVAE_OK;
  p6_fill_row_kernel<<<cdiv(k_end - k_begin, 256), 256, 0, (hipStream_t)stream_>>>((unsigned char*)img, nks, row, value, k_begin, k_end);
  return check_launch("p6_fill_row_kernel");
}

// does a GEMM with M output rows pay for the image path? (the same machine-filling rule as the split-bf16 kernel)
extern "C" int evae_gemm_p6_applies(int M, int N, int gated) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EVAE_P6"); on = (e && atoi(e) == 0) ? 0 : 1; }
  return (on && gemm_x6_enabled() && gemm_x6_fills(M, N, gated != 0)) ? 1 : 0;
}

// out = (x Wh^T + bh) * sigmoid(x Wg^T + bg) with x given as x^T's image (x_nks k-steps along its M rows) and the weights as
// the gated image of evae_p6_pack_rows (K = columns of x)
extern "C" int evae_gated_dense_fwd_p6t(const void* xT_img, int x_nks, int M, int K, const void* w_img, const float* bh,
                                        const float* bg, int N, float* out, float* save_s, evae_stream_t stream_) {
  EVAE_REQUIRE(M >= 0 && K > 0 && N > 0, "gated_dense_fwd_p6t: bad sizes M=%d K=%d N=%d", M, K, N);
  if (M == 0) return EVAE_OK;
  EVAE_REQUIRE(xT_img && w_img && out && x_nks >= cdiv(M, 128) * 8, "gated_dense_fwd_p6t: bad arguments (k-steps %d)", x_nks);
  GemmArgs g = {};
  g.ones_col = -1; g.npairs = 1;
  g.A[0] = (const float*)xT_img; g.lda[0] = x_nks; g.B[0] = (const float*)w_img; g.Kc[0] = p6_nks(K) * P6_KS;
  g.M = M; g.N = N; g.bias0 = bh; g.bias1 = bg; g.out0 = out; g.out2 = save_s; g.ldo = N;
  return launch_gemm_p6<EPI_GATED, 128, true>(g, 1, (hipStream_t)stream_, "gated_dense_fwd(p6)");
}

// dx = [dh | dg] W with [dh | dg] (N columns) given as its transpose's image and W^T as image(rows = K outputs, k = N)
// (evae_p6_pack_cols of the stacked banks); out_prev != NULL: the gate derivative of the layer below in the epilogue, (dh, dg) of
// that layer to dh / dg (row stride ldo) or, u8_img != NULL, as the tile images of evae_dense_bwd_weight_u8
extern "C" int evae_dense_bwd_data_p6t(const void* dyT_img, int dy_nks, int M, int N, const void* wT_img, int K,
                                       const float* out_prev, const float* s_prev, float* dh, float* dg, int ldo, void* u8_img,
                                       int u8_nslab, int u8_mbase, evae_stream_t stream_) {
  EVAE_REQUIRE(M >= 0 && K > 0 && N > 0, "dense_bwd_data_p6t: bad sizes M=%d K=%d N=%d", M, K, N);
  if (M == 0) return EVAE_OK;
  EVAE_REQUIRE(dyT_img && wT_img && dy_nks >= cdiv(M, 128) * 8, "dense_bwd_data_p6t: bad arguments (k-steps %d)", dy_nks);
  const bool gate = out_prev != nullptr;
  EVAE_REQUIRE(u8_img ? (gate && s_prev && (u8_mbase % 8) == 0) : (dh && ldo >= K && (!gate || (s_prev && dg))),
               "dense_bwd_data_p6t: bad outputs");
  hipStream_t stream = (hipStream_t)stream_;
  GemmArgs g = {};
  g.ones_col = -1; g.npairs = 1;
  g.A[0] = (const float*)dyT_img; g.lda[0] = dy_nks; g.B[0] = (const float*)wT_img; g.Kc[0] = p6_nks(N) * P6_KS;
  g.M = M; g.N = K; g.out0 = dh; g.out1 = gate ? dg : nullptr; g.ldo = ldo; g.e0 = out_prev; g.e1 = s_prev;
  g.img = (unsigned short*)u8_img; g.img_nslab = u8_nslab; g.img_mbase = u8_mbase;
  const bool narrow = gemm_x6_pick_bn(M, K) == 64;
  if (u8_img) {
    if (narrow) return launch_gemm_p6<EPI_GATE_BWD_IMG, 64, true>(g, 1, stream, "dense_bwd_data(p6, gate, byte-layer images)");
    return launch_gemm_p6<EPI_GATE_BWD_IMG, 128, true>(g, 1, stream, "dense_bwd_data(p6, gate, byte-layer images)");
  }
  if (gate) {
    if (narrow) return launch_gemm_p6<EPI_GATE_BWD, 64, true>(g, 1, stream, "dense_bwd_data(p6, gate)");
    return launch_gemm_p6<EPI_GATE_BWD, 128, true>(g, 1, stream, "dense_bwd_data(p6, gate)");
  }
  if (narrow) return launch_gemm_p6<EPI_LINEAR, 64, true>(g, 1, stream, "dense_bwd_data(p6)");
  return launch_gemm_p6<EPI_LINEAR, 128, true>(g, 1, stream, "dense_bwd_data(p6)");
}

// dw [N x K] = dy^T x, db [N] = column sums of dy, from the images of dy^T (N rows) and x^T (K rows + the all-ones row K when
// db != NULL), both with nks k-steps along the batch rows.  Split over the batch rows into one round of blocks; fixed-order finish.
struct P6WgradPlan { int kc, nz, ksplit, local; };
static P6WgradPlan p6_wgrad_plan(int nks, int N, int K, bool with_db) {
  P6WgradPlan p;
  p.kc = K + (with_db ? 1 : 0);
  const int tiles = cdiv(N, BM) * cdiv(p.kc, 64);
  static int slots = -1, local = -1;
  if (slots < 0) { const char* e = getenv("EVAE_P6_WGRAD_SLOTS"); slots = e ? atoi(e) : 256; }
  if (local < 0) { const char* e = getenv("EVAE_P6_WGRAD_LOCAL"); local = e ? atoi(e) : 16; }      // slices of the XCD-local form (0: off)
  int nz = std::max(1, std::min(slots / std::max(tiles, 1), nks / 8));
  // a long contraction: 8 or 16 slices, each slice's tiles on one XCD (gemm_p6_kernel, sk_local); two blocks per CU are resident,
  // so up to 64 blocks per XCD run in one round
  p.local = 0;
  if (local >= 8 && nz >= 8) {
    const int want = (local >= 16 && tiles * 2 <= 64 && nks / 16 >= 8) ? 16 : 8;
    if (tiles * (want / 8) <= 64) { nz = want; p.local = 1; }
  }
  p.ksplit = cdiv(nks, nz);
  p.nz = cdiv(nks, p.ksplit);
  return p;
}
extern "C" size_t evae_dense_bwd_weight_p6_workspace_bytes(int nks, int N, int K) {
  if (nks <= 0 || N <= 0 || K <= 0) return 256;
  const P6WgradPlan p = p6_wgrad_plan(nks, N, K, true);
  return align_up((size_t)p.nz * N * p.kc * sizeof(float), 256) + 256;
}
extern "C" int evae_dense_bwd_weight_p6(const void* dyT_img, const void* xT_img, int nks, int N, int K, float* dw, float* db,
                                        void* ws, size_t ws_bytes, evae_stream_t stream_) {
  EVAE_REQUIRE(nks > 0 && N > 0 && K > 0 && dyT_img && xT_img && dw, "dense_bwd_weight_p6: bad arguments");
  const P6WgradPlan p = p6_wgrad_plan(nks, N, K, db != nullptr);
  if (ws == nullptr || ws_bytes < (size_t)p.nz * N * p.kc * sizeof(float)) { set_error("dense_bwd_weight_p6: workspace too small (%zu)", ws_bytes); return EVAE_EWORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  GemmArgs g = {};
  g.ones_col = -1; g.npairs = 1;
  g.A[0] = (const float*)dyT_img; g.B[0] = (const float*)xT_img; g.Kc[0] = nks * P6_KS; g.M = N; g.N = p.kc;
  g.out0 = (float*)ws; g.ldo = p.kc; g.ksplit = p.ksplit; g.sk_local = p.local ? p.nz : 0;
  int rc = launch_gemm_p6<EPI_RAW, 64>(g, p.nz, stream, "dense_bwd_weight(p6)");
  if (rc) return rc;
  FinishArgs f = {};
  f.part = (const float*)ws; f.nz = p.nz; f.M = N; f.N = p.kc; f.ldo = p.kc; f.epi = EPI_RAW; f.out0 = dw;
  f.ones_col = db ? K : -1; f.out_db = db;
  return launch_finish(f, stream);
}
