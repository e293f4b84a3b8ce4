// Exemplar prior at large latent sizes (z_dim > 64: fully_conv's z = 256 / 294) as a dense contraction on the matrix
// cores.  Reference: utils/distributions.py:12-25 (the expanded-form distance is the reference's own, there in fp64),
// models/BaseModel.py:98-128.  With z' = z/s - mu, c' = c/s - mu (mu = the mean of the call's queries in sigma units;
// the distance is translation invariant):
//   forward   S = C' Z'^T on the fp32-MFMA GEMM of evae_gemm_kernel.h with a log-sum-exp epilogue (EPI_PRIOR_LSE): one
//             (max, sum exp, #masked) partial per 128-exemplar tile and query, merged by prior_merge_sel_kernel;
//   backward  the same GEMM with the epilogue P[e][q] = g_q exp(log N(z_q | c_e) - lse_q) (EPI_PRIOR_P), then two
//             library-shaped GEMMs of the dense family: T = P [Z' | 1] (evae_dense_bwd_data) and U = P^T C' with the
//             row sums of P in the bias-gradient column (evae_dense_bwd_weight); dC, dZ, dlogvar come out of T and U
//             exactly as in prior_bwd_mfma_kernel (evae_prior.hip).
// The fp32 expanded form is guarded like the small-z kernels: if any centred query norm exceeds the limit, a device flag
// cancels the GEMM launches and releases the direct-difference VALU kernels of evae_prior.hip instead -- both sets of
// launches are enqueued, the flag decides on the device, nothing is read back (hipGraph-capturable).
#include "evae_gemm_x6.h"
#include "evae_prior_gemm.h"

namespace evae {

// ---- mu[k] = mean over the B queries of z[q][k] exp(-log_var[k]/2); also clears the guard flag ------------------------
// z: [B x zdim] rows, or (count != B) the [B x zp] partial column sums of prior_colsum_part_kernel over `count` rows
__global__ __launch_bounds__(1024) void prior_colmean_kernel(const float* __restrict__ z, int B, int zdim, int zp,
                                                             const float* __restrict__ log_var, float* __restrict__ mu,
                                                             unsigned* __restrict__ flag, int count) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  float a = 0.f;
  const int ld = count != B ? zp : zdim;
  if (k < zdim)
    for (int q = part; q < B; q += 16) a += z[(size_t)q * ld + k];
  red[part][lane] = a;
  __syncthreads();
  if (part == 0 && k < zp) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) t += red[p][lane];      // fixed order
    mu[k] = k < zdim ? (t / (float)count) * expf(-0.5f * log_var[k]) : 0.f;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *flag = 0u;
}

// Many queries (the IWAE evaluator: 20 000 samples): one block per 64 columns would walk all rows alone (330 us at
// B = 20 000).  Two stages instead, both in a fixed order: kCmParts row slices summed by a grid of blocks into part[slice][k],
// then the kernel above over the kCmParts partial rows (count = B).
constexpr int kCmParts = 64;
__global__ __launch_bounds__(1024) void prior_colsum_part_kernel(const float* __restrict__ z, int B, int zdim, int zp,
                                                                 int rows_per, float* __restrict__ part) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per, r1 = min(B, r0 + rows_per);
  float a = 0.f;
  if (k < zdim)
    for (int q = r0 + sub; q < r1; q += 16) a += z[(size_t)q * zdim + k];
  red[sub][lane] = a;
  __syncthreads();
  if (sub == 0 && k < zp) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) t += red[p][lane];
    part[(size_t)blockIdx.y * zp + k] = t;
  }
}
// mu (and the cleared guard flag) of a call: one launch for few queries, two for many
static void launch_colmean(const float* z, int B, int zdim, int zp, const float* log_var, float* mu, unsigned* flag,
                           hipStream_t stream) {
  if (B <= 4096) {
    prior_colmean_kernel<<<cdiv(zp, 64), 1024, 0, stream>>>(z, B, zdim, zp, log_var, mu, flag, B);
    return;
  }
  float* part = mu + zp;                          // [kCmParts][zp] behind mu (the layouts reserve it)
  const int rows_per = cdiv(B, kCmParts);
  prior_colsum_part_kernel<<<dim3(cdiv(zp, 64), kCmParts), 1024, 0, stream>>>(z, B, zdim, zp, rows_per, part);
  // the partial rows are zp wide with zeros beyond zdim: read them as a [kCmParts x zp] matrix, divide by the real count
  prior_colmean_kernel<<<cdiv(zp, 64), 1024, 0, stream>>>(part, kCmParts, zdim, zp, log_var, mu, flag, B);
}

// ---- scaled, centred, padded copies + squared norms; one wave per row ------------------------------------------------
// rows [0, Bp): queries -> Zs [Bp x ldz] (ones column at zp when ldz > zp; rows B..Bp are zero padding, Bp = B rounded
// up to 4: the matrix is an operand of a GEMM whose contraction runs over its rows), zn; a norm above `limit` raises the flag.
// rows [Bp, Bp + C): exemplars -> Cs [C x zp], cn.
__global__ __launch_bounds__(256) void prior_prep_kernel(const float* __restrict__ z, int B, int Bp, float* __restrict__ Zs,
                                                         int ldz, float* __restrict__ zn, const float* __restrict__ c, int C,
                                                         float* __restrict__ Cs, float* __restrict__ cn, int zdim, int zp,
                                                         const float* __restrict__ log_var, const float* __restrict__ mu,
                                                         float limit, unsigned* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Bp + C) return;
  const bool isq = row < Bp;
  if (isq && row >= B) {
    for (int k = lane; k < ldz; k += 64) Zs[(size_t)row * ldz + k] = 0.f;
    return;
  }
  const float* src = isq ? z + (size_t)row * zdim : c + (size_t)(row - Bp) * zdim;
  float* dst = isq ? Zs + (size_t)row * ldz : Cs + (size_t)(row - Bp) * zp;
  const int ld = isq ? ldz : zp;
  float s = 0.f;
  for (int k = lane; k < ld; k += 64) {
    float v = 0.f;
    if (k < zdim) {
      v = fmaf(src[k], expf(-0.5f * log_var[k]), -mu[k]);
      s = fmaf(v, v, s);
    } else if (isq && k == zp) {
      v = 1.f;                       // the ones column of [Z' | 1]
    }
    dst[k] = v;
  }
  s = wave_sum(s);
  if (lane == 0) {
    if (isq) {
      zn[row] = s;
      if (s > limit) atomicOr(flag, 1u);
    } else {
      cn[row - Bp] = s;
    }
  }
}

// ---- evaluator-sized, unmasked calls at small latent sizes (IWAE: thousands of samples x all exemplars, z <= 48) ---------
// The streaming counterpart of EPI_PRIOR_LSE on the split-bf16 pipe (csrc/evae_gemm_x6.h): a block owns 128 queries, keeps
// their three-term fragments in REGISTERS, and walks over its share of the exemplar tiles with the log-sum-exp state of
// every query column in registers too -- no per-tile partial planes.  Per tile: the (prefetched) fp32 rows are split into
// bf16 planes in LDS (double-buffered: one barrier per tile), 6 partial products per k-step on v_mfma_f32_32x32x16_bf16
// give c'.z' at fp32-GEMM accuracy, and the epilogue is the online LSE on t = c'.z' - |c'|^2/2 (the query's own norm is added
// at the end).  prior_fwd_mfma_kernel spends 2/3 of its time in fp32 MFMAs at these sizes; six bf16 MFMAs per 16 k replace
// sixteen fp32 ones.  Output: one partial row per split, in the (max log N, sum exp, #masked = 0) convention of the merge.
template <int KS>     // k-steps of 16: K <= 16 KS <= 48
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void prior_x6_lse_kernel(
    const float* __restrict__ Cs, int C, const float* __restrict__ cn, const float* __restrict__ Zs, int B,
    const float* __restrict__ zn, int zp, int tiles_per_split, const float* __restrict__ cst_dev,
    const unsigned* __restrict__ skip_flag, float* __restrict__ pm, float* __restrict__ ps, float* __restrict__ pn, int ldp) {
  // block = 8 waves (2 exemplar halves x 4 query quarters): 128 exemplars x 256 queries per tile, wave tile 64 x 64.  The
  // query planes (256 rows) are only needed until their fragments sit in registers: the same LDS then holds the two
  // exemplar buffers, so the block needs 96 KB and its 8 waves give every SIMD two.
  constexpr int NS = (KS + 1) / 2;                 // 32-wide slabs staged per tile
  constexpr int SLAB = 3 * X6_PLANE;               // bytes of one slab's three planes (128 rows)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (*skip_flag != 0u) return;                    // the norm guard chose the direct-difference kernel
  char* const lds = reinterpret_cast<char*>(smem);
  char* const Ep = lds;                            // two buffers of [NS][3 planes][128 rows]; before the loop: the query planes
  float* const hc_s = reinterpret_cast<float*>(lds + 2 * NS * SLAB);       // [2][128] |c'|^2 / 2 of the tile's rows
  float* const red = hc_s + 256;                                           // [2 wave rows][256][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3, l31 = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x, q0 = blockIdx.y * 256;
  const int ntiles = (C + 127) / 128;
  const int tile_begin = split * tiles_per_split;
  const int tile_end = min(tile_begin + tiles_per_split, ntiles);

  // staging roles: chunk (row, c8) = float4 c8 of a slab's 32 k of tile row `row`; rows (tid >> 3) + 64 i
  const int c8 = tid & 7;
  auto st_off = [&](int row) -> unsigned {
    return (unsigned)(row * 64 + ((((c8 >> 1) ^ ((row >> 2) & 3))) << 4) + (c8 & 1) * 8);
  };
  auto load_chunk = [&](const float* src, int r, int nrows, int sl) -> float4 {
    const int k = sl * 32 + 4 * c8;
    return (r < nrows && k + 4 <= zp) ? *reinterpret_cast<const float4*>(src + (size_t)r * zp + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto stage_chunk = [&](char* p, const float4 v) {
    unsigned a0, a1, a2, b0, b1, b2;
    x6_split2(v.x, v.y, a0, a1, a2);
    x6_split2(v.z, v.w, b0, b1, b2);
    x6_u32x2 t0 = {a0, b0}, t1 = {a1, b1}, t2 = {a2, b2};
    *reinterpret_cast<x6_u32x2*>(p) = t0;
    *reinterpret_cast<x6_u32x2*>(p + X6_PLANE) = t1;
    *reinterpret_cast<x6_u32x2*>(p + 2 * X6_PLANE) = t2;
  };
  // fragment offsets inside a 128-row plane: rows of this lane, 16-byte slot of k-step `step` (0 / 1)
  unsigned fa[2][2], fb[2][2];
#pragma unroll
  for (int step = 0; step < 2; ++step) {
    const int ks = 2 * step + lh;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = wr * 64 + t * 32 + l31, c = (wc & 1) * 64 + t * 32 + l31;      // query rows: inside the half (wc >> 1)
      fa[step][t] = (unsigned)(r * 64 + ((ks ^ ((r >> 2) & 3)) << 4));
      fb[step][t] = (unsigned)(c * 64 + ((ks ^ ((c >> 2) & 3)) << 4));
    }
  }

  // queries: two halves of 128 rows, half h staged where exemplar buffer h will live
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 3) + 64 * i;
        float4 v = load_chunk(Zs, q0 + h * 128 + row, B, sl);
        v.x *= kLog2e; v.y *= kLog2e; v.z *= kLog2e; v.w *= kLog2e;      // the products come out in base-2 units: exp2(t - max) directly
        stage_chunk(Ep + h * NS * SLAB + sl * SLAB + st_off(row), v);
      }
  float4 rv[NS][2];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
#pragma unroll
      for (int i = 0; i < 2; ++i) rv[sl][i] = load_chunk(Cs, t * 128 + (tid >> 3) + 64 * i, C, sl);
  };
  if (tile_begin < tile_end) load_tile(tile_begin);
  __syncthreads();
  x6_bf16x8 bq[KS][2][3];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        bq[ks][nt][p] = *reinterpret_cast<const x6_bf16x8*>(Ep + (wc >> 1) * NS * SLAB + (ks >> 1) * SLAB + p * X6_PLANE + fb[ks & 1][nt]);
  __syncthreads();              // every wave holds its query fragments: the planes become the exemplar buffers

  float tm_[2] = {-INFINITY, -INFINITY}, ssum[2] = {0.f, 0.f};       // running max of t and sum exp(t - max) per query column
  for (int t = tile_begin; t < tile_end; ++t) {
    const int pb = (t - tile_begin) & 1, e0 = t * 128;
    char* const Eb = Ep + pb * NS * SLAB;
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
#pragma unroll
      for (int i = 0; i < 2; ++i) stage_chunk(Eb + sl * SLAB + st_off((tid >> 3) + 64 * i), rv[sl][i]);
    if (tid < 128) hc_s[pb * 128 + tid] = (e0 + tid < C) ? (0.5f * kLog2e) * cn[e0 + tid] : INFINITY;   // absent rows: t = -inf
    __syncthreads();            // the tile is staged; the other buffer (read two tiles ago) is free for the next staging
    if (t + 1 < tile_end) load_tile(t + 1);
    // the accumulators start at -log2(e) |c'|^2 / 2 of their row: the products then leave t2 = log2(e) (c'.z' - |c'|^2 / 2) and the
    // epilogue is max, subtract, exp2, add per pair (subtract and add two pairs per instruction)
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float h = -hc_s[pb * 128 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
        acc[mt][0][r] = h; acc[mt][1][r] = h;
      }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      x6_bf16x8 af[2][3];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          af[mt][p] = *reinterpret_cast<const x6_bf16x8*>(Eb + (ks >> 1) * SLAB + p * X6_PLANE + fa[ks & 1][mt]);
#pragma unroll
      for (int tt = 0; tt < 6; ++tt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][PA[tt]], bq[ks][nt][PB[tt]], acc[mt][nt], 0, 0, 0);
    }
    // online log-sum-exp (base 2) over this tile's 64 rows of the wave, per query column
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float vmax = -INFINITY;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) vmax = fmaxf(vmax, acc[mt][nt][r]);
      if (vmax > tm_[nt]) {
        ssum[nt] *= fast_exp2(tm_[nt] - vmax);        // first tile: 0 * exp2(-inf) = 0
        tm_[nt] = vmax;
      }
      if (tm_[nt] != -INFINITY) {
        const x6_f32x2 mk = {tm_[nt], tm_[nt]};
        x6_f32x2 part = {0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const x6_f32x2 v = {acc[mt][nt][r], acc[mt][nt][r + 1]};
            const x6_f32x2 d = v - mk;
            const x6_f32x2 e = {fast_exp2(d[0]), fast_exp2(d[1])};
            part = part + e;
          }
        ssum[nt] += part[0] + part[1];
      }
    }
  }
  // lanes l / l + 32 hold the same column, then the two wave rows through LDS
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const float ot = __shfl_xor(tm_[nt], 32, 64), os = __shfl_xor(ssum[nt], 32, 64);
    const float mx = fmaxf(tm_[nt], ot);
    const float fa_ = (tm_[nt] == mx) ? 1.f : fast_exp2(tm_[nt] - mx);
    const float fb_ = (ot == mx) ? 1.f : fast_exp2(ot - mx);
    if (lh == 0) {
      float* cb = red + (wr * 256 + wc * 64 + nt * 32 + l31) * 2;
      cb[0] = mx; cb[1] = ssum[nt] * fa_ + os * fb_;
    }
  }
  __syncthreads();
  if (tid < 256 && q0 + tid < B) {
    const float t0 = red[tid * 2], t1 = red[(256 + tid) * 2];
    const float mx = fmaxf(t0, t1);
    float sacc = 0.f;
    if (t0 != -INFINITY) sacc += red[tid * 2 + 1] * fast_exp2(t0 - mx);
    if (t1 != -INFINITY) sacc += red[(256 + tid) * 2 + 1] * fast_exp2(t1 - mx);
    const size_t o = (size_t)split * ldp + q0 + tid;
    pm[o] = (mx == -INFINITY) ? -INFINITY : *cst_dev + (mx * kLn2 - 0.5f * zn[q0 + tid]);      // back from base-2 units
    ps[o] = sacc;
    pn[o] = 0.f;
  }
}

// number of exemplar splits of the streaming kernel: fill the 256 CUs (one block each) with as little tail as possible,
// at least eight tiles per block
static int prior_x6_splits(int B, int C) {
  const int nq = cdiv(B, 256), ntiles = cdiv(C, 128);
  int best = 1;
  double best_eff = 0.0;
  for (int ns = 1; ns <= 32 && ns * 8 <= std::max(ntiles, 8); ++ns) {
    const int tps = cdiv(ntiles, ns), nse = cdiv(ntiles, tps);
    const long blocks = (long)nse * nq;
    const double eff = (double)blocks / (double)(cdiv((int)blocks, 256) * 256) * ((double)ntiles / ((double)tps * nse));
    if (eff > best_eff + 1e-9) { best_eff = eff; best = nse; }
  }
  return best;
}
bool prior_stream_applies(int B, int C, int zdim, bool masked) {
  return !masked && zdim <= 48 && (zdim & 3) == 0 && B >= 1024 && (int64_t)B * C >= ((int64_t)1 << 26) && gemm_x6_enabled() &&
         prior_gemm_applies(B, C, zdim);
}

// ---- merge of per-tile partials [R x ldp] -> (max, sumexp, nmask) per query; R is chosen on the device ----------------
// lanes <-> consecutive queries (coalesced rows of the partial planes), the 16 waves of a block stride over R.
__global__ __launch_bounds__(1024) void prior_merge_sel_kernel(const float* __restrict__ pm, const float* __restrict__ ps,
                                                               const float* __restrict__ pn, int ldp, int B, int R_gemm,
                                                               int R_valu, int ld_valu, const unsigned* __restrict__ flag,
                                                               float* __restrict__ om, float* __restrict__ os,
                                                               float* __restrict__ on) {
  __shared__ float red[16][64][3];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int q = blockIdx.x * 64 + lane;
  const bool valu = *flag != 0u;
  const int R = valu ? R_valu : R_gemm;
  const int ld = valu ? ld_valu : ldp;
  float m = -INFINITY, s = 0.f, n = 0.f;
  if (q < B) {
    for (int r = part; r < R; r += 16) {
      const float mr = pm[(size_t)r * ld + q], sr = ps[(size_t)r * ld + q];
      n += pn[(size_t)r * ld + q];
      if (mr > m) { s = s * __expf(m - mr) + sr; m = mr; }      // m == -inf: s == 0, exp(-inf) = 0
      else if (mr != -INFINITY) s += sr * __expf(mr - m);
    }
  }
  red[part][lane][0] = m; red[part][lane][1] = s; red[part][lane][2] = n;
  __syncthreads();
  if (part == 0 && q < B) {
    float mm = -INFINITY, ss = 0.f, nn = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) mm = fmaxf(mm, red[p][lane][0]);
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const float mp = red[p][lane][0];
      if (mp != -INFINITY) ss += red[p][lane][1] * __expf(mp - mm);
      nn += red[p][lane][2];
    }
    om[q] = mm; os[q] = ss; on[q] = nn;
  }
}

// ---- backward finish -------------------------------------------------------------------------------------------------
// T [C x ldt] = P [Z' | 1] (column zp = column sums of P), U [B x zp] = P^T C', rs [B] = row sums of P.
//   rows [0, C):      dC[e][k]  = (T[e][k] - T[e][zp] C'[e][k]) / s_k ;  dV partial += C'_ek (T[e][zp] C'_ek - 2 T[e][k])
//   rows [C, C + B):  dZ[q][k]  = (U[q][k] - rs_q Z'[q][k]) / s_k     ;  dV partial += rs_q Z'_qk^2 ;  sum_P partial += rs_q
// One wave per row, 4 rows per block; per-block partials of dV [zp] and of sum P go to dvp [nblocks x (zp + 1)].
__global__ __launch_bounds__(256) void prior_gemm_bwd_rows_kernel(const float* __restrict__ T, int ldt, const float* __restrict__ Cs,
                                                                  int C, const float* __restrict__ U, const float* __restrict__ rs,
                                                                  const float* __restrict__ Zs, int ldz, int B, int zdim, int zp,
                                                                  const float* __restrict__ log_var, float* __restrict__ dc,
                                                                  float* __restrict__ dz, float* __restrict__ dvp,
                                                                  const unsigned* __restrict__ flag) {
  extern __shared__ float sh[];            // [4][zp + 1]
  if (*flag != 0u) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  float* mine = sh + w * (zp + 1);
  for (int k = lane; k <= zp; k += 64) mine[k] = 0.f;
  if (row < C) {
    const float csum = T[(size_t)row * ldt + zp];
    for (int k = lane; k < zdim; k += 64) {
      const float t = T[(size_t)row * ldt + k], c_ = Cs[(size_t)row * zp + k];
      dc[(size_t)row * zdim + k] = (t - csum * c_) * expf(-0.5f * log_var[k]);
      mine[k] = c_ * (csum * c_ - 2.0f * t);
    }
  } else if (row < C + B) {
    const int q = row - C;
    const float rsum = rs[q];
    for (int k = lane; k < zdim; k += 64) {
      const float z_ = Zs[(size_t)q * ldz + k];
      dz[(size_t)q * zdim + k] = (U[(size_t)q * zp + k] - rsum * z_) * expf(-0.5f * log_var[k]);
      mine[k] = rsum * z_ * z_;
    }
    if (lane == 0) mine[zp] = rsum;
  }
  __syncthreads();
  for (int k = threadIdx.x; k <= zp; k += 256)
    dvp[(size_t)blockIdx.x * (zp + 1) + k] = sh[k] + sh[(zp + 1) + k] + sh[2 * (zp + 1) + k] + sh[3 * (zp + 1) + k];
}

// dlogvar[k] = 0.5 sum_blocks dV[k] - 0.5 sum_blocks sumP; one block (1024 threads = 64 k x 16 parts) per 64 k
__global__ __launch_bounds__(1024) void prior_gemm_bwd_dlv_kernel(const float* __restrict__ dvp, int nblocks, int zdim, int zp,
                                                                  float* __restrict__ dlogvar, const unsigned* __restrict__ flag) {
  __shared__ float red[16][64][2];
  if (*flag != 0u) return;
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  float sv = 0.f, sg = 0.f;
  for (int b = part; b < nblocks; b += 16) {
    if (k < zdim) sv += dvp[(size_t)b * (zp + 1) + k];
    sg += dvp[(size_t)b * (zp + 1) + zp];
  }
  red[part][lane][0] = sv; red[part][lane][1] = sg;
  __syncthreads();
  if (part == 0 && k < zdim) {
    float a = 0.f, g_ = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) { a += red[p][lane][0]; g_ += red[p][lane][1]; }
    dlogvar[k] = 0.5f * a - 0.5f * g_;
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
static int zpad(int zdim) { return (zdim + 3) / 4 * 4; }
static int bpad(int B) { return (B + 3) / 4 * 4; }

PriorGemmFwdLayout prior_gemm_fwd_layout(int B, int C, int zdim, int ns_valu, bool stream) {
  PriorGemmFwdLayout L;
  const int zp = zpad(zdim);
  // partial rows: one per 128-exemplar tile (GEMM epilogue) or one per split of the streaming kernel
  L.zp = zp; L.ldp = bpad(B); L.tiles_m = stream ? prior_x6_splits(B, C) : cdiv(C, BM);
  L.stream = stream ? 1 : 0;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes, 256); return at; };
  L.flag = take(256); L.mu = take((size_t)zp * 4 * (1 + kCmParts));
  L.Zs = take((size_t)L.ldp * zp * 4); L.zn = take((size_t)L.ldp * 4);
  L.Cs = take((size_t)C * zp * 4); L.cn = take((size_t)C * 4);
  const size_t rows = (size_t)(L.tiles_m > ns_valu ? L.tiles_m : ns_valu);
  L.pm = take(rows * L.ldp * 4); L.ps = take(rows * L.ldp * 4); L.pn = take(rows * L.ldp * 4);
  L.total = o + 256;
  return L;
}

bool prior_gemm_applies(int B, int C, int zdim) {
  static int off = -1;
  if (off < 0) { const char* e = getenv("EVAE_PRIOR_VALU"); off = (e && atoi(e)) ? 1 : 0; }
  // 31-bit buffer offsets of the GEMM's operand loads; the P matrix of the backward is [C x B]
  return !off && B > 0 && C > 0 && (int64_t)C * zpad(zdim) < ((int64_t)1 << 29) - (1 << 22) &&
         (int64_t)B * (zpad(zdim) + 4) < ((int64_t)1 << 29) - (1 << 22);
}

static int launch_prior_gemm(GemmArgs& g, int B, bool lse, hipStream_t stream) {
  // both operands are contraction-contiguous (centred exemplars x centred queries): the split-bf16 kernel when the launch
  // fills the machine (the IWAE evaluator: 5000 samples x 100000 exemplars)
  if (gemm_x6_use(g) && B > 64) {
    if (lse) return launch_gemm_x6<EPI_PRIOR_LSE>(g, 1, stream, "prior_gemm(lse, x6)");
    return launch_gemm_x6<EPI_PRIOR_P>(g, 1, stream, "prior_gemm(P, x6)");
  }
  if (lse) {
    if (B <= 64) return launch_gemm_w<true, true, EPI_PRIOR_LSE, true, 64, 8>(g, 1, stream, "prior_gemm(lse)");
    return launch_gemm_w<true, true, EPI_PRIOR_LSE, true, 128, 8>(g, 1, stream, "prior_gemm(lse)");
  }
  if (B <= 64) return launch_gemm_w<true, true, EPI_PRIOR_P, true, 64, 8>(g, 1, stream, "prior_gemm(P)");
  return launch_gemm_w<true, true, EPI_PRIOR_P, true, 128, 8>(g, 1, stream, "prior_gemm(P)");
}

// cst = -1/2 sum_k (log_var_k + log 2 pi) is needed on the host side of the GEMM arguments but lives on the device: the
// epilogues take it from a one-float device buffer instead (written by this kernel), so nothing is read back.
__global__ void prior_cst_kernel(const float* __restrict__ log_var, int zdim, float* __restrict__ out) {
  float part = 0.f;
  for (int k = threadIdx.x; k < zdim; k += 64) part += log_var[k] + kLog2Pi;
  part = wave_sum(part);
  if (threadIdx.x == 0) *out = -0.5f * part;
}

int prior_gemm_fwd(const float* z, int B, const float* centres, int C, int zdim, const float* log_var,
                   const int64_t* z_idx, const int64_t* c_idx, float norm_limit, char* ws, const PriorGemmFwdLayout& L,
                   hipStream_t stream) {
  if (L.stream) EVAE_REQUIRE(z_idx == nullptr || c_idx == nullptr, "prior_gemm_fwd: the streaming kernel takes unmasked calls only");
  unsigned* flag = (unsigned*)(ws + L.flag);
  float* cstp = (float*)(ws + L.flag) + 16;
  float* mu = (float*)(ws + L.mu);
  float* Zs = (float*)(ws + L.Zs); float* zn = (float*)(ws + L.zn);
  float* Cs = (float*)(ws + L.Cs); float* cn = (float*)(ws + L.cn);
  launch_colmean(z, B, zdim, L.zp, log_var, mu, flag, stream);
  prior_cst_kernel<<<1, 64, 0, stream>>>(log_var, zdim, cstp);
  prior_prep_kernel<<<cdiv(L.ldp + C, 4), 256, 0, stream>>>(z, B, L.ldp, Zs, L.zp, zn, centres, C, Cs, cn, zdim, L.zp, log_var,
                                                            mu, norm_limit, flag);
  int rc = check_launch("prior_prep_kernel");
  if (rc) return rc;
  if (L.stream) {
    const int ns6 = L.tiles_m, tps = cdiv(cdiv(C, 128), ns6), ks = cdiv(L.zp, 16);
    const size_t lds = (size_t)2 * ((ks + 1) / 2) * 3 * X6_PLANE + (256 + 1024) * sizeof(float);
    float* pm = (float*)(ws + L.pm); float* ps = (float*)(ws + L.ps); float* pn = (float*)(ws + L.pn);
    const dim3 grid(ns6, cdiv(B, 256));
#define EVAE_STREAM_LAUNCH(KS_)                                                                                                   \
    do {                                                                                                                          \
      static bool attr_done = false;                                                                                              \
      if (!attr_done) {                                                                                                           \
        (void)hipFuncSetAttribute((const void*)prior_x6_lse_kernel<KS_>, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024); \
        attr_done = true;                                                                                                         \
      }                                                                                                                           \
      prior_x6_lse_kernel<KS_><<<grid, 512, lds, stream>>>(Cs, C, cn, Zs, B, zn, L.zp, tps, cstp, flag, pm, ps, pn, L.ldp);       \
    } while (0)
    if (ks <= 1) EVAE_STREAM_LAUNCH(1);
    else if (ks == 2) EVAE_STREAM_LAUNCH(2);
    else EVAE_STREAM_LAUNCH(3);
#undef EVAE_STREAM_LAUNCH
    return check_launch("prior_x6_lse_kernel");
  }
  GemmArgs g = {};
  g.ones_col = -1;
  g.A[0] = Cs; g.B[0] = Zs; g.lda[0] = L.zp; g.ldb[0] = L.zp; g.Kc[0] = L.zp; g.npairs = 1;
  g.M = C; g.N = B; g.e0 = cn; g.e1 = zn; g.ksplit = 0;
  g.out0 = (float*)(ws + L.pm); g.out1 = (float*)(ws + L.ps); g.out2 = (float*)(ws + L.pn); g.ldo = L.ldp;
  g.pr_ridx = c_idx; g.pr_cidx = z_idx; g.pr_cst_dev = cstp; g.skip_flag = flag;
  return launch_prior_gemm(g, B, true, stream);
}

void prior_gemm_merge(const char* ws, const PriorGemmFwdLayout& L, int B, int ns_valu, float* om, float* os, float* on,
                      hipStream_t stream) {
  prior_merge_sel_kernel<<<cdiv(B, 64), 1024, 0, stream>>>((const float*)(ws + L.pm), (const float*)(ws + L.ps),
                                                          (const float*)(ws + L.pn), L.ldp, B, L.tiles_m, ns_valu, B,
                                                          (const unsigned*)(ws + L.flag), om, os, on);
}

PriorGemmBwdLayout prior_gemm_bwd_layout(int B, int C, int zdim) {
  PriorGemmBwdLayout L;
  const int zp = zpad(zdim);
  L.zp = zp; L.ldz = zp + 4; L.ldp = bpad(B);
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes, 256); return at; };
  L.flag = take(256); L.mu = take((size_t)zp * 4 * (1 + kCmParts));
  L.Zs = take((size_t)L.ldp * L.ldz * 4); L.zn = take((size_t)L.ldp * 4);
  L.Cs = take((size_t)C * zp * 4); L.cn = take((size_t)C * 4);
  L.P = take((size_t)C * L.ldp * 4);
  L.T = take((size_t)C * L.ldz * 4);
  L.U = take((size_t)L.ldp * zp * 4); L.rs = take((size_t)L.ldp * 4);
  L.nblk = cdiv(C + B, 4);
  L.dvp = take((size_t)L.nblk * (zp + 1) * 4);
  L.ws_data_bytes = evae_dense_bwd_data_workspace_bytes(C, L.ldp, L.ldz, 1);
  L.ws_data = take(L.ws_data_bytes);
  L.ws_weight_bytes = evae_dense_bwd_weight_workspace_bytes(C, L.ldp, zp);
  L.ws_weight = take(L.ws_weight_bytes);
  L.total = o + 256;
  return L;
}

int prior_gemm_bwd(const float* z, int B, const float* centres, int C, int zdim, const float* log_var,
                   const int64_t* z_idx, const int64_t* c_idx, const float* lse, const float* gout, float norm_limit,
                   float* dz, float* dc, float* dlogvar, char* ws, const PriorGemmBwdLayout& L, hipStream_t stream) {
  unsigned* flag = (unsigned*)(ws + L.flag);
  float* cstp = (float*)(ws + L.flag) + 16;
  float* mu = (float*)(ws + L.mu);
  float* Zs = (float*)(ws + L.Zs); float* zn = (float*)(ws + L.zn);
  float* Cs = (float*)(ws + L.Cs); float* cn = (float*)(ws + L.cn);
  float* P = (float*)(ws + L.P); float* T = (float*)(ws + L.T); float* U = (float*)(ws + L.U); float* rs = (float*)(ws + L.rs);
  float* dvp = (float*)(ws + L.dvp);
  launch_colmean(z, B, zdim, L.zp, log_var, mu, flag, stream);
  prior_cst_kernel<<<1, 64, 0, stream>>>(log_var, zdim, cstp);
  prior_prep_kernel<<<cdiv(L.ldp + C, 4), 256, 0, stream>>>(z, B, L.ldp, Zs, L.ldz, zn, centres, C, Cs, cn, zdim, L.zp, log_var,
                                                            mu, norm_limit, flag);
  int rc = check_launch("prior_prep_kernel");
  if (rc) return rc;
  GemmArgs g = {};
  g.ones_col = -1;
  g.A[0] = Cs; g.B[0] = Zs; g.lda[0] = L.zp; g.ldb[0] = L.ldz; g.Kc[0] = L.zp; g.npairs = 1;
  g.M = C; g.N = B; g.e0 = cn; g.e1 = zn; g.ksplit = 0;
  g.out0 = P; g.ldo = L.ldp; g.bias0 = lse; g.bias1 = gout;
  g.pr_ridx = c_idx; g.pr_cidx = z_idx; g.pr_cst_dev = cstp; g.skip_flag = flag;
  rc = launch_prior_gemm(g, B, false, stream);
  if (rc) return rc;
  // T [C x ldz] = P [C x ldp] . Zs1 [ldp x ldz] (P's padding columns and Zs1's padding rows are zero)
  rc = evae_dense_bwd_data(P, Zs, nullptr, nullptr, C, L.ldp, L.ldp, L.ldz, nullptr, nullptr, T, nullptr, L.ldz,
                           ws + L.ws_data, L.ws_data_bytes, (evae_stream_t)stream);
  if (rc) return rc;
  // U [B x zp] = P^T C', rs [B] = column sums of P^T's operand = row sums over the exemplars (the bias-gradient column)
  rc = evae_dense_bwd_weight(P, C, L.ldp, L.ldp, Cs, nullptr, L.zp, L.zp, U, rs, 0, ws + L.ws_weight, L.ws_weight_bytes,
                             (evae_stream_t)stream);
  if (rc) return rc;
  prior_gemm_bwd_rows_kernel<<<L.nblk, 256, (size_t)4 * (L.zp + 1) * sizeof(float), stream>>>(
      T, L.ldz, Cs, C, U, rs, Zs, L.ldz, B, zdim, L.zp, log_var, dc, dz, dvp, flag);
  prior_gemm_bwd_dlv_kernel<<<cdiv(zdim, 64), 1024, 0, stream>>>(dvp, L.nblk, zdim, L.zp, dlogvar, flag);
  return check_launch("prior_gemm_bwd finish");
}

}  // namespace evae
