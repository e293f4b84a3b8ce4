// Exemplar prior on gfx950: fused all-pairs distance + leave-one-out mask + online log-sum-exp,
// its backward by recomputation, and the shard merge.  See include/evae_hip.h for the contract
// and the reference lines replaced (utils/distributions.py:12-25, models/BaseModel.py:98-128).
//
// Kernel shape (z_dim = 40 is a thin contraction, so the distance runs on the VALU in the exact
// direct-difference form sum_k (z-c)^2, not as ||z||^2+||c||^2-2zc on MFMA):
//   block = 256 threads = 16 (exemplar lanes) x 16 (query groups); block tile = 128 queries x 64
//   exemplars; thread tile = 8 queries x 4 exemplars (32 fp32 accumulators).
//   The query tile and each exemplar tile are staged in LDS pre-divided by sigma, row-major with a
//   row stride of (kc + pad) floats chosen so stride/4 is odd: the 16 lanes of a ds_read_b128 group
//   read 16 different rows and land on all 64 banks (conflict-free); query reads are broadcasts.
//   Global reads of the [C x z] cache are flat float4 streams (each 64-row tile is one contiguous
//   10 KB span), i.e. fully coalesced.
//   Each thread keeps a running (dmin, sum exp(-(d-dmin)/2), #masked) per query; the 16 exemplar
//   lanes are combined with wave shuffles once per block; blocks (exemplar splits) are combined by
//   the merge kernel, which is the same code that merges GPU shards.
#include <algorithm>
#include <type_traits>
#include "evae_tile.h"
#include "evae_prior_gemm.h"

namespace evae {

// c_idx value that masks an exemplar slot for EVERY query (EVAE_PRIOR_MASK_ALL in evae_hip.h): a duplicate slot of the
// fixed-size exemplar list of the approximate prior -- excluded from the sum and counted in nmask like a leave-one-out hit
constexpr long long kMaskAll = EVAE_PRIOR_MASK_ALL;

// inv_sigma[k] = exp(-log_var[k]/2) (zero-padded to nchunk*kc); returns -1/2 sum_k (lv_k + log 2pi)
__device__ __forceinline__ float setup_sigma(float* __restrict__ inv_sigma, float* __restrict__ red,
                                             const float* __restrict__ log_var, int zdim, int zpad) {
  float part = 0.f;
  for (int k = threadIdx.x; k < zpad; k += NT) {
    float lv = k < zdim ? log_var[k] : 0.f;
    inv_sigma[k] = k < zdim ? expf(-0.5f * lv) : 0.f;
    if (k < zdim) part += lv + kLog2Pi;
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  float tot = red[0] + red[1] + red[2] + red[3];
  return -0.5f * tot;
}

// ------------------------------------------------------------------------------------------------
// forward: per (exemplar split, query tile) partial (max, sumexp, nmask)
// ------------------------------------------------------------------------------------------------
template <int KC>
__global__ __launch_bounds__(NT) void prior_fwd_kernel(
    const float* __restrict__ z, int B, const float* __restrict__ centres, int C, int zdim,
    const float* __restrict__ log_var, const int64_t* __restrict__ z_idx,
    const int64_t* __restrict__ c_idx, int tiles_per_split, int nsplit, PriorGeom g,
    float* __restrict__ part_m, float* __restrict__ part_s, float* __restrict__ part_n,
    float* __restrict__ out_prob, const unsigned* __restrict__ run_flag) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // fallback of the GEMM path (evae_prior_gemm.hip): runs only when that path's norm guard raised the flag
  if (run_flag != nullptr && *run_flag == 0u) return;
  float* Qs = smem;
  float* Es = Qs + BQ * Geom<KC>::ks;
  float* inv_sigma = Es + BE * Geom<KC>::ks;     // [ZDIM_MAX]
  float* red = inv_sigma + ZDIM_MAX;     // [16]

  const int split = blockIdx.x;
  const int q0 = blockIdx.y * BQ;
  const int te = threadIdx.x & 15;
  const int tq = threadIdx.x >> 4;
  const bool vec_ok = (zdim & 3) == 0 && (((uintptr_t)z | (uintptr_t)centres) & 15) == 0;
  const bool masked = (z_idx != nullptr) && (c_idx != nullptr);
  const int zpad = g.nchunk * KC;

  const float cst = setup_sigma(inv_sigma, red, log_var, zdim, zpad);

  int64_t zi[TQ];
  bool qvalid[TQ];
#pragma unroll
  for (int i = 0; i < TQ; ++i) {
    int q = q0 + tq + 16 * i;
    qvalid[i] = q < B;
    zi[i] = (masked && qvalid[i]) ? z_idx[q] : -1;
  }

  float dmin[TQ], ssum[TQ], nmask[TQ];
#pragma unroll
  for (int i = 0; i < TQ; ++i) { dmin[i] = INFINITY; ssum[i] = 0.f; nmask[i] = 0.f; }

  if (g.nchunk == 1) stage_rows<KC>(Qs, z, q0, B, BQ, zdim, 0, inv_sigma, vec_ok);

  const int tile_begin = split * tiles_per_split;
  const int ntiles = (C + BE - 1) / BE;
  int tile_end = tile_begin + tiles_per_split;
  if (tile_end > ntiles) tile_end = ntiles;

  for (int t = tile_begin; t < tile_end; ++t) {
    const int e0 = t * BE;
    float acc[TQ][TE];
#pragma unroll
    for (int i = 0; i < TQ; ++i)
#pragma unroll
      for (int j = 0; j < TE; ++j) acc[i][j] = 0.f;

    for (int ch = 0; ch < g.nchunk; ++ch) {
      __syncthreads();  // previous readers of Es (and Qs when re-staged) are done
      if (g.nchunk > 1)
        stage_rows<KC>(Qs, z, q0, B, BQ, zdim, ch * KC, inv_sigma + ch * KC, vec_ok);
      stage_rows<KC>(Es, centres, e0, C, BE, zdim, ch * KC, inv_sigma + ch * KC, vec_ok);
      __syncthreads();
      dist_chunk<KC>(acc, Qs, Es, tq, te);
    }

    int64_t ci[TE];
    bool evalid[TE];
#pragma unroll
    for (int j = 0; j < TE; ++j) {
      int e = e0 + te + 16 * j;
      evalid[j] = e < C;
      ci[j] = (masked && evalid[j]) ? c_idx[e] : -2;
    }
#pragma unroll
    for (int i = 0; i < TQ; ++i) {
      bool ok[TE];
      float cmin = INFINITY;
#pragma unroll
      for (int j = 0; j < TE; ++j) {
        bool hit = masked && (zi[i] == ci[j] || ci[j] == kMaskAll) && evalid[j];
        ok[j] = evalid[j] && !hit;
        if (hit) nmask[i] += 1.f;
        if (ok[j]) cmin = fminf(cmin, acc[i][j]);
      }
      if (cmin < dmin[i]) {
        ssum[i] *= fast_exp2((cmin - dmin[i]) * kHalfLog2e);   // dmin==inf -> 0*0 = 0
        dmin[i] = cmin;
      }
#pragma unroll
      for (int j = 0; j < TE; ++j)
        if (ok[j]) ssum[i] += fast_exp2((dmin[i] - acc[i][j]) * kHalfLog2e);
      if (out_prob != nullptr && qvalid[i]) {
        size_t rowoff = (size_t)(q0 + tq + 16 * i) * C;
#pragma unroll
        for (int j = 0; j < TE; ++j)
          if (evalid[j]) out_prob[rowoff + e0 + te + 16 * j] = ok[j] ? cst - 0.5f * acc[i][j] : -INFINITY;
      }
    }
  }

  // combine the 16 exemplar lanes (te = lane & 15) of each query group
#pragma unroll
  for (int i = 0; i < TQ; ++i) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      float od = __shfl_xor(dmin[i], o, 64);
      float os = __shfl_xor(ssum[i], o, 64);
      float on = __shfl_xor(nmask[i], o, 64);
      float m = fminf(dmin[i], od);
      float fa = (dmin[i] == m) ? 1.f : fast_exp2((m - dmin[i]) * kHalfLog2e);
      float fb = (od == m) ? 1.f : fast_exp2((m - od) * kHalfLog2e);
      ssum[i] = ssum[i] * fa + os * fb;
      dmin[i] = m;
      nmask[i] += on;
    }
    if (te == 0 && qvalid[i]) {
      size_t o = (size_t)split * B + (q0 + tq + 16 * i);
      part_m[o] = (dmin[i] == INFINITY) ? -INFINITY : cst - 0.5f * dmin[i];
      part_s[o] = ssum[i];
      part_n[o] = nmask[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward on the matrix cores (z_dim <= 64, multiple of 4, no probability matrix requested):
//   d2_ij = |z_i'|^2 + |c_j'|^2 - 2 z_i'.c_j' with x' = x/s - mu, the dot products on v_mfma_f32_32x32x2_f32.
//   This is the reference's own formulation (utils/distributions.py:12-18 expands the square, in fp64).  In fp32 the
//   cancellation costs ~2e-7 (|z'|^2 + |c'|^2) absolute on d2/2, so two things keep it inside the 1e-5 bar whatever
//   the latents look like:
//     * centring -- the distance is translation invariant, and every block subtracts the mean mu of ITS query tile
//       (in sigma units) from the queries and from every exemplar it stages: a common offset of the latent cloud no
//       longer enters the norms;
//     * a guard -- only the QUERY's centred norm enters the error that matters: an exemplar far from the tile mean is
//       as far from the query (d >= |c'| - |z'|), its error eps |c'|^2 is a relative error eps of its own d2, so
//       |err(log p)| <~ eps (1.5 |z'|^2 + 2 |log p|).  A block whose max|z'|^2 exceeds kPriorNormLimit (a query tile
//       spread widely at a small prior variance) computes its tiles as direct differences sum_k (z'-c')^2 on the VALU
//       from the staged tiles instead (no cancellation; ~3x the time of a matrix-core tile).  Block-uniform.
//   The running log-sum-exp state of a query is kept on u = -d2/2 (the MFMA epilogue works on t = s - |c'|^2/2 =
//   u + |z'|^2/2: the query's norm drops out of every difference).
//   Block = 512 threads = 8 waves (4 x 2), tile = 128 exemplars (MFMA rows) x 128 queries (MFMA columns):
//   a lane owns ONE query column of each 32 x 32 result tile and 16 exemplar rows of it in registers, so
//   the running (max, sum exp, #masked) of a query is lane-local; lanes l / l+32 and the 4 wave rows are
//   combined once per block.  Scaled tiles are staged row-major [row][8 KG + 4] (stride/4 odd ->
//   conflict-free ds_read_b128 fragments); the next exemplar tile is prefetched into registers.
//   ~5x the direct-difference VALU kernel at S = 5000 importance samples x 50 000 exemplars.
typedef float f32x16_t __attribute__((ext_vector_type(16)));
constexpr int MFQ = 128, MFE = 128, MFT = 512;
// max|z'|^2 (centred, sigma units) of a query tile above which the expanded form is not trusted: the fp32 cancellation
// error on log p is ~3e-7 of it, i.e. < 1e-3 absolute -- 1e-5 of a log-density of magnitude ~1e2
constexpr float kPriorNormLimit = 2048.f;

// Centre the staged (sigma-scaled) query tile: mu[k] = mean over its nvalid live rows, subtracted from those rows in place;
// then the squared row norms and their maximum over the tile.  `scratch` = 512 floats of LDS nobody uses yet.
// Fixed reduction order.  Contains barriers: every thread of the block calls it.
template <int KP, int KS2, int NTH>
__device__ __forceinline__ float centre_queries(float* __restrict__ Qs, float* __restrict__ mu_s, float* __restrict__ zn,
                                                float* __restrict__ zmx, float* __restrict__ scratch, int nvalid) {
  const int tid = threadIdx.x, col = tid & 63, part = tid >> 6;
  if (col < KP) {
    float a = 0.f;
    constexpr int RP = MFQ / (NTH / 64);          // rows per wave
#pragma unroll 4
    for (int r = part * RP; r < part * RP + RP; ++r) a += (r < nvalid) ? Qs[r * KS2 + col] : 0.f;
    scratch[part * 64 + col] = a;
  }
  __syncthreads();
  if (tid < KP) {
    float a = 0.f;
#pragma unroll
    for (int p = 0; p < NTH / 64; ++p) a += scratch[p * 64 + tid];
    mu_s[tid] = a / (float)nvalid;
  }
  __syncthreads();
  {
    constexpr int CPR = KP / 4;
    for (int f = tid; f < MFQ * CPR; f += NTH) {
      const int r = f / CPR, c = f - r * CPR;
      if (r < nvalid) {
        float4 v = *reinterpret_cast<float4*>(Qs + r * KS2 + 4 * c);
        const float4 m = *reinterpret_cast<const float4*>(mu_s + 4 * c);
        v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w;
        *reinterpret_cast<float4*>(Qs + r * KS2 + 4 * c) = v;
      }
    }
  }
  __syncthreads();
  if (tid < MFQ) {
    constexpr int CPR = KP / 4;
    float sacc = 0.f;
#pragma unroll
    for (int c = 0; c < CPR; ++c) {
      const float4 t = *reinterpret_cast<const float4*>(Qs + tid * KS2 + 4 * c);
      sacc += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
    }
    zn[tid] = sacc;
    const float wm = wave_max(sacc);
    if ((tid & 63) == 0) zmx[tid >> 6] = wm;
  }
  __syncthreads();
  return fmaxf(zmx[0], zmx[1]);
}

// a[nt][r] <- sum_k (Es[e(r)][k] - Qs[q(nt)][k])^2 from the staged, centred tiles (the guard's direct-difference path):
// rows of this lane e = wr*32 + (r&3) + 8*(r>>2) + 4*lh (half-wave broadcast reads), columns q = wc*64 + nt*32 + l31.
template <int KP, int KS2>
__device__ __forceinline__ void direct_tile(f32x16_t (&a)[2], const float* __restrict__ Es, const float* __restrict__ Qs,
                                            int wr, int wc, int l31, int lh) {
#pragma unroll
  for (int r = 0; r < 16; ++r) { a[0][r] = 0.f; a[1][r] = 0.f; }
  const float* q0p = Qs + (wc * 64 + l31) * KS2;
  const float* q1p = q0p + 32 * KS2;
  const float* ep = Es + (wr * 32 + 4 * lh) * KS2;
  for (int c = 0; c < KP / 4; ++c) {
    const float4 q0 = *reinterpret_cast<const float4*>(q0p + 4 * c);
    const float4 q1 = *reinterpret_cast<const float4*>(q1p + 4 * c);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 e = *reinterpret_cast<const float4*>(ep + ((r & 3) + 8 * (r >> 2)) * KS2 + 4 * c);
      float d, s0 = a[0][r], s1 = a[1][r];
      d = e.x - q0.x; s0 = fmaf(d, d, s0);
      d = e.y - q0.y; s0 = fmaf(d, d, s0);
      d = e.z - q0.z; s0 = fmaf(d, d, s0);
      d = e.w - q0.w; s0 = fmaf(d, d, s0);
      d = e.x - q1.x; s1 = fmaf(d, d, s1);
      d = e.y - q1.y; s1 = fmaf(d, d, s1);
      d = e.z - q1.z; s1 = fmaf(d, d, s1);
      d = e.w - q1.w; s1 = fmaf(d, d, s1);
      a[0][r] = s0; a[1][r] = s1;
    }
  }
}

// WR = wave rows of a block: 4 -> 8 waves, tile 128 exemplars x 128 queries; 2 -> 4 waves, tile 64 x 128 -- three of those
// blocks fit a CU (the kernel needs ~168 VGPRs: 3 waves per SIMD), and blocks in different phases are what overlaps one
// block's exp / log-sum-exp epilogue (VALU) with another's products (matrix pipe)
template <int KG, int WR>
__global__ __launch_bounds__(128 * WR) void prior_fwd_mfma_kernel(
    const float* __restrict__ z, int B, const float* __restrict__ centres, int C, int zdim,
    const float* __restrict__ log_var, const int64_t* __restrict__ z_idx,
    const int64_t* __restrict__ c_idx, int tiles_per_split, float norm_limit,
    float* __restrict__ part_m, float* __restrict__ part_s, float* __restrict__ part_n) {
  constexpr int KP = KG * 8, KS2 = KP + 4, CPR = KP / 4;       // chunks (float4) per row
  constexpr int FE = 32 * WR, FT = 128 * WR;                   // exemplars per tile, threads per block
  constexpr int NV = (FE * CPR + FT - 1) / FT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;                          // [128][KS2]
  float* Es = Qs + MFQ * KS2;                // [128][KS2]
  float* zn = Es + FE * KS2;                // [128]
  float* cn = zn + MFQ;                      // [2][128]
  float* inv_sigma = cn + 2 * FE;           // [64]
  float* mu_s = inv_sigma + 64;              // [64]  mean of the query tile (sigma units)
  float* red = mu_s + 64;                    // [16]
  float* zmx = red + 16;                     // [2] (+6 padding)
  long long* ci_s = reinterpret_cast<long long*>(zmx + 8);             // [2][128]
  float* Es1 = reinterpret_cast<float*>(ci_s + 2 * FE);                // [128][KS2] second exemplar tile (pipelined loop)
  // the cross-wave combine buffer [4][128][3] is only needed after the last tile: it reuses the exemplar tile, which
  // keeps the block at 48 KB of LDS -- three blocks (24 waves) per CU instead of two, and it is waves of OTHER blocks
  // that fill a SIMD while one block sits in its exp / log-sum-exp epilogue
  float* comb = Es;
  static_assert(WR * MFQ * 3 <= FE * KS2, "combine buffer must fit in the exemplar tile");
  static_assert(512 <= FE * KS2, "centre_queries scratch must fit in the exemplar tile");

  const int split = blockIdx.x;
  const int q0 = blockIdx.y * MFQ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, lh = lane >> 5;
  const bool masked = (z_idx != nullptr) && (c_idx != nullptr);
  const float cst = setup_sigma(inv_sigma, red, log_var, zdim, KP);   // contains a barrier

  // stage one 128-row tile of `src` (rows r0.., nrows total), scaled (and centred), zero-padded
  auto load_tile = [&](const float* src, int r0, int nrows, float4 (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + FT * i;
      const int r = f / CPR, c = f - r * CPR;
      const bool ok = f < FE * CPR && r0 + r < nrows && 4 * c + 4 <= zdim;
      v[i] = ok ? *reinterpret_cast<const float4*>(src + (size_t)(r0 + r) * zdim + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&](float* tile, const float4 (&v)[NV], const bool centre) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + FT * i;
      const int r = f / CPR, c = f - r * CPR;
      if (f < FE * CPR) {
        const float4 s4 = *reinterpret_cast<const float4*>(inv_sigma + 4 * c);
        float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (centre) m4 = *reinterpret_cast<const float4*>(mu_s + 4 * c);
        *reinterpret_cast<float4*>(tile + r * KS2 + 4 * c) =
            make_float4(fmaf(v[i].x, s4.x, -m4.x), fmaf(v[i].y, s4.y, -m4.y), fmaf(v[i].z, s4.z, -m4.z), fmaf(v[i].w, s4.w, -m4.w));
      }
    }
  };
  auto row_norms = [&](const float* tile, float* out) {     // threads 0..127, one row each
    if (tid < FE) {
      float sacc = 0.f;
#pragma unroll
      for (int c = 0; c < CPR; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(tile + tid * KS2 + 4 * c);
        sacc += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
      }
      out[tid] = sacc;
    }
  };

  float4 rv[NV];
  load_tile(z, q0, B, rv);
  store_tile(Qs, rv, false);
  const int ntiles = (C + FE - 1) / FE;
  const int tile_begin = split * tiles_per_split;
  int tile_end = tile_begin + tiles_per_split;
  if (tile_end > ntiles) tile_end = ntiles;
  if (tile_begin < tile_end) load_tile(centres, tile_begin * FE, C, rv);
  __syncthreads();
  // the guard (block-uniform): queries too far from their tile mean for the expanded form -> direct differences
  const bool slow_block = centre_queries<KP, KS2, FT>(Qs, mu_s, zn, zmx, Es, (B - q0) < MFQ ? (B - q0) : MFQ) > norm_limit;

  // this lane's two query columns
  float hz[2];
  long long zi[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int ql = wc * 64 + nt * 32 + l31;
    hz[nt] = 0.5f * zn[ql];
    zi[nt] = (masked && q0 + ql < B) ? (long long)z_idx[q0 + ql] : -1;
  }
  float dmin[2], ssum[2] = {0.f, 0.f}, nmask[2] = {0.f, 0.f};
  float um[2] = {-INFINITY, -INFINITY};        // running max of u = -d2/2 per query column

  // the tile loop exists twice, with the guard's verdict as a compile-time constant inside (the epilogue is bound by VALU
  // issue: not one select on `slow` may survive in the matrix-core variant)
  auto tile_loop = [&](auto SLOW_) {
  constexpr bool slow = decltype(SLOW_)::value;
  for (int t = tile_begin; t < tile_end; ++t) {
    const int e0 = t * FE;
    const int pb = (t - tile_begin) & 1;
    store_tile(Es, rv, true);
    if (masked && tid < FE) ci_s[pb * FE + tid] = (e0 + tid < C) ? (long long)c_idx[e0 + tid] : -2;
    __syncthreads();
    if (t + 1 < tile_end) load_tile(centres, (t + 1) * FE, C, rv);
    if constexpr (!slow) row_norms(Es, cn + pb * FE);

    f32x16_t acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    if constexpr (!slow) {
      const float* ea = Es + (wr * 32 + l31) * KS2 + lh * 4;
      const float* qb = Qs + (wc * 64 + l31) * KS2 + lh * 4;
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) {
        const float4 a = *reinterpret_cast<const float4*>(ea + kg * 8);
        const float4 b0 = *reinterpret_cast<const float4*>(qb + kg * 8);
        const float4 b1 = *reinterpret_cast<const float4*>(qb + 32 * KS2 + kg * 8);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc[1], 0, 0, 0);
      }
    } else {
      direct_tile<KP, KS2>(acc, Es, Qs, wr, wc, l31, lh);      // acc <- d2
    }
    __syncthreads();     // cn (and ci_s) of this tile are complete; every wave is done reading Es

    // rows of this lane: e = wr*32 + (r&3) + 8*(r>>2) + 4*lh; bit r of `live` = that exemplar exists.
    // The epilogue works on v with u = -d2/2 = v - off: matrix-core tiles v = t = s - |c'|^2/2 (s = the dot product), off =
    // |z'|^2/2, so the query's norm drops out of every difference and an element costs sub, max, fma, exp2, add -- VALU
    // issue is what this kernel is bound by; direct tiles v = -d2/2, off = 0.
    float hc[16];
    unsigned live = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int el = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      hc[r] = slow ? 0.f : 0.5f * cn[pb * FE + el];
      if (e0 + el < C) live |= 1u << r;
    }
    if constexpr (slow) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] *= -0.5f; acc[1][r] *= -0.5f; }
    }
    if (!masked && e0 + FE <= C) {       // whole tile present, nothing to mask: no per-element predicates
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float off = slow ? 0.f : hz[nt];
        float v[16];
        float vmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = acc[nt][r] - hc[r];
          vmax = fmaxf(vmax, v[r]);
        }
        const float ut = vmax - off;
        if (ut > um[nt]) {
          ssum[nt] *= fast_exp2((um[nt] - ut) * kLog2e);      // um == -inf -> 0 * 0 = 0
          um[nt] = ut;
        }
        const float mk = -(um[nt] + off) * kLog2e;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // direct tiles: the difference FIRST (exact near the maximum) -- fma(v, log2e, -max log2e) carries the rounding of
          // max log2e, half an ulp of 1e8 .. 1e9 = 8 .. 128 in the exponent at the magnitudes this path exists for (r03)
          if constexpr (slow) ssum[nt] += fast_exp2((v[r] - um[nt]) * kLog2e);
          else ssum[nt] += fast_exp2(fmaf(v[r], kLog2e, mk));
        }
      }
      continue;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float off = slow ? 0.f : hz[nt];
      float v[16];
      unsigned use = live;
      float vmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r] = acc[nt][r] - hc[r];
        if (masked) {
          const int el = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const long long ce = ci_s[pb * FE + el];
          if (((use >> r) & 1u) && (ce == zi[nt] || ce == kMaskAll)) { nmask[nt] += 1.f; use &= ~(1u << r); }
        }
        if ((use >> r) & 1u) vmax = fmaxf(vmax, v[r]);
      }
      const float ut = vmax - off;
      if (ut > um[nt]) {
        ssum[nt] *= fast_exp2((um[nt] - ut) * kLog2e);
        um[nt] = ut;
      }
      const float mk = -(um[nt] + off) * kLog2e;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if ((use >> r) & 1u) {
          if constexpr (slow) ssum[nt] += fast_exp2((v[r] - um[nt]) * kLog2e);
          else ssum[nt] += fast_exp2(fmaf(v[r], kLog2e, mk));
        }
    }
  }
  };
  // Unmasked calls with several tiles per block (the IWAE evaluator: thousands of importance samples against every exemplar)
  // run the loop SOFTWARE-PIPELINED: the products of tile t+1 are issued into a second accumulator set while the epilogue of
  // tile t (sub, max, fma, exp2, add per pair -- as much VALU time as the products take on the matrix pipe) runs on the
  // first, from a second exemplar buffer in LDS; one barrier per tile.  Only the last tile of the range can be ragged.
  auto pipe_loop = [&]() {
    auto mfma_tile = [&](f32x16_t (&a)[2], const float* E) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { a[0][r] = 0.f; a[1][r] = 0.f; }
      const float* ea = E + (wr * 32 + l31) * KS2 + lh * 4;
      const float* qb = Qs + (wc * 64 + l31) * KS2 + lh * 4;
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) {
        const float4 x = *reinterpret_cast<const float4*>(ea + kg * 8);
        const float4 b0 = *reinterpret_cast<const float4*>(qb + kg * 8);
        const float4 b1 = *reinterpret_cast<const float4*>(qb + 32 * KS2 + kg * 8);
        a[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, b0.x, a[0], 0, 0, 0);
        a[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, b1.x, a[1], 0, 0, 0);
        a[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, b0.y, a[0], 0, 0, 0);
        a[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, b1.y, a[1], 0, 0, 0);
        a[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, b0.z, a[0], 0, 0, 0);
        a[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, b1.z, a[1], 0, 0, 0);
        a[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, b0.w, a[0], 0, 0, 0);
        a[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, b1.w, a[1], 0, 0, 0);
      }
    };
    // online log-sum-exp update from one tile's products; branch-free when the whole tile exists
    auto epilogue = [&](const f32x16_t (&a)[2], const float* cnp, const int e0) {
      float hc[16];
      unsigned live = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int el = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        hc[r] = 0.5f * cnp[el];
        if (e0 + el < C) live |= 1u << r;
      }
      const bool full = e0 + FE <= C;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float v[16];
        float vmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = a[nt][r] - hc[r];
          vmax = fmaxf(vmax, (full || ((live >> r) & 1u)) ? v[r] : -INFINITY);
        }
        const float newm = fmaxf(um[nt], vmax - hz[nt]);
        if (newm != -INFINITY) {                  // (a ragged tile can leave a lane group without a live row)
          ssum[nt] *= fast_exp2((um[nt] - newm) * kLog2e);      // um == -inf -> 0 * 0
          um[nt] = newm;
          const float mk = -(newm + hz[nt]) * kLog2e;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e = fast_exp2(fmaf(v[r], kLog2e, mk));
            ssum[nt] += (full || ((live >> r) & 1u)) ? e : 0.f;
          }
        }
      }
    };
    // the same for a tile that is known to be whole: straight-line code, no predicates
    auto epilogue_full = [&](const f32x16_t (&a)[2], const float* cnp) {
      float hc[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) hc[r] = 0.5f * cnp[wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float v[16];
        float vmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = a[nt][r] - hc[r];
          vmax = fmaxf(vmax, v[r]);
        }
        const float newm = fmaxf(um[nt], vmax - hz[nt]);
        ssum[nt] *= fast_exp2((um[nt] - newm) * kLog2e);        // um == -inf -> 0 * 0
        um[nt] = newm;
        const float mk = -(newm + hz[nt]) * kLog2e;
#pragma unroll
        for (int r = 0; r < 16; ++r) ssum[nt] += fast_exp2(fmaf(v[r], kLog2e, mk));
      }
    };
    f32x16_t accA[2], accB[2];
    float* Eb[2] = {Es, Es1};
    // tile tile_begin: staged and multiplied up front
    store_tile(Eb[0], rv, true);
    __syncthreads();
    if (tile_begin + 1 < tile_end) load_tile(centres, (tile_begin + 1) * FE, C, rv);
    row_norms(Eb[0], cn);
    mfma_tile(accA, Eb[0]);
    // one pipelined step: tile t (whole, products in `cur`) gets its epilogue while tile t+1 (exists) is multiplied into `nxt`
    auto step = [&](f32x16_t (&cur)[2], f32x16_t (&nxt)[2], const int t) {
      const int pb = (t - tile_begin) & 1;
      store_tile(Eb[pb ^ 1], rv, true);
      __syncthreads();           // tile t+1 is staged; the norms of tile t are complete; every wave is done with buffer pb^1's old tile
      if (t + 2 < tile_end) load_tile(centres, (t + 2) * FE, C, rv);
      row_norms(Eb[pb ^ 1], cn + (pb ^ 1) * FE);
      // From here on ONE basic block in which the 8 KG matrix instructions of tile t+1 and the epilogue of tile t (16 x sub /
      // max, the rescale, 32 x fma / exp2 / add) alternate instruction by instruction: an MFMA occupies the matrix pipe for 64
      // cycles, and what the wave issues behind it runs on the vector ALU meanwhile.  The order is pinned with scheduling
      // barriers (the machine scheduler otherwise clusters the MFMAs -- into two DEPENDENT chains -- and appends the VALU work).
      const float* cnp = cn + pb * FE;
      const float* ea = Eb[pb ^ 1] + (wr * 32 + l31) * KS2 + lh * 4;
      const float* qb = Qs + (wc * 64 + l31) * KS2 + lh * 4;
      float hc[16], v0[16], v1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) hc[r] = 0.5f * cnp[wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
      float4 fa = *reinterpret_cast<const float4*>(ea), fb0 = *reinterpret_cast<const float4*>(qb),
             fb1 = *reinterpret_cast<const float4*>(qb + 32 * KS2);
      float4 ga = fa, gb0 = fb0, gb1 = fb1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { nxt[0][r] = 0.f; nxt[1][r] = 0.f; }
      float vmax0 = -INFINITY, vmax1 = -INFINITY, mk0 = 0.f, mk1 = 0.f;
      constexpr int NS = 8 * KG, NP = 33, PPS = (NP + NS - 1) / NS;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int kg = i >> 3, j = i & 7;
        if (j == 0 && kg + 1 < KG) {             // fragments of the next k-group, one group ahead
          ga = *reinterpret_cast<const float4*>(ea + (kg + 1) * 8);
          gb0 = *reinterpret_cast<const float4*>(qb + (kg + 1) * 8);
          gb1 = *reinterpret_cast<const float4*>(qb + 32 * KS2 + (kg + 1) * 8);
        }
        const float av = (j >> 1) == 0 ? fa.x : (j >> 1) == 1 ? fa.y : (j >> 1) == 2 ? fa.z : fa.w;
        const float4 fb = (j & 1) ? fb1 : fb0;
        const float bv = (j >> 1) == 0 ? fb.x : (j >> 1) == 1 ? fb.y : (j >> 1) == 2 ? fb.z : fb.w;
        nxt[j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, nxt[j & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = i * PPS; p < (i + 1) * PPS; ++p) {
          if (p < 16) {
            v0[p] = cur[0][p] - hc[p]; v1[p] = cur[1][p] - hc[p];
            vmax0 = fmaxf(vmax0, v0[p]); vmax1 = fmaxf(vmax1, v1[p]);
          } else if (p == 16) {
            const float n0 = fmaxf(um[0], vmax0 - hz[0]), n1 = fmaxf(um[1], vmax1 - hz[1]);
            ssum[0] *= fast_exp2((um[0] - n0) * kLog2e); ssum[1] *= fast_exp2((um[1] - n1) * kLog2e);     // um == -inf -> 0 * 0
            um[0] = n0; um[1] = n1;
            mk0 = -(n0 + hz[0]) * kLog2e; mk1 = -(n1 + hz[1]) * kLog2e;
          } else if (p < NP) {
            ssum[0] += fast_exp2(fmaf(v0[p - 17], kLog2e, mk0));
            ssum[1] += fast_exp2(fmaf(v1[p - 17], kLog2e, mk1));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (j == 7) { fa = ga; fb0 = gb0; fb1 = gb1; }
      }
    };
    int t = tile_begin;
    for (; t + 2 < tile_end; t += 2) {
      step(accA, accB, t);
      step(accB, accA, t + 1);
    }
    bool last_in_b = false;
    if (t + 1 < tile_end) { step(accA, accB, t); ++t; last_in_b = true; }
    __syncthreads();             // the norms of the last tile are complete
    if (last_in_b) epilogue(accB, cn + ((t - tile_begin) & 1) * FE, t * FE);
    else epilogue(accA, cn + ((t - tile_begin) & 1) * FE, t * FE);
    __syncthreads();             // comb aliases the first exemplar buffer
  };
  if (!masked && !slow_block && tile_end - tile_begin >= 2) pipe_loop();
  else if (slow_block) tile_loop(std::true_type{});
  else tile_loop(std::false_type{});
  // back to squared distances for the combine below: d2_min = -2 u_max  (no live exemplar: +inf)
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) dmin[nt] = (um[nt] == -INFINITY) ? INFINITY : fmaxf(-2.0f * um[nt], 0.f);

  // combine lanes l and l+32 (same query, other rows), then the four wave rows through LDS
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const float od = __shfl_xor(dmin[nt], 32, 64), os = __shfl_xor(ssum[nt], 32, 64), on = __shfl_xor(nmask[nt], 32, 64);
    const float m = fminf(dmin[nt], od);
    const float fa = (dmin[nt] == m) ? 1.f : fast_exp2((m - dmin[nt]) * kHalfLog2e);
    const float fb = (od == m) ? 1.f : fast_exp2((m - od) * kHalfLog2e);
    ssum[nt] = ssum[nt] * fa + os * fb;
    dmin[nt] = m;
    nmask[nt] += on;
    if (lh == 0) {
      float* cb = comb + (wr * MFQ + wc * 64 + nt * 32 + l31) * 3;
      cb[0] = dmin[nt]; cb[1] = ssum[nt]; cb[2] = nmask[nt];
    }
  }
  __syncthreads();
  if (tid < MFQ && q0 + tid < B) {
    float m = INFINITY, sacc = 0.f, nacc = 0.f;
#pragma unroll
    for (int w = 0; w < WR; ++w) m = fminf(m, comb[(w * MFQ + tid) * 3]);
#pragma unroll
    for (int w = 0; w < WR; ++w) {
      const float dw = comb[(w * MFQ + tid) * 3];
      if (dw != INFINITY) sacc += comb[(w * MFQ + tid) * 3 + 1] * fast_exp2((m - dw) * kHalfLog2e);
      nacc += comb[(w * MFQ + tid) * 3 + 2];
    }
    const size_t o = (size_t)split * B + (q0 + tid);
    part_m[o] = (m == INFINITY) ? -INFINITY : cst - 0.5f * m;
    part_s[o] = sacc;
    part_n[o] = nacc;
  }
}

// evae_prior_set_norm_limit overrides kPriorNormLimit (tests: 0 sends every tile through the direct-difference path)
static float g_norm_limit = kPriorNormLimit;
static float prior_norm_limit() { return g_norm_limit; }

template <int KG, int WR>
static int launch_prior_mfma_w(const float* z, int B, const float* centres, int C, int zdim, const float* log_var,
                               const int64_t* z_idx, const int64_t* c_idx, int ns_max, float* pm, float* ps, float* pn,
                               int* ns_out, hipStream_t stream) {
  constexpr int KS2 = KG * 8 + 4, FE = 32 * WR;
  const size_t lds = (size_t)((128 + 2 * FE) * KS2 + 128 + 2 * FE + 64 + 64 + 16 + 8) * sizeof(float) + 2 * FE * sizeof(long long);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)prior_fwd_mfma_kernel<KG, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int nq = cdiv(B, MFQ), ntiles = cdiv(C, FE);
  int ns = cdiv(WR == 4 ? 512 : 1536, nq);          // two rounds of the blocks a chip holds
  if (ns > ntiles) ns = ntiles;
  if (ns > ns_max) ns = ns_max;
  if (ns < 1) ns = 1;
  const int tps = cdiv(ntiles, ns);
  ns = cdiv(ntiles, tps);
  *ns_out = ns;
  prior_fwd_mfma_kernel<KG, WR><<<dim3(ns, nq), 128 * WR, lds, stream>>>(z, B, centres, C, zdim, log_var, z_idx, c_idx, tps,
                                                                       prior_norm_limit(), pm, ps, pn);
  return check_launch("prior_fwd_mfma_kernel");
}

template <int KG>
static int launch_prior_mfma(const float* z, int B, const float* centres, int C, int zdim, const float* log_var,
                             const int64_t* z_idx, const int64_t* c_idx, int ns_max, float* pm, float* ps, float* pn,
                             int* ns_out, hipStream_t stream) {
  // 8-wave blocks (WR = 4).  Measured (r02, S = 20 000 x C = 50 000 x z = 40): 1.18 ms; 4-wave blocks with 64-exemplar tiles,
  // three of them per CU: 1.26 ms -- co-resident blocks in different phases did not buy back the halved tile.
  return launch_prior_mfma_w<KG, 4>(z, B, centres, C, zdim, log_var, z_idx, c_idx, ns_max, pm, ps, pn, ns_out, stream);
}

// ------------------------------------------------------------------------------------------------
// merge: one wave per row combines R partials.  finalize=0 -> (max, sumexp, nmask);
// finalize=1 -> logprior = lse - log(c_total - nmask) and lse.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void prior_merge_kernel(const float* __restrict__ pm,
                                                         const float* __restrict__ ps,
                                                         const float* __restrict__ pn, int R, int B,
                                                         int finalize, float c_total,
                                                         float* __restrict__ o0, float* __restrict__ o1,
                                                         float* __restrict__ o2) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= B) return;
  float m = -INFINITY;
  for (int r = lane; r < R; r += 64) m = fmaxf(m, pm[(size_t)r * B + row]);
  m = wave_max(m);
  float s = 0.f, n = 0.f;
  for (int r = lane; r < R; r += 64) {
    float mr = pm[(size_t)r * B + row];
    float sr = ps[(size_t)r * B + row];
    if (mr != -INFINITY) s += sr * expf(mr - m);
    n += pn[(size_t)r * B + row];
  }
  s = wave_sum(s);
  n = wave_sum(n);
  if (lane == 0) {
    if (finalize) {
      const float ls = logf(s);
      o0[row] = (m + ls) - logf(c_total - n);
      // the forward -> backward token, 2 B floats: row max and log of the normalised sum, kept APART (lse = tok0 + tok1).
      // The backward forms (p_ij - tok0) - tok1: p_ij - tok0 is exact for the pairs that matter (p_ij is the same
      // single-rounded cst - d2/2 the forward took its maximum over), so the softmax weights stay normalised at any
      // magnitude of the log-density; the rounded sum m + log s lost ulp(lse) -- 8 nats at |lse| ~ 1e8 (r03, golden G21).
      if (o1 != nullptr) { o1[row] = m; o1[B + row] = (m == -INFINITY) ? 0.f : ls; }
    } else {
      o0[row] = m; o1[row] = s; o2[row] = n;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Tail of the training step's forward in ONE launch (single block, 16 waves): merge R partial rows per query (the splits
// of one device, or the gathered shard partials), logp = lse - log(c_total - nmask), KL = logq - logp, loss = beta KL - RE,
// and the three batch means (models/BaseModel.py:71-75,124-125).  One wave per row, rows strided over the waves.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void prior_elbo_fwd_kernel(const float* __restrict__ pm, const float* __restrict__ ps,
                                                              const float* __restrict__ pn, int R, int ldp, int B,
                                                              float c_total, const float* __restrict__ RE,
                                                              const float* __restrict__ logq,
                                                              const float* __restrict__ beta_dev, float beta_host,
                                                              float* __restrict__ logp, float* __restrict__ lse_out,
                                                              float* __restrict__ loss, float* __restrict__ KL,
                                                              float* __restrict__ means, float* __restrict__ cRE,
                                                              float* __restrict__ cKL, float* __restrict__ neg_cKL) {
  // thread = (query row within a chunk of 128, one of 8 groups of partial rows): consecutive lanes read consecutive
  // queries of one partial row (coalesced), every thread keeps an online (max, sum exp, #masked) over its rows r = g, g+8, ..
  __shared__ float cm[8][128], cs[8][128], cn[8][128];
  __shared__ float red[3][2];
  const int q = threadIdx.x & 127, grp = threadIdx.x >> 7;
  const float beta = beta_dev ? beta_dev[0] : beta_host;
  float sl = 0.f, sr = 0.f, sk = 0.f;          // threads of group 0 accumulate their rows
  for (int row0 = 0; row0 < B; row0 += 128) {
    const int row = row0 + q;
    float m = -INFINITY, s = 0.f, n = 0.f;
    if (row < B) {
      // four partial rows per round, their loads issued together: the merge is a dependent chain, and one global-memory
      // round trip per partial row was most of this launch (18 us at 196 splits)
      for (int r0 = grp; r0 < R; r0 += 32) {
        float mr[4], sr_[4], nr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = r0 + 8 * u;
          const bool ok = r < R;
          const size_t o = (size_t)(ok ? r : grp) * ldp + row;
          mr[u] = ok ? pm[o] : -INFINITY; sr_[u] = ok ? ps[o] : 0.f; nr[u] = ok ? pn[o] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {                                        // same order as a one-by-one walk
          n += nr[u];
          if (mr[u] > m) { s = s * expf(m - mr[u]) + sr_[u]; m = mr[u]; }      // m == -inf: s == 0 and exp(-inf) = 0
          else if (mr[u] != -INFINITY) s += sr_[u] * expf(mr[u] - m);
        }
      }
    }
    cm[grp][q] = m; cs[grp][q] = s; cn[grp][q] = n;
    __syncthreads();
    if (grp == 0 && row < B) {
      float mm = -INFINITY, ss = 0.f, nn = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) mm = fmaxf(mm, cm[g][q]);
#pragma unroll
      for (int g = 0; g < 8; ++g) {                // fixed order
        if (cm[g][q] != -INFINITY) ss += cs[g][q] * expf(cm[g][q] - mm);
        nn += cn[g][q];
      }
      const float ls = logf(ss);
      const float lse = mm + ls;
      const float lp = lse - logf(c_total - nn);
      logp[row] = lp;
      if (lse_out) { lse_out[row] = mm; lse_out[B + row] = (mm == -INFINITY) ? 0.f : ls; }      // token (max, log sum): prior_merge_kernel
      const float kl = logq[row] - lp;
      const float l = beta * kl - RE[row];
      KL[row] = kl;
      loss[row] = l;
      if (cRE != nullptr) {      // backward coefficients of "the batch mean of the loss, upstream gradient 1" (evae_elbo_bwd)
        const float gl = 1.0f / (float)B;
        cRE[row] = 0.f - gl; cKL[row] = 0.f + beta * gl; neg_cKL[row] = -(0.f + beta * gl);
      }
      sl += l; sr += RE[row]; sk += kl;
    }
    __syncthreads();
  }
  if (means == nullptr) return;
  if (grp == 0) {                                  // waves 0 and 1
    sl = wave_sum(sl); sr = wave_sum(sr); sk = wave_sum(sk);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sl; red[1][threadIdx.x >> 6] = sr; red[2][threadIdx.x >> 6] = sk; }
  }
  __syncthreads();
  if (threadIdx.x < 3) means[threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) / (float)B;
}

// ------------------------------------------------------------------------------------------------
// The same tail as TWO launches for a step on two streams (r04): the merge of the R partial rows needs nothing of the
// reconstruction term, and the prior's backward needs nothing but the merge (token + coefficients) -- so the merge runs on
// the prior's stream right behind the partials, spread over the queries (16 per block, 64 groups of partial rows per query:
// every thread's <= 4 rows of loads are in flight at once instead of a six-round dependent chain in one block), and the ELBO
// assembly (loss, KL, batch means) runs beside the prior's backward on the stream that produced RE.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void prior_merge_coef_kernel(const float* __restrict__ pm, const float* __restrict__ ps,
                                                                const float* __restrict__ pn, int R, int ldp, int B,
                                                                float c_total, const float* __restrict__ beta_dev, float beta_host,
                                                                float* __restrict__ logp, float* __restrict__ lse_out,
                                                                float* __restrict__ cRE, float* __restrict__ cKL,
                                                                float* __restrict__ neg_cKL) {
  __shared__ float cm[64][17], cs[64][17], cn[64][17];
  const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int row = blockIdx.x * 16 + q;
  float m = -INFINITY, s = 0.f, n = 0.f;
  if (row < B) {
    for (int r0 = grp; r0 < R; r0 += 256) {
      float mr[4], sr_[4], nr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + 64 * u;
        const bool ok = r < R;
        const size_t o = (size_t)(ok ? r : grp) * ldp + row;
        mr[u] = ok ? pm[o] : -INFINITY; sr_[u] = ok ? ps[o] : 0.f; nr[u] = ok ? pn[o] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        n += nr[u];
        if (mr[u] > m) { s = s * expf(m - mr[u]) + sr_[u]; m = mr[u]; }
        else if (mr[u] != -INFINITY) s += sr_[u] * expf(mr[u] - m);
      }
    }
  }
  cm[grp][q] = m; cs[grp][q] = s; cn[grp][q] = n;
  __syncthreads();
  if (grp == 0 && row < B) {
    float mm = -INFINITY, ss = 0.f, nn = 0.f;
    for (int g = 0; g < 64; ++g) mm = fmaxf(mm, cm[g][q]);
    for (int g = 0; g < 64; ++g) {                 // fixed order
      if (cm[g][q] != -INFINITY) ss += cs[g][q] * expf(cm[g][q] - mm);
      nn += cn[g][q];
    }
    const float ls = logf(ss);
    logp[row] = (mm + ls) - logf(c_total - nn);
    lse_out[row] = mm; lse_out[B + row] = (mm == -INFINITY) ? 0.f : ls;       // token (max, log sum): prior_merge_kernel
    if (cRE != nullptr) {      // coefficients of "the batch mean of the loss, upstream gradient 1" (evae_elbo_bwd)
      const float beta = beta_dev ? beta_dev[0] : beta_host;
      const float gl = 1.0f / (float)B;
      cRE[row] = 0.f - gl; cKL[row] = 0.f + beta * gl; neg_cKL[row] = -(0.f + beta * gl);
    }
  }
}

__global__ __launch_bounds__(128) void elbo_assemble_kernel(const float* __restrict__ logp, const float* __restrict__ RE,
                                                            const float* __restrict__ logq, const float* __restrict__ beta_dev,
                                                            float beta_host, int B, float* __restrict__ loss,
                                                            float* __restrict__ KL, float* __restrict__ means) {
  __shared__ float red[3][2];
  const float beta = beta_dev ? beta_dev[0] : beta_host;
  float sl = 0.f, sr = 0.f, sk = 0.f;
  for (int row = threadIdx.x; row < B; row += 128) {          // the same rows per lane, in the same order, as prior_elbo_fwd_kernel
    const float kl = logq[row] - logp[row];
    const float l = beta * kl - RE[row];
    KL[row] = kl; loss[row] = l;
    sl += l; sr += RE[row]; sk += kl;
  }
  if (means == nullptr) return;
  sl = wave_sum(sl); sr = wave_sum(sr); sk = wave_sum(sk);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sl; red[1][threadIdx.x >> 6] = sr; red[2][threadIdx.x >> 6] = sk; }
  __syncthreads();
  if (threadIdx.x < 3) means[threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) / (float)B;
}

__global__ void prior_fill_empty_kernel(float* m, float* s, float* n, int B) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { m[i] = -INFINITY; s[i] = 0.f; n[i] = 0.f; }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// grid (nsplit, nq).  Per exemplar tile: recompute distances, gw_ij = g_i exp(p_ij - lse_i) into LDS,
// then  dC[e][k] += sum_i gw (zs_ik - cs_ek)   (thread <-> (e, k-group), direct differences)
//       dV[k]    += sum_ie gw (zs_ik - cs_ek)^2
//       dZ[i][k] += sum_e gw (cs_ek - zs_ik)   (thread <-> (i, k-group), registers across tiles)
// all in sigma-scaled coordinates; the finishing kernels apply 1/sigma and reduce the splits.
constexpr int KG_C = NT / BE;  // 4 k-groups for the dC phase
constexpr int KG_Z = NT / BQ;  // 2 k-groups for the dZ phase
constexpr int GWS = BE + 1;    // gw row stride (conflict-free column reads)

template <int KC>
__global__ __launch_bounds__(NT) void prior_bwd_kernel(
    const float* __restrict__ z, int B, const float* __restrict__ centres, int C, int zdim,
    const float* __restrict__ log_var, const int64_t* __restrict__ z_idx,
    const int64_t* __restrict__ c_idx, const float* __restrict__ lse, const float* __restrict__ gout,
    int tiles_per_split, int nsplit, PriorGeom g, int use_atomic_dc,
    float* __restrict__ dz_part /* [nsplit][B][zdim] */, float* __restrict__ dc /* [C][zdim] */,
    float* __restrict__ dlv_part /* [nq*nsplit][zdim+1] */, const unsigned* __restrict__ run_flag) {
  constexpr int ks = Geom<KC>::ks;
  if (run_flag != nullptr && *run_flag == 0u) return;      // see prior_fwd_kernel
  constexpr int KPC = KC / KG_C;   // dims per thread in the dC phase
  constexpr int KPZ = KC / KG_Z;   // dims per thread in the dZ phase
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;
  float* Es = Qs + BQ * ks;
  float* inv_sigma = Es + BE * ks;
  float* red = inv_sigma + ZDIM_MAX;
  float* GW = red + 64;  // [BQ][GWS]

  const int split = blockIdx.x;
  const int q0 = blockIdx.y * BQ;
  const int te = threadIdx.x & 15;
  const int tq = threadIdx.x >> 4;
  const bool vec_ok = (zdim & 3) == 0 && (((uintptr_t)z | (uintptr_t)centres) & 15) == 0;
  const bool masked = (z_idx != nullptr) && (c_idx != nullptr);
  const int zpad = g.nchunk * KC;
  const float cst = setup_sigma(inv_sigma, red, log_var, zdim, zpad);
  const int nq_valid = (B - q0) < BQ ? (B - q0) : BQ;      // live query rows of this tile

  int64_t zi[TQ];
  float gi[TQ], li[TQ], l2[TQ];
#pragma unroll
  for (int i = 0; i < TQ; ++i) {
    int q = q0 + tq + 16 * i;
    bool v = q < B;
    zi[i] = (masked && v) ? z_idx[q] : -1;
    gi[i] = v ? gout[q] : 0.f;
    li[i] = v ? lse[q] : 0.f;               // token: row max ...
    l2[i] = v ? lse[B + q] : 0.f;           // ... and log of the normalised sum (prior_merge_kernel)
  }

  // phase-2 roles
  const int ce = threadIdx.x & (BE - 1);      // exemplar row for dC
  const int ckg = threadIdx.x >> 6;           // 0..3 (one wave per k-group)
  const int zi_row = threadIdx.x & (BQ - 1);  // query row for dZ
  const int zkg = threadIdx.x >> 7;           // 0..1

  float gwsum = 0.f;                           // sum of gw seen in the dC role (for dlogvar)
  const int tile_begin = split * tiles_per_split;
  const int ntiles = (C + BE - 1) / BE;
  int tile_end = tile_begin + tiles_per_split;
  if (tile_end > ntiles) tile_end = ntiles;

  if (g.nchunk == 1) stage_rows<KC>(Qs, z, q0, B, BQ, zdim, 0, inv_sigma, vec_ok);

  for (int ch2 = 0; ch2 < g.nchunk; ++ch2) {
    float accZ[KPZ];
    float accV[KPC];
#pragma unroll
    for (int k = 0; k < KPZ; ++k) accZ[k] = 0.f;
#pragma unroll
    for (int k = 0; k < KPC; ++k) accV[k] = 0.f;

    for (int t = tile_begin; t < tile_end; ++t) {
      const int e0 = t * BE;
      float acc[TQ][TE];
#pragma unroll
      for (int i = 0; i < TQ; ++i)
#pragma unroll
        for (int j = 0; j < TE; ++j) acc[i][j] = 0.f;
      // distances over all chunks IN THE FORWARD'S ORDER (0, 1, ...): d2 has to come out bit for bit as in prior_fwd_kernel --
      // the weights below are exp((p_ij - max_i) - log sum_i), and at |log p| ~ 1e8 one ulp of d2 is 16 nats (r03: staging
      // chunk ch2 last, as phase 2 wants it, reordered the sum: gradients 1e8 x off at z = 294, |z| ~ 1e4, golden G21);
      // chunk ch2 is then staged once more for phase 2 unless it was the last one anyway
      for (int ch = 0; ch < g.nchunk; ++ch) {
        __syncthreads();
        if (g.nchunk > 1) stage_rows<KC>(Qs, z, q0, B, BQ, zdim, ch * KC, inv_sigma + ch * KC, vec_ok);
        stage_rows<KC>(Es, centres, e0, C, BE, zdim, ch * KC, inv_sigma + ch * KC, vec_ok);
        __syncthreads();
        dist_chunk<KC>(acc, Qs, Es, tq, te);
      }
      if (g.nchunk > 1 && ch2 != g.nchunk - 1) {
        __syncthreads();
        stage_rows<KC>(Qs, z, q0, B, BQ, zdim, ch2 * KC, inv_sigma + ch2 * KC, vec_ok);
        stage_rows<KC>(Es, centres, e0, C, BE, zdim, ch2 * KC, inv_sigma + ch2 * KC, vec_ok);
        __syncthreads();
      }
      // gw tile -> LDS
#pragma unroll
      for (int j = 0; j < TE; ++j) {
        int e = e0 + te + 16 * j;
        bool ev = e < C;
        int64_t cj = (masked && ev) ? c_idx[e] : -2;
#pragma unroll
        for (int i = 0; i < TQ; ++i) {
          bool ok = ev && !(masked && (zi[i] == cj || cj == kMaskAll));
          float w = ok ? gi[i] * __expf(((cst - 0.5f * acc[i][j]) - li[i]) - l2[i]) : 0.f;
          GW[(tq + 16 * i) * GWS + te + 16 * j] = w;
        }
      }
      __syncthreads();
      // ---- dC / dV: thread <-> (exemplar ce, dims ckg*KPC ..)
      {
        float cs[KPC], aC[KPC];
#pragma unroll
        for (int k = 0; k < KPC; ++k) { cs[k] = Es[ce * ks + ckg * KPC + k]; aC[k] = 0.f; }
        float wsum = 0.f;
#pragma unroll 4
        for (int i = 0; i < nq_valid; ++i) {
          const float w = GW[i * GWS + ce];
          wsum += w;
          const float* qrow = Qs + i * ks + ckg * KPC;
#pragma unroll
          for (int k = 0; k < KPC; ++k) {
            const float d = qrow[k] - cs[k];
            const float tw = w * d;
            aC[k] += tw;
            accV[k] = fmaf(tw, d, accV[k]);
          }
        }
        if (ch2 == 0 && ckg == 0) gwsum += wsum;
        const int e = e0 + ce;
        if (e < C) {
#pragma unroll
          for (int k = 0; k < KPC; ++k) {
            const int kk = ch2 * KC + ckg * KPC + k;
            if (kk < zdim) {
              const float v = aC[k] * inv_sigma[kk];
              if (use_atomic_dc) atomicAdd(&dc[(size_t)e * zdim + kk], v);
              else dc[(size_t)e * zdim + kk] = v;
            }
          }
        }
      }
      // ---- dZ: thread <-> (query zi_row, dims zkg*KPZ ..)
      if (zi_row < nq_valid) {
        float zs[KPZ];
#pragma unroll
        for (int k = 0; k < KPZ; ++k) zs[k] = Qs[zi_row * ks + zkg * KPZ + k];
#pragma unroll 4
        for (int e = 0; e < BE; ++e) {
          const float w = GW[zi_row * GWS + e];
          const float* erow = Es + e * ks + zkg * KPZ;
#pragma unroll
          for (int k = 0; k < KPZ; ++k) accZ[k] = fmaf(w, erow[k] - zs[k], accZ[k]);
        }
      }
    }
    // flush chunk ch2
    {
      const int q = q0 + zi_row;
      if (q < B) {
#pragma unroll
        for (int k = 0; k < KPZ; ++k) {
          const int kk = ch2 * KC + zkg * KPZ + k;
          if (kk < zdim) dz_part[((size_t)split * B + q) * zdim + kk] = accZ[k];
        }
      }
      // reduce accV over the 64 exemplar lanes of this wave (wave == k-group ckg)
      float* dlv = dlv_part + (size_t)(blockIdx.y * nsplit + split) * (zdim + 1);
#pragma unroll
      for (int k = 0; k < KPC; ++k) {
        const float v = wave_sum(accV[k]);
        const int kk = ch2 * KC + ckg * KPC + k;
        if ((threadIdx.x & 63) == 0 && kk < zdim) dlv[kk] = v;
      }
      if (ch2 == 0) {
        const float sg = wave_sum(gwsum);
        if (threadIdx.x == 0) dlv[zdim] = sg;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward on the matrix cores (z_dim <= 56, multiple of 4).  Same tiling as prior_fwd_mfma_kernel: block = 8 waves,
// tile = 128 exemplars x 128 queries, S = (c/s).(z/s)^T on v_mfma_f32_32x32x2_f32.  Then, per tile,
//   P[e][q] = g_q exp(log N(z_q | c_e) - lse_q)      (0 for masked / absent pairs), written once to LDS,
//   T = P  . [Zs | 1]  -> T[e][k] = sum_q P Zs_qk,  T[e][KP] = colsum_e(P)       (contraction over the 128 queries)
//   U = P^T. [Cs | 1]  -> U[q][k] = sum_e P Cs_ek,  U[q][KP] = rowsum_q(P)       (contraction over the 128 exemplars;
//                                                                                  accumulated over the split's tiles)
// both again on the matrix cores -- the column of ones that yields the row / column sums sits in the padding column KP
// of the staged tiles, outside the k range of S.  From those:
//   dC[e][k] = (T[e][k] - colsum_e Cs_ek) / s_k          dZ'[q][k] = U[q][k] - rowsum_q Zs_qk   (finish: / s_k, sum of splits)
//   dV[k]    = sum_q rowsum_q Zs_qk^2 + sum_e colsum_e Cs_ek^2 - 2 sum_e Cs_ek T[e][k]          (= sum P (Zs - Cs)^2)
// i.e. the same partial outputs as prior_bwd_kernel, so the finish kernel is shared.
// ------------------------------------------------------------------------------------------------
template <int KG>
__global__ __launch_bounds__(MFT) void prior_bwd_mfma_kernel(
    const float* __restrict__ z, int B, const float* __restrict__ centres, int C, int zdim,
    const float* __restrict__ log_var, const int64_t* __restrict__ z_idx, const int64_t* __restrict__ c_idx,
    const float* __restrict__ lse, const float* __restrict__ gout, int tiles_per_split, int nsplit, int use_atomic_dc,
    float norm_limit, float* __restrict__ dz_part /* [nsplit][B][zdim] */, float* __restrict__ dc /* [C][zdim] */,
    float* __restrict__ dlv_part /* [nq*nsplit][zdim+1] */) {
  constexpr int KP = KG * 8, KS2 = KP + 4, CPR = KP / 4, PP = 132;
  constexpr int NV = (MFE * CPR + MFT - 1) / MFT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;                          // [128][KS2]   scaled queries, column KP = 1
  float* Es = Qs + MFQ * KS2;                // [128][KS2]   scaled exemplars, column KP = 1
  float* Ps = Es + MFE * KS2;                // [128 e][PP]
  float* zn = Ps + MFE * PP;                 // [128]
  float* cn = zn + MFQ;                      // [128]
  float* cs = cn + MFE;                      // [128] column sums of P (per exemplar)
  float* rs = cs + MFE;                      // [128] row sums of P (per query)
  float* inv_sigma = rs + MFQ;               // [64]
  float* mu_s = inv_sigma + 64;              // [64]  mean of the query tile (sigma units), see prior_fwd_mfma_kernel
  float* red = mu_s + 64;                    // [16]
  float* zmx = red + 16;                     // [2] (+6 padding)
  float* dvs = zmx + 8;                      // [8][64] per-(wave row, lane half) partials of dV
  long long* ci_s = reinterpret_cast<long long*>(dvs + 8 * 64);   // [128]

  const int split = blockIdx.x;
  const int q0 = blockIdx.y * MFQ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, lh = lane >> 5;
  const bool masked = (z_idx != nullptr) && (c_idx != nullptr);
  const float cst = setup_sigma(inv_sigma, red, log_var, zdim, KP);   // contains a barrier

  auto load_tile = [&](const float* src, int r0, int nrows, float4 (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + MFT * i;
      const int r = f / CPR, c = f - r * CPR;
      const bool ok = f < MFE * CPR && r0 + r < nrows && 4 * c + 4 <= zdim;
      v[i] = ok ? *reinterpret_cast<const float4*>(src + (size_t)(r0 + r) * zdim + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&](float* tile, const float4 (&v)[NV], const bool centre) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + MFT * i;
      const int r = f / CPR, c = f - r * CPR;
      if (f < MFE * CPR) {
        const float4 s4 = *reinterpret_cast<const float4*>(inv_sigma + 4 * c);
        float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (centre) m4 = *reinterpret_cast<const float4*>(mu_s + 4 * c);
        *reinterpret_cast<float4*>(tile + r * KS2 + 4 * c) =
            make_float4(fmaf(v[i].x, s4.x, -m4.x), fmaf(v[i].y, s4.y, -m4.y), fmaf(v[i].z, s4.z, -m4.z), fmaf(v[i].w, s4.w, -m4.w));
      }
    }
    if (tid < MFE) *reinterpret_cast<float4*>(tile + tid * KS2 + KP) = make_float4(1.f, 0.f, 0.f, 0.f);   // the ones column
  };
  auto row_norms = [&](const float* tile, float* out) {
    if (tid < MFE) {
      float sacc = 0.f;
#pragma unroll
      for (int c = 0; c < CPR; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(tile + tid * KS2 + 4 * c);
        sacc += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
      }
      out[tid] = sacc;
    }
  };

  float4 rv[NV];
  load_tile(z, q0, B, rv);
  store_tile(Qs, rv, false);
  const int ntiles = (C + MFE - 1) / MFE;
  const int tile_begin = split * tiles_per_split;
  int tile_end = tile_begin + tiles_per_split;
  if (tile_end > ntiles) tile_end = ntiles;
  if (tile_begin < tile_end) load_tile(centres, tile_begin * MFE, C, rv);
  __syncthreads();
  // centred coordinates (every formula below is translation invariant); rows past B stay zero.  The guard of
  // prior_fwd_mfma_kernel (block-uniform): queries too far from their tile mean for the expanded form -> direct differences
  const bool slow = centre_queries<KP, KS2, MFT>(Qs, mu_s, zn, zmx, Ps, (B - q0) < MFQ ? (B - q0) : MFQ) > norm_limit;

  // this lane's two query columns of S
  float znq[2], gq[2], lq[2], lq2[2];
  long long zi[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int ql = wc * 64 + nt * 32 + l31;
    const bool v = q0 + ql < B;
    znq[nt] = zn[ql];
    gq[nt] = v ? gout[q0 + ql] : 0.f;
    lq[nt] = v ? lse[q0 + ql] : 0.f;            // token: row max, log of the normalised sum (prior_merge_kernel)
    lq2[nt] = v ? lse[B + q0 + ql] * kLog2e : 0.f;
    zi[nt] = (masked && v) ? (long long)z_idx[q0 + ql] : -1;
  }
  // column of the T / U tiles this lane holds (clamped to the zero padding column for the operand reads)
  const int ncol = wc * 32 + l31;
  const int nread = ncol < KS2 ? ncol : KS2 - 1;
  const float isg = ncol < zdim ? inv_sigma[ncol] : 0.f;

  f32x16_t U;
#pragma unroll
  for (int r = 0; r < 16; ++r) U[r] = 0.f;
  float dv_acc = 0.f;        // this lane's share of dV[ncol]
  float gw_acc = 0.f;        // threads 0..127: sum of the column sums they saw

  for (int t = tile_begin; t < tile_end; ++t) {
    const int e0 = t * MFE;
    store_tile(Es, rv, true);
    if (tid < MFE) ci_s[tid] = (masked && e0 + tid < C) ? (long long)c_idx[e0 + tid] : -2;
    __syncthreads();
    if (t + 1 < tile_end) load_tile(centres, (t + 1) * MFE, C, rv);
    row_norms(Es, cn);

    // ---- S = Es . Qs^T
    f32x16_t acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    {
      const float* ea = Es + (wr * 32 + l31) * KS2 + lh * 4;
      const float* qb = Qs + (wc * 64 + l31) * KS2 + lh * 4;
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) {
        const float4 a = *reinterpret_cast<const float4*>(ea + kg * 8);
        const float4 b0 = *reinterpret_cast<const float4*>(qb + kg * 8);
        const float4 b1 = *reinterpret_cast<const float4*>(qb + 32 * KS2 + kg * 8);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc[1], 0, 0, 0);
      }
    }
    __syncthreads();     // cn, ci_s complete

    if (slow) direct_tile<KP, KS2>(acc, Es, Qs, wr, wc, l31, lh);      // acc <- d2 (Es stays put until the end of the tile)

    // ---- P into LDS (rows of this lane: e = wr*32 + (r&3) + 8*(r>>2) + 4*lh)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int ql = wc * 64 + nt * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int el = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float d = slow ? acc[nt][r] : fmaxf(cn[el] + znq[nt] - 2.0f * acc[nt][r], 0.f);
        bool ok = e0 + el < C;
        if (masked) ok = ok && (ci_s[el] != zi[nt]) && (ci_s[el] != kMaskAll);
        // exp(p - lse), p = cst - d/2 rounded once as in the forward; (p - max) first: exact where the weight is not ~0
        const float w = gq[nt] * fast_exp2(fmaf(fmaf(-0.5f, d, cst) - lq[nt], kLog2e, -lq2[nt]));
        Ps[el * PP + ql] = ok ? w : 0.f;
      }
    }
    __syncthreads();

    // ---- T = P . [Zs | 1]  (wave: exemplar rows wr*32.., columns wc*32..)   and   U += P^T . [Cs | 1]
    f32x16_t T;
#pragma unroll
    for (int r = 0; r < 16; ++r) T[r] = 0.f;
    {
      const float* pa = Ps + (wr * 32 + l31) * PP + lh * 4;        // P[e][q..q+3]
      const float* pu = Ps + (lh * 4) * PP + wr * 32 + l31;        // P[e..e+3][q]
      const float* qb = Qs + (lh * 4) * KS2 + nread;
      const float* eb = Es + (lh * 4) * KS2 + nread;
#pragma unroll 4
      for (int kg = 0; kg < 16; ++kg) {
        const float4 a = *reinterpret_cast<const float4*>(pa + kg * 8);
        const float* q8 = qb + kg * 8 * KS2;
        T = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, q8[0], T, 0, 0, 0);
        T = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, q8[KS2], T, 0, 0, 0);
        T = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, q8[2 * KS2], T, 0, 0, 0);
        T = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, q8[3 * KS2], T, 0, 0, 0);
        const float* p8 = pu + kg * 8 * PP;
        const float* e8 = eb + kg * 8 * KS2;
        U = __builtin_amdgcn_mfma_f32_32x32x2f32(p8[0], e8[0], U, 0, 0, 0);
        U = __builtin_amdgcn_mfma_f32_32x32x2f32(p8[PP], e8[KS2], U, 0, 0, 0);
        U = __builtin_amdgcn_mfma_f32_32x32x2f32(p8[2 * PP], e8[2 * KS2], U, 0, 0, 0);
        U = __builtin_amdgcn_mfma_f32_32x32x2f32(p8[3 * PP], e8[3 * KS2], U, 0, 0, 0);
      }
    }
    // column sums of P: the lanes that hold column KP of T
    if (ncol == KP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) cs[wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = T[r];
    }
    __syncthreads();
    if (tid < MFE) gw_acc += cs[tid];
    if (ncol < zdim) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int el = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float c_ = Es[el * KS2 + ncol], csum = cs[el];
        dv_acc += c_ * (csum * c_ - 2.0f * T[r]);
        const int e = e0 + el;
        if (e < C) {
          const float v = (T[r] - csum * c_) * isg;
          if (use_atomic_dc) atomicAdd(&dc[(size_t)e * zdim + ncol], v);
          else dc[(size_t)e * zdim + ncol] = v;
        }
      }
    }
    __syncthreads();     // everybody is done with Es / Ps / cs of this tile
  }

  // ---- row sums, dZ' partial of this split, the query part of dV
  if (ncol == KP) {
#pragma unroll
    for (int r = 0; r < 16; ++r) rs[wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = U[r];
  }
  __syncthreads();
  if (ncol < zdim) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ql = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float z_ = Qs[ql * KS2 + ncol], rsum = rs[ql];
      dv_acc += rsum * z_ * z_;
      if (q0 + ql < B) dz_part[((size_t)split * B + q0 + ql) * zdim + ncol] = U[r] - rsum * z_;
    }
  }
  dvs[(wr * 2 + lh) * 64 + ncol] = dv_acc;
  __syncthreads();
  float* dlv = dlv_part + (size_t)(blockIdx.y * nsplit + split) * (zdim + 1);
  if (tid < zdim) {
    float v = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) v += dvs[p * 64 + tid];     // fixed order
    dlv[tid] = v;
  }
  __syncthreads();
  // sum of all of P seen by this block: threads 0..127 hold the column sums of their exemplar slot
  if (wave < 2) {
    const float sg = wave_sum(gw_acc);
    if (lane == 0) dvs[wave] = sg;
  }
  __syncthreads();
  if (tid == 0) dlv[zdim] = dvs[0] + dvs[1];
}

template <int KG>
static int launch_prior_bwd_mfma(const float* z, int B, const float* centres, int C, int zdim, const float* log_var,
                                 const int64_t* z_idx, const int64_t* c_idx, const float* lse, const float* gout,
                                 int ns_max, int use_atomic, float* dz_part, float* dc, float* dlv_part, int* ns_out,
                                 hipStream_t stream) {
  constexpr int KS2 = KG * 8 + 4;
  const size_t lds = (size_t)(2 * 128 * KS2 + 128 * 132 + 4 * 128 + 64 + 64 + 16 + 8 + 8 * 64) * sizeof(float) + 128 * sizeof(long long);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)prior_bwd_mfma_kernel<KG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int nq = cdiv(B, MFQ), ntiles = cdiv(C, MFE);
  int ns = cdiv(512, nq);
  if (ns > ntiles) ns = ntiles;
  if (ns > ns_max) ns = ns_max;
  if (ns < 1) ns = 1;
  const int tps = cdiv(ntiles, ns);
  ns = cdiv(ntiles, tps);
  *ns_out = ns;
  prior_bwd_mfma_kernel<KG><<<dim3(ns, nq), MFT, lds, stream>>>(z, B, centres, C, zdim, log_var, z_idx, c_idx, lse, gout, tps, ns,
                                                                use_atomic, prior_norm_limit(), dz_part, dc, dlv_part);
  return check_launch("prior_bwd_mfma_kernel");
}

// Both reductions that follow prior_bwd_kernel, in one launch.
//   blocks [0, nb_dz): dz[e] = exp(-logvar/2) * sum over splits of dz_part (16 split-lanes per output, fixed assignment
//                      => fixed order);
//   the blocks after:  dlogvar[k] = 0.5 * sum_blocks dV[k] - 0.5 * sum_blocks gwsum, one wave per k.
__global__ __launch_bounds__(1024) void prior_bwd_finish_kernel(const float* __restrict__ dz_part, int nsplit, int n,
                                                                int zdim, const float* __restrict__ log_var,
                                                                float* __restrict__ dz, int nb_dz,
                                                                const float* __restrict__ dlv_part, int nblocks,
                                                                float* __restrict__ dlogvar,
                                                                const unsigned* __restrict__ run_flag,
                                                                const float* __restrict__ fold_src = nullptr, const int64_t* __restrict__ fold_rep = nullptr,
                                                                const float* __restrict__ fold_mult = nullptr, float* __restrict__ fold_dst = nullptr,
                                                                int fold_rows = 0, int nb_lv = 0) {
  __shared__ float red[16][64];
  if (run_flag != nullptr && *run_flag == 0u) return;
  const int lane = threadIdx.x & 63;
  const int part = threadIdx.x >> 6;
  if (fold_dst != nullptr && (int)blockIdx.x >= nb_dz + nb_lv) {
    // the blocks behind the two reductions (r06): dst[u] = mult[u] * src[rep[u]] -- the per-draw centre gradients folded onto the
    // DISTINCT rows of the draw (a distinct row's gradient = multiplicity x one of its draws'; padding rows: multiplicity 0)
    const int q = zdim >> 2;
    const size_t i = (size_t)((int)blockIdx.x - nb_dz - nb_lv) * 1024 + threadIdx.x;
    if (i < (size_t)fold_rows * q) {
      const int u = (int)(i / q), c = (int)(i - (size_t)u * q);
      const float m = fold_mult[u];
      const float4 v = *reinterpret_cast<const float4*>(fold_src + (size_t)fold_rep[u] * zdim + 4 * c);
      *reinterpret_cast<float4*>(fold_dst + (size_t)u * zdim + 4 * c) = make_float4(m * v.x, m * v.y, m * v.z, m * v.w);
    }
    return;
  }
  if ((int)blockIdx.x >= nb_dz) {
    const int k = ((int)blockIdx.x - nb_dz) * 16 + part;
    if (k >= zdim) return;
    float sv = 0.f, sg = 0.f;
    for (int b = lane; b < nblocks; b += 64) {
      sv += dlv_part[(size_t)b * (zdim + 1) + k];
      sg += dlv_part[(size_t)b * (zdim + 1) + zdim];
    }
    sv = wave_sum(sv);
    sg = wave_sum(sg);
    if (lane == 0) dlogvar[k] = 0.5f * sv - 0.5f * sg;
    return;
  }
  const int e = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (e < n)
    for (int r = part; r < nsplit; r += 16) s += dz_part[(size_t)r * n + e];
  red[part][lane] = s;
  __syncthreads();
  if (part == 0 && e < n) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) t += red[p][threadIdx.x];
    dz[e] = t * expf(-0.5f * log_var[e % zdim]);
  }
}

__global__ void zero_kernel(float* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

}  // namespace evae
#include "evae_prior_train.h"
namespace evae {

static void choose_splits(int B, int C, int* nsplit, int* tiles_per_split, int* nq) {
  int ntiles = cdiv(C, BE);
  *nq = cdiv(B, BQ);
  int target = cdiv(1024, *nq);  // aim for ~4 blocks per CU in total
  if (target < 1) target = 1;
  int ns = ntiles < target ? ntiles : target;
  if (ns < 1) ns = 1;
  *tiles_per_split = cdiv(ntiles, ns);
  if (*tiles_per_split < 1) *tiles_per_split = 1;
  *nsplit = cdiv(ntiles, *tiles_per_split);
  if (*nsplit < 1) *nsplit = 1;
}

}  // namespace evae

using namespace evae;

extern "C" int evae_prior_set_norm_limit(float limit) {
  g_norm_limit = limit >= 0.f ? limit : kPriorNormLimit;
  return EVAE_OK;
}

// large latent sizes: the GEMM path of evae_prior_gemm.hip (forward z > 64, backward z > 56)
static bool fwd_uses_gemm(int B, int C, int zdim) { return zdim > 64 && prior_gemm_applies(B, C, zdim); }
static bool bwd_uses_gemm(int B, int C, int zdim) {
  return zdim > 56 && prior_gemm_applies(B, C, zdim) && (int64_t)B * C <= ((int64_t)1 << 28);
}

extern "C" size_t evae_prior_lse_fwd_workspace_bytes(int B, int C, int zdim) {
  if (B <= 0 || C <= 0) return 256;
  int ns, tps, nq;
  choose_splits(B, C, &ns, &tps, &nq);
  const size_t valu = align_up((size_t)3 * ns * B * sizeof(float), 256) + 256;
  if (fwd_uses_gemm(B, C, zdim)) return std::max(valu, prior_gemm_fwd_layout(B, C, zdim, ns).total);
  if (prior_stream_applies(B, C, zdim, false)) return std::max(valu, prior_gemm_fwd_layout(B, C, zdim, ns, true).total);
  return valu;
}

// splits_out != nullptr: leave the per-split partials in the workspace (planes of *splits_out rows x B at ws, ws + R B,
// ws + 2 R B floats, R = the split count of choose_splits) instead of merging them into out_*; z <= 64 / VALU paths only
static int prior_lse_fwd_core(const float* z, int B, const float* centres, int C, int zdim,
                              const float* log_var, const int64_t* z_idx, const int64_t* c_idx,
                              float* out_max, float* out_sumexp, float* out_nmask, float* out_prob,
                              void* ws, size_t ws_bytes, int* splits_out, hipStream_t stream) {
  EVAE_REQUIRE(B >= 0 && C >= 0 && zdim > 0, "prior_lse_fwd: bad sizes B=%d C=%d zdim=%d", B, C, zdim);
  EVAE_REQUIRE(zdim <= ZDIM_MAX, "prior_lse_fwd: zdim %d > %d unsupported", zdim, ZDIM_MAX);
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(z && log_var && (splits_out || (out_max && out_sumexp && out_nmask)), "prior_lse_fwd: null pointer");
  if (C == 0) {
    if (splits_out) {
      EVAE_REQUIRE(ws != nullptr && ws_bytes >= (size_t)3 * B * sizeof(float), "prior_lse_fwd: workspace too small");
      float* w = (float*)ws;
      prior_fill_empty_kernel<<<cdiv(B, 256), 256, 0, stream>>>(w, w + B, w + 2 * B, B);
      *splits_out = 1;
    } else {
      prior_fill_empty_kernel<<<cdiv(B, 256), 256, 0, stream>>>(out_max, out_sumexp, out_nmask, B);
    }
    return check_launch("prior_fill_empty");
  }
  EVAE_REQUIRE(centres != nullptr, "prior_lse_fwd: null centres");
  if (ws_bytes < evae_prior_lse_fwd_workspace_bytes(B, C, zdim) || ws == nullptr) {
    set_error("prior_lse_fwd: workspace too small (%zu)", ws_bytes);
    return EVAE_EWORKSPACE;
  }
  int ns, tps, nq;
  choose_splits(B, C, &ns, &tps, &nq);
  PriorGeom g = prior_geom(zdim);
  float* pm = (float*)ws;
  float* ps = pm + (size_t)ns * B;
  float* pn = ps + (size_t)ns * B;
  // matrix-core path (see prior_fwd_mfma_kernel); EVAE_PRIOR_VALU=1 forces the direct-difference kernel
  static int force_valu = -1;
  if (force_valu < 0) { const char* e = getenv("EVAE_PRIOR_VALU"); force_valu = (e && atoi(e)) ? 1 : 0; }
  const bool masked_call = z_idx != nullptr && c_idx != nullptr;
  const bool stream6 = !force_valu && out_prob == nullptr && splits_out == nullptr && prior_stream_applies(B, C, zdim, masked_call);
  if (!stream6 && !force_valu && out_prob == nullptr && zdim <= 64 && (zdim & 3) == 0 &&
      ((((uintptr_t)z) | ((uintptr_t)centres)) & 15) == 0) {
    int ns2 = 1, rc;
    switch ((zdim + 7) / 8) {
      case 1: rc = launch_prior_mfma<1>(z, B, centres, C, zdim, log_var, z_idx, c_idx, ns, pm, ps, pn, &ns2, stream); break;
      case 2: rc = launch_prior_mfma<2>(z, B, centres, C, zdim, log_var, z_idx, c_idx, ns, pm, ps, pn, &ns2, stream); break;
      case 3: rc = launch_prior_mfma<3>(z, B, centres, C, zdim, log_var, z_idx, c_idx, ns, pm, ps, pn, &ns2, stream); break;
      case 4: rc = launch_prior_mfma<4>(z, B, centres, C, zdim, log_var, z_idx, c_idx, ns, pm, ps, pn, &ns2, stream); break;
      case 5: rc = launch_prior_mfma<5>(z, B, centres, C, zdim, log_var, z_idx, c_idx, ns, pm, ps, pn, &ns2, stream); break;
      case 6: rc = launch_prior_mfma<6>(z, B, centres, C, zdim, log_var, z_idx, c_idx, ns, pm, ps, pn, &ns2, stream); break;
      case 7: rc = launch_prior_mfma<7>(z, B, centres, C, zdim, log_var, z_idx, c_idx, ns, pm, ps, pn, &ns2, stream); break;
      default: rc = launch_prior_mfma<8>(z, B, centres, C, zdim, log_var, z_idx, c_idx, ns, pm, ps, pn, &ns2, stream); break;
    }
    if (rc) return rc;
    if (splits_out) {
      // the kernel wrote ns2 <= ns rows into planes laid out for ns rows: hand over the plane stride with the count
      *splits_out = ns2 | (ns << 16);
      return EVAE_OK;
    }
    prior_merge_kernel<<<cdiv(B, NT / 64), NT, 0, stream>>>(pm, ps, pn, ns2, B, 0, 0.f, out_max, out_sumexp, out_nmask);
    return check_launch("prior_merge_kernel(splits)");
  }
  size_t lds = prior_lds_bytes(g, false);
  if (stream6 || (!force_valu && out_prob == nullptr && fwd_uses_gemm(B, C, zdim))) {
    EVAE_REQUIRE(splits_out == nullptr, "prior_lse_fwd: raw split partials are not available on the GEMM path (z_dim > 64)");
    // matrix-core GEMM with a log-sum-exp epilogue (or, for evaluator-sized calls at small z, the streaming split-bf16
    // kernel); its norm guard hands over to the direct-difference kernel on the device
    const PriorGemmFwdLayout L = prior_gemm_fwd_layout(B, C, zdim, ns, stream6);
    int rc = prior_gemm_fwd(z, B, centres, C, zdim, log_var, z_idx, c_idx, prior_norm_limit(), (char*)ws, L, stream);
    if (rc) return rc;
    const unsigned* flag = (const unsigned*)((char*)ws + L.flag);
    float* gm = (float*)((char*)ws + L.pm); float* gs = (float*)((char*)ws + L.ps); float* gn = (float*)((char*)ws + L.pn);
    EVAE_DISPATCH_KC(g.kc, (prior_fwd_kernel<KC_><<<dim3(ns, nq), NT, lds, stream>>>(
                               z, B, centres, C, zdim, log_var, z_idx, c_idx, tps, ns, g, gm, gs, gn, nullptr, flag)));
    rc = check_launch("prior_fwd_kernel(guard fallback)");
    if (rc) return rc;
    prior_gemm_merge((const char*)ws, L, B, ns, out_max, out_sumexp, out_nmask, stream);
    return check_launch("prior_merge_sel_kernel");
  }
  EVAE_DISPATCH_KC(g.kc, (prior_fwd_kernel<KC_><<<dim3(ns, nq), NT, lds, stream>>>(
                             z, B, centres, C, zdim, log_var, z_idx, c_idx, tps, ns, g, pm, ps, pn, out_prob, nullptr)));
  int rc = check_launch("prior_fwd_kernel");
  if (rc) return rc;
  if (splits_out) { *splits_out = ns | (ns << 16); return EVAE_OK; }
  prior_merge_kernel<<<cdiv(B, NT / 64), NT, 0, stream>>>(pm, ps, pn, ns, B, 0, 0.f, out_max, out_sumexp,
                                                        out_nmask);
  return check_launch("prior_merge_kernel(splits)");
}

extern "C" int evae_prior_lse_fwd(const float* z, int B, const float* centres, int C, int zdim,
                                  const float* log_var, const int64_t* z_idx, const int64_t* c_idx,
                                  float* out_max, float* out_sumexp, float* out_nmask, float* out_prob,
                                  void* ws, size_t ws_bytes, evae_stream_t stream_) {
  return prior_lse_fwd_core(z, B, centres, C, zdim, log_var, z_idx, c_idx, out_max, out_sumexp, out_nmask, out_prob, ws,
                            ws_bytes, nullptr, (hipStream_t)stream_);
}

extern "C" int evae_prior_lse_fwd_splits(const float* z, int B, const float* centres, int C, int zdim,
                                         const float* log_var, const int64_t* z_idx, const int64_t* c_idx,
                                         void* ws, size_t ws_bytes, int* nsplit, int* plane_rows, evae_stream_t stream_) {
  EVAE_REQUIRE(nsplit && plane_rows, "prior_lse_fwd_splits: null pointer");
  EVAE_REQUIRE(zdim <= 64 || !fwd_uses_gemm(B, C, zdim), "prior_lse_fwd_splits: z_dim > 64 runs on the GEMM path; use evae_prior_lse_fwd");
  int packed = 0;
  *nsplit = 0; *plane_rows = 0;
  if (B == 0) return EVAE_OK;
  int rc = prior_lse_fwd_core(z, B, centres, C, zdim, log_var, z_idx, c_idx, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes,
                              &packed, (hipStream_t)stream_);
  if (rc) return rc;
  if (C == 0) { *nsplit = 1; *plane_rows = 1; return EVAE_OK; }
  *nsplit = packed & 0xFFFF; *plane_rows = packed >> 16;
  return EVAE_OK;
}

extern "C" int evae_prior_elbo_fwd(const float* pmax, const float* psum, const float* pnmask, int R, int ldp, int B,
                                   float c_total, const float* RE, const float* logq, const float* beta_dev,
                                   float beta_host, float* logp, float* lse, float* loss, float* KL, float* means,
                                   evae_stream_t stream_) {
  EVAE_REQUIRE(R >= 1 && B >= 0 && ldp >= B, "prior_elbo_fwd: bad sizes R=%d B=%d ldp=%d", R, B, ldp);
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(pmax && psum && pnmask && RE && logq && logp && loss && KL, "prior_elbo_fwd: null pointer");
  prior_elbo_fwd_kernel<<<1, 1024, 0, (hipStream_t)stream_>>>(pmax, psum, pnmask, R, ldp, B, c_total, RE, logq, beta_dev,
                                                             beta_host, logp, lse, loss, KL, means, nullptr, nullptr, nullptr);
  return check_launch("prior_elbo_fwd_kernel");
}

// The same, and the three coefficient vectors evae_elbo_bwd would produce when what is back-propagated is the batch mean of
// the loss with upstream gradient 1 (a captured training step: loss.backward(ones)): cRE = -1/B, cKL = beta/B, neg_cKL = -beta/B.
// The backward pass then starts one launch later.
extern "C" int evae_prior_elbo_fwd_coef(const float* pmax, const float* psum, const float* pnmask, int R, int ldp, int B,
                                        float c_total, const float* RE, const float* logq, const float* beta_dev,
                                        float beta_host, float* logp, float* lse, float* loss, float* KL, float* means,
                                        float* cRE, float* cKL, float* neg_cKL, evae_stream_t stream_) {
  EVAE_REQUIRE(R >= 1 && B >= 0 && ldp >= B, "prior_elbo_fwd_coef: bad sizes R=%d B=%d ldp=%d", R, B, ldp);
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(pmax && psum && pnmask && RE && logq && logp && loss && KL && cRE && cKL && neg_cKL, "prior_elbo_fwd_coef: null pointer");
  prior_elbo_fwd_kernel<<<1, 1024, 0, (hipStream_t)stream_>>>(pmax, psum, pnmask, R, ldp, B, c_total, RE, logq, beta_dev,
                                                             beta_host, logp, lse, loss, KL, means, cRE, cKL, neg_cKL);
  return check_launch("prior_elbo_fwd_kernel");
}



// prior_elbo_fwd(_coef) as two launches (prior_merge_coef_kernel above): evae_prior_merge_coef on the prior's stream, evae_elbo_assemble
// wherever RE is.  cRE / cKL / neg_cKL may all be NULL (no coefficients).
extern "C" int evae_prior_merge_coef(const float* pmax, const float* psum, const float* pnmask, int R, int ldp, int B,
                                     float c_total, const float* beta_dev, float beta_host, float* logp, float* lse,
                                     float* cRE, float* cKL, float* neg_cKL, evae_stream_t stream_) {
  EVAE_REQUIRE(R >= 1 && B >= 0 && ldp >= B, "prior_merge_coef: bad sizes R=%d B=%d ldp=%d", R, B, ldp);
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(pmax && psum && pnmask && logp && lse, "prior_merge_coef: null pointer");
  EVAE_REQUIRE((cRE && cKL && neg_cKL) || (!cRE && !cKL && !neg_cKL), "prior_merge_coef: all three coefficient vectors or none");
  prior_merge_coef_kernel<<<cdiv(B, 16), 1024, 0, (hipStream_t)stream_>>>(pmax, psum, pnmask, R, ldp, B, c_total, beta_dev, beta_host,
                                                                        logp, lse, cRE, cKL, neg_cKL);
  return check_launch("prior_merge_coef_kernel");
}

extern "C" int evae_elbo_assemble(const float* logp, const float* RE, const float* logq, const float* beta_dev, float beta_host,
                                  int B, float* loss, float* KL, float* means, evae_stream_t stream_) {
  EVAE_REQUIRE(B >= 0, "elbo_assemble: bad size B=%d", B);
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(logp && RE && logq && loss && KL, "elbo_assemble: null pointer");
  elbo_assemble_kernel<<<1, 128, 0, (hipStream_t)stream_>>>(logp, RE, logq, beta_dev, beta_host, B, loss, KL, means);
  return check_launch("elbo_assemble_kernel");
}

extern "C" int evae_prior_merge(const float* max, const float* sumexp, const float* nmask, int R, int B,
                                float c_total, float* out_logprior, float* out_lse,
                                evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(R >= 1 && B >= 0, "prior_merge: bad sizes R=%d B=%d", R, B);
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(max && sumexp && nmask && out_logprior, "prior_merge: null pointer");
  prior_merge_kernel<<<cdiv(B, NT / 64), NT, 0, stream>>>(max, sumexp, nmask, R, B, 1, c_total,
                                                        out_logprior, out_lse, nullptr);
  return check_launch("prior_merge_kernel");
}

extern "C" size_t evae_prior_lse_bwd_workspace_bytes(int B, int C, int zdim) {
  if (B <= 0 || C <= 0) return 256;
  int ns, tps, nq;
  choose_splits(B, C, &ns, &tps, &nq);
  size_t a = align_up((size_t)ns * B * zdim * sizeof(float), 256);
  size_t b = align_up((size_t)ns * nq * (zdim + 1) * sizeof(float), 256);
  size_t tot = a + b + 256;
  if (bwd_uses_gemm(B, C, zdim)) tot += prior_gemm_bwd_layout(B, C, zdim).total;     // GEMM buffers behind the VALU partials
  return tot;
}

// phase 0: everything.  phase 1 / 2 (evae_prior_lse_bwd_phased): 1 = up to and including dcentres, 2 = what is left (the
// reduction of the per-split dz / dlogvar partials in the workspace) -- the matrix-core path splits there, every other path
// does all its work in phase 1.
static int prior_lse_bwd_core(const float* z, int B, const float* centres, int C, int zdim,
                              const float* log_var, const int64_t* z_idx, const int64_t* c_idx,
                              const float* lse, const float* grad_out, float* dz, float* dcentres,
                              float* dlogvar, void* ws, size_t ws_bytes, int phase, hipStream_t stream) {
  EVAE_REQUIRE(B >= 0 && C >= 0 && zdim > 0, "prior_lse_bwd: bad sizes");
  EVAE_REQUIRE(zdim <= ZDIM_MAX, "prior_lse_bwd: zdim %d > %d unsupported", zdim, ZDIM_MAX);
  if (B == 0 && C == 0) return EVAE_OK;
  EVAE_REQUIRE((dz || B == 0) && dlogvar && log_var, "prior_lse_bwd: null pointer");   // an empty batch has no dz
  if (B == 0 || C == 0) {
    if (phase == 2) return EVAE_OK;
    if (B > 0) zero_kernel<<<cdiv(B * zdim, 256), 256, 0, stream>>>(dz, (size_t)B * zdim);
    if (C > 0 && dcentres) zero_kernel<<<cdiv(C * zdim, 256), 256, 0, stream>>>(dcentres, (size_t)C * zdim);
    zero_kernel<<<1, 256, 0, stream>>>(dlogvar, (size_t)zdim);
    return check_launch("prior_bwd zero");
  }
  EVAE_REQUIRE(z && centres && lse && grad_out && dcentres, "prior_lse_bwd: null pointer");
  if (ws_bytes < evae_prior_lse_bwd_workspace_bytes(B, C, zdim) || ws == nullptr) {
    set_error("prior_lse_bwd: workspace too small (%zu)", ws_bytes);
    return EVAE_EWORKSPACE;
  }
  int ns, tps, nq;
  choose_splits(B, C, &ns, &tps, &nq);
  PriorGeom g = prior_geom(zdim);
  float* dz_part = (float*)ws;
  float* dlv_part = (float*)((char*)ws + align_up((size_t)ns * B * zdim * sizeof(float), 256));
  int use_atomic = nq > 1;
  // matrix-core path (see prior_bwd_mfma_kernel); EVAE_PRIOR_VALU=1 forces the direct-difference kernel
  static int force_valu = -1;
  if (force_valu < 0) { const char* e = getenv("EVAE_PRIOR_VALU"); force_valu = (e && atoi(e)) ? 1 : 0; }
  const bool mfma = !force_valu && zdim <= 56 && (zdim & 3) == 0 && ((((uintptr_t)z) | ((uintptr_t)centres)) & 15) == 0;
  if (phase == 2 && !mfma) return EVAE_OK;           // everything happened in phase 1
  if (use_atomic && phase != 2) {
    zero_kernel<<<cdiv(C * zdim, 256), 256, 0, stream>>>(dcentres, (size_t)C * zdim);
    int rc = check_launch("zero dcentres");
    if (rc) return rc;
  }
  if (mfma) {
    int ns2 = 1, rc;
    if (phase == 2) {
      // the split count of launch_prior_bwd_mfma, recomputed (deterministic in (B, C))
      const int nq2 = cdiv(B, MFQ), ntiles = cdiv(C, MFE);
      int n2 = cdiv(512, nq2);
      if (n2 > ntiles) n2 = ntiles;
      if (n2 > ns) n2 = ns;
      if (n2 < 1) n2 = 1;
      ns2 = cdiv(ntiles, cdiv(ntiles, n2));
      const int nb = cdiv(B * zdim, 64);
      prior_bwd_finish_kernel<<<nb + cdiv(zdim, 16), 1024, 0, stream>>>(dz_part, ns2, B * zdim, zdim, log_var, dz, nb, dlv_part,
                                                                       ns2 * nq, dlogvar, nullptr);
      return check_launch("prior_bwd_finish_kernel");
    }
#define EVAE_BWD_MFMA(KG_) rc = launch_prior_bwd_mfma<KG_>(z, B, centres, C, zdim, log_var, z_idx, c_idx, lse, grad_out, ns, \
                                                          use_atomic, dz_part, dcentres, dlv_part, &ns2, stream)
    switch ((zdim + 7) / 8) {
      case 1: EVAE_BWD_MFMA(1); break;
      case 2: EVAE_BWD_MFMA(2); break;
      case 3: EVAE_BWD_MFMA(3); break;
      case 4: EVAE_BWD_MFMA(4); break;
      case 5: EVAE_BWD_MFMA(5); break;
      case 6: EVAE_BWD_MFMA(6); break;
      default: EVAE_BWD_MFMA(7); break;
    }
#undef EVAE_BWD_MFMA
    if (rc) return rc;
    if (phase == 1) return EVAE_OK;
    const int nb = cdiv(B * zdim, 64);
    prior_bwd_finish_kernel<<<nb + cdiv(zdim, 16), 1024, 0, stream>>>(dz_part, ns2, B * zdim, zdim, log_var, dz, nb, dlv_part,
                                                                     ns2 * nq, dlogvar, nullptr);
    return check_launch("prior_bwd_finish_kernel");
  }
  size_t lds = prior_lds_bytes(g, true);
  const unsigned* run_flag = nullptr;
  if (!force_valu && bwd_uses_gemm(B, C, zdim)) {
    // three GEMMs on the matrix cores (evae_prior_gemm.hip); its norm guard releases the kernels below on the device
    const PriorGemmBwdLayout L = prior_gemm_bwd_layout(B, C, zdim);
    char* gws = (char*)ws + align_up((size_t)ns * B * zdim * sizeof(float), 256) +
                align_up((size_t)ns * nq * (zdim + 1) * sizeof(float), 256);
    int rc = prior_gemm_bwd(z, B, centres, C, zdim, log_var, z_idx, c_idx, lse, grad_out, prior_norm_limit(), dz, dcentres,
                            dlogvar, gws, L, stream);
    if (rc) return rc;
    run_flag = (const unsigned*)(gws + L.flag);
  }
  EVAE_DISPATCH_KC(g.kc, {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)prior_bwd_kernel<KC_>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
      attr = true;
    }
    prior_bwd_kernel<KC_><<<dim3(ns, nq), NT, lds, stream>>>(z, B, centres, C, zdim, log_var, z_idx, c_idx, lse,
                                                           grad_out, tps, ns, g, use_atomic, dz_part, dcentres,
                                                           dlv_part, run_flag);
  });
  int rc = check_launch("prior_bwd_kernel");
  if (rc) return rc;
  const int nb_dz = cdiv(B * zdim, 64);
  prior_bwd_finish_kernel<<<nb_dz + cdiv(zdim, 16), 1024, 0, stream>>>(dz_part, ns, B * zdim, zdim, log_var, dz, nb_dz,
                                                                       dlv_part, ns * nq, dlogvar, run_flag);
  return check_launch("prior_bwd_finish_kernel");
}

extern "C" int evae_prior_lse_bwd(const float* z, int B, const float* centres, int C, int zdim,
                                  const float* log_var, const int64_t* z_idx, const int64_t* c_idx,
                                  const float* lse, const float* grad_out, float* dz, float* dcentres,
                                  float* dlogvar, void* ws, size_t ws_bytes, evae_stream_t stream_) {
  return prior_lse_bwd_core(z, B, centres, C, zdim, log_var, z_idx, c_idx, lse, grad_out, dz, dcentres, dlogvar, ws, ws_bytes,
                            0, (hipStream_t)stream_);
}

extern "C" int evae_prior_lse_bwd_phased(const float* z, int B, const float* centres, int C, int zdim,
                                         const float* log_var, const int64_t* z_idx, const int64_t* c_idx,
                                         const float* lse, const float* grad_out, float* dz, float* dcentres,
                                         float* dlogvar, void* ws, size_t ws_bytes, int phase, evae_stream_t stream_) {
  EVAE_REQUIRE(phase == 1 || phase == 2, "prior_lse_bwd_phased: phase must be 1 or 2");
  return prior_lse_bwd_core(z, B, centres, C, zdim, log_var, z_idx, c_idx, lse, grad_out, dz, dcentres, dlogvar, ws, ws_bytes,
                            phase, (hipStream_t)stream_);
}


// ------------------------------------------------------------------------------------------------
// The prior of a captured training step as one launch + the dz' / dlogvar reduction (evae_prior_train.h)
// ------------------------------------------------------------------------------------------------
template <int KG> static size_t prior_train_lds() {
  constexpr int KS2 = KG * 8 + 4;
  return (size_t)(2 * 128 * KS2 + 128 * 132 + 4 * 128 + 64 + 64 + 16 + 8 + 8 * 64 + 2 * 128) * sizeof(float) + 128 * sizeof(long long);
}
template <int KG> static int prior_train_occ() {
  int n = 0;
  (void)hipFuncSetAttribute((const void*)prior_train_kernel<KG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)prior_train_lds<KG>());
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, prior_train_kernel<KG>, MFT, prior_train_lds<KG>()) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}
// Blocks of prior_train_kernel<KG> the CURRENT device keeps resident at once (its inter-block exchange spins on a generation word:
// the whole grid must be co-resident -- ADVICE r04): CUs of the device (a CPX / DPX partition or a smaller part reports fewer) x
// blocks per CU from the occupancy query for the kernel's registers and LDS, minus a margin of one CU in sixteen for launches of
// other streams that hold CUs while the grid starts, never above PT_MAX_BLOCKS.  0 = no device / query failed: the one-launch
// form is then not offered and callers take the three-launch prior.
static int prior_train_block_limit(int kg) {
  static int cache[16][8];            // [device][kg]: 0 = not asked yet, -1 = none
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (dev < 0 || dev >= 16 || kg < 1 || kg > 7) return 0;
  if (cache[dev][kg] == 0) {
    int cus = 0, occ = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); cus = 0; }
    switch (kg) {
      case 1: occ = prior_train_occ<1>(); break;
      case 2: occ = prior_train_occ<2>(); break;
      case 3: occ = prior_train_occ<3>(); break;
      case 4: occ = prior_train_occ<4>(); break;
      case 5: occ = prior_train_occ<5>(); break;
      case 6: occ = prior_train_occ<6>(); break;
      default: occ = prior_train_occ<7>(); break;
    }
    const long lim = (long)cus * occ - (cus + 15) / 16;
    cache[dev][kg] = lim > 0 ? (int)std::min<long>(lim, PT_MAX_BLOCKS) : -1;
  }
  return cache[dev][kg] > 0 ? cache[dev][kg] : 0;
}

extern "C" int evae_prior_train_applies(int B, int C, int zdim) {
  static int off = -1;
  if (off < 0) { const char* e = getenv("EVAE_PRIOR_TRAIN"); off = (e && atoi(e) == 0) ? 1 : 0; }
  if (off) return 0;
  if (!(B >= 1 && B <= MFQ && C >= 1 && zdim >= 4 && zdim <= 56 && (zdim & 3) == 0)) return 0;
  return cdiv(C, MFE) <= prior_train_block_limit((zdim + 7) / 8);
}

// Blocks that gave up waiting in any evae_prior_train_step launch on this state block since it was zeroed (state[10]); read and
// cleared by the host at a point where it synchronises anyway (end of epoch, replica check): non-zero = some step's dz /
// dcentres / dlogvar were formed with a stale token -- the caller must raise.
extern "C" int evae_prior_train_gave_up(void* state, evae_stream_t stream_) {
  EVAE_REQUIRE(state, "prior_train_gave_up: null state");
  unsigned v = 0;
  hipStream_t stream = (hipStream_t)stream_;
  if (hipMemcpyAsync(&v, (const unsigned*)state + 10, sizeof(unsigned), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess) { set_error("prior_train_gave_up: read-back failed"); return EVAE_EINVAL; }
  if (v) (void)hipMemsetAsync((unsigned*)state + 10, 0, sizeof(unsigned), stream);
  return (int)std::min<unsigned>(v, 0x3FFFFFFFu);
}

static size_t prior_train_layout(int B, int C, int zdim, size_t* o_gpart, size_t* o_dz, size_t* o_dlv) {
  const size_t nblk = (size_t)cdiv(C, MFE);
  size_t o = align_up(3 * nblk * MFQ * sizeof(float), 256);
  *o_gpart = o; o += align_up((size_t)3 * 8 * MFQ * sizeof(float), 256);
  *o_dz = o;    o += align_up(nblk * B * zdim * sizeof(float), 256);
  *o_dlv = o;   o += align_up(nblk * (zdim + 1) * sizeof(float), 256);
  return o;
}

extern "C" size_t evae_prior_train_workspace_bytes(int B, int C, int zdim) {
  if (B <= 0 || C <= 0 || zdim <= 0) return 256;
  size_t a, b, c;
  return prior_train_layout(B, C, zdim, &a, &b, &c);
}

template <int KG>
static int launch_prior_train(const float* z, int B, const float* centres, const int64_t* row_map, int C, int zdim, const float* log_var,
                              const int64_t* z_idx, const int64_t* c_idx, float c_total, const float* beta_dev, float beta_host,
                              unsigned* state, float* part, float* gpart, float* logp, float* token, float* cRE, float* cKL,
                              float* ncKL, float* dz_part, float* dc, float* dlv_part, hipStream_t stream) {
  const size_t lds = prior_train_lds<KG>();
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)prior_train_kernel<KG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  prior_train_kernel<KG><<<cdiv(C, MFE), MFT, lds, stream>>>(z, B, centres, row_map, C, zdim, log_var, z_idx, c_idx, c_total, beta_dev, beta_host,
                                                            prior_norm_limit(), state, part, gpart, logp, token, cRE, cKL, ncKL,
                                                            dz_part, dc, dlv_part);
  return check_launch("prior_train_kernel");
}

// rows_inv != NULL: the step encoded each DISTINCT image of its draw once -- `centres` holds the n_rows distinct rows' encodings,
// exemplar j of the prior is row rows_inv[j]; the per-draw centre gradients go to dc_draws [C x zdim] (scratch) and the reduction launch
// folds them onto the distinct rows: dcentres[u] = rows_mult[u] * dc_draws[rows_rep[u]], u < n_rows
static int prior_train_step_core(const float* z, int B, const float* centres, int C, int zdim, const float* log_var,
                                 const int64_t* z_idx, const int64_t* c_idx, float c_total, const float* beta_dev,
                                 float beta_host, float* logp, float* token, float* cRE, float* cKL, float* neg_cKL,
                                 float* dz, float* dcentres, float* dlogvar, void* state, void* ws, size_t ws_bytes,
                                 int phase, evae_stream_t stream_, const int64_t* rows_inv, const int64_t* rows_rep,
                                 const float* rows_mult, int n_rows, float* dc_draws) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(phase >= 0 && phase <= 2, "prior_train_step: phase must be 0, 1 or 2");
  EVAE_REQUIRE(evae_prior_train_applies(B, C, zdim), "prior_train_step: B=%d C=%d zdim=%d is outside the one-launch form (B <= 128, "
               "C <= %d = the blocks this device keeps resident x 128, zdim <= 56 and a multiple of 4)", B, C, zdim,
               prior_train_block_limit((zdim + 7) / 8) * MFE);
  EVAE_REQUIRE(z && centres && log_var && logp && token && dz && dcentres && dlogvar && state && ws, "prior_train_step: null pointer");
  EVAE_REQUIRE((cRE && cKL && neg_cKL) || (!cRE && !cKL && !neg_cKL), "prior_train_step: all three coefficient vectors or none");
  EVAE_REQUIRE(((((uintptr_t)z) | ((uintptr_t)centres)) & 15) == 0, "prior_train_step: z and centres must be 16-byte aligned");
  size_t o_gpart, o_dz, o_dlv;
  const size_t need = prior_train_layout(B, C, zdim, &o_gpart, &o_dz, &o_dlv);
  if (ws_bytes < need) { set_error("prior_train_step: workspace too small (%zu < %zu)", ws_bytes, need); return EVAE_EWORKSPACE; }
  float* part = (float*)ws;
  float* gpart = (float*)((char*)ws + o_gpart);
  float* dz_part = (float*)((char*)ws + o_dz);
  float* dlv_part = (float*)((char*)ws + o_dlv);
  const int nblk = cdiv(C, MFE);
  int rc = EVAE_OK;
  if (phase != 2) {
#define EVAE_PT(KG_) rc = launch_prior_train<KG_>(z, B, centres, rows_inv, C, zdim, log_var, z_idx, c_idx, c_total, beta_dev, beta_host, \
                                                  (unsigned*)state, part, gpart, logp, token, cRE, cKL, neg_cKL, dz_part, \
                                                  rows_inv ? dc_draws : dcentres, dlv_part, stream)
    switch ((zdim + 7) / 8) {
      case 1: EVAE_PT(1); break;
      case 2: EVAE_PT(2); break;
      case 3: EVAE_PT(3); break;
      case 4: EVAE_PT(4); break;
      case 5: EVAE_PT(5); break;
      case 6: EVAE_PT(6); break;
      default: EVAE_PT(7); break;
    }
#undef EVAE_PT
    if (rc) return rc;
  }
  if (phase == 1) return EVAE_OK;
  const int nb = cdiv(B * zdim, 64), nlv = cdiv(zdim, 16);
  const int nfold = rows_inv ? cdiv(n_rows * (zdim / 4), 1024) : 0;
  prior_bwd_finish_kernel<<<nb + nlv + nfold, 1024, 0, stream>>>(dz_part, nblk, B * zdim, zdim, log_var, dz, nb, dlv_part, nblk,
                                                                dlogvar, nullptr, rows_inv ? dc_draws : nullptr, rows_rep, rows_mult,
                                                                rows_inv ? dcentres : nullptr, n_rows, nlv);
  return check_launch("prior_bwd_finish_kernel");
}

extern "C" int evae_prior_train_step(const float* z, int B, const float* centres, int C, int zdim, const float* log_var,
                                     const int64_t* z_idx, const int64_t* c_idx, float c_total, const float* beta_dev,
                                     float beta_host, float* logp, float* token, float* cRE, float* cKL, float* neg_cKL,
                                     float* dz, float* dcentres, float* dlogvar, void* state, void* ws, size_t ws_bytes,
                                     int phase, evae_stream_t stream_) {
  return prior_train_step_core(z, B, centres, C, zdim, log_var, z_idx, c_idx, c_total, beta_dev, beta_host, logp, token, cRE, cKL, neg_cKL,
                               dz, dcentres, dlogvar, state, ws, ws_bytes, phase, stream_, nullptr, nullptr, nullptr, 0, nullptr);
}

extern "C" int evae_prior_train_step_rows(const float* z, int B, const float* centres, int n_rows, const int64_t* rows_inv,
                                          const int64_t* rows_rep, const float* rows_mult, int C, int zdim, const float* log_var,
                                          const int64_t* z_idx, const int64_t* c_idx, float c_total, const float* beta_dev,
                                          float beta_host, float* logp, float* token, float* cRE, float* cKL, float* neg_cKL,
                                          float* dz, float* dcentres, float* dc_draws, float* dlogvar, void* state, void* ws,
                                          size_t ws_bytes, evae_stream_t stream_) {
  EVAE_REQUIRE(rows_inv && rows_rep && rows_mult && dc_draws && n_rows >= 1, "prior_train_step_rows: null row tables");
  EVAE_REQUIRE((((uintptr_t)dc_draws | (uintptr_t)dcentres) & 15) == 0, "prior_train_step_rows: gradient buffers must be 16-byte aligned");
  return prior_train_step_core(z, B, centres, C, zdim, log_var, z_idx, c_idx, c_total, beta_dev, beta_host, logp, token, cRE, cKL, neg_cKL,
                               dz, dcentres, dlogvar, state, ws, ws_bytes, 0, stream_, rows_inv, rows_rep, rows_mult, n_rows, dc_draws);
}
