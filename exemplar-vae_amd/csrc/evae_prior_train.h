// The exemplar prior of a TRAINING step as one launch (r04): forward partials, merge and backward, which the step otherwise runs
// as prior_fwd_mfma_kernel -> merge / ELBO launch -> prior_bwd_mfma_kernel (15 + 11 + 29 us at 100 queries x 25 000 exemplars,
// every one of them latency-bound, plus two launch seams on the step's critical path).  Included by evae_prior.hip inside
// namespace evae (it uses that file's staging helpers and the two kernels' conventions: 128 exemplars x 128 queries per block,
// eight waves as 4 wave rows x 2 wave columns, S = Es . Qs^T on v_mfma_f32_32x32x2f32).
//
// What makes one launch possible: with B <= 128 queries a block's whole tile of S lives in its accumulators, and the ONLY thing
// the backward needs from outside the block is the merged (row max, log sum) token of its queries.  So: every block computes S
// for its 128 exemplars once, forms its partial (max, sum exp, #masked) row exactly as prior_fwd_mfma_kernel does, publishes
// it, and the rows are merged in two fixed-order levels by last arrivers -- group g = the blocks with index = g (mod 8) (the
// blocks one XCD usually runs: a pure speed choice), then the <= 8 group rows -- after which every block reads the token and
// carries on with prior_bwd_mfma_kernel's tail on the S it still holds (P, T = P [Zs | 1], U = P^T [Cs | 1], dcentres, the
// dz' / dlogvar partials of prior_bwd_finish_kernel).  Merge order is by block index, never by arrival: deterministic.
//
// Inter-block protocol (never placement-dependent): everything that crosses blocks inside the launch -- a row is 1.5 KB -- goes
// through write-through (sc1) stores and L1-bypassing (sc1) loads, so no L2 write-back / L1 invalidate fence is needed (the
// release fence writes back whatever ELSE is dirty in the XCD's L2: 5.7 us per arrival measured inside the step, the whole
// exchange 13.7 -> see DESIGN): payload stores -> every wave s_waitcnt vmcnt(0) -> barrier -> lane 0: relaxed agent ticket.  A last
// arriver reads the rows with sc1 loads.  The final one publishes the token, zeroes the nine counters (all arrivals of this
// launch are behind it), drains, and bumps the generation word; every block read that word when it started (it cannot change
// before all blocks have arrived) and polls it relaxed with s_sleep.  state = 64 words, zeroed ONCE by the caller.
// The grid must be co-resident: one block per CU (LDS), so the entry point refuses more than 240 blocks; the poll is bounded
// (state[10] counts blocks that gave up -- tests read it).
#pragma once

namespace evae {

constexpr int PT_MAX_BLOCKS = 240;
#ifndef EVAE_PT_STAMPS
#define EVAE_PT_STAMPS 0
#endif
constexpr bool PT_STAMPS = EVAE_PT_STAMPS != 0;

// what crosses blocks inside the launch: write-through (sc1) stores and L1-bypassing (sc1) loads
__device__ __forceinline__ void pt_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float pt_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void pt_merge_one(float& m, float& s, float& n, float mr, float sr, float nr) {
  n += nr;
  if (mr > m) { s = s * expf(m - mr) + sr; m = mr; }           // m == -inf: s == 0 and exp(-inf) = 0
  else if (mr != -INFINITY) s += sr * expf(mr - m);
}

template <int KG>
__global__ __launch_bounds__(MFT) void prior_train_kernel(
    const float* __restrict__ z, int B, const float* __restrict__ centres, const int64_t* __restrict__ row_map /* [C] or NULL */, int C, int zdim,
    const float* __restrict__ log_var, const int64_t* __restrict__ z_idx, const int64_t* __restrict__ c_idx,
    float c_total, const float* __restrict__ beta_dev, float beta_host, float norm_limit,
    unsigned* __restrict__ state, float* __restrict__ part /* [3][nblk][128] */, float* __restrict__ gpart /* [3][8][128] */,
    float* __restrict__ logp, float* __restrict__ token /* [2 B] */, float* __restrict__ cRE, float* __restrict__ cKL,
    float* __restrict__ ncKL, float* __restrict__ dz_part /* [nblk][B][zdim] */, float* __restrict__ dc /* [C][zdim] */,
    float* __restrict__ dlv_part /* [nblk][zdim + 1] */) {
  constexpr int KP = KG * 8, KS2 = KP + 4, CPR = KP / 4, PP = 132;
  constexpr int NV = (MFE * CPR + MFT - 1) / MFT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;                          // [128][KS2]   scaled queries, column KP = 1
  float* Es = Qs + MFQ * KS2;                // [128][KS2]   scaled exemplars, column KP = 1
  float* Ps = Es + MFE * KS2;                // [128 e][PP]  (before the token exists: combine / merge scratch)
  float* zn = Ps + MFE * PP;                 // [128]
  float* cn = zn + MFQ;                      // [128]
  float* cs = cn + MFE;                      // [128] column sums of P (per exemplar)
  float* rs = cs + MFE;                      // [128] row sums of P (per query)
  float* inv_sigma = rs + MFQ;               // [64]
  float* mu_s = inv_sigma + 64;              // [64]
  float* red = mu_s + 64;                    // [16]  ([8..11]: flags of the inter-block protocol)
  float* zmx = red + 16;                     // [2] (+6 padding)
  float* dvs = zmx + 8;                      // [8][64]
  float* tok = dvs + 8 * 64;                 // [2][128] the merged token of this block's queries
  long long* ci_s = reinterpret_cast<long long*>(tok + 2 * MFQ);   // [128]
  unsigned* flg = reinterpret_cast<unsigned*>(red + 8);

  const int nblk = gridDim.x, blk = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, lh = lane >> 5;
  const bool masked = (z_idx != nullptr) && (c_idx != nullptr);
  unsigned gen0 = 0;
  if (tid == 0) gen0 = __hip_atomic_load(&state[9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // phase stamps of block 0 (100 MHz wall clock, words 16..31 of the state block; tools/prior_train_probe.py reads them)
  auto stamp = [&](int i) { if (PT_STAMPS && blk == 0 && tid == 0) state[16 + i] = (unsigned)wall_clock64(); };
  stamp(0);
  const float cst = setup_sigma(inv_sigma, red, log_var, zdim, KP);   // contains a barrier

  // (map: exemplar j of the prior is row map[j] of src -- the draws of a step that encoded each DISTINCT image once, r06)
  auto load_tile = [&](const float* src, const int64_t* map, int r0, int nrows, float4 (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + MFT * i;
      const int r = f / CPR, c = f - r * CPR;
      const bool ok = f < MFE * CPR && r0 + r < nrows && 4 * c + 4 <= zdim;
      const size_t row = ok ? (map ? (size_t)map[r0 + r] : (size_t)(r0 + r)) : 0;
      v[i] = ok ? *reinterpret_cast<const float4*>(src + row * zdim + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
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

  float4 rv[NV];
  load_tile(z, nullptr, 0, B, rv);
  store_tile(Qs, rv, false);
  const int e0 = blk * MFE;
  load_tile(centres, row_map, e0, C, rv);
  __syncthreads();
  const bool slow = centre_queries<KP, KS2, MFT>(Qs, mu_s, zn, zmx, Ps, B < MFQ ? B : MFQ) > norm_limit;
  stamp(1);

  store_tile(Es, rv, true);
  if (tid < MFE) ci_s[tid] = (masked && e0 + tid < C) ? (long long)c_idx[e0 + tid] : -2;
  __syncthreads();
  if (tid < MFE) {                           // squared row norms of the exemplar tile
    float sacc = 0.f;
#pragma unroll
    for (int c = 0; c < CPR; ++c) {
      const float4 t = *reinterpret_cast<const float4*>(Es + tid * KS2 + 4 * c);
      sacc += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
    }
    cn[tid] = sacc;
  }

  // ---- S = Es . Qs^T (slow: squared distances by direct differences), kept in registers until the end
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
  if (slow) direct_tile<KP, KS2>(acc, Es, Qs, wr, wc, l31, lh);      // acc <- d2
  stamp(2);

  // this lane's two query columns
  float znq[2];
  long long zi[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int ql = wc * 64 + nt * 32 + l31;
    znq[nt] = zn[ql];
    zi[nt] = (masked && ql < B) ? (long long)z_idx[ql] : -1;
  }

  // ---- forward: this block's partial (max, sum exp, #masked) per query: the arithmetic of prior_fwd_mfma_kernel's tile loop for
  //      one tile, branch-free -- the pairs in use are one 16-bit mask per query column, which the backward part below takes over
  unsigned use2[2];
  float cnr[16];
  {
    unsigned live = 0, hit0 = 0, hit1 = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int el = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      cnr[r] = cn[el];
      if (e0 + el < C) live |= 1u << r;
      if (masked) {
        const long long ce = ci_s[el];
        hit0 |= (unsigned)((ce == zi[0]) | (ce == kMaskAll)) << r;
        hit1 |= (unsigned)((ce == zi[1]) | (ce == kMaskAll)) << r;
      }
    }
    use2[0] = live & ~hit0; use2[1] = live & ~hit1;
    float dmin[2], ssum[2], nmask[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const unsigned use = use2[nt];
      nmask[nt] = (float)__popc(live & (nt ? hit1 : hit0));
      const float off = slow ? 0.f : 0.5f * znq[nt];
      float v[16];
      float vmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r] = slow ? -0.5f * acc[nt][r] : acc[nt][r] - 0.5f * cnr[r];
        vmax = fmaxf(vmax, ((use >> r) & 1u) ? v[r] : -INFINITY);
      }
      const float um = vmax - off;                    // -inf: no pair in use
      const float mk = -(um + off) * kLog2e;
      float ss = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = slow ? fast_exp2((v[r] - um) * kLog2e) : fast_exp2(fmaf(v[r], kLog2e, mk));
        ss += ((use >> r) & 1u) ? e : 0.f;
      }
      ssum[nt] = ss;
      dmin[nt] = (um == -INFINITY) ? INFINITY : fmaxf(-2.0f * um, 0.f);
    }
    float* comb = Ps;                         // [4][128][3]
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
    if (tid < MFQ) {
      float m = INFINITY, sacc = 0.f, nacc = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) m = fminf(m, comb[(w * MFQ + tid) * 3]);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float dw = comb[(w * MFQ + tid) * 3];
        if (dw != INFINITY) sacc += comb[(w * MFQ + tid) * 3 + 1] * fast_exp2((m - dw) * kHalfLog2e);
        nacc += comb[(w * MFQ + tid) * 3 + 2];
      }
      const size_t o = (size_t)blk * MFQ + tid;
      const size_t pl = (size_t)nblk * MFQ;
      pt_st(part + o, (m == INFINITY) ? -INFINITY : cst - 0.5f * m);
      pt_st(part + pl + o, sacc);
      pt_st(part + 2 * pl + o, nacc);
    }
  }

  stamp(3);
  // ---- publish the row; two levels of last arrivers merge in block order
  const int grp = blk & 7;
  const int ngrp = nblk < 8 ? nblk : 8;
  const int n_g = (nblk - grp + 7) >> 3;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(&state[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flg[0] = (old == (unsigned)(n_g - 1)) ? 1u : 0u;
    flg[1] = 0u;
  }
  __syncthreads();
  if (flg[0]) {                                                  // block-uniform: last of its group
    float* mm = Ps;                                              // [3][4][128]
    {
      const int q = tid & 127, sub = tid >> 7;
      const size_t pl = (size_t)nblk * MFQ;
      float m = -INFINITY, s = 0.f, n = 0.f;
      for (int k0 = sub; k0 < n_g; k0 += 16) {                   // this quarter's rows k0, k0 + 4, ..: four loads in flight
        float mr[4], sr[4], nr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + 4 * u;
          const bool ok = k < n_g;
          const size_t o = (size_t)(grp + 8 * (ok ? k : 0)) * MFQ + q;
          mr[u] = ok ? pt_ld(part + o) : -INFINITY; sr[u] = ok ? pt_ld(part + pl + o) : 0.f; nr[u] = ok ? pt_ld(part + 2 * pl + o) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) pt_merge_one(m, s, n, mr[u], sr[u], nr[u]);
      }
      mm[(0 * 4 + sub) * MFQ + q] = m; mm[(1 * 4 + sub) * MFQ + q] = s; mm[(2 * 4 + sub) * MFQ + q] = n;
    }
    __syncthreads();
    if (tid < MFQ) {
      float m = -INFINITY, s = 0.f, n = 0.f;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) pt_merge_one(m, s, n, mm[(0 * 4 + sub) * MFQ + tid], mm[(1 * 4 + sub) * MFQ + tid], mm[(2 * 4 + sub) * MFQ + tid]);
      pt_st(gpart + (0 * 8 + grp) * MFQ + tid, m); pt_st(gpart + (1 * 8 + grp) * MFQ + tid, s); pt_st(gpart + (2 * 8 + grp) * MFQ + tid, n);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(&state[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flg[1] = (old == (unsigned)(ngrp - 1)) ? 1u : 0u;
    }
    __syncthreads();
    if (flg[1]) {                                                // the last group: the token, log p, the coefficients
      if (tid < MFQ) {
        float mr[8], sr[8], nr[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const bool ok = g < ngrp;
          mr[g] = ok ? pt_ld(gpart + (0 * 8 + g) * MFQ + tid) : -INFINITY;
          sr[g] = ok ? pt_ld(gpart + (1 * 8 + g) * MFQ + tid) : 0.f;
          nr[g] = ok ? pt_ld(gpart + (2 * 8 + g) * MFQ + tid) : 0.f;
        }
        float m = -INFINITY, s = 0.f, n = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) pt_merge_one(m, s, n, mr[g], sr[g], nr[g]);
        if (tid < B) {
          const float ls = logf(s);
          logp[tid] = (m + ls) - logf(c_total - n);
          pt_st(token + tid, m); pt_st(token + B + tid, (m == -INFINITY) ? 0.f : ls);     // (max, log sum): prior_merge_kernel's token
          if (cRE != nullptr) {                 // coefficients of "batch mean of the loss, upstream gradient 1" (evae_elbo_bwd)
            const float beta = beta_dev ? beta_dev[0] : beta_host;
            const float gl = 1.0f / (float)B;
            cRE[tid] = 0.f - gl; cKL[tid] = 0.f + beta * gl; ncKL[tid] = -(0.f + beta * gl);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        // every arrival of this launch is behind us: the counters start the next launch at zero
#pragma unroll
        for (int i = 0; i < 9; ++i) __hip_atomic_store(&state[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&state[9], gen0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  stamp(4);
  // ---- every block: wait for the generation word to move, then take the token
  if (tid == 0) {
    int spins = 0;
    while (__hip_atomic_load(&state[9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 21)) { __hip_atomic_fetch_add(&state[10], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
  stamp(5);
  const float beta = beta_dev ? beta_dev[0] : beta_host;
  if (tid < MFQ) {
    tok[tid] = tid < B ? pt_ld(token + tid) : 0.f;
    tok[MFQ + tid] = tid < B ? pt_ld(token + B + tid) : 0.f;
  }
  __syncthreads();

  // ---- backward: prior_bwd_mfma_kernel's tile body on the S this block holds; upstream coefficient of log p = -beta / B
  float gq[2], lq[2], lq2[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int ql = wc * 64 + nt * 32 + l31;
    gq[nt] = ql < B ? -(0.f + beta * (1.0f / (float)B)) : 0.f;
    lq[nt] = tok[ql];
    lq2[nt] = tok[MFQ + ql] * kLog2e;
  }
  const int ncol = wc * 32 + l31;
  const int nread = ncol < KS2 ? ncol : KS2 - 1;
  const float isg = ncol < zdim ? inv_sigma[ncol] : 0.f;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int ql = wc * 64 + nt * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int el = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float d = slow ? acc[nt][r] : fmaxf(cnr[r] + znq[nt] - 2.0f * acc[nt][r], 0.f);
      const bool ok = (use2[nt] >> r) & 1u;
      const float w = gq[nt] * fast_exp2(fmaf(fmaf(-0.5f, d, cst) - lq[nt], kLog2e, -lq2[nt]));
      Ps[el * PP + ql] = ok ? w : 0.f;
    }
  }
  __syncthreads();

  stamp(6);
  f32x16_t T, U;
#pragma unroll
  for (int r = 0; r < 16; ++r) { T[r] = 0.f; U[r] = 0.f; }
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
  if (ncol == KP) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      cs[wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = T[r];
      rs[wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = U[r];
    }
  }
  __syncthreads();
  stamp(7);
  float dv_acc = 0.f;
  const float gw_acc = tid < MFE ? cs[tid] : 0.f;
  if (ncol < zdim) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int el = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float c_ = Es[el * KS2 + ncol], csum = cs[el];
      dv_acc += c_ * (csum * c_ - 2.0f * T[r]);
      const int e = e0 + el;
      if (e < C) dc[(size_t)e * zdim + ncol] = (T[r] - csum * c_) * isg;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ql = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float z_ = Qs[ql * KS2 + ncol], rsum = rs[ql];
      dv_acc += rsum * z_ * z_;
      if (ql < B) dz_part[((size_t)blk * B + ql) * zdim + ncol] = U[r] - rsum * z_;
    }
  }
  dvs[(wr * 2 + lh) * 64 + ncol] = dv_acc;
  __syncthreads();
  float* dlv = dlv_part + (size_t)blk * (zdim + 1);
  if (tid < zdim) {
    float v = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) v += dvs[p * 64 + tid];     // fixed order
    dlv[tid] = v;
  }
  __syncthreads();
  if (wave < 2) {
    const float sg = wave_sum(gw_acc);
    if (lane == 0) dvs[wave] = sg;
  }
  __syncthreads();
  if (tid == 0) dlv[zdim] = dvs[0] + dvs[1];
  stamp(8);
}

}  // namespace evae
