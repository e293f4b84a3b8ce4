// Top-K by screening: the all-pairs distances are a GEMM (cache rows x queries on the matrix cores, the kernels of
// evae_gemm_kernel.h / evae_gemm_x6.h with a distance epilogue), but only to FIND the few exemplars that can be among the k
// nearest; those are then re-evaluated exactly -- fp64 direct differences rounded once to fp32, the arithmetic of the scan
// kernel in evae_topk.hip -- and ordered by (value, index).  Indices and values are bit-identical to the scan kernel
// (and to the reference's fp64 distance + topk) because the candidate set provably contains the true top-k:
//   1. squared norms of the queries (fp32); those of the cache rows are accumulated by the GEMM while it stages them (split-
//      bf16 kernel) or by a pass of their own (fp32 kernel), with the largest exemplar norm;
//   2. ONE screening GEMM: per query the minimum approximate distance of every 128-exemplar tile, and all approximate
//      distances kept.  The k-th smallest tile minimum T is an upper bound of the k-th smallest approximate distance (k tiles
//      hold one element <= T each);
//   3. with E = gamma (|q|^2 + max|c|^2) bounding the error of one approximate distance (gamma derived in topk_screen for
//      either kernel), every true top-k member has approximate distance <= T + 2E: a scan of the kept distances appends
//      exactly those rows to the query's candidate list;
//   4. exact distances of the candidates, each ranked by the (value, index) pairs in front of it.
// Cost: one pass over the [N x z] cache at streaming speed + a write and a read of the [N x B] distances.
// Launches (r04): TWO on the split-bf16 kernel -- the screening GEMM (cache norms and their per-tile maxima accumulated while
// it stages the rows; distances stored query-major, without the query's own norm) and topk_finish_kernel, one 1024-thread block
// per query for steps 1 (its norm), 2's threshold, 3 and 4.  c5 size 89 -> 66 us, c2 size 40 -> 30 us against the five launches
// (query norms + clears, GEMM, threshold, collect, exact) that remain the form of the fp32 kernel (EVAE_X6=0, EVAE_TOPK_TWO_LAUNCH=0).
#include "evae_gemm_x6.h"
#include "evae_topk_screen.h"

namespace evae {

// |row|^2 in fp32; rows of the cache also feed the global maximum (positive floats order like their bit patterns).
// zdim % 4 == 0.  Wide rows (>= 128 floats): one wave per row, float4 per lane, four rows in flight per wave; narrow rows:
// one thread per row (a 64-lane read then covers 64 short rows; every 128-byte line is used in full over the loop).
// `zero_ints` (query launch only): per-query candidate counters to clear, `zero_bits` the running maximum -- the query launch
// runs first, so the two housekeeping stores need no launch of their own.
__global__ __launch_bounds__(256) void sq_norms_kernel(const float* __restrict__ x, int rows, int zdim, float* __restrict__ out,
                                                       unsigned* __restrict__ max_bits, int* __restrict__ zero_ints, int nzero,
                                                       unsigned* __restrict__ zero_bits) {
  const int lane = threadIdx.x & 63;
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
  if (zero_ints)
    for (int i = gtid; i < nzero; i += gridDim.x * blockDim.x) zero_ints[i] = 0;
  if (zero_bits && gtid == 0) *zero_bits = 0u;
  float wmax = 0.f;
  const int cpr = zdim >> 2;
  if (zdim >= 128) {
    const int wpb = blockDim.x >> 6, nw = gridDim.x * wpb;
    for (int row0 = (blockIdx.x * wpb + (threadIdx.x >> 6)) * 4; row0 < rows; row0 += nw * 4) {
      float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (row0 + r < rows)
          for (int c = lane; c < cpr; c += 64) {
            const float4 v = *reinterpret_cast<const float4*>(x + (size_t)(row0 + r) * zdim + 4 * c);
            s[r] += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
          }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = wave_sum(s[r]);
        if (lane == 0 && row0 + r < rows) out[row0 + r] = t;
        wmax = fmaxf(wmax, t);
      }
    }
  } else {
    for (int row = gtid; row < rows; row += gridDim.x * blockDim.x) {
      float t = 0.f;
      for (int c = 0; c < cpr; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * zdim + 4 * c);
        t += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
      out[row] = t;
      wmax = fmaxf(wmax, t);
    }
    wmax = wave_max(wmax);
  }
  // one atomic per BLOCK: same-address atomics from a whole device serialise at the memory side (16 384 of them cost
  // more than streaming the 100 MB cache)
  if (max_bits) {
    __shared__ float wm[4];
    if (lane == 0) wm[threadIdx.x >> 6] = wmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
      if (m > 0.f) atomicMax(max_bits, __float_as_uint(m));
    }
  }
}

// thr[n] = (k-th smallest tile minimum of query n) + 2 gamma (qn[n] + cnmax); one BLOCK per query, tmin query-major [B][ldt]
// (ldt >= ntiles: contiguous loads).  Only the VALUE of the k-th smallest minimum matters, so it is found by a bit-wise search
// over the order-preserving integer image of a float: 32 steps of "how many minima lie below t" = a ballot + population count
// per register, the four waves' counts met through LDS (one barrier per step, counters double-buffered) -- no cross-lane data
// movement (the k rounds of wave-wide (value, tile) arg-min reductions of r01 cost 20 us; one wave per query with 13 registers
// of minima 16 us at 782 tiles).  A thread holds <= 4 minima (up to 1024 tiles = 131 072 exemplars); longer caches are
// re-read in every step.
__device__ __forceinline__ unsigned ordered_key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
// the k-th smallest of row[0 .. ntiles) as an ordered key; every thread of the 256-thread block returns it
template <int NT = 256>
__device__ __forceinline__ unsigned kth_smallest_key(const float* __restrict__ row, int ntiles, int k) {
  constexpr int NWV = NT / 64;
  __shared__ int cnt[2][NWV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int TR = 1024 / NT;
  const bool in_regs = ntiles <= NT * TR;
  const int used = in_regs ? (ntiles + NT - 1) / NT : 0;          // registers that hold anything (block-uniform)
  unsigned key[TR];
#pragma unroll
  for (int j = 0; j < TR; ++j) {
    const int t = threadIdx.x + NT * j;
    key[j] = (in_regs && t < ntiles) ? ordered_key(row[t]) : 0xFFFFFFFFu;
  }
  // the k-th smallest key = the largest t with #(key < t) <= k - 1 (fewer than k minima exist: the largest key, +inf's image)
  unsigned ans = 0u;
  for (int b = 31; b >= 0; --b) {
    const unsigned t = ans | (1u << b);
    int below = 0;
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < TR; ++j)
        if (j < used) below += __popcll(__ballot(key[j] < t));
    } else {
      for (int tt = threadIdx.x; tt < ((ntiles + NT - 1) / NT) * NT; tt += NT)
        below += __popcll(__ballot(tt < ntiles && ordered_key(row[tt]) < t));
    }
    if (lane == 0) cnt[b & 1][wave] = below;
    __syncthreads();
    int all = 0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) all += cnt[b & 1][w];
    if (all <= k - 1) ans = t;
  }
  return ans;
}
__global__ __launch_bounds__(256) void kth_threshold_kernel(const float* __restrict__ tmin, int ntiles, int ldt, int B, int k,
                                                            const float* __restrict__ qn, const unsigned* __restrict__ cnmax_bits,
                                                            float gamma, float* __restrict__ thr) {
  const int n = blockIdx.x;
  const unsigned ans = kth_smallest_key(tmin + (size_t)n * ldt, ntiles, k);
  if (threadIdx.x == 0) thr[n] = key_to_float(ans) + 2.0f * gamma * (qn[n] + __uint_as_float(*cnmax_bits));
}

// exact re-ranking of one query's candidates: block = 256 threads
constexpr int RANK_MAX = 1024;
// qs = the query row in LDS (z_dim <= 512 on this path, multiple of 4), cl / vl = query n's candidate rows / a value per candidate
template <int NT = 256>
__device__ __forceinline__ void topk_exact_body(const float* qs, const float* __restrict__ cache, int zdim, int k, unsigned flags,
                                                int64_t index_base, int n, int M, const int* cl, float* vl,
                                                int64_t* __restrict__ out_idx, float* __restrict__ out_val) {
  __shared__ float sv[NT / 64];
  __shared__ int si[NT / 64];
  const bool do_sqrt = (flags & EVAE_TOPK_SQRT) != 0;
  for (int c = threadIdx.x; c < M; c += NT) {
    const float* cr = cache + (size_t)cl[c] * zdim;
    double a = 0.0;
    for (int d = 0; d < zdim; d += 4) {    // same order and operations as dist_chunk_f64 (evae_topk.hip), 16-byte loads
      const float4 cv = *reinterpret_cast<const float4*>(cr + d);
      const float4 qv = *reinterpret_cast<const float4*>(qs + d);
      double df = (double)qv.x - (double)cv.x; a = fma(df, df, a);
      df = (double)qv.y - (double)cv.y; a = fma(df, df, a);
      df = (double)qv.z - (double)cv.z; a = fma(df, df, a);
      df = (double)qv.w - (double)cv.w; a = fma(df, df, a);
    }
    float v = (float)a;
    if (do_sqrt) v = sqrtf(v);
    vl[c] = v;
  }
  __syncthreads();
  if (M <= RANK_MAX) {
    // few candidates (the usual case: a few dozen): every candidate counts the (value, index) pairs in front of it -- the
    // candidates of a query are distinct rows, so the pairs are totally ordered and rank j IS output slot j; no rounds, no
    // barriers.  The values are the ones just stored by this block.
    __shared__ float rv[RANK_MAX];
    __shared__ int ri[RANK_MAX];
    for (int c = threadIdx.x; c < M; c += NT) { rv[c] = vl[c]; ri[c] = cl[c]; }
    __syncthreads();
    for (int c = threadIdx.x; c < M; c += NT) {
      const float v = rv[c];
      const int id = ri[c];
      int rank = 0;
      for (int o = 0; o < M; ++o) rank += ((rv[o] < v) || (rv[o] == v && ri[o] < id)) ? 1 : 0;
      if (rank < k) {
        out_idx[(size_t)n * k + rank] = (int64_t)id + index_base;
        if (out_val) out_val[(size_t)n * k + rank] = v;
      }
    }
    for (int j = M + threadIdx.x; j < k; j += NT) {      // fewer candidates than k: as the rounds below would leave them
      out_idx[(size_t)n * k + j] = (int64_t)-1;
      if (out_val) out_val[(size_t)n * k + j] = INFINITY;
    }
    return;
  }
  float lastv = -INFINITY;
  int lasti = -1;
  for (int j = 0; j < k; ++j) {
    float bv = INFINITY;
    int bi = INT_MAX;
    for (int c = threadIdx.x; c < M; c += NT) {
      const float v = vl[c];
      const int id = cl[c];
      const bool after = (v > lastv) || (v == lastv && id > lasti);
      if (after && ((v < bv) || (v == bv && id < bi))) { bv = v; bi = id; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if ((ov < bv) || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    bv = sv[0]; bi = si[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w)
      if ((sv[w] < bv) || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
    __syncthreads();
    lastv = bv; lasti = bi;
    if (threadIdx.x == 0) {
      out_idx[(size_t)n * k + j] = (bi == INT_MAX) ? (int64_t)-1 : (int64_t)bi + index_base;
      if (out_val) out_val[(size_t)n * k + j] = bv;
    }
  }
}

__global__ __launch_bounds__(256) void topk_exact_kernel(const float* __restrict__ q, const float* __restrict__ cache,
                                                         int zdim, int k, unsigned flags, int64_t index_base,
                                                         const int* __restrict__ cnt, const int* __restrict__ cand,
                                                         float* __restrict__ val, size_t ldc,
                                                         int64_t* __restrict__ out_idx, float* __restrict__ out_val) {
  const int n = blockIdx.x;
  __shared__ __attribute__((aligned(16))) float qs[512];
  for (int d = threadIdx.x; d < zdim; d += 256) qs[d] = q[(size_t)n * zdim + d];
  __syncthreads();
  topk_exact_body(qs, cache, zdim, k, flags, index_base, n, cnt[n], cand + (size_t)n * ldc, val + (size_t)n * ldc, out_idx, out_val);
}

// The two-launch form's second launch (r04): everything behind the screening GEMM for ONE query per block -- its squared norm,
// the largest cache-row norm (from the GEMM's per-tile maxima), the threshold (k-th smallest tile minimum + the error bound),
// the scan of ITS row of the query-major distances for candidates, their exact distances and ranks.  The distances carry no
// |q|^2 (the GEMM did not have it): a constant per query, it changes neither the order nor the bound.
constexpr int FIN_NT = 1024;      // sixteen waves: the scan of a query's row of distances wants many loads in flight on its one CU
__global__ __launch_bounds__(FIN_NT) void topk_finish_kernel(const float* __restrict__ q, const float* __restrict__ cache, int N,
                                                             int zdim, int k, unsigned flags, int64_t index_base,
                                                             const float* __restrict__ tmin, int ntiles, int ldm,
                                                             const float* __restrict__ tile_max, const float* __restrict__ dist,
                                                             int ldq, float gamma, int* __restrict__ cand, float* __restrict__ val,
                                                             size_t ldc, int64_t* __restrict__ out_idx, float* __restrict__ out_val) {
  constexpr int NWV = FIN_NT / 64;
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ __attribute__((aligned(16))) float qs[512];
  __shared__ float red[2][NWV];
  __shared__ int ncand;
  float qq = 0.f, cm = 0.f;
  for (int d = threadIdx.x; d < zdim; d += FIN_NT) { const float v = q[(size_t)n * zdim + d]; qs[d] = v; qq = fmaf(v, v, qq); }
  for (int t = threadIdx.x; t < 2 * ntiles; t += FIN_NT) cm = fmaxf(cm, tile_max[t]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { qq += __shfl_xor(qq, o, 64); cm = fmaxf(cm, __shfl_xor(cm, o, 64)); }
  if (lane == 0) { red[0][wave] = qq; red[1][wave] = cm; }
  if (threadIdx.x == 0) ncand = 0;
  __syncthreads();
  float qn = 0.f, cnmax = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) { qn += red[0][w]; cnmax = fmaxf(cnmax, red[1][w]); }
  const unsigned key = kth_smallest_key<FIN_NT>(tmin + (size_t)n * ldm, ntiles, k);
  const float thr = key_to_float(key) + 2.0f * gamma * (qn + cnmax);
  int* cl = cand + (size_t)n * ldc;
  const float* drow = dist + (size_t)n * ldq;
  // ldq is a multiple of 128: whole float4s, rows beyond N hold +inf.  Four loads per thread in flight (64 KB per sweep of the block)
  for (int m0 = 0; m0 < ldq; m0 += 4 * 4 * FIN_NT) {
    float4 d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + 4 * (threadIdx.x + j * FIN_NT);
      d[j] = m < ldq ? *reinterpret_cast<const float4*>(drow + m) : make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + 4 * (threadIdx.x + j * FIN_NT);
      if (fminf(fminf(d[j].x, d[j].y), fminf(d[j].z, d[j].w)) <= thr) {
        if (d[j].x <= thr) cl[atomicAdd(&ncand, 1)] = m;
        if (d[j].y <= thr) cl[atomicAdd(&ncand, 1)] = m + 1;
        if (d[j].z <= thr) cl[atomicAdd(&ncand, 1)] = m + 2;
        if (d[j].w <= thr) cl[atomicAdd(&ncand, 1)] = m + 3;
      }
    }
  }
  __syncthreads();
  topk_exact_body<FIN_NT>(qs, cache, zdim, k, flags, index_base, n, ncand, cl, val + (size_t)n * ldc, out_idx, out_val);
}

// Candidates from the stored approximate distances D[m][n] (row stride ldt): m joins the list of query n when D <= thr[n].
// One thread per four queries of a row; the per-query counters take one atomic per candidate (a few dozen per query).
__global__ __launch_bounds__(256) void dist_collect_kernel(const float* __restrict__ D, int N, int B, int ldt,
                                                           const float* __restrict__ thr, int* __restrict__ cnt,
                                                           int* __restrict__ cand, size_t ldc) {
  const int q4 = ldt / 4;
  const size_t total = (size_t)N * q4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / q4), n0 = 4 * (int)(i - (size_t)m * q4);
    if (n0 >= B) continue;
    const float4 d = *reinterpret_cast<const float4*>(D + (size_t)m * ldt + n0);
    const float4 t = *reinterpret_cast<const float4*>(thr + n0);
    if (d.x <= t.x) cand[(size_t)n0 * ldc + atomicAdd(&cnt[n0], 1)] = m;
    if (n0 + 1 < B && d.y <= t.y) cand[(size_t)(n0 + 1) * ldc + atomicAdd(&cnt[n0 + 1], 1)] = m;
    if (n0 + 2 < B && d.z <= t.z) cand[(size_t)(n0 + 2) * ldc + atomicAdd(&cnt[n0 + 2], 1)] = m;
    if (n0 + 3 < B && d.w <= t.w) cand[(size_t)(n0 + 3) * ldc + atomicAdd(&cnt[n0 + 3], 1)] = m;
  }
}

__global__ void zero_ints_kernel(int* p, int n, unsigned* q) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
  if (i == 0 && q) *q = 0u;
}

struct ScreenLayout {
  size_t cn, qn, cnmax, tmin, thr, cnt, cand, val, dist, tile_max, total;
  int ntiles, ldt, ldm, ldq;  // ldt: row stride of the distances (queries, padded); ldm: of the query-major tile minima;
                              // ldq: of the query-major distances (whole row tiles)
};
static bool screen_applies(int B, int N, int zdim, int k) {
  static int off = -1;
  if (off < 0) { const char* e = getenv("EVAE_TOPK_EXACT_SCAN"); off = (e && atoi(e)) ? 1 : 0; }
  if (off) return false;
  const int ntiles = cdiv(N, BM);
  return N >= 2048 && ntiles >= k && (zdim & 3) == 0 && (int64_t)B * N <= ((int64_t)1 << 26) &&
         zdim <= 512 && (int64_t)N * zdim < ((int64_t)1 << 29) - (1 << 22) && (int64_t)B * zdim < ((int64_t)1 << 29) - (1 << 22);
}
static ScreenLayout screen_layout(int B, int N) {
  ScreenLayout L;
  L.ntiles = cdiv(N, BM);
  L.ldt = (B + 63) / 64 * 64;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes, 256); return at; };
  L.cn = take((size_t)N * 4); L.qn = take((size_t)L.ldt * 4); L.cnmax = take(256);
  L.ldm = (L.ntiles + 63) / 64 * 64;
  L.tmin = take((size_t)L.ldm * L.ldt * 4); L.thr = take((size_t)L.ldt * 4); L.cnt = take((size_t)L.ldt * 4);
  L.cand = take((size_t)B * N * 4); L.val = take((size_t)B * N * 4);
  L.dist = take((size_t)N * L.ldt * 4);        // approximate distances of the screening GEMM, scanned by the collect pass
  L.ldq = L.ntiles * BM;
  if ((size_t)L.ldq * B > (size_t)N * L.ldt) L.dist = take((size_t)L.ldq * B * 4);      // query-major form, a few rows more
  L.tile_max = take((size_t)2 * L.ntiles * 4);
  L.total = o + 256;
  return L;
}

size_t topk_screen_workspace_bytes(int B, int N, int zdim, int k) {
  if (!screen_applies(B, N, zdim, k)) return 0;
  return screen_layout(B, N).total;
}

int topk_screen(const float* q, int B, const float* cache, int N, int zdim, int k, unsigned flags, int64_t index_base,
                int64_t* out_idx, float* out_val, void* ws, size_t ws_bytes, hipStream_t stream, int* handled) {
  *handled = 0;
  if (!screen_applies(B, N, zdim, k) || (((uintptr_t)q | (uintptr_t)cache) & 15) != 0) return EVAE_OK;
  const ScreenLayout L = screen_layout(B, N);
  if (ws == nullptr || ws_bytes < L.total) return EVAE_OK;
  *handled = 1;
  char* w = (char*)ws;
  float* cn = (float*)(w + L.cn); float* qn = (float*)(w + L.qn); unsigned* cnmax = (unsigned*)(w + L.cnmax);
  float* tmin = (float*)(w + L.tmin); float* thr = (float*)(w + L.thr); int* cnt = (int*)(w + L.cnt);
  int* cand = (int*)(w + L.cand); float* val = (float*)(w + L.val);
  float* dist = (float*)(w + L.dist);
  GemmArgs g = {};
  g.ones_col = -1;
  g.A[0] = cache; g.B[0] = q; g.lda[0] = zdim; g.ldb[0] = zdim; g.Kc[0] = zdim; g.npairs = 1;
  g.M = N; g.N = B; g.ksplit = 0;
  g.out0 = tmin; g.out1 = dist; g.ldo = L.ldt; g.ldo2 = L.ldm;
  static int two = -1;
  if (two < 0) { const char* e = getenv("EVAE_TOPK_TWO_LAUNCH"); two = (e && atoi(e) == 0) ? 0 : 1; }
  if (two && gemm_x6_enabled() && gemm_x6_ok(g)) {
    // TWO launches: the screening product on the split-bf16 kernel (two-term products: a filter needs a bound, not fp32
    // accuracy; the cache rows' norms accumulated while it stages them, their maxima per half tile; distances query-major
    // and without |q|^2), then one block per query for everything else.  The same kernel at any size: a 196-tile launch
    // over a 40-wide cache is one short round of blocks, and three launches less than the fp32 form.
    g.tile_max = (float*)(w + L.tile_max); g.ldq = L.ldq;
    int rc = launch_gemm_x6<EPI_DIST_TILEMIN, 0, 128, 2>(g, 1, stream, "topk_screen(tile minima, split-bf16, query-major)");
    if (rc) return rc;
    const float u = 5.9604645e-08f;
    const float gamma = 2.0f * ((4.0f * zdim + 3.0f) * u + 3.0517578e-05f);         // derived below
    topk_finish_kernel<<<B, FIN_NT, 0, stream>>>(q, cache, N, zdim, k, flags, index_base, tmin, L.ntiles, L.ldm, g.tile_max, dist, L.ldq,
                                              gamma, cand, val, (size_t)N, out_idx, out_val);
    return check_launch("topk_finish_kernel");
  }
  // query norms first: that launch also clears the candidate counters and the running maximum of the cache norms
  const int rpb = zdim >= 128 ? 16 : 256;       // rows one block covers per sweep
  sq_norms_kernel<<<std::min(cdiv(B, rpb), 2048), 256, 0, stream>>>(q, B, zdim, qn, nullptr, cnt, L.ldt, cnmax);
  g.e0 = cn; g.e1 = qn;
  // the screening product once, on the split-bf16 kernel when the launch fills the machine; its distances are kept
  const bool x6 = B > 64 && gemm_x6_use(g);
  int rc;
  if (x6) {
    // two-term products (a filter needs a bound, not fp32 accuracy), cache norms accumulated by the kernel while it stages
    // the rows: no pass of its own over the cache
    g.e0 = nullptr; g.aux_cnt = reinterpret_cast<int*>(cnmax);
    rc = launch_gemm_x6<EPI_DIST_TILEMIN, 0, 128, 2>(g, 1, stream, "topk_screen(tile minima, split-bf16)");
  } else {
    sq_norms_kernel<<<std::min(cdiv(N, rpb), 1024), 256, 0, stream>>>(cache, N, zdim, cn, cnmax, nullptr, 0, nullptr);
    rc = check_launch("sq_norms_kernel");
    if (rc) return rc;
    if (B <= 64) rc = launch_gemm_w<true, true, EPI_DIST_TILEMIN, true, 64, 8>(g, 1, stream, "topk_screen(tile minima)");
    else rc = launch_gemm_w<true, true, EPI_DIST_TILEMIN, true, 128, 8>(g, 1, stream, "topk_screen(tile minima)");
  }
  if (rc) return rc;
  // error bound of one approximate distance d = |q|^2 + |c|^2 - 2 q.c, in units of (|q|^2 + |c|^2), doubled for safety.
  // fp32 kernel: K roundings in the norms, K in the product, 3 in the combination, each <= u = 2^-24.  Two-term split-bf16
  // kernel: K in the norms, 3K accumulated partial products (each exact in fp32), 3 in the combination, plus what the split
  // drops: with round-to-nearest terms |a1| <= 2^-8 |a| and |a - a0 - a1| <= 2^-17 |a|, so per product
  // |a b - (a0 b0 + a0 b1 + a1 b0)| <= |a1 b1| + 2 * 2^-17 |a b| (1 + ...) <= 2^-15 |a b|; summed over k that is
  // <= 2^-15 sum |q_k c_k| <= 2^-16 (|q|^2 + |c|^2), twice that in d.
  const float u = 5.9604645e-08f;
  const float gamma = x6 ? 2.0f * ((4.0f * zdim + 3.0f) * u + 3.0517578e-05f) : 2.0f * (2.0f * zdim + 3.0f) * u;
  kth_threshold_kernel<<<B, 256, 0, stream>>>(tmin, L.ntiles, L.ldm, B, k, qn, cnmax, gamma, thr);
  rc = check_launch("kth_threshold_kernel");
  if (rc) return rc;
  dist_collect_kernel<<<(unsigned)std::min<size_t>(((size_t)N * (L.ldt / 4) + 255) / 256, 8192), 256, 0, stream>>>(
      dist, N, B, L.ldt, thr, cnt, cand, (size_t)N);
  rc = check_launch("dist_collect_kernel");
  if (rc) return rc;
  topk_exact_kernel<<<B, 256, 0, stream>>>(q, cache, zdim, k, flags, index_base, cnt, cand, val, (size_t)N, out_idx, out_val);
  return check_launch("topk_exact_kernel");
}

}  // namespace evae
