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

// ======================================================================================================================
// r06: the cache scan + top-K as ONE launch.  The two-launch form above writes the [N x B] approximate
// distances (40 MB at c5) and reads them back: 182 MB of traffic for 102 MB of cache.  Here nothing but candidates leaves a block:
//   * a block streams ITS contiguous range of cache rows once (fp32 rows -> two bf16 terms while staging, their squared norms
//     accumulated on the way), 128-row tiles against all <= 128 queries; the queries' fragments (two terms) stay in registers,
//     wave w owning queries 32 w .. 32 w + 31, so that a lane pair (l, l ^ 32) holds all 128 distances of ONE query;
//   * a valid upper bound T' of the k-th smallest approximate distance comes for free: every lane keeps MG = ceil(k / 2) running
//     GROUP minima over the rows it has seen (register j of a tile belongs to group j mod MG); the 2 MG minima of a lane pair are 2 MG
//     >= k distinct rows, so max of them >= the k-th smallest distance of the whole cache -- the tile-minima argument of the
//     two-launch form, inside a lane;
//   * a row is a candidate when d <= T' + 2 gamma (|q|^2 + largest row norm seen): (value, row) goes to the query's list in LDS,
//     flushed once per block (one atomic per query to reserve the global range; a full LDS list spills straight to the global one);
//   * blocks 0 .. B-1 then wait for every block to have flushed (an arrival counter; the others exit) and finish ONE query each:
//     exact k-th smallest approximate distance among its candidates (every row with one of the k smallest approximate distances
//     is a candidate: it is <= any T'), the final candidates d <= T + 2 gamma (|q|^2 + max |c|^2), their exact fp64 distances and
//     ranks (topk_exact_body: bit-identical indices and values to the scan kernel by the argument at the head of this file).
// HBM traffic = the cache + the queries + a few MB of candidates.
constexpr int TS_NT = 256, TS_SLOTS = 48, TS_NQ = 128, TS_MAX_BLOCKS = 1024;
struct TopkStreamArgs {
  const float* q; const float* cache;
  int B, N, zdim, k; unsigned flags; int64_t index_base;
  int rows_per_block; float gamma;
  int abl;                            // tools only (EVAE_TS_ABL): 1 = leave before the finish stage, 2 = also emit nothing, 4 = also no MFMAs
  unsigned* counters;                 // [0, 128): entries of a query's OVERFLOW list; [256, 256 + blocks): a block's flag (its lists are flushed)
  float* cmaxb;                       // [blocks] the largest |c|^2 of a block's rows
  int* gcnt;                          // [B][blocks] candidates of (query, block)
  float* gval; int* gidx;             // [B][blocks][TS_SLOTS] (approximate distance without |q|^2, row): fixed regions, no atomics
  float* oval; int* oidx; size_t cap; // [B][cap] candidates beyond a block's TS_SLOTS (rare; appended with an atomic)
  int* cand; float* val; size_t ldc;  // the exact stage's lists
  int64_t* out_idx; float* out_val;
};
template <int KS> constexpr int ts_row_bytes() { return KS * 32 + 16; }                  // a staged row of one term: KS x 16 bf16 + pad
template <int KS> constexpr int ts_stage_bytes() { return 2 * 32 * ts_row_bytes<KS>(); }    // a chunk: 32 rows, both terms
template <int KS> constexpr size_t ts_lds_bytes() {
  return 2 * (size_t)ts_stage_bytes<KS>() + 2 * 128 * 4 + 16 + 2 * TS_NQ * 4 + 2 * ((size_t)TS_NQ * TS_SLOTS + TS_NT) * 4;
}
// (x, y) -> the two bf16 terms of each, term t of x in the low half of p[t]
__device__ __forceinline__ void ts_split2(float x, float y, unsigned& p0, unsigned& p1) {
  x6_f32x2 r = {x, y};
  p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, x6_bf16x2));
  x6_f32x2 h = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xFFFF0000u)};
  r = r - h;
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, x6_bf16x2));
}

// a candidate (approximate distance, row) of query nq: the block's LDS list, or the query's global overflow list when that is full.
// Out of line: the 64 call sites of a tile's epilogue are a compare and a branch each (the unrolled form was 1000 instructions of
// a kernel whose straight-line code then no longer fitted the instruction cache: 27 us of fetch stalls per block at c2)
__device__ __attribute__((noinline)) void ts_emit(float d, int row, int nq, int* lcnt, float* lv, int* li, unsigned* counters,
                                                  float* oval, int* oidx, size_t cap) {
  const int slot = atomicAdd(&lcnt[nq], 1);
  if (slot < TS_SLOTS) { lv[nq * TS_SLOTS + slot] = d; li[nq * TS_SLOTS + slot] = row; }
  else {
    const unsigned gs = atomicAdd(&counters[nq], 1u);
    oval[(size_t)nq * cap + gs] = d; oidx[(size_t)nq * cap + gs] = row;
  }
}

// v of the lane DPP control CTRL names (0x120 + n: rotate right by n inside the lane's row of 16)
template <int CTRL>
__device__ __forceinline__ float ts_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

template <int I, int N, typename F>
__device__ __forceinline__ void ts_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); ts_static_for<I + 1, N>(f); }
}

template <int KS, int MG>
__global__ __launch_bounds__(TS_NT) void topk_stream_kernel(const TopkStreamArgs a) {
  constexpr int NCH = 4;                           // chunks of a tile: 32 rows x the whole (padded) row each -> one MFMA row tile
  constexpr int RS = ts_row_bytes<KS>(), TERM = 32 * RS, STAGE = 2 * TERM;
  constexpr int F4R = 4 * KS;                      // float4s of a padded row
  constexpr int NJ = 32 * F4R / TS_NT;             // float4s a thread stages per chunk: f = 256 j + tid -> row f / F4R, float4 f % F4R
  static_assert(NJ >= 1 && 64 % F4R == 0, "a row's float4s sit in one wave's load");
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const stA = lds;
  float* const cn_l = reinterpret_cast<float*>(lds + 2 * STAGE);         // [2][128]
  unsigned* const cmax_l = reinterpret_cast<unsigned*>(cn_l + 256);
  int* const lcnt = reinterpret_cast<int*>(cmax_l + 4);                   // [128]
  int* const lbase = lcnt + TS_NQ;
  float* const lv = reinterpret_cast<float*>(lbase + TS_NQ);              // [128][SLOTS]
  int* const li = reinterpret_cast<int*>(lv + TS_NQ * TS_SLOTS + TS_NT);   // (+ a dummy slot per thread behind either list)
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, half = l >> 5;
  const int nq = 32 * w + (l & 31);
  const bool q_live = nq < a.B;
  const int zdim = a.zdim;
  if (tid < TS_NQ) lcnt[tid] = 0;
  if (tid == 0) *cmax_l = 0u;
  unsigned long long* const stamp = reinterpret_cast<unsigned long long*>(a.counters + 256 + TS_MAX_BLOCKS) + 32 * (blockIdx.x == 0 ? 0 : 1);
  const bool stamping = (a.abl & 64) && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100);
  int nstamp = 0;
  auto mark = [&]() { if (stamping && nstamp < 32) stamp[nstamp++] = wall_clock64(); };
  mark();
  const int rb = blockIdx.x * a.rows_per_block;
  const int row_end = min(a.N, rb + a.rows_per_block);
  const int ntile = rb < row_end ? (row_end - rb + 127) / 128 : 0;
  // ---- the queries' fragments: B operand of v_mfma_f32_32x32x16_bf16, lane l: query l & 31 of the wave, k = 16 s + 8 (l >> 5) .. + 7
  x6_bf16x8 bq[KS][2];
  float qn = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k0 = 16 * s + 8 * half;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (q_live && k0 < zdim) v0 = *reinterpret_cast<const float4*>(a.q + (size_t)nq * zdim + k0);
    if (q_live && k0 + 4 < zdim) v1 = *reinterpret_cast<const float4*>(a.q + (size_t)nq * zdim + k0 + 4);
    qn += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w + v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
    unsigned t0[4], t1[4];
    ts_split2(v0.x, v0.y, t0[0], t1[0]); ts_split2(v0.z, v0.w, t0[1], t1[1]);
    ts_split2(v1.x, v1.y, t0[2], t1[2]); ts_split2(v1.z, v1.w, t0[3], t1[3]);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    bq[s][0] = __builtin_bit_cast(x6_bf16x8, (u32x4){t0[0], t0[1], t0[2], t0[3]});
    bq[s][1] = __builtin_bit_cast(x6_bf16x8, (u32x4){t1[0], t1[1], t1[2], t1[3]});
  }
  qn += __shfl_xor(qn, 32, 64);
  mark();
  // ---- staging.  A chunk = 32 consecutive cache rows in full: 32 KB of contiguous memory at z = 256, read by 64-lane loads of
  // one whole row each (the k-chunked first form of this kernel visited every DRAM page four times, 256 bytes a visit: 2.3 us per
  // 32 KB chunk).  Register set S holds a chunk on its way to LDS stage S.  Pipeline (one barrier per chunk): while the MFMAs of chunk
  // c run from stage c & 1, the same wave splits chunk c + 1 (loaded two iterations ago) into the other stage -- a float4 every
  // other k-step, behind that k-step's MFMAs, so that vector and matrix pipes overlap within ONE wave per SIMD -- and refills its
  // registers with chunk c + 3.  rows_per_block is a multiple of 128: every tile of a block is whole, except the cache's very last
  // one -- its rows beyond N are read from row N - 1 and masked in the epilogue; columns beyond zdim are zeroed by a select.
  float4 R[2][NJ];
  auto load1 = [&](auto SET, auto J, int tile, int ck) {
    constexpr int S = decltype(SET)::value, j = decltype(J)::value;
    const int f = TS_NT * j + tid, kk = 4 * (f % F4R);
    const int row = min(rb + tile * 128 + 32 * ck + f / F4R, a.N - 1);
    float4 v = *reinterpret_cast<const float4*>(a.cache + (size_t)row * zdim + min(kk, zdim - 4));
    if (kk >= zdim) v = make_float4(0.f, 0.f, 0.f, 0.f);
    R[S][j] = v;
  };
  // ... its two bf16 terms into the stage, its row's squared norm (the F4R lanes of a row reduce among themselves) into the
  // tile's norm table and the running maximum
  auto stage1 = [&](auto SET, auto J, char* st, int tile, int ck) {
    constexpr int S = decltype(SET)::value, j = decltype(J)::value;
    const int f = TS_NT * j + tid, rr = f / F4R, kq = f % F4R;
    const float4 v = R[S][j];
    float t = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    unsigned p0a, p1a, p0b, p1b;
    ts_split2(v.x, v.y, p0a, p1a); ts_split2(v.z, v.w, p0b, p1b);
    char* dst = st + rr * RS + kq * 8;
    *reinterpret_cast<uint2*>(dst) = make_uint2(p0a, p0b);
    *reinterpret_cast<uint2*>(dst + TERM) = make_uint2(p1a, p1b);
    // (rotations inside the 16-lane DPP rows: vector instructions, no trip through the LDS crossbar; then across rows)
    t += ts_dpp<0x128>(t); t += ts_dpp<0x124>(t); t += ts_dpp<0x122>(t); t += ts_dpp<0x121>(t);
    if constexpr (F4R >= 32) t += __shfl_xor(t, 16, 64);
    if constexpr (F4R >= 64) t += __shfl_xor(t, 32, 64);
    if (kq == 0) { cn_l[(tile & 1) * 128 + 32 * ck + rr] = t; atomicMax(cmax_l, __float_as_uint(t)); }
  };
  f32x16 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
  float gmin[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) gmin[g] = INFINITY;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if (ntile > 0) {
    ts_static_for<0, NJ>([&](auto J) { load1(I0{}, J, 0, 0); });
    ts_static_for<0, NJ>([&](auto J) { load1(I1{}, J, 0, 1); });
    ts_static_for<0, NJ>([&](auto J) { stage1(I0{}, J, stA, 0, 0); load1(I0{}, J, 0, 2); });
  }
  __syncthreads();
  for (int tile = 0; tile < ntile; ++tile) {
    ts_static_for<0, NCH>([&](auto CK_) {
      constexpr int ck = decltype(CK_)::value;
      constexpr int SN = (ck + 1) & 1;                             // register set / stage of the NEXT chunk
      const char* const st = stA + ((ck & 1) ? STAGE : 0);
      char* const stn = stA + (SN ? STAGE : 0);
      // next chunk = (tile, ck + 1) or (tile + 1, 0); the chunk that refills its registers is three ahead of this one
      const int n_tile = (ck + 1 < NCH) ? tile : tile + 1, n_ck = (ck + 1) % NCH;
      const int f_tile = (ck + 3 < NCH) ? tile : tile + 1, f_ck = (ck + 3) % NCH;
      const bool n_live = n_tile < ntile, f_live = f_tile < ntile;
      x6_bf16x8 a0[2], a1[2];
      auto frags = [&](int buf, int s) {
        const char* src = st + (l & 31) * RS + s * 32 + 16 * half;
        a0[buf] = *reinterpret_cast<const x6_bf16x8*>(src);
        a1[buf] = *reinterpret_cast<const x6_bf16x8*>(src + TERM);
      };
      frags(0, 0);
      ts_static_for<0, KS>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        if constexpr (s + 1 < KS) frags((s + 1) & 1, s + 1);
        if (!(a.abl & 4)) {
          acc[ck] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s & 1], bq[s][0], acc[ck], 0, 0, 0);
          acc[ck] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[s & 1], bq[s][1], acc[ck], 0, 0, 0);
          acc[ck] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[s & 1], bq[s][0], acc[ck], 0, 0, 0);
        }
        // a float4 of the next chunk into the other stage every KS / NJ k-steps, behind this k-step's MFMAs; its register refilled
        if constexpr (s % (KS / NJ) == 0) {
          if (n_live) {
            constexpr int j = s / (KS / NJ);
            stage1(std::integral_constant<int, SN>{}, std::integral_constant<int, j>{}, stn, n_tile, n_ck);
            if (f_live) load1(std::integral_constant<int, SN>{}, std::integral_constant<int, j>{}, f_tile, f_ck);
          }
        }
      });
      if constexpr (ck == NCH - 1) {
        // ---- the tile's epilogue: d = |c|^2 - 2 q.c (no |q|^2: a constant per query), group minima, threshold, candidates.
        // (its norms went to LDS one barrier ago)
        mark();
        const float* cnp = cn_l + (tile & 1) * 128;
        const float cmax_seen = __uint_as_float(*cmax_l);
        const int rows_left = row_end - rb - tile * 128;            // rows of this tile inside the cache (>= 128 except at its end)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int rl = 32 * mt + 8 * g4 + 4 * half;
            const float4 c4 = *reinterpret_cast<const float4*>(cnp + rl);
            const float cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float d = fmaf(-2.0f, acc[mt][4 * g4 + i], cc[i]);
              if (rl + i >= rows_left) d = __builtin_nanf("");     // a row beyond the cache: below no threshold, ignored by fminf
              acc[mt][4 * g4 + i] = d;
              gmin[4 * g4 + i] = fminf(gmin[4 * g4 + i], d);
            }
          }
        }
        // the MG-th smallest of the lane's 16 group minima (MG distinct rows at or below it): a min / max insertion chain
        float sm[MG];
#pragma unroll
        for (int j = 0; j < MG; ++j) sm[j] = INFINITY;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          float v = gmin[g];
#pragma unroll
          for (int j = 0; j < MG; ++j) { const float lo = fminf(sm[j], v); v = fmaxf(sm[j], v); sm[j] = lo; }
        }
        float tb = sm[MG - 1];
        tb = fmaxf(tb, __shfl_xor(tb, 32, 64));
        const float thr = (q_live && !(a.abl & 2)) ? tb + 2.0f * a.gamma * (qn + cmax_seen) : -INFINITY;
        // candidates: count them, reserve the lane's slots with ONE LDS atomic, then write every value -- those that do not pass
        // (or do not fit) to a dummy slot: no branch and no atomic round trip per register
        int cpass = 0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int i = 0; i < 16; ++i) cpass += (acc[mt][i] <= thr) ? 1 : 0;
        int base = 0;
        if (cpass > 0) base = atomicAdd(&lcnt[nq], cpass);
        const bool fits = base + cpass <= TS_SLOTS;                 // (almost always: the block's list of this query has room)
        int off = base;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float d = acc[mt][i];
            const bool pass = d <= thr;
            const int row = rb + tile * 128 + 32 * mt + 8 * (i >> 2) + 4 * half + (i & 3);
            const int at = (pass && off < TS_SLOTS) ? nq * TS_SLOTS + off : TS_NQ * TS_SLOTS + tid;
            lv[at] = d; li[at] = row;
            if (!fits && pass && off >= TS_SLOTS) {                 // rare: the query's global overflow list
              const unsigned gs = atomicAdd(&a.counters[nq], 1u);
              a.oval[(size_t)nq * a.cap + gs] = d; a.oidx[(size_t)nq * a.cap + gs] = row;
            }
            off += pass ? 1 : 0;
            acc[mt][i] = 0.f;
          }
        }
      }
      __syncthreads();
    });
  }
  mark();
  // ---- flush this block's lists into ITS regions of the queries' candidate stores (no atomics), then raise its flag: a wave
  // takes every fourth query, its lanes the slots
  const int nblk = gridDim.x;
  for (int qq = w; qq < a.B; qq += TS_NT / 64) {
    const int c = min(lcnt[qq], TS_SLOTS);
    const size_t o = ((size_t)qq * nblk + blockIdx.x) * TS_SLOTS;
    if (l < c) { a.gval[o + l] = lv[qq * TS_SLOTS + l]; a.gidx[o + l] = li[qq * TS_SLOTS + l]; }
    if (l == 0) a.gcnt[(size_t)qq * nblk + blockIdx.x] = c;
  }
  if (tid == 0) a.cmaxb[blockIdx.x] = __uint_as_float(*cmax_l);
  mark();
  __threadfence();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(&a.counters[256 + blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  mark();
  if ((int)blockIdx.x >= a.B || (a.abl & 1)) return;
  // ---- one query per block from here (blocks beyond B have left): wait until every block has raised its flag
  for (;;) {
    int ok = 1;
    for (int b = tid; b < nblk; b += TS_NT)
      ok &= (__hip_atomic_load(&a.counters[256 + b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1 : 0;
    if (__syncthreads_and(ok)) break;
    __builtin_amdgcn_s_sleep(4);
  }
  __threadfence();
  mark();
  float* const qs = reinterpret_cast<float*>(lds);                 // [<= 256] the query row (the stream's LDS is free now)
  float* const tmn = qs + 512;                                     // [256] thread minima
  float* const sv = tmn + TS_NT;                                   // [1024] candidates' values below the coarse bound
  int* const scnt = reinterpret_cast<int*>(sv + 1024);             // [2]
  float* const red = reinterpret_cast<float*>(scnt + 4);           // [8]
  float* const fv = red + 8;                                       // [TS_SLOTS][256] a thread's block list
  for (int n = blockIdx.x; n < a.B; n += nblk) {
    // a thread owns the lists of blocks tid, tid + 256, ... and a stride of the overflow list
    const int Mo = (int)__hip_atomic_load(&a.counters[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float* ov = a.oval + (size_t)n * a.cap;
    const int* oi = a.oidx + (size_t)n * a.cap;
    float qq2 = 0.f, cm = 0.f;
    for (int d = tid; d < zdim; d += TS_NT) { const float v = a.q[(size_t)n * zdim + d]; qs[d] = v; qq2 = fmaf(v, v, qq2); }
    for (int b = tid; b < nblk; b += TS_NT) cm = fmaxf(cm, a.cmaxb[b]);
    qq2 = wave_sum(qq2); cm = wave_max(cm);
    if (l == 0) { red[w] = qq2; red[4 + w] = cm; }
    if (tid == 0) { scnt[0] = 0; scnt[1] = 0; }
    // this thread's block list (one per thread: <= 256 blocks; more are walked from memory): its TS_SLOTS values into LDS by
    // independent 16-byte loads (column tid of fv[slot][256]: conflict-free), then short loops over the count
    int cnt0 = 0;
    {
      const bool has = tid < nblk;
      cnt0 = has ? a.gcnt[(size_t)n * nblk + tid] : 0;
      const float4* gv4 = reinterpret_cast<const float4*>(a.gval + ((size_t)n * nblk + (has ? tid : 0)) * TS_SLOTS);
      float4 t4[TS_SLOTS / 4];
#pragma unroll
      for (int i = 0; i < TS_SLOTS / 4; ++i) t4[i] = gv4[i];
#pragma unroll
      for (int i = 0; i < TS_SLOTS / 4; ++i) {
        fv[(4 * i) * TS_NT + tid] = t4[i].x; fv[(4 * i + 1) * TS_NT + tid] = t4[i].y;
        fv[(4 * i + 2) * TS_NT + tid] = t4[i].z; fv[(4 * i + 3) * TS_NT + tid] = t4[i].w;
      }
    }
    auto sweep = [&](auto&& f) {                  // f(value, pointer to its row index)
      const int* gi0 = a.gidx + ((size_t)n * nblk + tid) * TS_SLOTS;
      for (int i = 0; i < cnt0; ++i) f(fv[i * TS_NT + tid], gi0 + i);
      for (int b = tid + TS_NT; b < nblk; b += TS_NT) {
        const int c = a.gcnt[(size_t)n * nblk + b];
        const float* gv = a.gval + ((size_t)n * nblk + b) * TS_SLOTS;
        for (int i = 0; i < c; ++i) f(gv[i], a.gidx + ((size_t)n * nblk + b) * TS_SLOTS + i);
      }
      for (int i = tid; i < Mo; i += TS_NT) f(ov[i], oi + i);
    };
    mark();
    float mn = INFINITY;
    sweep([&](float v, const int*) { mn = fminf(mn, v); });
    tmn[tid] = mn;
    __syncthreads();
    const float qnn = red[0] + red[1] + red[2] + red[3];
    const float cmax = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    // coarse bound: the k-th smallest of the 256 thread minima (distinct candidates, k <= 32 of them at or below it)
    const float t2 = key_to_float(kth_smallest_key<TS_NT>(tmn, TS_NT, a.k));
    sweep([&](float v, const int*) { if (v <= t2 && v < INFINITY) { const int s_ = atomicAdd(&scnt[0], 1); if (s_ < 1024) sv[s_] = v; } });
    __syncthreads();
    const int ns_ = scnt[0];
    // the k-th smallest approximate distance of the whole cache is among those at or below the coarse bound.  More than 1024 of
    // them (masses of equal distances): the bound itself serves -- looser, still an upper bound of it
    const float tk = ns_ <= 1024 ? key_to_float(kth_smallest_key<TS_NT>(sv, ns_, a.k)) : t2;
    const float thr = tk + 2.0f * a.gamma * (qnn + cmax);
    int* cl = a.cand + (size_t)n * a.ldc;
    mark();
    sweep([&](float v, const int* rowp) { if (v <= thr && v < INFINITY) cl[atomicAdd(&scnt[1], 1)] = *rowp; });
    __syncthreads();
    mark();
    topk_exact_body<TS_NT>(qs, a.cache, zdim, a.k, a.flags, a.index_base, n, scnt[1], cl, a.val + (size_t)n * a.ldc, a.out_idx, a.out_val);
    __syncthreads();
    mark();
  }
}

// OPT-IN (EVAE_TOPK_STREAM=1, read per call so that a test can switch it).  Measured r06 (rocprofv3, tools/ts_prof.sh; phase stamps
// tools/ts_stamps.py): c5 size 146 us against the two-launch form's 67, c2 size 62 against 30 -- correct (bit-exact indices and
// values, tests/test_gpu_kernels.py::test_topk_stream_kernel_equals_the_two_launch_form), not yet fast.  Where the time goes at c5
// (block 0, us): queries 4, four tiles 19 each (15 without any MFMA: the staging of a 32 KB chunk -- loads, two-term split, LDS
// writes, row norms -- takes 4 us of ONE wave per SIMD, the compiler's code for it is a chain of waits and uniform branches), the
// last epilogue 5, flush 4, fence + flag 11, waiting for the slowest block 14, lists to LDS 6, bounds 15, final sweep 4, exact
// ranks 10.  Steps tried on the way: global atomics per (block, query) reservation 435 us -> fixed regions 159; candidates by
// an LDS atomic each 14 us per tile -> counted + one atomic per lane 5; k-chunked staging (four visits per DRAM page) vs whole rows:
// equal.  What it needs next: the staging of chunk c + 1 issued as straight-line code between the MFMAs of chunk c (hand-scheduled:
// sched_group_barrier), two waves per SIMD (queries' fragments in LDS instead of 128 registers), and a finish that starts per query
// as soon as ITS candidates are in -- DESIGN section 7.
static bool stream_applies(int B, int N, int zdim, int k) {
  const char* e = getenv("EVAE_TOPK_STREAM");
  const bool on = e && atoi(e) != 0;
  return on && B <= TS_NQ && zdim <= 256 && k <= 32 && N >= 2048;
}
static int stream_blocks(int N, int* rows_per_block) {
  static int cus = 0;
  if (!cus) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256; }
  const int rpb = std::max(128, (int)align_up((size_t)cdiv(N, cus), 128));      // whole 128-row tiles
  *rows_per_block = rpb;
  return cdiv(N, rpb);
}
struct StreamLayout { size_t counters, cmaxb, gcnt, gval, gidx, oval, oidx, cand, val, total; };
static StreamLayout stream_layout(int B, int N) {
  StreamLayout L; size_t o = 0;
  int rpb; const int nb = stream_blocks(N, &rpb);
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes, 256); return at; };
  L.counters = take((256 + TS_MAX_BLOCKS) * 4 + 8192);            // (+ tools-only time stamps, EVAE_TS_ABL & 64)
  L.cmaxb = take((size_t)TS_MAX_BLOCKS * 4); L.gcnt = take((size_t)B * nb * 4);
  L.gval = take((size_t)B * nb * TS_SLOTS * 4); L.gidx = take((size_t)B * nb * TS_SLOTS * 4);
  L.oval = take((size_t)B * N * 4); L.oidx = take((size_t)B * N * 4);
  L.cand = take((size_t)B * N * 4); L.val = L.oval;              // the exact stage's values re-use the drained overflow values
  L.total = o + 256;
  return L;
}
template <int KS>
static int launch_stream(const TopkStreamArgs& a, int nblocks, hipStream_t stream) {
  const size_t lds = ts_lds_bytes<KS>();
  auto go = [&](auto kern) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    kern<<<nblocks, TS_NT, lds, stream>>>(a);
  };
  if (a.k <= 10) go(topk_stream_kernel<KS, 5>);
  else if (a.k <= 20) go(topk_stream_kernel<KS, 10>);
  else go(topk_stream_kernel<KS, 16>);
  return check_launch("topk_stream_kernel");
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
  const size_t two = screen_layout(B, N).total;
  const bool shape = B <= TS_NQ && zdim <= 256 && k <= 32 && N >= 2048;      // (whatever EVAE_TOPK_STREAM says now: a later call may differ)
  return shape ? std::max(two, stream_layout(B, N).total) : two;
}

int topk_screen(const float* q, int B, const float* cache, int N, int zdim, int k, unsigned flags, int64_t index_base,
                int64_t* out_idx, float* out_val, void* ws, size_t ws_bytes, hipStream_t stream, int* handled) {
  *handled = 0;
  if (!screen_applies(B, N, zdim, k) || (((uintptr_t)q | (uintptr_t)cache) & 15) != 0) return EVAE_OK;
  if (stream_applies(B, N, zdim, k) && ws != nullptr && ws_bytes >= stream_layout(B, N).total) {
    // ONE launch (+ a 1 KB memset node for its counters)
    const StreamLayout S = stream_layout(B, N);
    char* w_ = (char*)ws;
    TopkStreamArgs a = {};
    a.q = q; a.cache = cache; a.B = B; a.N = N; a.zdim = zdim; a.k = k; a.flags = flags; a.index_base = index_base;
    const int nb = stream_blocks(N, &a.rows_per_block);
    { static int abl = -1; if (abl < 0) { const char* e = getenv("EVAE_TS_ABL"); abl = e ? atoi(e) : 0; } a.abl = abl; }
    const float u_ = 5.9604645e-08f;
    a.gamma = 2.0f * ((4.0f * zdim + 3.0f) * u_ + 3.0517578e-05f);      // the two-term bound derived below
    a.counters = (unsigned*)(w_ + S.counters); a.cmaxb = (float*)(w_ + S.cmaxb); a.gcnt = (int*)(w_ + S.gcnt);
    a.gval = (float*)(w_ + S.gval); a.gidx = (int*)(w_ + S.gidx);
    a.oval = (float*)(w_ + S.oval); a.oidx = (int*)(w_ + S.oidx); a.cap = (size_t)N;
    a.cand = (int*)(w_ + S.cand); a.val = (float*)(w_ + S.val); a.ldc = (size_t)N;
    a.out_idx = out_idx; a.out_val = out_val;
    if (nb > TS_MAX_BLOCKS) { set_error("topk_stream: %d blocks", nb); return EVAE_EINVAL; }
    if (hipMemsetAsync(a.counters, 0, (256 + TS_MAX_BLOCKS) * 4, stream) != hipSuccess) return check_launch("topk_stream(counters)");
    *handled = 1;
    if (zdim <= 64) return launch_stream<4>(a, nb, stream);
    if (zdim <= 128) return launch_stream<8>(a, nb, stream);
    return launch_stream<16>(a, nb, stream);
  }
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
