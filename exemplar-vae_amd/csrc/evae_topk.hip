// Fused all-pairs distance + top-K on gfx950 (models/BaseModel.py:263-264, utils/knn_on_latent.py:4-9).
//
// Bit-exactness: the reference takes top-k of fp32(fp64 distance).  Here the distance is the direct
// difference sum_k (q-c)^2 accumulated in fp64 (each fp32 operand converts exactly, each squared
// difference is exact in fp64) and rounded ONCE to fp32; candidates are ordered by
// (fp32 value ascending, index ascending).
//
// Shape: same 128-query x 64-exemplar LDS tile as the prior kernel (coalesced float4 streaming of the
// [N x z] cache, conflict-free ds_read_b128), thread tile 8 x 4 with fp64 accumulators.  The fp32
// distances of a tile go to LDS; each wave then owns 32 queries and keeps their running top-k lists in
// LDS (lane s holds the s-th best): one broadcast read of the current k-th value, one ballot, and only
// lanes that beat it are inserted by a wave-cooperative shift-insert.  A block scans a contiguous
// split of the cache; per-split lists are merged by the same insertion code in a second kernel, which
// is also the cross-shard (all-gathered) merge.
#include "evae_tile.h"
#include "evae_topk_screen.h"

namespace evae {

constexpr int DS = BE + 1;  // distance tile row stride

// Lists and candidates are (fp32 value, index) pairs ordered by (value, index); IdT is int inside the scan
// kernel (shard-local exemplar index, one shuffle) and int64_t in the merge kernel (global indices).
template <typename IdT> struct IdLimit;
template <> struct IdLimit<int> { static constexpr int max() { return INT_MAX; } };
template <> struct IdLimit<int64_t> { static constexpr int64_t max() { return INT64_MAX; } };

template <typename IdT>
__device__ __forceinline__ bool tk_less(float av, IdT ai, float bv, IdT bi) {
  return (av < bv) || (av == bv && ai < bi);
}

// insert (cv, cid), known to be < entry k-1, into the sorted list held by lanes 0..k-1
template <typename IdT>
__device__ __forceinline__ void tk_insert(float& lv, IdT& li, float cv, IdT cid, int k, int lane) {
  bool less = (lane < k) && tk_less(lv, li, cv, cid);
  int pos = __popcll(__ballot(less));
  float upv = __shfl_up(lv, 1, 64);
  IdT upi = __shfl_up(li, 1, 64);
  if (lane > pos && lane < k) { lv = upv; li = upi; }
  else if (lane == pos) { lv = cv; li = cid; }
}

// full bitonic sort of one (value, index) per lane, ascending by (value, index): 21 compare-exchange stages
template <typename IdT>
__device__ __forceinline__ void bitonic_sort64(float& v, IdT& id, int lane) {
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const float ov = __shfl_xor(v, stride, 64);
      const IdT oi = __shfl_xor(id, stride, 64);
      const bool want_min = (((lane & stride) == 0) == ((lane & size) == 0));
      const bool take = want_min ? tk_less(ov, oi, v, id) : tk_less(v, id, ov, oi);
      if (take) { v = ov; id = oi; }
    }
  }
}

// bitonic merge (6 stages) of a 64-lane bitonic sequence into ascending order
template <typename IdT>
__device__ __forceinline__ void bitonic_merge64(float& v, IdT& id, int lane) {
#pragma unroll
  for (int stride = 32; stride > 0; stride >>= 1) {
    const float ov = __shfl_xor(v, stride, 64);
    const IdT oi = __shfl_xor(id, stride, 64);
    const bool want_min = (lane & stride) == 0;
    const bool take = want_min ? tk_less(ov, oi, v, id) : tk_less(v, id, ov, oi);
    if (take) { v = ov; id = oi; }
  }
}

// Offer one candidate per lane to the sorted list (lanes 0..k-1).  Few qualifying lanes: wave-cooperative
// shift-inserts.  Many (a fresh list, an early tile): sort the 64 candidates, lay list (ascending, lanes
// 0..k-1), +inf filler and the k best candidates (descending, lanes 63..64-k) out as ONE bitonic sequence and
// merge it -- ~27 compare-exchange stages regardless of how many candidates enter.
template <typename IdT>
__device__ __forceinline__ void tk_offer(float& lv, IdT& li, float v, IdT id, bool valid, int k, int lane) {
  float tv = __shfl(lv, k - 1, 64);
  IdT ti = __shfl(li, k - 1, 64);
  unsigned long long mask = __ballot(valid && tk_less(v, id, tv, ti));
  if (mask == 0ull) return;
  if (__popcll(mask) > 4 && k <= 32) {
    float cv = valid ? v : INFINITY;
    IdT ci = valid ? id : IdLimit<IdT>::max();
    bitonic_sort64(cv, ci, lane);
    const float rv = __shfl(cv, 63 - lane, 64);       // candidates reversed: lane 63-j holds the j-th best
    const IdT ri = __shfl(ci, 63 - lane, 64);
    if (lane >= 64 - k) { lv = rv; li = ri; }
    else if (lane >= k) { lv = INFINITY; li = IdLimit<IdT>::max(); }
    bitonic_merge64(lv, li, lane);
    if (lane >= k) { lv = INFINITY; li = IdLimit<IdT>::max(); }
    return;
  }
  while (mask) {
    const int L = __ffsll((long long)mask) - 1;
    const float cv = __shfl(v, L, 64);
    const IdT cid = __shfl(id, L, 64);
    if (tk_less(cv, cid, tv, ti)) {
      tk_insert(lv, li, cv, cid, k, lane);
      tv = __shfl(lv, k - 1, 64);
      ti = __shfl(li, k - 1, 64);
    }
    mask &= mask - 1;
  }
}

template <int KC>
__device__ __forceinline__ void dist_chunk_f64(double (&acc)[TQ][TE], const float* __restrict__ Qs,
                                               const float* __restrict__ Es, int tq, int te) {
  constexpr int ks = Geom<KC>::ks;
#pragma unroll 2
  for (int k = 0; k < KC; k += 4) {
    double ed[TE][4];
#pragma unroll
    for (int j = 0; j < TE; ++j) {
      const float4 e4 = *reinterpret_cast<const float4*>(Es + (te + 16 * j) * ks + k);
      ed[j][0] = e4.x; ed[j][1] = e4.y; ed[j][2] = e4.z; ed[j][3] = e4.w;
    }
#pragma unroll
    for (int i = 0; i < TQ; ++i) {
      const float4 q4 = *reinterpret_cast<const float4*>(Qs + (tq + 16 * i) * ks + k);
      const double q0 = q4.x, q1 = q4.y, q2 = q4.z, q3 = q4.w;
#pragma unroll
      for (int j = 0; j < TE; ++j) {
        double d0 = q0 - ed[j][0], d1 = q1 - ed[j][1], d2 = q2 - ed[j][2], d3 = q3 - ed[j][3];
        double a = acc[i][j];
        a = fma(d0, d0, a);
        a = fma(d1, d1, a);
        a = fma(d2, d2, a);
        a = fma(d3, d3, a);
        acc[i][j] = a;
      }
    }
  }
}

static size_t topk_lds_bytes(const PriorGeom& g, int k) {
  size_t fl = (size_t)(BQ + BE) * geom_ks(g.kc) + (size_t)BQ * DS + (size_t)BQ * k;  // + list values
  size_t bytes = fl * sizeof(float);
  bytes = align_up(bytes, 16) + (size_t)BQ * k * sizeof(int);
  return bytes;
}

template <int KC>
__global__ __launch_bounds__(NT) void pairdist_topk_kernel(
    const float* __restrict__ q, int B, const float* __restrict__ cache, int N, int zdim, int k,
    unsigned flags, int tiles_per_split, PriorGeom g, float* __restrict__ cand_val,
    int64_t* __restrict__ cand_idx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;
  constexpr int ks = Geom<KC>::ks;
  float* Es = Qs + BQ * ks;
  float* D = Es + BE * ks;            // [BQ][DS]
  float* Lv = D + BQ * DS;            // [BQ][k]
  int* Li = reinterpret_cast<int*>(   // shard-local exemplar index; INT_MAX = empty slot
      reinterpret_cast<char*>(smem) + align_up(((size_t)(BQ + BE) * ks + (size_t)BQ * DS + (size_t)BQ * k) * 4, 16));

  const int split = blockIdx.x;
  const int q0 = blockIdx.y * BQ;
  const int te = threadIdx.x & 15;
  const int tq = threadIdx.x >> 4;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const bool vec_ok = (zdim & 3) == 0 && (((uintptr_t)q | (uintptr_t)cache) & 15) == 0;
  const bool do_sqrt = (flags & EVAE_TOPK_SQRT) != 0;

  for (int i = threadIdx.x; i < BQ * k; i += NT) { Lv[i] = INFINITY; Li[i] = INT_MAX; }
  if (g.nchunk == 1) stage_rows<KC, false>(Qs, q, q0, B, BQ, zdim, 0, nullptr, vec_ok);

  const int ntiles = (N + BE - 1) / BE;
  const int tile_begin = split * tiles_per_split;
  int tile_end = tile_begin + tiles_per_split;
  if (tile_end > ntiles) tile_end = ntiles;

  for (int t = tile_begin; t < tile_end; ++t) {
    const int e0 = t * BE;
    double acc[TQ][TE];
#pragma unroll
    for (int i = 0; i < TQ; ++i)
#pragma unroll
      for (int j = 0; j < TE; ++j) acc[i][j] = 0.0;
    for (int ch = 0; ch < g.nchunk; ++ch) {
      __syncthreads();
      if (g.nchunk > 1) stage_rows<KC, false>(Qs, q, q0, B, BQ, zdim, ch * KC, nullptr, vec_ok);
      stage_rows<KC, false>(Es, cache, e0, N, BE, zdim, ch * KC, nullptr, vec_ok);
      __syncthreads();
      dist_chunk_f64<KC>(acc, Qs, Es, tq, te);
    }
#pragma unroll
    for (int i = 0; i < TQ; ++i)
#pragma unroll
      for (int j = 0; j < TE; ++j) {
        float v = (float)acc[i][j];
        if (do_sqrt) v = sqrtf(v);
        D[(tq + 16 * i) * DS + te + 16 * j] = v;
      }
    __syncthreads();
    // wave w owns queries w*32 .. w*32+31
    const int e = e0 + lane;
    const bool valid = e < N;
    for (int qq = 0; qq < BQ / 4; ++qq) {
      const int ql = wave * (BQ / 4) + qq;
      if (q0 + ql >= B) break;
      const float v = D[ql * DS + lane];
      const float tv = Lv[ql * k + k - 1];
      const int ti = Li[ql * k + k - 1];
      if (__ballot(valid && tk_less(v, e, tv, ti)) == 0ull) continue;
      float lv = lane < k ? Lv[ql * k + lane] : INFINITY;
      int li = lane < k ? Li[ql * k + lane] : INT_MAX;
      tk_offer<int>(lv, li, v, e, valid, k, lane);
      if (lane < k) { Lv[ql * k + lane] = lv; Li[ql * k + lane] = li; }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BQ * k; i += NT) {
    int ql = i / k;
    if (q0 + ql < B) {
      size_t o = ((size_t)split * B + q0 + ql) * k + (i - ql * k);
      cand_val[o] = Lv[i];
      cand_idx[o] = Li[i] == INT_MAX ? INT64_MAX : (int64_t)Li[i];
    }
  }
}

// materialised [B x N] distance matrix (API completeness: utils/distributions.py:12-18); fp64 accumulate
template <int KC>
__global__ __launch_bounds__(NT) void pairdist_kernel(const float* __restrict__ q, int B,
                                                      const float* __restrict__ cache, int N, int zdim,
                                                      PriorGeom g, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;
  float* Es = Qs + BQ * Geom<KC>::ks;
  const int e0 = blockIdx.x * BE, q0 = blockIdx.y * BQ;
  const int te = threadIdx.x & 15, tq = threadIdx.x >> 4;
  const bool vec_ok = (zdim & 3) == 0 && (((uintptr_t)q | (uintptr_t)cache) & 15) == 0;
  double acc[TQ][TE];
#pragma unroll
  for (int i = 0; i < TQ; ++i)
#pragma unroll
    for (int j = 0; j < TE; ++j) acc[i][j] = 0.0;
  for (int ch = 0; ch < g.nchunk; ++ch) {
    __syncthreads();
    stage_rows<KC, false>(Qs, q, q0, B, BQ, zdim, ch * KC, nullptr, vec_ok);
    stage_rows<KC, false>(Es, cache, e0, N, BE, zdim, ch * KC, nullptr, vec_ok);
    __syncthreads();
    dist_chunk_f64<KC>(acc, Qs, Es, tq, te);
  }
#pragma unroll
  for (int i = 0; i < TQ; ++i)
#pragma unroll
    for (int j = 0; j < TE; ++j) {
      const int qi = q0 + tq + 16 * i, e = e0 + te + 16 * j;
      if (qi < B && e < N) out[(size_t)qi * N + e] = (float)acc[i][j];
    }
}

// one wave per query merges R lists of k (any order inside a list is fine)
__global__ __launch_bounds__(NT) void topk_merge_kernel(const float* __restrict__ val,
                                                        const int64_t* __restrict__ idx, int R, int B,
                                                        int k, int64_t index_base,
                                                        int64_t* __restrict__ out_idx,
                                                        float* __restrict__ out_val) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= B) return;
  float lv = INFINITY;
  int64_t li = INT64_MAX;
  const int total = R * k;
  for (int c0 = 0; c0 < total; c0 += 64) {
    int c = c0 + lane;
    bool valid = c < total;
    float v = INFINITY;
    int64_t id = INT64_MAX;
    if (valid) {
      int r = c / k, s = c - r * k;
      size_t o = ((size_t)r * B + row) * k + s;
      v = val[o];
      id = idx[o];
      valid = id >= 0 && id != INT64_MAX;
    }
    tk_offer<int64_t>(lv, li, v, id, valid, k, lane);
  }
  if (lane < k) {
    out_idx[(size_t)row * k + lane] = (li == INT64_MAX) ? (int64_t)-1 : li + index_base;
    if (out_val) out_val[(size_t)row * k + lane] = lv;
  }
}

// first occurrence of every position keeps its exemplar, repeats become masked slots (see evae_select_exemplars).
// A block decides 64 slots: it keeps the positions in front of its last slot in LDS (padded to fours with -1 - index, which
// never matches); the four waves share the scan of a slot's prefix (wave w takes the fours w, w + 4, ...: every lane of a wave
// reads the same address, a broadcast), their verdicts meet in LDS.  (r03: one thread per slot scanning its whole prefix took
// 30 us for the 1000 slots of c2a -- 250 dependent LDS round trips in the last thread -- on the step's critical path.)
__global__ __launch_bounds__(256) void select_exemplars_kernel(const int64_t* __restrict__ pos, int n,
                                                               const int64_t* __restrict__ cand_idx, int C,
                                                               int64_t* __restrict__ sel_rows, int64_t* __restrict__ c_idx,
                                                               int* __restrict__ n_unique) {
  extern __shared__ __attribute__((aligned(16))) int sp[];
  const int n4 = (n + 3) & ~3;
  int* const flag = sp + n4;
  const int hi = min(n4, (int)((blockIdx.x * 64 + 64 + 3) & ~3u));
  for (int i = threadIdx.x; i < hi; i += 256) sp[i] = i < n ? (int)pos[i] : -1 - i;
  __syncthreads();
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  bool dup = false;
  int p = 0;
  if (i < n) {
    p = sp[i];
    const int full = i & ~3, nf = full >> 2;
#pragma unroll 4
    for (int f = part; f < nf; f += 4) {
      const int4 q = *reinterpret_cast<const int4*>(sp + 4 * f);
      dup |= (q.x == p) | (q.y == p) | (q.z == p) | (q.w == p);
    }
    if (part == 0)
      for (int j = full; j < i; ++j) dup |= (sp[j] == p);
  }
  flag[threadIdx.x] = dup ? 1 : 0;
  __syncthreads();
  if (part != 0) return;
  bool first = false;
  if (i < n) {
    dup = (flag[lane] | flag[64 + lane] | flag[128 + lane] | flag[192 + lane]) != 0;
    const int64_t row = (p >= 0 && p < C) ? cand_idx[p] : (int64_t)0;
    sel_rows[i] = row;
    c_idx[i] = dup ? (int64_t)EVAE_PRIOR_MASK_ALL : row;
    first = !dup;
  }
  if (n_unique) {
    const unsigned long long m = __ballot(first);
    if (lane == 0 && m) atomicAdd(n_unique, __popcll(m));
  }
}

static void topk_splits(int B, int N, int* nsplit, int* tps, int* nq) {
  int ntiles = cdiv(N, BE);
  *nq = cdiv(B, BQ);
  int target = cdiv(768, *nq);
  int ns = ntiles < target ? ntiles : target;
  if (ns < 1) ns = 1;
  *tps = cdiv(ntiles, ns);
  if (*tps < 1) *tps = 1;
  *nsplit = cdiv(ntiles, *tps);
  if (*nsplit < 1) *nsplit = 1;
}


}  // namespace evae

using namespace evae;

extern "C" size_t evae_pairdist_topk_workspace_bytes(int B, int N, int zdim, int k) {
  (void)zdim;
  if (B <= 0 || N <= 0 || k <= 0) return 256;
  int ns, tps, nq;
  topk_splits(B, N, &ns, &tps, &nq);
  size_t n = (size_t)ns * B * k;
  const size_t scan = align_up(n * sizeof(float), 256) + align_up(n * sizeof(int64_t), 256) + 256;
  const size_t screen = topk_screen_workspace_bytes(B, N, zdim, k);      // 0 when the screening path does not apply
  return scan > screen ? scan : screen;
}

extern "C" int evae_pairdist_topk(const float* q, int B, const float* cache, int N, int zdim, int k,
                                  unsigned flags, int64_t index_base, int64_t* out_idx, float* out_val,
                                  void* ws, size_t ws_bytes, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(B >= 0 && N >= 0 && zdim > 0, "pairdist_topk: bad sizes B=%d N=%d zdim=%d", B, N, zdim);
  EVAE_REQUIRE(k >= 1 && k <= 64, "pairdist_topk: k=%d outside [1,64]", k);
  EVAE_REQUIRE(k <= N || N == 0, "pairdist_topk: k=%d > N=%d", k, N);
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(N > 0, "pairdist_topk: empty cache");
  EVAE_REQUIRE(q && cache && out_idx, "pairdist_topk: null pointer");
  EVAE_REQUIRE((size_t)N < (size_t)INT_MAX, "pairdist_topk: N too large for one shard");
  if (ws == nullptr || ws_bytes < evae_pairdist_topk_workspace_bytes(B, N, zdim, k)) {
    set_error("pairdist_topk: workspace too small (%zu)", ws_bytes);
    return EVAE_EWORKSPACE;
  }
  {   // large caches: fp32 matrix-core screening + exact re-ranking of the survivors (evae_topk_screen.hip)
    int handled = 0;
    int rc = topk_screen(q, B, cache, N, zdim, k, flags, index_base, out_idx, out_val, ws, ws_bytes, stream, &handled);
    if (rc || handled) return rc;
  }
  int ns, tps, nq;
  topk_splits(B, N, &ns, &tps, &nq);
  PriorGeom g = prior_geom(zdim);
  size_t n = (size_t)ns * B * k;
  float* cv = (float*)ws;
  int64_t* ci = (int64_t*)((char*)ws + align_up(n * sizeof(float), 256));
  size_t lds = topk_lds_bytes(g, k);
  EVAE_DISPATCH_KC(g.kc, {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)pairdist_topk_kernel<KC_>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
      attr = true;
    }
    pairdist_topk_kernel<KC_><<<dim3(ns, nq), NT, lds, stream>>>(q, B, cache, N, zdim, k, flags, tps, g, cv, ci);
  });
  int rc = check_launch("pairdist_topk_kernel");
  if (rc) return rc;
  topk_merge_kernel<<<cdiv(B, NT / 64), NT, 0, stream>>>(cv, ci, ns, B, k, index_base, out_idx, out_val);
  return check_launch("topk_merge_kernel");
}

extern "C" int evae_topk_merge(const float* val, const int64_t* idx, int R, int B, int k,
                               int64_t* out_idx, float* out_val, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(R >= 1 && B >= 0 && k >= 1 && k <= 64, "topk_merge: bad sizes R=%d B=%d k=%d", R, B, k);
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(val && idx && out_idx, "topk_merge: null pointer");
  topk_merge_kernel<<<cdiv(B, NT / 64), NT, 0, stream>>>(val, idx, R, B, k, 0, out_idx, out_val);
  return check_launch("topk_merge_kernel");
}

extern "C" int evae_select_exemplars(const int64_t* pos, int n, const int64_t* cand_idx, int C, int64_t* sel_rows,
                                     int64_t* c_idx, int* n_unique, evae_stream_t stream_) {
  EVAE_REQUIRE(n >= 0 && n <= 16384 && C >= 0, "select_exemplars: bad sizes n=%d (<= 16384) C=%d", n, C);
  if (n == 0) return EVAE_OK;
  EVAE_REQUIRE(pos && cand_idx && sel_rows && c_idx, "select_exemplars: null pointer");
  if (n_unique) {
    hipError_t e = hipMemsetAsync(n_unique, 0, sizeof(int), (hipStream_t)stream_);
    EVAE_REQUIRE(e == hipSuccess, "select_exemplars: memset failed");
  }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)select_exemplars_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr = true;
  }
  select_exemplars_kernel<<<cdiv(n, 64), 256, (size_t)(((n + 3) & ~3) + 256) * sizeof(int), (hipStream_t)stream_>>>(pos, n, cand_idx, C,
                                                                                                              sel_rows, c_idx, n_unique);
  return check_launch("select_exemplars_kernel");
}

extern "C" int evae_pairwise_distance(const float* q, int B, const float* cache, int N, int zdim,
                                      float* out, evae_stream_t stream_) {
  EVAE_REQUIRE(B >= 0 && N >= 0 && zdim > 0, "pairwise_distance: bad sizes");
  if (B == 0 || N == 0) return EVAE_OK;
  EVAE_REQUIRE(q && cache && out, "pairwise_distance: null pointer");
  PriorGeom g = prior_geom(zdim);
  size_t lds = (size_t)(BQ + BE) * geom_ks(g.kc) * sizeof(float);
  EVAE_DISPATCH_KC(g.kc, (pairdist_kernel<KC_><<<dim3(cdiv(N, BE), cdiv(B, BQ)), NT, lds, (hipStream_t)stream_>>>(
                             q, B, cache, N, zdim, g, out)));
  return check_launch("pairdist_kernel");
}
