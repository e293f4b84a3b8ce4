// Shared core of the fp32-MFMA GEMM family (evae_gemm_kernel.h: dense layers and channels-last convolutions; evae_conv.hip: NCHW convolutions):
// tile constants, the per-slab MFMA loop over LDS tiles, the split-K finish kernel and the host planner.
#pragma once
#include <stdlib.h>

#include "evae_common.h"
#include "evae_p6_image.h"

namespace evae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BK = 32;
constexpr int KS = BK + 4;  // KC tile row stride (floats); 36/4 = 9 odd


enum { EPI_LINEAR = 0, EPI_GATED = 1, EPI_GATE_BWD = 2, EPI_RAW = 3, EPI_RAW_GATED = 4,
       // squared distances |a_m|^2 + |b_n|^2 - 2 a_m.b_n from the dot products (top-K screening, evae_topk_screen.hip):
       EPI_DIST_TILEMIN = 5,   // out0[tile_m][n] = min over the tile's rows
       EPI_DIST_COLLECT = 6,   // rows with distance <= bias0[n] are appended to the candidate list of column n
       // exemplar prior at large latent sizes (evae_prior_gemm.hip): rows = exemplars, columns = queries, acc = c'.z'
       EPI_PRIOR_LSE = 7,      // per (row tile, column): max / sum exp / #masked of log N(z_n | c_m) over the tile's rows
       EPI_PRIOR_P = 8,        // out0[m][n] = bias1[n] exp(log N(z_n | c_m) - bias0[n])   (0 where masked)
       // EPI_GATE_BWD whose (dh, dg) leave as the three bf16 terms of [dh | dg]^T in the B-tile order of u8_gemm_kernel
       // (csrc/evae_dense_u8.hip): the weight gradient of the byte-store layer reads them without a transposing pre-pass
       EPI_GATE_BWD_IMG = 9 };


// one K-slab of MFMAs: wave tile 64 x (32 NT) at rows wr*64.., cols wc*32*NT..
// (k-groups [KG0, KG1) of 8 within the slab, so that LDS stores / global loads can be placed between them)
// Register fragments of one k-group (8 k values) of a wave tile, so the LDS reads of k-group g+1 can be
// issued before the MFMAs of k-group g (the compiler otherwise reuses one register set and exposes the
// LDS latency once per k-group per wave).
template <int MT, int NT>
struct Frag {
  float a[MT][4], b[NT][4];
};

template <bool A_KC, bool B_KC, int MT, int NT, int BN_>
__device__ __forceinline__ void load_frag(Frag<MT, NT>& f, const float* __restrict__ As,
                                          const float* __restrict__ Bs, int wr, int wc, int lane, int kgi) {
  constexpr int ARS = BM + 4, BRS = BN_ + 4;
  const int l31 = lane & 31;
  const int k = kgi * 8 + (lane >> 5) * 4;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    if (A_KC) {
      const float4 v = *reinterpret_cast<const float4*>(As + (wr * 32 * MT + t * 32 + l31) * KS + k);
      f.a[t][0] = v.x; f.a[t][1] = v.y; f.a[t][2] = v.z; f.a[t][3] = v.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) f.a[t][s] = As[(k + s) * ARS + wr * 32 * MT + t * 32 + l31];
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (B_KC) {
      const float4 v = *reinterpret_cast<const float4*>(Bs + (wc * 32 * NT + t * 32 + l31) * KS + k);
      f.b[t][0] = v.x; f.b[t][1] = v.y; f.b[t][2] = v.z; f.b[t][3] = v.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) f.b[t][s] = Bs[(k + s) * BRS + wc * 32 * NT + t * 32 + l31];
    }
  }
}

template <int MT, int NT>
__device__ __forceinline__ void mma_frag(f32x16 (&acc)[MT][NT], const Frag<MT, NT>& f) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[mt][s], f.b[nt][s], acc[mt][nt], 0, 0, 0);
}

// one of the four k-steps of a fragment (MT x NT MFMAs)
template <int MT, int NT>
__device__ __forceinline__ void mma_step(f32x16 (&acc)[MT][NT], const Frag<MT, NT>& f, int s) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[mt][s], f.b[nt][s], acc[mt][nt], 0, 0, 0);
}
// the LDS reads of one fragment, split so they can be slotted between MFMAs: part 0..MT-1 = A tiles, MT.. = B tiles
template <bool A_KC, bool B_KC, int MT, int NT, int BN_>
__device__ __forceinline__ void load_frag_part(Frag<MT, NT>& f, const float* __restrict__ As,
                                               const float* __restrict__ Bs, int wr, int wc, int lane, int kgi, int part) {
  constexpr int ARS = BM + 4, BRS = BN_ + 4;
  const int l31 = lane & 31;
  const int k = kgi * 8 + (lane >> 5) * 4;
  if (part < MT) {
    const int t = part;
    if (A_KC) {
      const float4 v = *reinterpret_cast<const float4*>(As + (wr * 32 * MT + t * 32 + l31) * KS + k);
      f.a[t][0] = v.x; f.a[t][1] = v.y; f.a[t][2] = v.z; f.a[t][3] = v.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) f.a[t][s] = As[(k + s) * ARS + wr * 32 * MT + t * 32 + l31];
    }
  } else {
    const int t = part - MT;
    if (B_KC) {
      const float4 v = *reinterpret_cast<const float4*>(Bs + (wc * 32 * NT + t * 32 + l31) * KS + k);
      f.b[t][0] = v.x; f.b[t][1] = v.y; f.b[t][2] = v.z; f.b[t][3] = v.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) f.b[t][s] = Bs[(k + s) * BRS + wc * 32 * NT + t * 32 + l31];
    }
  }
}

__device__ __forceinline__ float apply_act(float v, int act, float lo, float hi) {
  if (act == EVAE_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  if (act == EVAE_ACT_HARDTANH) return fminf(fmaxf(v, lo), hi);
  return v;
}

constexpr int A_TILE_FLOATS = BM * KS;  // 4608 (>= 32 * 132 for the RC layout)
constexpr int b_tile_floats(int bn) { return bn * KS; }  // >= 32 * (bn + 4)
constexpr size_t gemm_lds_bytes(int bn) { return 2 * (size_t)(A_TILE_FLOATS + b_tile_floats(bn)) * sizeof(float); }


// ---- split-K finish: sum the partial planes in a fixed order, then the real epilogue ----------------
struct FinishArgs {
  const float* part;
  int nz, M, N, ldo;
  int epi;
  const float* bias0;
  const float* bias1;
  float* out0;
  float* out1;
  float* out2;
  const float* e0;
  const float* e1;
  int act;
  float lo, hi;
  int accumulate;
  int ones_col;            // EPI_RAW: column that carries the bias gradient (-1 = none)
  float* out_db;
  int perm_c, perm_taps;   // EPI_RAW of a convolution weight gradient: column t*perm_c + c -> c*perm_taps + t (OIHW order)
  int perm_k;              //   ... real filter size C*taps (row stride of dw); columns [perm_k, ones_col) are padding
  P6Sink tsink;            // EPI_GATED / EPI_GATE_BWD, img != NULL: the result also as the pre-split image of its transpose (as
                           // GemmArgs::tsink; element-wise 2-byte stores: the finish runs behind thin launches only)
};

// one element of a result into X^T's image: X's column c = image row, X's row m = k index
__device__ __forceinline__ void p6_sink_element(const P6Sink& s, int row, int k, float v) {
  unsigned short t0, t1, t2;
  p6_split1(v, t0, t1, t2);
  unsigned short* o = reinterpret_cast<unsigned short*>(s.img + p6_off(row, k, s.nks));
  o[0] = t0; o[P6_CHUNK / 2] = t1; o[P6_CHUNK] = t2;
}

// LANES = 1: one thread per output element walks the nz planes.  LANES = 8 (many planes, small output -- e.g.
// the [40 x 301] head gradient over 25 000 rows): eight lanes share an element, lane j sums planes j, j+8, ... and
// a fixed butterfly adds the eight sums, so the walk is 8x shorter and still deterministic.
// (the body is a device function of the flat thread index: gemm_finish_kernel runs it on its own grid, the grouped finish of a
// training step's weight gradients -- csrc/evae_dense_u8.hip::wgrad_finish_group_kernel -- on a per-job index)
template <int LANES>
__device__ __forceinline__ void gemm_finish_body(const FinishArgs& f, const size_t t) {
  const size_t plane = (size_t)f.M * f.N;
  const size_t i = t / LANES;
  const int j = (int)(t % LANES);
  const bool live = i < plane;
  if (LANES == 1 && !live) return;
  const int m = live ? (int)(i / f.N) : 0, n = live ? (int)(i - (size_t)m * f.N) : 0;
  const size_t o = (size_t)m * f.ldo + n;
  if (f.epi == EPI_GATED) {
    float h = 0.f, gg = 0.f;
    if (live)
      for (int z = j; z < f.nz; z += LANES) {
        h += f.part[(size_t)z * 2 * plane + i];
        gg += f.part[(size_t)z * 2 * plane + plane + i];
      }
    if (LANES > 1) {
#pragma unroll
      for (int d = 1; d < LANES; d <<= 1) { h += __shfl_xor(h, d); gg += __shfl_xor(gg, d); }
      if (!live || j != 0) return;
    }
    h += f.bias0 ? f.bias0[n] : 0.f;
    const float s = 1.0f / (1.0f + expf(-(gg + (f.bias1 ? f.bias1[n] : 0.f))));
    f.out0[o] = h * s;
    if (f.out1) f.out1[o] = h;
    if (f.out2) f.out2[o] = s;
    if (f.tsink.img) p6_sink_element(f.tsink, f.tsink.row0 + n, f.tsink.kbase + m, h * s);
    return;
  }
  float v = 0.f;
  if (live)
    for (int z = j; z < f.nz; z += LANES) v += f.part[(size_t)z * plane + i];
  if (LANES > 1) {
#pragma unroll
    for (int d = 1; d < LANES; d <<= 1) v += __shfl_xor(v, d);
    if (!live || j != 0) return;
  }
  if (f.epi == EPI_LINEAR) {
    const float pre = v + (f.bias0 ? f.bias0[n] : 0.f);
    if (f.out1) f.out1[o] = pre;
    f.out0[o] = apply_act(pre, f.act, f.lo, f.hi);
  } else if (f.epi == EPI_GATE_BWD) {
    const float go = f.e0[i], s = f.e1[i];
    if (f.out0) { f.out0[o] = v * s; f.out1[o] = v * go * (1.0f - s); }
    if (f.tsink.img) {
      p6_sink_element(f.tsink, f.tsink.row0 + n, f.tsink.kbase + m, v * s);
      p6_sink_element(f.tsink, f.tsink.row0 + f.N + n, f.tsink.kbase + m, v * go * (1.0f - s));
    }
  } else {  // EPI_RAW: plain sum (weight gradient), optional accumulate; column ones_col is db
    if (f.ones_col >= 0) {
      if (n == f.ones_col) { if (f.out_db) f.out_db[m] = (f.accumulate ? f.out_db[m] : 0.f) + v; }
      else if (f.perm_taps > 0) {
        if (n < f.perm_k) {
          const int t = n / f.perm_c;
          const size_t ow = (size_t)m * f.perm_k + (n - t * f.perm_c) * f.perm_taps + t;
          f.out0[ow] = (f.accumulate ? f.out0[ow] : 0.f) + v;
        }
      } else {
        const size_t ow = (size_t)m * f.ones_col + n;
        f.out0[ow] = (f.accumulate ? f.out0[ow] : 0.f) + v;
      }
    } else {
      f.out0[o] = (f.accumulate ? f.out0[o] : 0.f) + v;
    }
  }
}

template <int LANES>
static __global__ __launch_bounds__(256) void gemm_finish_kernel(const FinishArgs f) {
  gemm_finish_body<LANES>(f, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// which variant launch_finish takes for a job, and the blocks of 256 threads it needs
static inline int finish_lanes(const FinishArgs& f) { return (f.nz >= 16 && (size_t)f.M * f.N * 8 <= ((size_t)1 << 22)) ? 8 : 1; }
static inline unsigned finish_blocks(const FinishArgs& f) { return (unsigned)(((size_t)f.M * f.N * finish_lanes(f) + 255) / 256); }



struct Plan {
  int bn;      // 128 or 64
  int nz;      // split-K factor (blockIdx.z extent)
  int ksplit;  // slabs per split
};

// Wave-quantisation model: a launch runs in rounds of 512 resident blocks (256 CUs x 2); a block costs
// (slabs + fixed prologue/epilogue) slab-times, a BN=64 slab ~0.6 of a BN=128 slab; split-K adds the
// finish kernel (launch + partial traffic).  Deterministic in (M, N, slabs, gated, must_split).
static double plan_finish_cost() {
  static double c = -1.0;
  if (c < 0) { const char* e = getenv("EVAE_PLAN_FINISH"); c = e ? atof(e) : 3.0; }
  return c;
}

static Plan make_plan(int M, int N, int slabs, bool gated, bool must_split, int planes) {
  Plan best = {128, 1, slabs};
  const double fin = plan_finish_cost();
  double best_t = 1e30;
  const int bns[2] = {128, 64};
  for (int bi = 0; bi < (gated ? 1 : 2); ++bi) {
    const int bn = bns[bi];
    const long tiles = (long)cdiv(M, BM) * cdiv(N, gated ? 64 : bn);
    const double slab_cost = bn == 64 ? 0.6 : 1.0;
    for (int nz = 1; nz <= slabs && nz <= 512; ++nz) {
      const int ks = cdiv(slabs, nz);
      const int nze = cdiv(slabs, ks);
      if (nze != nz) continue;
      const long rounds = (tiles * nze + 511) / 512;
      double t = rounds * (ks + 1.5) * slab_cost;
      // the finish launch: ~3 slab-times of a dependent launch inside a graph (measured: 2 -> 3 takes 0.7 % off the c2 step)
      if (nze > 1 || must_split) t += fin + (double)M * N * planes * nze * 4.0 / 12e6;
      if (t < best_t) { best_t = t; best = {bn, nze, ks}; }
    }
  }
  return best;
}


// Split-K plan for the XCD-local block mapping of gemm_kernel (GemmArgs::sk_local): ONE round of resident blocks in which every
// XCD holds whole units (a unit = the column tiles of one (slice, row tile)): 64 block slots per XCD -> floor(64 / tiles_n)
// units per XCD -> at most 8 floor(64 / tiles_n) / tiles_m slices.  Returns nz = 0 when the shape does not fit the scheme.
static Plan make_plan_local(int M, int N, int slabs) {
  Plan best = {64, 0, slabs};
  double best_t = 1e30;
  const int bns[2] = {128, 64};
  for (int bi = 0; bi < 2; ++bi) {
    const int bn = bns[bi];
    const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, bn);
    const int per_xcd = 64 / tiles_n;
    if (per_xcd < 1) continue;
    int nz = (8 * per_xcd) / tiles_m;
    if (nz > slabs) nz = slabs;
    if (nz < 2) continue;
    const int ks = cdiv(slabs, nz), nze = cdiv(slabs, ks);
    if (nze < 2) continue;
    const double slab_cost = bn == 64 ? 0.6 : 1.0;
    // blocks per CU decide the pace: a column-tile count that leaves CUs with one block wastes them
    const double fill = (double)(cdiv(nze * tiles_m, 8) * tiles_n) / 64.0;
    const double t = (ks + 1.5) * slab_cost / (fill > 0.5 ? 1.0 : 2.0 * fill) + 3.0 + (double)M * N * nze * 4.0 / 12e6;
    if (t < best_t) { best_t = t; best = {bn, nze, ks}; }
  }
  return best;
}

static bool sk_local_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("EVAE_SK_LOCAL"); on = (e && atoi(e) == 0) ? 0 : 1; }
  return on == 1;
}

static int launch_finish(const FinishArgs& f, hipStream_t stream) {
  if (finish_lanes(f) == 8) gemm_finish_kernel<8><<<finish_blocks(f), 256, 0, stream>>>(f);
  else gemm_finish_kernel<1><<<finish_blocks(f), 256, 0, stream>>>(f);
  return check_launch("gemm_finish_kernel");
}

static int elt_grid(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b < 4096 ? (b ? b : 1) : 4096);
}


}  // namespace evae
