// 128-query x 64-exemplar LDS tile shared by the prior (fp32) and top-K (fp64) kernels.
#pragma once
#include "evae_common.h"

namespace evae {

constexpr int TQ = 8;     // queries per thread
constexpr int TE = 4;     // exemplars per thread
constexpr int BQ = 128;   // queries per block tile
constexpr int BE = 64;    // exemplars per block tile
constexpr int NT = 256;
constexpr int KC_MAX = 64;  // z-dims staged per chunk
constexpr int ZDIM_MAX = 8 * KC_MAX;   // largest latent size of the VALU prior kernels (512: fully_conv on 28x28 has 6*7*7 = 294)
constexpr float kHalfLog2e = 0.5f * kLog2e;

// Chunk geometry is a compile-time parameter (KC = z-dims staged per chunk, one of 16/32/40/64) so that
// every inner loop is fully unrolled with static register indices and batched LDS reads.
// LDS row stride KS = KC + pad with KS/4 odd: the 16 lanes of a ds_read_b128 group hit 16 different
// rows and cover all 64 banks.
template <int KC> struct Geom {
  static constexpr int kc = KC;
  static constexpr int ks = KC + ((((KC / 4) + 1) & 1) ? 4 : 8);
};

struct PriorGeom {
  int kc;       // chunk width actually instantiated (16, 32, 40 or 64)
  int nchunk;   // ceil(zdim / kc)
};

static PriorGeom prior_geom(int zdim) {
  PriorGeom g;
  g.kc = zdim <= 16 ? 16 : zdim <= 32 ? 32 : zdim <= 40 ? 40 : 64;
  g.nchunk = (zdim + g.kc - 1) / g.kc;
  return g;
}

static int geom_ks(int kc) { return kc == 16 ? Geom<16>::ks : kc == 32 ? Geom<32>::ks : kc == 40 ? Geom<40>::ks : Geom<64>::ks; }

static size_t prior_lds_bytes(const PriorGeom& g, bool bwd) {
  size_t fl = (size_t)(BQ + BE) * geom_ks(g.kc) + ZDIM_MAX /*inv_sigma*/ + 128;
  if (bwd) fl += (size_t)BQ * (BE + 1);
  return fl * sizeof(float);
}

// Stage tile_rows x KC dims (dims k0.. of rows r0..) of src[nrows_total x zdim] into LDS, optionally
// multiplied by inv_sigma.  Rows >= nrows_total and dims >= zdim are zero-filled.
template <int KC, bool SCALE = true>
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, const float* __restrict__ src,
                                           int r0, int nrows_total, int tile_rows, int zdim, int k0,
                                           const float* __restrict__ inv_sigma_lds, bool vec_ok) {
  constexpr int ks = Geom<KC>::ks;
  constexpr int kq = KC >> 2;
  const int total = tile_rows * kq;
  for (int f = threadIdx.x; f < total; f += NT) {
    int row = f / kq;
    int kk = (f - row * kq) << 2;
    int gr = r0 + row;
    int gk = k0 + kk;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gr < nrows_total) {
      const float* p = src + (size_t)gr * zdim + gk;
      if (vec_ok && gk + 3 < zdim) {
        v = *reinterpret_cast<const float4*>(p);
      } else {
        if (gk + 0 < zdim) v.x = p[0];
        if (gk + 1 < zdim) v.y = p[1];
        if (gk + 2 < zdim) v.z = p[2];
        if (gk + 3 < zdim) v.w = p[3];
      }
      if (SCALE) {
        const float4 is = *reinterpret_cast<const float4*>(inv_sigma_lds + kk);
        v.x *= is.x; v.y *= is.y; v.z *= is.z; v.w *= is.w;
      }
    }
    *reinterpret_cast<float4*>(dst + row * ks + kk) = v;
  }
}

// acc[i][j] += sum_k (q_i[k] - e_j[k])^2 over one staged chunk
template <int KC>
__device__ __forceinline__ void dist_chunk(float (&acc)[TQ][TE], const float* __restrict__ Qs,
                                           const float* __restrict__ Es, int tq, int te) {
  constexpr int ks = Geom<KC>::ks;
#pragma unroll 2
  for (int k = 0; k < KC; k += 4) {
    float4 e4[TE];
#pragma unroll
    for (int j = 0; j < TE; ++j)
      e4[j] = *reinterpret_cast<const float4*>(Es + (te + 16 * j) * ks + k);
#pragma unroll
    for (int i = 0; i < TQ; ++i) {
      const float4 q4 = *reinterpret_cast<const float4*>(Qs + (tq + 16 * i) * ks + k);
#pragma unroll
      for (int j = 0; j < TE; ++j) {
        float d0 = q4.x - e4[j].x, d1 = q4.y - e4[j].y, d2 = q4.z - e4[j].z, d3 = q4.w - e4[j].w;
        float a = acc[i][j];
        a = fmaf(d0, d0, a);
        a = fmaf(d1, d1, a);
        a = fmaf(d2, d2, a);
        a = fmaf(d3, d3, a);
        acc[i][j] = a;
      }
    }
  }
}

// dispatch a kernel template over the supported chunk widths
#define EVAE_DISPATCH_KC(kc, ...)                                   \
  do {                                                              \
    if ((kc) == 16) { constexpr int KC_ = 16; __VA_ARGS__; }        \
    else if ((kc) == 32) { constexpr int KC_ = 32; __VA_ARGS__; }   \
    else if ((kc) == 40) { constexpr int KC_ = 40; __VA_ARGS__; }   \
    else { constexpr int KC_ = 64; __VA_ARGS__; }                   \
  } while (0)

}  // namespace evae
