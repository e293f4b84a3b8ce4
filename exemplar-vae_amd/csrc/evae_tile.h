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
constexpr float kHalfLog2e = 0.5f * kLog2e;

struct PriorGeom {
  int kc;       // chunk width (multiple of 4, <= KC_MAX)
  int ks;       // LDS row stride in floats, ks/4 odd
  int nchunk;   // ceil(zdim / kc)
};

static PriorGeom prior_geom(int zdim) {
  PriorGeom g;
  int zp = (zdim + 3) & ~3;
  g.kc = zp < KC_MAX ? zp : KC_MAX;
  g.nchunk = (zdim + g.kc - 1) / g.kc;
  int q = g.kc / 4;
  g.ks = g.kc + (((q + 1) & 1) ? 4 : 8);
  return g;
}

static size_t prior_lds_bytes(const PriorGeom& g, bool bwd) {
  size_t fl = (size_t)(BQ + BE) * g.ks + 2 * KC_MAX * 4 /*inv_sigma, lv scratch*/ + 64;
  if (bwd) fl += (size_t)BQ * (BE + 1);
  return fl * sizeof(float);
}

// Stage `nrows` rows x `kc` dims (dims k0..k0+kc of rows r0..) of src[nrows_total x zdim] into LDS,
// multiplied by inv_sigma.  Rows >= nrows_total and dims >= zdim are zero-filled.
template <bool SCALE = true>
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, const float* __restrict__ src,
                                           int r0, int nrows_total, int tile_rows, int zdim, int k0,
                                           int kc, int ks, const float* __restrict__ inv_sigma_lds,
                                           bool vec_ok) {
  const int kq = kc >> 2;
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
__device__ __forceinline__ void dist_chunk(float (&acc)[TQ][TE], const float* __restrict__ Qs,
                                           const float* __restrict__ Es, int tq, int te, int kc,
                                           int ks) {
  for (int k = 0; k < kc; k += 4) {
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


}  // namespace evae
