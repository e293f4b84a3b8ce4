// The pre-split bf16 operand format of evae_gemm_p6.h ("p6 image") and the epilogue-side writer of one: what a layer's
// epilogue needs to leave its output as the operand of the next GEMMs (evae_gemm_p6.h says why and how it is read).
//
// An operand X [R rows x K contraction], both padded (R to 128, K to 16, padding = zeros), is stored as chunks of 16 rows x
// 16 k (512 bytes: rows of 32 bytes whose two 16-byte halves are swapped on the odd group of eight rows), three planes
// (the three bf16 terms of the element) each, ordered (row group, k-step, plane):
//     byte offset of (r, k), plane p = (((r >> 4) * nks + (k >> 4)) * 3 + p) * 512 + (r & 15) * 32
//                                      + ((((k >> 3) & 1) ^ ((r >> 3) & 1)) << 4) + (k & 7) * 2,        nks = K / 16
#pragma once
#include "evae_common.h"

namespace evae {

constexpr int P6_KS = 16;                  // contraction elements per k-step
constexpr int P6_PLANE = 128 * 32;         // bytes of one plane of a (128-row tile, k-step) in LDS
constexpr int P6_TILE = 3 * P6_PLANE;      // 12 KB
constexpr int P6_CHUNK = 512;              // bytes of one (16 rows, k-step, plane) chunk in HBM
constexpr int P6_GROUP = 3 * P6_CHUNK;     // the three planes of a (row group, k-step): 1.5 KB

__host__ __device__ inline int p6_nks(int K) { return (K + P6_KS - 1) / P6_KS; }
// k-steps of an image whose contraction index is a batch-row index (read in 128-row tiles through the transpose path): whole
// groups of eight, plus one (= 1 mod 8: the eight row groups of a tile, nks * 1536 bytes apart, then fall on all sixteen L2
// channels -- evae_gemm_p6.h)
__host__ __device__ inline int p6_nks_rows(int M) { return (M + 127) / 128 * 8 + 1; }
__host__ __device__ inline size_t p6_image_bytes(int rows, int nks) { return (size_t)((rows + 127) / 128 * 8) * nks * P6_GROUP; }
// byte offset of element (r, k) in plane 0 of an image with nks k-steps (plane p: + p * P6_CHUNK)
__host__ __device__ inline size_t p6_off(int r, int k, int nks) {
  return ((size_t)(r >> 4) * nks + (k >> 4)) * P6_GROUP + (size_t)((r & 15) * 32 + ((((k >> 3) & 1) ^ ((r >> 3) & 1)) << 4) + (k & 7) * 2);
}

// round-to-nearest three-term split a = a0 + a1 + a2 (bf16 each; exact: 3 x 8 significant bits) of one fp32
__device__ __forceinline__ unsigned p6_rn(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ void p6_split1(float x, unsigned short& t0, unsigned short& t1, unsigned short& t2) {
  const unsigned a0 = p6_rn(x);
  const float r1 = x - __uint_as_float(a0 << 16);
  const unsigned a1 = p6_rn(r1);
  const float r2 = r1 - __uint_as_float(a1 << 16);
  t0 = (unsigned short)a0; t1 = (unsigned short)a1; t2 = (unsigned short)p6_rn(r2);
}
// ... of two (hardware conversion, v_cvt_pk_bf16_f32): term q of x in the low half of p[q], of y in the high half
__device__ __forceinline__ void p6_split2(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  f32x2_ r = {x, y};
  p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_));
  f32x2_ h = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xFFFF0000u)};
  r = r - h;
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_));
  h[0] = __uint_as_float(p1 << 16); h[1] = __uint_as_float(p1 & 0xFFFF0000u);
  r = r - h;
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_));
}

// Where an epilogue leaves X^T's image: X's column c = image row row0 + c, X's row m = image k index kbase + m (kbase a multiple
// of 8); k indices at or beyond klim are written as zeros (the rows a partial last tile does not have)
struct P6Sink { unsigned char* img; int nks, row0, kbase, klim; };

// One 32 x 32 accumulator tile in the matrix core's C layout -- this lane's column, rows (r & 3) + 8 (r >> 2) + 4 lh of the
// tile, r = 0..15 -- into X^T's image: image row `row` (this lane's column; !row_ok: the lane only takes part in the exchange),
// k indices k0 .. k0 + 31 (k0 = image k of the tile's row 0, a multiple of 8).  A lane holds rows 8 j + 4 lh .. + 3, its
// partner lane ^ 32 the other four of the same eight: the two swap halves (one cross-lane move per dword), then every lane
// stores whole 16-byte slots (8 consecutive k of one image row, one per plane) -- lanes lh = 0 the even j, lh = 1 the odd j; the
// 32 lanes x 2 of one store instruction cover whole 512-byte chunks.  ALL 64 lanes must call this together.
__device__ __forceinline__ void p6_emit_tile(const P6Sink& s, const int row, const bool row_ok, const int k0, const float (&v)[16],
                                             const int lh) {
  unsigned t[4][3][2];                              // [row group j][term][dword]: four rows of one term = 8 bytes
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + 8 * j + 4 * lh + 2 * h;
      const float a = (k < s.klim) ? v[4 * j + 2 * h] : 0.f, b = (k + 1 < s.klim) ? v[4 * j + 2 * h + 1] : 0.f;
      p6_split2(a, b, t[j][0][h], t[j][1][h], t[j][2][h]);
    }
#pragma unroll
  for (int q2 = 0; q2 < 2; ++q2) {                  // the pair of row groups (2 q2, 2 q2 + 1): lh = 0 keeps the even one
    const int k = k0 + 8 * (2 * q2 + lh);           // first of the eight k this lane stores
    unsigned char* const base = s.img + p6_off(row, k, s.nks);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      // send the group the partner keeps, receive the partner's half of mine
      const unsigned s0 = lh ? t[2 * q2][q][0] : t[2 * q2 + 1][q][0];
      const unsigned s1 = lh ? t[2 * q2][q][1] : t[2 * q2 + 1][q][1];
      const unsigned r0 = (unsigned)__shfl_xor((int)s0, 32), r1 = (unsigned)__shfl_xor((int)s1, 32);
      const unsigned m0 = lh ? t[2 * q2 + 1][q][0] : t[2 * q2][q][0];
      const unsigned m1 = lh ? t[2 * q2 + 1][q][1] : t[2 * q2][q][1];
      // k 0..3 of the eight come from the lh = 0 lane, k 4..7 from the lh = 1 lane
      const uint4 val = lh ? make_uint4(r0, r1, m0, m1) : make_uint4(m0, m1, r0, r1);
      if (row_ok && k < s.klim) *reinterpret_cast<uint4*>(base + q * P6_CHUNK) = val;
    }
  }
}

// ---- image builders (weights in the step head; activations in tests and on fallback paths -- the layers' epilogues write
// theirs themselves) ----------------------------------------------------------------------------------------------------------
// X [R x K] fp32, K-contiguous (row stride ld) -> image(rows = R, k = K).  One thread per (image row, 8 k): a 16-byte slot per
// plane; the grid covers the padded image (rows_img rows, whole k-steps), padding = zeros.
// gated > 0: the row order of a gated layer's weight tiles -- image row (tn, c) with c = wc * 64 + hg * 32 + j holds weight row
// tn * 64 + wc * 32 + j of bank hg (x = bank h, x2 = bank g; evae_gemm_x6.h's [wc][h | g][32] LDS rows).
__device__ __forceinline__ void p6_pack_rows_element(size_t t, const float* __restrict__ x, const float* __restrict__ x2, int R, int K,
                                                     long long ld, int gated, int rows_img, int nks, unsigned char* __restrict__ img) {
  const int kslots = nks * 2;
  if (t >= (size_t)rows_img * kslots) return;
  const int ri = (int)(t / kslots), ks8 = (int)(t - (size_t)ri * kslots);
  const int k0 = ks8 * 8;
  int r = ri;
  const float* src = x;
  if (gated) {
    const int tn = ri >> 7, c = ri & 127, wc = c >> 6, hg = (c >> 5) & 1, j = c & 31;
    r = tn * 64 + wc * 32 + j;
    src = hg ? x2 : x;
  }
  unsigned short p0[8], p1[8], p2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float v = (r < R && k0 + i < K) ? src[(size_t)r * ld + k0 + i] : 0.f;
    p6_split1(v, p0[i], p1[i], p2[i]);
  }
  unsigned char* o = img + p6_off(ri, k0, nks);
  *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(p0);
  *reinterpret_cast<uint4*>(o + P6_CHUNK) = *reinterpret_cast<const uint4*>(p1);
  *reinterpret_cast<uint4*>(o + 2 * P6_CHUNK) = *reinterpret_cast<const uint4*>(p2);
}

// X^T: X [Kd x R] fp32 with the CONTRACTION along its rows (row stride ld; rows Kd .. 2 Kd - 1 from x2 when given: the [h | g]
// banks of a gated layer stacked along the contraction) -> image(rows = R, k).  ones_row >= 0: that image row is 1 for k < Kd_all
// (the bias-gradient row of a weight gradient's x operand).  One thread per (image row, 8 k); consecutive threads = consecutive
// image rows (coalesced along a row of X).
__device__ __forceinline__ void p6_pack_cols_element(size_t t, const float* __restrict__ x, const float* __restrict__ x2, int Kd, int R,
                                                     long long ld, int ones_row, int rows_img, int nks, unsigned char* __restrict__ img) {
  if (t >= (size_t)rows_img * nks * 2) return;
  const int ks8 = (int)(t / rows_img), r = (int)(t - (size_t)ks8 * rows_img);
  const int k0 = ks8 * 8;
  const int Kall = x2 ? 2 * Kd : Kd;
  unsigned short p0[8], p1[8], p2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = k0 + i;
    float v = 0.f;
    if (r < R && k < Kall) v = k < Kd ? x[(size_t)k * ld + r] : x2[(size_t)(k - Kd) * ld + r];
    if (r == ones_row && k < Kall) v = 1.f;
    p6_split1(v, p0[i], p1[i], p2[i]);
  }
  unsigned char* o = img + p6_off(r, k0, nks);
  *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(p0);
  *reinterpret_cast<uint4*>(o + P6_CHUNK) = *reinterpret_cast<const uint4*>(p1);
  *reinterpret_cast<uint4*>(o + 2 * P6_CHUNK) = *reinterpret_cast<const uint4*>(p2);
}

}  // namespace evae
