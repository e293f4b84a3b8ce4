// The weight split of the byte-store layer (csrc/evae_dense_u8.hip) as a device function: shared by u8_prepare_kernel and by
// the step-head kernel of evae_loss.hip, which does it in the same launch as the batch prologue.
#pragma once
#include "evae_common.h"

namespace evae {

constexpr int U8_BN = 64, U8_BK = 32;

__device__ __forceinline__ unsigned short bf16_rn(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__host__ __device__ static inline int u8_prepare_nslab(int K) { return (K + U8_BK - 1) / U8_BK; }
static inline size_t u8_prepare_elems(int N, int K) { return (size_t)((N + U8_BN - 1) / U8_BN) * u8_prepare_nslab(K) * 128 * 32; }

// B-tile images: image (tn, s) = [term p][column c = wc*64 + hg*32 + j][slot^swz][8 k], 24 KB each.  Element e = one
// (tn, s, c, k), all three terms: w = w0 + w1 + w2, round-to-nearest bf16 terms (exact: 3 x 8 significant bits).
__device__ __forceinline__ void u8_prepare_element(size_t e, const float* __restrict__ wh, const float* __restrict__ wg, int N,
                                                   int K, int nslab, unsigned short* __restrict__ img) {
  const size_t per_img = (size_t)128 * 32;
  const size_t im = e / per_img;
  const int rem = (int)(e - im * per_img);
  const int c = rem >> 5, k = rem & 31;
  const int tn = (int)(im / nslab), s = (int)(im - (size_t)tn * nslab);
  const int wc = c >> 6, hg = (c >> 5) & 1, j = c & 31;
  const int n = tn * U8_BN + wc * 32 + j, kk = s * U8_BK + k;
  float w = 0.f;
  if (n < N && kk < K) w = (hg ? wg : wh)[(size_t)n * K + kk];
  const unsigned short w0 = bf16_rn(w);
  const float r1 = w - bf16_f(w0);
  const unsigned short w1 = bf16_rn(r1);
  const float r2 = r1 - bf16_f(w1);
  const unsigned short w2 = bf16_rn(r2);
  const int slot = (k >> 3) ^ ((c >> 2) & 3);
  unsigned short* o = img + im * (size_t)(3 * 128 * 32) + (size_t)c * 32 + slot * 8 + (k & 7);
  o[0] = w0; o[128 * 32] = w1; o[2 * 128 * 32] = w2;
}

// A weight transposition riding in the step-head launch: dst[p][k][n] = w_p[n][k] (row stride ldt), the layout the split-bf16
// data gradients read (evae_dense_bwd_data_wt).  Tiles of 32 x 32 through LDS, one block per tile, as transpose_pairs_kernel.
struct WtJob {
  const float* w1; const float* w2;      // [N x K] each (w2 NULL: one matrix)
  float* dst;
  int N, K, ldt, tx, ty, ntiles;         // tx = tiles along K, ty = tiles along ldt, ntiles = tx * ty * (w2 ? 2 : 1)
};
__device__ __forceinline__ void wt_job_tile(const WtJob& j, int t, float (*tile)[33]) {
  const int per = j.tx * j.ty, z = t / per, r = t - z * per;
  const int by = r / j.tx, bx = r - by * j.tx;
  const float* w = z ? j.w2 : j.w1;
  float* o = j.dst + (size_t)z * j.K * j.ldt;
  const int n0 = by * 32, k0 = bx * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int rr = ty; rr < 32; rr += 8)
    tile[rr][tx] = (n0 + rr < j.N && k0 + tx < j.K) ? w[(size_t)(n0 + rr) * j.K + k0 + tx] : 0.f;
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8)
    if (k0 + rr < j.K && n0 + tx < j.ldt) o[(size_t)(k0 + rr) * j.ldt + n0 + tx] = tile[tx][rr];
}


// ---- finish of the byte layer's split-K weight gradient (csrc/evae_dense_u8.hip), as a device function so that the grouped
// finish of a training step (csrc/evae_dense.hip::wgrad_finish_group_kernel) can run it beside the fp32 layers' finishes.
// dw[n][k] = x_scale * sum_z part[z][k][n] (k < K), db[n] = sum_z part[z][K][n]; fixed order.  32 x 32 tiles through LDS: the
// planes are read along n and dw is written along k, both in full 128-byte runs (r02 wrote dw with a stride of K floats per
// lane); eight planes' worth of loads in flight before the first add (two blocks per CU: one dependent load per plane was 14
// memory latencies in a row).  256 threads, tile (bx, by) of the [K + 1] x [N] plane.
struct U8FinishArgs { const float* part; int nz, K, N; float x_scale; float* dw; float* db; };

__device__ __forceinline__ void u8_wgrad_finish_body(const U8FinishArgs& u, const int bx, const int by, float (*tile)[33]) {
  const float* __restrict__ part = u.part;
  const int nz = u.nz, K = u.K, N = u.N;
  const int k0 = bx * 32, n0 = by * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t plane = (size_t)(K + 1) * N;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  const int n = n0 + tx;
  for (int z0 = 0; z0 < nz; z0 += 8) {
    float v[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + 8 * i;
        const bool ok = z0 + j < nz && k <= K && n < N;
        v[j][i] = ok ? part[(size_t)(z0 + j) * plane + (size_t)k * N + n] : 0.f;
      }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] += v[j][i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty + 8 * i;
    tile[ty + 8 * i][tx] = a[i];
    if (k == K && n < N && u.db) u.db[n] = a[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int nn = n0 + ty + 8 * i, k = k0 + tx;
    if (nn < N && k < K) u.dw[(size_t)nn * K + k] = tile[tx][ty + 8 * i] * u.x_scale;
  }
}

// host side (csrc/evae_dense_u8.hip): the finish of evae_dense_bwd_weight_u8(M, N, K) over the partial planes in `ws`
int u8_wgrad_finish_job(int M, int N, int K, float x_scale, float* dw, float* db, void* ws, size_t ws_bytes, U8FinishArgs* out,
                        int* tiles_x, int* tiles_y);

}  // namespace evae
