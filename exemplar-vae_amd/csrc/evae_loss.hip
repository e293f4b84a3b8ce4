// Row-wise latent sampling and log-density kernels (one wave per row, shuffle reductions).
// Replaces models/BaseModel.py:79-82 (reparameterize) and utils/distributions.py:28-33,44-51
// (log_normal_diag, log_bernoulli) and their autograd.
#include "evae_common.h"
#include "evae_u8_prepare.h"
#include "evae_p6_image.h"

namespace evae {

constexpr int LNT = 256;  // 4 rows (waves) per block

__global__ __launch_bounds__(LNT) void reparam_logq_fwd_kernel(const float* __restrict__ mu,
                                                               const float* __restrict__ logvar,
                                                               const float* __restrict__ eps, int B,
                                                               int zdim, float* __restrict__ z,
                                                               float* __restrict__ logq) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (LNT / 64) + (threadIdx.x >> 6);
  if (row >= B) return;
  float acc = 0.f;
  for (int k = lane; k < zdim; k += 64) {
    const size_t o = (size_t)row * zdim + k;
    const float m = mu[o], lv = logvar[o];
    const float zz = eps[o] * expf(0.5f * lv) + m;
    z[o] = zz;
    const float d = zz - m;
    acc += -0.5f * (lv + kLog2Pi + d * d / expf(lv));
  }
  acc = wave_sum(acc);
  if (lane == 0 && logq) logq[row] = acc;
}

__global__ void reparam_logq_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                        const float* __restrict__ eps, const float* __restrict__ z,
                                        const float* __restrict__ dz, const float* __restrict__ dlogq,
                                        int B, int zdim, float* __restrict__ dmu,
                                        float* __restrict__ dlogvar) {
  const size_t n = (size_t)B * zdim;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int row = (int)(i / zdim);
  const float m = mu[i], lv = logvar[i];
  const float var = expf(lv), sd = expf(0.5f * lv);
  const float d = z[i] - m;
  const float gq = dlogq ? dlogq[row] : 0.f;
  const float gz = (dz ? dz[i] : 0.f) + gq * (-(d / var));       // total gradient reaching z
  dmu[i] = gz + gq * (d / var);
  dlogvar[i] = gz * eps[i] * sd * 0.5f + gq * (-0.5f) * (1.0f - d * d / var);
}

__global__ __launch_bounds__(LNT) void log_normal_diag_fwd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ mu,
                                                                  const float* __restrict__ logvar,
                                                                  int B, int zdim,
                                                                  float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (LNT / 64) + (threadIdx.x >> 6);
  if (row >= B) return;
  float acc = 0.f;
  for (int k = lane; k < zdim; k += 64) {
    const size_t o = (size_t)row * zdim + k;
    const float lv = logvar[o];
    const float d = x[o] - mu[o];
    acc += -0.5f * (lv + kLog2Pi + d * d / expf(lv));
  }
  acc = wave_sum(acc);
  if (lane == 0) out[row] = acc;
}

__global__ void log_normal_diag_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                           const float* __restrict__ logvar,
                                           const float* __restrict__ dout, int B, int zdim,
                                           float* __restrict__ dx, float* __restrict__ dmu,
                                           float* __restrict__ dlogvar) {
  const size_t n = (size_t)B * zdim;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = dout[i / zdim];
  const float var = expf(logvar[i]);
  const float d = x[i] - mu[i];
  const float t = g * (d / var);
  if (dx) dx[i] = -t;
  if (dmu) dmu[i] = t;
  if (dlogvar) dlogvar[i] = g * (-0.5f) * (1.0f - d * d / var);
}

// ... with the Hardtanh(lo, hi) of the log-variance head folded in (dlv_pre = gradient of the head's pre-activation)
__global__ void log_normal_diag_bwd_ht_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                              const float* __restrict__ logvar, const float* __restrict__ lv_pre, float lo, float hi,
                                              const float* __restrict__ dout, int B, int zdim, float* __restrict__ dx,
                                              float* __restrict__ dmu, float* __restrict__ dlv_pre) {
  const size_t n = (size_t)B * zdim;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = dout[i / zdim];
  const float var = expf(logvar[i]);
  const float d = x[i] - mu[i];
  const float t = g * (d / var);
  if (dx) dx[i] = -t;
  dmu[i] = t;
  const float pre = lv_pre[i];
  dlv_pre[i] = (pre > lo && pre < hi) ? g * (-0.5f) * (1.0f - d * d / var) : 0.f;
}

constexpr float kMinEps = 1e-5f, kMaxEps = 1.0f - 1e-5f;

__global__ __launch_bounds__(LNT) void bernoulli_ll_fwd_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ mean, int B,
                                                               int D, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (LNT / 64) + (threadIdx.x >> 6);
  if (row >= B) return;
  float acc = 0.f;
  for (int k = lane; k < D; k += 64) {
    const size_t o = (size_t)row * D + k;
    const float p = fminf(fmaxf(mean[o], kMinEps), kMaxEps);
    const float xv = x[o];
    acc += xv * logf(p) + (1.0f - xv) * logf(1.0f - p);
  }
  acc = wave_sum(acc);
  if (lane == 0) out[row] = acc;
}

__global__ void bernoulli_ll_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                        const float* __restrict__ dout, int B, int D,
                                        float* __restrict__ dmean) {
  const size_t n = (size_t)B * D;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float mv = mean[i];
  const bool inside = (mv >= kMinEps) && (mv <= kMaxEps);
  const float p = fminf(fmaxf(mv, kMinEps), kMaxEps);
  const float xv = x[i];
  dmean[i] = inside ? dout[i / D] * (xv / p - (1.0f - xv) / (1.0f - p)) : 0.f;
}

// ---- 256-bin discretised logistic (utils/distributions.py:54-66), summed over dim 1; one wave per row ---------------
//   scale = exp(logvar), xs = (floor(256 x)/256 - mean)/scale, ll = log(sigmoid(xs + 1/(256 scale)) - sigmoid(xs) + 1e-7)
// logvar: [B x D] (lv_scalar = 0) or ONE value (lv_scalar = 1: fully_conv's decoder_logstd, models/AbsModel.py:35-37)
__device__ __forceinline__ float ll256_terms(float xv, float m, float lv, float& cp, float& cm, float& u, float& v) {
  const float bin = 1.0f / 256.0f;
  const float is = expf(-lv);
  v = (floorf(xv / bin) * bin - m) * is;
  u = v + bin * is;
  cp = 1.0f / (1.0f + expf(-u));
  cm = 1.0f / (1.0f + expf(-v));
  return cp - cm + 1e-7f;
}

__global__ __launch_bounds__(LNT) void log_logistic256_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                                  const float* __restrict__ logvar, int lv_scalar, int B,
                                                                  int D, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (LNT / 64) + (threadIdx.x >> 6);
  if (row >= B) return;
  const float lv0 = lv_scalar ? logvar[0] : 0.f;
  float acc = 0.f;
  for (int k = lane; k < D; k += 64) {
    const size_t o = (size_t)row * D + k;
    float cp, cm, u, v;
    acc += logf(ll256_terms(x[o], mean[o], lv_scalar ? lv0 : logvar[o], cp, cm, u, v));
  }
  acc = wave_sum(acc);
  if (lane == 0) out[row] = acc;
}

// dmean [B x D]; dlogvar [B x D], or per-row partial sums [B] when logvar is one value (summed by the kernel below)
__global__ __launch_bounds__(LNT) void log_logistic256_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                                  const float* __restrict__ logvar, int lv_scalar,
                                                                  const float* __restrict__ dout, int B, int D,
                                                                  float* __restrict__ dmean, float* __restrict__ dlogvar) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (LNT / 64) + (threadIdx.x >> 6);
  if (row >= B) return;
  const float lv0 = lv_scalar ? logvar[0] : 0.f;
  const float g = dout[row];
  float acc = 0.f;
  for (int k = lane; k < D; k += 64) {
    const size_t o = (size_t)row * D + k;
    const float lv = lv_scalar ? lv0 : logvar[o];
    float cp, cm, u, v;
    const float r = g / ll256_terms(x[o], mean[o], lv, cp, cm, u, v);
    const float a = cp * (1.0f - cp), b = cm * (1.0f - cm);
    if (dmean) dmean[o] = -r * (a - b) * expf(-lv);
    const float dl = -r * (a * u - b * v);
    if (lv_scalar) acc += dl;
    else if (dlogvar) dlogvar[o] = dl;
  }
  if (lv_scalar && dlogvar) {
    acc = wave_sum(acc);
    if (lane == 0) dlogvar[row] = acc;
  }
}

// The same two kernels for long rows (D >= 4096: fully_conv's 3 x 64 x 64 images, B = 100 rows): one block of sixteen waves per row --
// a wave per row leaves a 100-row batch on 100 of the chip's 1024 SIMDs -- wave sums combined in wave order (deterministic).
constexpr int LWT = 1024;
__global__ __launch_bounds__(LWT) void log_logistic256_fwd_wide_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                                       const float* __restrict__ logvar, int lv_scalar, int B,
                                                                       int D, float* __restrict__ out) {
  __shared__ float red[LWT / 64];
  const int row = blockIdx.x;
  const float lv0 = lv_scalar ? logvar[0] : 0.f;
  float acc = 0.f;
  for (int k = threadIdx.x; k < D; k += LWT) {
    const size_t o = (size_t)row * D + k;
    float cp, cm, u, v;
    acc += logf(ll256_terms(x[o], mean[o], lv_scalar ? lv0 : logvar[o], cp, cm, u, v));
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < LWT / 64; ++w) t += red[w];
    out[row] = t;
  }
}

__global__ __launch_bounds__(LWT) void log_logistic256_bwd_wide_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                                       const float* __restrict__ logvar, int lv_scalar,
                                                                       const float* __restrict__ dout, int B, int D,
                                                                       float* __restrict__ dmean, float* __restrict__ dlogvar) {
  __shared__ float red[LWT / 64];
  const int row = blockIdx.x;
  const float lv0 = lv_scalar ? logvar[0] : 0.f;
  const float g = dout[row];
  float acc = 0.f;
  for (int k = threadIdx.x; k < D; k += LWT) {
    const size_t o = (size_t)row * D + k;
    const float lv = lv_scalar ? lv0 : logvar[o];
    float cp, cm, u, v;
    const float r = g / ll256_terms(x[o], mean[o], lv, cp, cm, u, v);
    const float a = cp * (1.0f - cp), b = cm * (1.0f - cm);
    if (dmean) dmean[o] = -r * (a - b) * expf(-lv);
    const float dl = -r * (a * u - b * v);
    if (lv_scalar) acc += dl;
    else if (dlogvar) dlogvar[o] = dl;
  }
  if (lv_scalar && dlogvar) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < LWT / 64; ++w) t += red[w];
      dlogvar[row] = t;
    }
  }
}

// out[0] = sum of v[0..n) in a fixed order (one block)
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += v[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

// ELU (alpha = 1) over a flat array: the activation in front of a fully_conv residual block's convolution
__global__ void elu_fwd_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 4 <= n) {
    float4 v = *reinterpret_cast<const float4*>(x + i);
    v.x = v.x > 0.f ? v.x : expm1f(v.x); v.y = v.y > 0.f ? v.y : expm1f(v.y);
    v.z = v.z > 0.f ? v.z : expm1f(v.z); v.w = v.w > 0.f ? v.w : expm1f(v.w);
    *reinterpret_cast<float4*>(out + i) = v;
  } else {
    for (size_t j = i; j < n; ++j) out[j] = x[j] > 0.f ? x[j] : expm1f(x[j]);
  }
}

// Bernoulli log-likelihood backward through the sigmoid that produced `mean`: d/dpre = d/dmean * mean * (1 - mean)
__global__ void bernoulli_sigmoid_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                             const float* __restrict__ dout, int B, int D,
                                             float* __restrict__ dpre) {
  const size_t n = (size_t)B * D;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float mv = mean[i];
  const bool inside = (mv >= kMinEps) && (mv <= kMaxEps);
  const float p = fminf(fmaxf(mv, kMinEps), kMaxEps);
  const float xv = x[i];
  const float dm = inside ? dout[i / D] * (xv / p - (1.0f - xv) / (1.0f - p)) : 0.f;
  dpre[i] = dm * mv * (1.0f - mv);
}

// reparam_logq_bwd with a second upstream gradient of z (dz + dz2) and the Hardtanh(lo, hi) that produced logvar
// from lv_pre folded in: dlogvar comes out as the gradient of the pre-activation
__global__ void reparam_logq_bwd_ht_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                           const float* __restrict__ eps, const float* __restrict__ z,
                                           const float* __restrict__ dz, const float* __restrict__ dz2,
                                           const float* __restrict__ dlogq, const float* __restrict__ lv_pre,
                                           float lo, float hi, int B, int zdim, float* __restrict__ dmu,
                                           float* __restrict__ dlv_pre) {
  const size_t n = (size_t)B * zdim;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int row = (int)(i / zdim);
  const float m = mu[i], lv = logvar[i];
  const float var = expf(lv), sd = expf(0.5f * lv);
  const float d = z[i] - m;
  const float gq = dlogq ? dlogq[row] : 0.f;
  const float up = (dz ? dz[i] : 0.f) + (dz2 ? dz2[i] : 0.f);
  const float gz = up + gq * (-(d / var));
  dmu[i] = gz + gq * (d / var);
  const float dl = gz * eps[i] * sd * 0.5f + gq * (-0.5f) * (1.0f - d * d / var);
  const float pre = lv_pre[i];
  dlv_pre[i] = (pre > lo && pre < hi) ? dl : 0.f;
}

// ---- merged element-wise launches of a captured step (r06: a replayed node costs the host 3.6 us whatever it computes) ----
// Reconstruction term of a step whose backward is loss.backward(ones) on the batch means: RE[b] (bernoulli_ll_fwd_kernel), the
// step's coefficient vectors (elbo_bwd_kernel of a unit upstream: cRE = -1/B, cKL = beta/B, neg_cKL = -beta/B) and the gradient
// of the sigmoid head's pre-activation (bernoulli_sigmoid_bwd_kernel with dout = cRE) -- the same arithmetic as the three
// launches, element by element, in one pass over x / mean.
__global__ __launch_bounds__(LNT) void bernoulli_unit_step_kernel(const float* __restrict__ x, const float* __restrict__ mean, int B,
                                                                  int D, const float* __restrict__ beta_dev, float beta_host,
                                                                  float* __restrict__ RE, float* __restrict__ cRE,
                                                                  float* __restrict__ cKL, float* __restrict__ neg_cKL,
                                                                  float* __restrict__ dpre) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (LNT / 64) + (threadIdx.x >> 6);
  if (row >= B) return;
  const float beta = beta_dev ? beta_dev[0] : beta_host;
  const float invb = 1.0f / (float)B;
  const float gl = 1.0f * invb;
  const float ck = 0.f + beta * gl;
  const float cre = 0.f - gl;
  float acc = 0.f;
  for (int k = lane; k < D; k += 64) {
    const size_t o = (size_t)row * D + k;
    const float mv = mean[o];
    const bool inside = (mv >= kMinEps) && (mv <= kMaxEps);
    const float p = fminf(fmaxf(mv, kMinEps), kMaxEps);
    const float xv = x[o];
    acc += xv * logf(p) + (1.0f - xv) * logf(1.0f - p);
    const float dm = inside ? cre * (xv / p - (1.0f - xv) / (1.0f - p)) : 0.f;
    dpre[o] = dm * mv * (1.0f - mv);
  }
  acc = wave_sum(acc);
  if (lane == 0) { RE[row] = acc; cRE[row] = cre; cKL[row] = ck; neg_cKL[row] = -ck; }
}

// reparam_logq_bwd_ht_kernel, and in one more block of the same launch the two single-block launches that sat beside it on the
// batch rows' chain: the ELBO's assembly (evae_elbo_assemble: same rows per lane, same order) and the sum of the prior's
// log-variance gradient row (sum_small_kernel's order: lane i the elements i, i + 64, ...).
__global__ __launch_bounds__(256) void reparam_logq_bwd_ht_tail_kernel(
    const float* __restrict__ mu, const float* __restrict__ logvar, const float* __restrict__ eps, const float* __restrict__ z,
    const float* __restrict__ dz, const float* __restrict__ dz2, const float* __restrict__ dlogq, const float* __restrict__ lv_pre,
    float lo, float hi, int B, int zdim, float* __restrict__ dmu, float* __restrict__ dlv_pre, int elt_blocks,
    const float* __restrict__ a_logp, const float* __restrict__ a_RE, const float* __restrict__ a_logq,
    const float* __restrict__ beta_dev, float beta_host, float* __restrict__ a_loss, float* __restrict__ a_KL,
    float* __restrict__ a_means, const float* __restrict__ sum_src, int sum_n, float* __restrict__ sum_dst) {
  if ((int)blockIdx.x < elt_blocks) {
    const size_t n = (size_t)B * zdim;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int row = (int)(i / zdim);
    const float m = mu[i], lv = logvar[i];
    const float var = expf(lv), sd = expf(0.5f * lv);
    const float d = z[i] - m;
    const float gq = dlogq ? dlogq[row] : 0.f;
    const float up = (dz ? dz[i] : 0.f) + (dz2 ? dz2[i] : 0.f);
    const float gz = up + gq * (-(d / var));
    dmu[i] = gz + gq * (d / var);
    const float dl = gz * eps[i] * sd * 0.5f + gq * (-0.5f) * (1.0f - d * d / var);
    const float pre = lv_pre[i];
    dlv_pre[i] = (pre > lo && pre < hi) ? dl : 0.f;
    return;
  }
  __shared__ float red[3][2];
  if (sum_dst && threadIdx.x >= 192) {                   // the block's last wave: the row's sum
    const int l = threadIdx.x - 192;
    float s = 0.f;
    for (int i = l; i < sum_n; i += 64) s += sum_src[i];
    s = wave_sum(s);
    if (l == 0) sum_dst[0] = s;
  }
  if (a_loss == nullptr) return;
  const float beta = beta_dev ? beta_dev[0] : beta_host;
  float sl = 0.f, sr = 0.f, sk = 0.f;
  if (threadIdx.x < 128)
    for (int row = threadIdx.x; row < B; row += 128) {
      const float kl = a_logq[row] - a_logp[row];
      const float l = beta * kl - a_RE[row];
      a_KL[row] = kl; a_loss[row] = l;
      sl += l; sr += a_RE[row]; sk += kl;
    }
  if (a_means == nullptr) return;
  sl = wave_sum(sl); sr = wave_sum(sr); sk = wave_sum(sk);
  if (threadIdx.x < 128 && (threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sl; red[1][threadIdx.x >> 6] = sr; red[2][threadIdx.x >> 6] = sk; }
  __syncthreads();
  if (threadIdx.x < 3) a_means[threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) / (float)B;
}

// ---- head of a training step: batch gather + dynamic binarisation + eps, one launch ---------------------------------
// Counter-based generator (Philox4x32-10): element e of stream s at step t always gets the same 128 random bits for a
// given seed, whatever the launch geometry -- a replayed hipGraph only needs the step counter in device memory.
__device__ __forceinline__ uint4 philox4x32(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-8f; }          // [0, 1)
__device__ __forceinline__ float u01_open(uint32_t r) { return (float)((r >> 8) + 1u) * 5.9604644775390625e-8f; }  // (0, 1]

__global__ __launch_bounds__(256) void batch_prologue_kernel(const float* __restrict__ data, int64_t ldd,
                                                             const int64_t* __restrict__ idx, int B, int D, int binarize,
                                                             const int64_t* __restrict__ seed_ctr, float* __restrict__ x_out,
                                                             int64_t ldx, float* __restrict__ eps_out, int zdim,
                                                             int64_t nq_img) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t seed = (uint64_t)seed_ctr[0], step = (uint64_t)seed_ctr[1];
  const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  if (t < nq_img) {                                   // four consecutive elements of the flattened [B x D] batch
    const int64_t e0 = t * 4, n = (int64_t)B * D;
    uint4 r = make_uint4(0, 0, 0, 0);
    if (binarize) r = philox4x32(make_uint4((uint32_t)t, (uint32_t)(t >> 32), (uint32_t)step, (uint32_t)(step >> 32) << 1), key);
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t e = e0 + j;
      if (e >= n) break;
      const int b = (int)(e / D), d = (int)(e - (int64_t)b * D);
      const float p = data[idx[b] * ldd + d];
      x_out[(int64_t)b * ldx + d] = binarize ? (u01(rr[j]) < p ? 1.0f : 0.0f) : p;
    }
    return;
  }
  if (eps_out == nullptr) return;
  const int64_t q = t - nq_img, e0 = q * 4, n = (int64_t)B * zdim;    // four standard normals (two Box-Muller pairs)
  if (e0 >= n) return;
  const uint4 r = philox4x32(make_uint4((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)step, ((uint32_t)(step >> 32) << 1) | 1u), key);
  const float r0 = sqrtf(-2.0f * logf(u01_open(r.x))), r1 = sqrtf(-2.0f * logf(u01_open(r.z)));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
  sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
  const float v[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (e0 + j < n) eps_out[e0 + j] = v[j];
}

// The same head of a step on the uint8-resident store (csrc/evae_dense_u8.hip): pixel = byte / x_div (IEEE division: the very
// float the fp32 dataset holds), same generator streams as above; besides the fp32 batch it writes the batch's BYTES into the
// store's staging rows (255 / 0 when binarised), which is where the first-layer kernels gather them from.
// The control block of a captured step (exemplar-vae_amd/evae/graph.py) handed over by the SAME launch: the host uploads step n's
// block into staging block n & 1; this kernel reads the batch indices and the generator's counter from the staging block the
// device-side parity word names, and further blocks copy that staging block into the control block every later launch of the
// step reads.  The parity is flipped by the step's LAST launch (evae_adam_normgrad_step_stats' toggle): a count of finished
// blocks in here -- one returning device-scope atomic per block on one word -- cost 95 ns per block, 238 us for the 2 500
// blocks of this launch at c2 (tools/prologue_probe.py).  (r06: the hand-over was a device-to-device copy node in front of
// the graph -- ~27 us of the c2 step between the blit and the seam behind it.)
// ... and the step's weight images of the pre-split GEMMs (evae_p6_pack_rows / _cols: weights only, like the split above) as
// further blocks: two launches and one cross-stream join less at the head of the c2 step.
struct PackJob {
  const float* x; const float* x2;
  unsigned char* img;
  long long ld;
  int cols;              // 0: evae_p6_pack_rows(x, x2, R, K, ld, flag = gated); 1: evae_p6_pack_cols(x, x2, Kd = K, R, ld, flag = ones_row, nks)
  int R, K, flag, rows_img, nks, blocks;
};

struct CtlJob {
  const uint4* s[2];
  uint4* ctl;
  size_t n16;
  const int* state;      // [0] parity of the staging block this step reads
  size_t idx_off, seed_off;   // in 8-byte words from the start of a block
  int blocks;
};

__device__ __forceinline__ void batch_prologue_u8_body(const unsigned char* __restrict__ data, int64_t ldd,
                                                       const int64_t* __restrict__ idx, int B, int D, int binarize,
                                                       const int64_t* __restrict__ seed_ctr, float x_div,
                                                       float* __restrict__ x_out, int64_t ldx,
                                                       unsigned char* __restrict__ stage, int64_t lds_,
                                                       float* __restrict__ eps_out, int zdim, int64_t nq_img,
                                                       int pro_blocks, const float* __restrict__ wh,
                                                       const float* __restrict__ wg, int wN, int wK,
                                                       unsigned short* __restrict__ prepared, size_t prep_elems,
                                                       int prep_blocks, const WtJob& j0, const WtJob& j1, const uint4* ctl_src,
                                                       const CtlJob& cj, const PackJob& p0, const PackJob& p1) {
  // blocks past the prologue's: the weight split of the byte-store layer (evae_u8_prepare.h) -- the two jobs are independent
  // and each is a few microseconds of one launch's latency at the head of every training step
  if ((int)blockIdx.x >= pro_blocks + prep_blocks) {
    // ... and past those: the weight transpositions the backward pass's split-bf16 data gradients want (weights only, too)
    __shared__ float tile[32][33];
    const int t = (int)blockIdx.x - pro_blocks - prep_blocks;
    if (t < j0.ntiles) wt_job_tile(j0, t, tile);
    else if (t - j0.ntiles < j1.ntiles) wt_job_tile(j1, t - j0.ntiles, tile);
    else if (t - j0.ntiles - j1.ntiles < p0.blocks + p1.blocks) {      // ... the weight images of the pre-split GEMMs
      const int pb = t - j0.ntiles - j1.ntiles;
      const PackJob& pj = pb < p0.blocks ? p0 : p1;
      const size_t e = (size_t)(pb < p0.blocks ? pb : pb - p0.blocks) * 256 + threadIdx.x;
      if (pj.cols) p6_pack_cols_element(e, pj.x, pj.x2, pj.K, pj.R, pj.ld, pj.flag, pj.rows_img, pj.nks, pj.img);
      else p6_pack_rows_element(e, pj.x, pj.x2, pj.R, pj.K, pj.ld, pj.flag, pj.rows_img, pj.nks, pj.img);
    } else if (ctl_src) {                               // ... and behind those: staging block -> control block
      const int cb = t - j0.ntiles - j1.ntiles - p0.blocks - p1.blocks;
      for (size_t i = (size_t)cb * 256 + threadIdx.x; i < cj.n16; i += (size_t)cj.blocks * 256) cj.ctl[i] = ctl_src[i];
    }
    return;
  }
  if ((int)blockIdx.x >= pro_blocks) {
    const size_t e = (size_t)(blockIdx.x - pro_blocks) * 256 + threadIdx.x;
    if (e < prep_elems) u8_prepare_element(e, wh, wg, wN, wK, u8_prepare_nslab(wK), prepared);
    return;
  }
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t seed = (uint64_t)seed_ctr[0], step = (uint64_t)seed_ctr[1];
  const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  if (t < nq_img) {
    const int64_t e0 = t * 4, n = (int64_t)B * D;
    uint4 r = make_uint4(0, 0, 0, 0);
    if (binarize) r = philox4x32(make_uint4((uint32_t)t, (uint32_t)(t >> 32), (uint32_t)step, (uint32_t)(step >> 32) << 1), key);
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t e = e0 + j;
      if (e >= n) break;
      const int b = (int)(e / D), d = (int)(e - (int64_t)b * D);
      const unsigned char q = data[idx[b] * ldd + d];
      const float p = __fdiv_rn((float)q, x_div);
      const bool one = u01(rr[j]) < p;
      x_out[(int64_t)b * ldx + d] = binarize ? (one ? 1.0f : 0.0f) : p;
      stage[(int64_t)b * lds_ + d] = binarize ? (one ? (unsigned char)255 : (unsigned char)0) : q;
    }
    return;
  }
  if (eps_out == nullptr) return;
  const int64_t q = t - nq_img, e0 = q * 4, n = (int64_t)B * zdim;
  if (e0 >= n) return;
  const uint4 r = philox4x32(make_uint4((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)step, ((uint32_t)(step >> 32) << 1) | 1u), key);
  const float r0 = sqrtf(-2.0f * logf(u01_open(r.x))), r1 = sqrtf(-2.0f * logf(u01_open(r.z)));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
  sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
  const float v[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (e0 + j < n) eps_out[e0 + j] = v[j];
}

__global__ __launch_bounds__(256) void batch_prologue_u8_kernel(const unsigned char* __restrict__ data, int64_t ldd,
                                                                const int64_t* idx, int B, int D, int binarize,
                                                                const int64_t* seed_ctr, float x_div,
                                                                float* __restrict__ x_out, int64_t ldx,
                                                                unsigned char* __restrict__ stage, int64_t lds_,
                                                                float* __restrict__ eps_out, int zdim, int64_t nq_img,
                                                                int pro_blocks, const float* __restrict__ wh,
                                                                const float* __restrict__ wg, int wN, int wK,
                                                                unsigned short* __restrict__ prepared, size_t prep_elems,
                                                                int prep_blocks, WtJob j0, WtJob j1, CtlJob cj, PackJob p0,
                                                                PackJob p1) {
  const uint4* src = nullptr;
  int par = 0;
  if (cj.state) {                                       // the staging block of THIS step: the index list and the counter are read there
    par = cj.state[0] & 1;
    src = cj.s[par];
    idx = (const int64_t*)src + cj.idx_off;
    seed_ctr = (const int64_t*)src + cj.seed_off;
  }
  batch_prologue_u8_body(data, ldd, idx, B, D, binarize, seed_ctr, x_div, x_out, ldx, stage, lds_, eps_out, zdim, nq_img, pro_blocks, wh,
                         wg, wN, wK, prepared, prep_elems, prep_blocks, j0, j1, src, cj, p0, p1);
}

// ELBO assembly on [B] rows in one launch: KL_i = logq_i - logp_i, loss_i = beta*KL_i - RE_i, and the three
// batch means (models/BaseModel.py:71-75).  beta comes from device memory when the step is graph-captured.
__global__ __launch_bounds__(256) void elbo_fwd_kernel(const float* __restrict__ RE, const float* __restrict__ logq,
                                                       const float* __restrict__ logp, const float* __restrict__ logq2,
                                                       const float* __restrict__ logp2,
                                                       const float* __restrict__ beta_dev, float beta_host, int B,
                                                       float* __restrict__ loss, float* __restrict__ KL,
                                                       float* __restrict__ means) {
  __shared__ float red[3][4];
  const float beta = beta_dev ? beta_dev[0] : beta_host;
  float sl = 0.f, sr = 0.f, sk = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    float kl = logq[i] - logp[i];
    if (logq2) kl += logq2[i] - logp2[i];            // two latent layers: (log q1 - log p1) + (log q2 - log p2), models/AbsHModel.py:88-106
    const float l = beta * kl - RE[i];
    KL[i] = kl;
    loss[i] = l;
    sl += l; sr += RE[i]; sk += kl;
  }
  if (means == nullptr) return;
  sl = wave_sum(sl); sr = wave_sum(sr); sk = wave_sum(sk);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sl; red[1][threadIdx.x >> 6] = sr; red[2][threadIdx.x >> 6] = sk; }
  __syncthreads();
  if (threadIdx.x < 3) means[threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]) / (float)B;
}

// Backward coefficients: cRE_i = d/dRE_i, cKL_i = d/dKL_i of the objective given upstream gradients of
// (loss, RE, KL), each NULL, a scalar (n = 1: gradient of the batch mean, spread as g/B) or a [B] vector.
__global__ void elbo_bwd_kernel(const float* __restrict__ dloss, int n_dloss, const float* __restrict__ dRE,
                                int n_dRE, const float* __restrict__ dKL, int n_dKL,
                                const float* __restrict__ beta_dev, float beta_host, int B,
                                float* __restrict__ cRE, float* __restrict__ cKL, float* __restrict__ neg_cKL) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const float beta = beta_dev ? beta_dev[0] : beta_host;
  const float invb = 1.0f / (float)B;
  const float gl = dloss ? (n_dloss == 1 ? dloss[0] * invb : dloss[i]) : 0.f;
  const float gr = dRE ? (n_dRE == 1 ? dRE[0] * invb : dRE[i]) : 0.f;
  const float gk = dKL ? (n_dKL == 1 ? dKL[0] * invb : dKL[i]) : 0.f;
  const float ck = gk + beta * gl;
  cRE[i] = gr - gl;
  cKL[i] = ck;
  neg_cKL[i] = -ck;
}

}  // namespace evae

using namespace evae;

#define ROWS_GRID(B) dim3((unsigned)cdiv((B), LNT / 64))
#define ELT_GRID(n) dim3((unsigned)(((n) + 255) / 256))

extern "C" int evae_reparam_logq_fwd(const float* mu, const float* logvar, const float* eps, int B,
                                     int zdim, float* z, float* logq, evae_stream_t s) {
  EVAE_REQUIRE(B >= 0 && zdim > 0, "reparam_logq_fwd: bad sizes");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(mu && logvar && eps && z, "reparam_logq_fwd: null pointer");
  reparam_logq_fwd_kernel<<<ROWS_GRID(B), LNT, 0, (hipStream_t)s>>>(mu, logvar, eps, B, zdim, z, logq);
  return check_launch("reparam_logq_fwd");
}

extern "C" int evae_reparam_logq_bwd(const float* mu, const float* logvar, const float* eps,
                                     const float* z, const float* dz, const float* dlogq, int B,
                                     int zdim, float* dmu, float* dlogvar, evae_stream_t s) {
  EVAE_REQUIRE(B >= 0 && zdim > 0, "reparam_logq_bwd: bad sizes");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(mu && logvar && eps && z && dmu && dlogvar, "reparam_logq_bwd: null pointer");
  reparam_logq_bwd_kernel<<<ELT_GRID((size_t)B * zdim), 256, 0, (hipStream_t)s>>>(mu, logvar, eps, z, dz, dlogq, B, zdim, dmu, dlogvar);
  return check_launch("reparam_logq_bwd");
}

extern "C" int evae_log_normal_diag_fwd(const float* x, const float* mu, const float* logvar, int B,
                                        int zdim, float* out, evae_stream_t s) {
  EVAE_REQUIRE(B >= 0 && zdim > 0, "log_normal_diag_fwd: bad sizes");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(x && mu && logvar && out, "log_normal_diag_fwd: null pointer");
  log_normal_diag_fwd_kernel<<<ROWS_GRID(B), LNT, 0, (hipStream_t)s>>>(x, mu, logvar, B, zdim, out);
  return check_launch("log_normal_diag_fwd");
}

extern "C" int evae_log_normal_diag_bwd(const float* x, const float* mu, const float* logvar,
                                        const float* dout, int B, int zdim, float* dx, float* dmu,
                                        float* dlogvar, evae_stream_t s) {
  EVAE_REQUIRE(B >= 0 && zdim > 0, "log_normal_diag_bwd: bad sizes");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(x && mu && logvar && dout, "log_normal_diag_bwd: null pointer");
  log_normal_diag_bwd_kernel<<<ELT_GRID((size_t)B * zdim), 256, 0, (hipStream_t)s>>>(x, mu, logvar, dout, B, zdim, dx, dmu, dlogvar);
  return check_launch("log_normal_diag_bwd");
}

extern "C" int evae_log_normal_diag_bwd_hardtanh(const float* x, const float* mu, const float* logvar, const float* lv_pre,
                                                 float lo, float hi, const float* dout, int B, int zdim, float* dx, float* dmu,
                                                 float* dlv_pre, evae_stream_t s) {
  EVAE_REQUIRE(B >= 0 && zdim > 0, "log_normal_diag_bwd_hardtanh: bad sizes");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(x && mu && logvar && lv_pre && dout && dmu && dlv_pre, "log_normal_diag_bwd_hardtanh: null pointer");
  log_normal_diag_bwd_ht_kernel<<<ELT_GRID((size_t)B * zdim), 256, 0, (hipStream_t)s>>>(x, mu, logvar, lv_pre, lo, hi, dout, B, zdim, dx,
                                                                                         dmu, dlv_pre);
  return check_launch("log_normal_diag_bwd_hardtanh");
}

extern "C" int evae_bernoulli_ll_fwd(const float* x, const float* mean, int B, int D, float* out,
                                     evae_stream_t s) {
  EVAE_REQUIRE(B >= 0 && D > 0, "bernoulli_ll_fwd: bad sizes");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(x && mean && out, "bernoulli_ll_fwd: null pointer");
  bernoulli_ll_fwd_kernel<<<ROWS_GRID(B), LNT, 0, (hipStream_t)s>>>(x, mean, B, D, out);
  return check_launch("bernoulli_ll_fwd");
}

extern "C" int evae_bernoulli_ll_bwd(const float* x, const float* mean, const float* dout, int B, int D,
                                     float* dmean, evae_stream_t s) {
  EVAE_REQUIRE(B >= 0 && D > 0, "bernoulli_ll_bwd: bad sizes");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(x && mean && dout && dmean, "bernoulli_ll_bwd: null pointer");
  bernoulli_ll_bwd_kernel<<<ELT_GRID((size_t)B * D), 256, 0, (hipStream_t)s>>>(x, mean, dout, B, D, dmean);
  return check_launch("bernoulli_ll_bwd");
}

extern "C" int evae_log_logistic256_fwd(const float* x, const float* mean, const float* logvar, int lv_scalar, int B, int D,
                                        float* out, evae_stream_t s) {
  EVAE_REQUIRE(B >= 0 && D > 0, "log_logistic256_fwd: bad sizes");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(x && mean && logvar && out, "log_logistic256_fwd: null pointer");
  if (D >= 4096) log_logistic256_fwd_wide_kernel<<<B, LWT, 0, (hipStream_t)s>>>(x, mean, logvar, lv_scalar, B, D, out);
  else log_logistic256_fwd_kernel<<<ROWS_GRID(B), LNT, 0, (hipStream_t)s>>>(x, mean, logvar, lv_scalar, B, D, out);
  return check_launch("log_logistic256_fwd");
}

extern "C" int evae_log_logistic256_bwd(const float* x, const float* mean, const float* logvar, int lv_scalar,
                                        const float* dout, int B, int D, float* dmean, float* dlogvar, float* ws_rows,
                                        evae_stream_t s) {
  EVAE_REQUIRE(B >= 0 && D > 0, "log_logistic256_bwd: bad sizes");
  if (B == 0) {
    if (lv_scalar && dlogvar) (void)hipMemsetAsync(dlogvar, 0, sizeof(float), (hipStream_t)s);
    return EVAE_OK;
  }
  EVAE_REQUIRE(x && mean && logvar && dout, "log_logistic256_bwd: null pointer");
  EVAE_REQUIRE(!(lv_scalar && dlogvar) || ws_rows, "log_logistic256_bwd: a scalar log-variance needs ws_rows [B]");
  float* const dl = lv_scalar ? (dlogvar ? ws_rows : nullptr) : dlogvar;
  if (D >= 4096) log_logistic256_bwd_wide_kernel<<<B, LWT, 0, (hipStream_t)s>>>(x, mean, logvar, lv_scalar, dout, B, D, dmean, dl);
  else log_logistic256_bwd_kernel<<<ROWS_GRID(B), LNT, 0, (hipStream_t)s>>>(x, mean, logvar, lv_scalar, dout, B, D, dmean, dl);
  if (lv_scalar && dlogvar) sum_rows_kernel<<<1, 256, 0, (hipStream_t)s>>>(ws_rows, B, dlogvar);
  return check_launch("log_logistic256_bwd");
}

extern "C" int evae_batch_prologue(const float* data, int64_t ldd, const int64_t* idx, int B, int D, int binarize,
                                   const int64_t* seed_ctr, float* x_out, int64_t ldx, float* eps_out, int zdim,
                                   evae_stream_t s) {
  EVAE_REQUIRE(B >= 0 && D > 0 && zdim >= 0 && ldd >= D && ldx >= D, "batch_prologue: bad sizes");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(data && idx && x_out && seed_ctr, "batch_prologue: null pointer");
  EVAE_REQUIRE(eps_out == nullptr || zdim > 0, "batch_prologue: eps_out needs zdim > 0");
  const int64_t nq_img = ((int64_t)B * D + 3) / 4;
  const int64_t nq_eps = eps_out ? ((int64_t)B * zdim + 3) / 4 : 0;
  batch_prologue_kernel<<<(unsigned)cdiv(nq_img + nq_eps, (int64_t)256), 256, 0, (hipStream_t)s>>>(
      data, ldd, idx, B, D, binarize, seed_ctr, x_out, ldx, eps_out, zdim, nq_img);
  return check_launch("batch_prologue");
}

// One launch for the whole head of a step on the byte store: the batch, optionally the first layer's weight split and the
// backward pass's weight transpositions, optionally the hand-over of the step's control block (CtlJob above).
static int launch_prologue_u8(const char* who, const unsigned char* data, int64_t ldd, const int64_t* idx, int B, int D, int binarize,
                              const int64_t* seed_ctr, float x_div, float* x_out, int64_t ldx, unsigned char* stage, int64_t lds_,
                              float* eps_out, int zdim, const float* wh, const float* wg, int N, int K, void* prepared,
                              size_t prepared_bytes, const evae_wt_job_t* jobs, int njobs, const evae_ctl_job_t* ctl,
                              const evae_p6_pack_job_t* packs, int npacks, evae_stream_t s) {
  EVAE_REQUIRE(B > 0 && D > 0 && zdim >= 0 && ldd >= D && ldx >= D && lds_ >= D && x_div > 0.f, "%s: bad sizes", who);
  EVAE_REQUIRE(data && x_out && stage && ((idx && seed_ctr) || ctl), "%s: null pointer", who);
  EVAE_REQUIRE(eps_out == nullptr || zdim > 0, "%s: eps_out needs zdim > 0", who);
  size_t elems = 0;
  if (wh || wg || prepared) {
    EVAE_REQUIRE(N > 0 && K > 0 && wh && wg && prepared, "%s: bad weight arguments", who);
    elems = u8_prepare_elems(N, K);
    EVAE_REQUIRE(prepared_bytes >= elems * 3 * sizeof(unsigned short), "%s: buffer too small (%zu)", who, prepared_bytes);
  }
  const int64_t nq_img = ((int64_t)B * D + 3) / 4;
  const int64_t nq_eps = eps_out ? ((int64_t)B * zdim + 3) / 4 : 0;
  const unsigned pro = (unsigned)cdiv(nq_img + nq_eps, (int64_t)256);
  const unsigned prep = (unsigned)((elems + 255) / 256);
  EVAE_REQUIRE(njobs >= 0 && njobs <= 2 && (njobs == 0 || jobs != nullptr), "%s: at most two transposition jobs", who);
  WtJob wj[2] = {WtJob{}, WtJob{}};
  for (int i = 0; i < njobs; ++i) {
    const evae_wt_job_t& q = jobs[i];
    EVAE_REQUIRE(q.w1 && q.dst && q.N > 0 && q.K > 0 && q.ldt >= q.N, "%s: bad transposition job", who);
    wj[i].w1 = q.w1; wj[i].w2 = q.w2; wj[i].dst = q.dst; wj[i].N = q.N; wj[i].K = q.K; wj[i].ldt = q.ldt;
    wj[i].tx = cdiv(q.K, 32); wj[i].ty = cdiv(q.ldt, 32); wj[i].ntiles = wj[i].tx * wj[i].ty * (q.w2 ? 2 : 1);
  }
  CtlJob cj{};
  if (ctl) {
    EVAE_REQUIRE(ctl->stage0 && ctl->stage1 && ctl->ctl && ctl->state && ctl->bytes > 0 && ctl->bytes % 16 == 0, "%s: bad control-block job", who);
    EVAE_REQUIRE((((uintptr_t)ctl->stage0 | (uintptr_t)ctl->stage1 | (uintptr_t)ctl->ctl) & 15) == 0, "%s: unaligned control block", who);
    EVAE_REQUIRE((ctl->idx_word + (size_t)B) * 8 <= ctl->bytes && (ctl->seed_word + 2) * 8 <= ctl->bytes, "%s: control-block fields outside the block", who);
    cj.s[0] = (const uint4*)ctl->stage0; cj.s[1] = (const uint4*)ctl->stage1; cj.ctl = (uint4*)ctl->ctl;
    cj.n16 = ctl->bytes / 16; cj.state = (const int*)ctl->state; cj.idx_off = ctl->idx_word; cj.seed_off = ctl->seed_word;
    cj.blocks = (int)std::min<size_t>(128, (cj.n16 + 255) / 256);
  }
  EVAE_REQUIRE(npacks >= 0 && npacks <= 2 && (npacks == 0 || packs != nullptr), "%s: at most two image jobs", who);
  PackJob pk[2] = {PackJob{}, PackJob{}};
  for (int i = 0; i < npacks; ++i) {
    const evae_p6_pack_job_t& q = packs[i];
    EVAE_REQUIRE(q.x && q.img && q.R > 0 && q.K > 0, "%s: bad image job", who);
    PackJob& o = pk[i];
    o.x = q.x; o.x2 = q.x2; o.img = (unsigned char*)q.img; o.ld = q.ld; o.cols = q.cols ? 1 : 0; o.R = q.R; o.K = q.K; o.flag = q.flag;
    if (o.cols) {          // evae_p6_pack_cols' checks and geometry
      EVAE_REQUIRE(q.ld >= q.R && q.nks >= p6_nks(q.x2 ? 2 * q.K : q.K), "%s: bad image job (columns)", who);
      o.rows_img = cdiv(std::max(q.R, q.flag + 1), 128) * 128; o.nks = q.nks;
    } else {               // evae_p6_pack_rows'
      EVAE_REQUIRE(q.ld >= q.K && (!q.flag || q.x2), "%s: bad image job (rows)", who);
      o.rows_img = q.flag ? cdiv(q.R, 64) * 128 : cdiv(q.R, 128) * 128; o.nks = p6_nks(q.K);
    }
    EVAE_REQUIRE(q.img_bytes >= p6_image_bytes(o.rows_img, o.nks), "%s: image buffer too small (%zu)", who, q.img_bytes);
    o.blocks = (int)(((size_t)o.rows_img * o.nks * 2 + 255) / 256);
  }
  batch_prologue_u8_kernel<<<pro + prep + (unsigned)(wj[0].ntiles + wj[1].ntiles + pk[0].blocks + pk[1].blocks + cj.blocks), 256, 0,
                             (hipStream_t)s>>>(data, ldd, idx, B, D, binarize, seed_ctr, x_div, x_out, ldx, stage, lds_, eps_out, zdim, nq_img,
                                               (int)pro, wh, wg, N, K, (unsigned short*)prepared, elems, (int)prep, wj[0], wj[1], cj, pk[0], pk[1]);
  return check_launch(who);
}

extern "C" int evae_batch_prologue_u8(const unsigned char* data, int64_t ldd, const int64_t* idx, int B, int D, int binarize,
                                      const int64_t* seed_ctr, float x_div, float* x_out, int64_t ldx, unsigned char* stage,
                                      int64_t lds_, float* eps_out, int zdim, evae_stream_t s) {
  EVAE_REQUIRE(B >= 0, "batch_prologue_u8: bad sizes");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(idx && seed_ctr, "batch_prologue_u8: null pointer");
  return launch_prologue_u8("batch_prologue_u8", data, ldd, idx, B, D, binarize, seed_ctr, x_div, x_out, ldx, stage, lds_, eps_out, zdim,
                            nullptr, nullptr, 0, 0, nullptr, 0, nullptr, 0, nullptr, nullptr, 0, s);
}

// The same launch also splits the first layer's weights into the byte kernels' bf16 tile images (evae_dense_u8_prepare): the
// head of a training step is then one launch instead of two.
extern "C" int evae_batch_prologue_u8_prepare(const unsigned char* data, int64_t ldd, const int64_t* idx, int B, int D, int binarize,
                                              const int64_t* seed_ctr, float x_div, float* x_out, int64_t ldx,
                                              unsigned char* stage, int64_t lds_, float* eps_out, int zdim, const float* wh,
                                              const float* wg, int N, int K, void* prepared, size_t prepared_bytes,
                                              const evae_wt_job_t* jobs, int njobs, evae_stream_t s) {
  EVAE_REQUIRE(idx && seed_ctr && wh && wg && prepared, "batch_prologue_u8_prepare: null pointer");
  return launch_prologue_u8("batch_prologue_u8_prepare", data, ldd, idx, B, D, binarize, seed_ctr, x_div, x_out, ldx, stage, lds_,
                            eps_out, zdim, wh, wg, N, K, prepared, prepared_bytes, jobs, njobs, nullptr, nullptr, 0, s);
}

// ... and the step's control block (wh / wg / prepared NULL: no weight split)
extern "C" int evae_batch_prologue_u8_step(const unsigned char* data, int64_t ldd, int B, int D, int binarize, float x_div,
                                           float* x_out, int64_t ldx, unsigned char* stage, int64_t lds_, float* eps_out, int zdim,
                                           const float* wh, const float* wg, int N, int K, void* prepared, size_t prepared_bytes,
                                           const evae_wt_job_t* jobs, int njobs, const evae_ctl_job_t* ctl,
                                           const evae_p6_pack_job_t* packs, int npacks, evae_stream_t s) {
  EVAE_REQUIRE(ctl, "batch_prologue_u8_step: no control-block job");
  return launch_prologue_u8("batch_prologue_u8_step", data, ldd, nullptr, B, D, binarize, nullptr, x_div, x_out, ldx, stage, lds_,
                            eps_out, zdim, wh, wg, N, K, prepared, prepared_bytes, jobs, njobs, ctl, packs, npacks, s);
}

extern "C" int evae_elu_fwd(const float* x, size_t n, float* out, evae_stream_t s) {
  if (n == 0) return EVAE_OK;
  EVAE_REQUIRE(x && out && (((uintptr_t)x | (uintptr_t)out) & 15) == 0, "elu_fwd: null or unaligned pointer");
  const size_t nthreads = (n + 3) / 4;
  elu_fwd_kernel<<<(unsigned)((nthreads + 255) / 256), 256, 0, (hipStream_t)s>>>(x, n, out);
  return check_launch("elu_fwd");
}

extern "C" int evae_bernoulli_sigmoid_bwd(const float* x, const float* mean, const float* dout, int B, int D,
                                          float* dpre, evae_stream_t s) {
  EVAE_REQUIRE(B >= 0 && D > 0, "bernoulli_sigmoid_bwd: bad sizes");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(x && mean && dout && dpre, "bernoulli_sigmoid_bwd: null pointer");
  bernoulli_sigmoid_bwd_kernel<<<ELT_GRID((size_t)B * D), 256, 0, (hipStream_t)s>>>(x, mean, dout, B, D, dpre);
  return check_launch("bernoulli_sigmoid_bwd");
}

extern "C" int evae_reparam_logq_bwd_hardtanh(const float* mu, const float* logvar, const float* eps, const float* z,
                                              const float* dz, const float* dz2, const float* dlogq,
                                              const float* lv_pre, float lo, float hi, int B, int zdim, float* dmu,
                                              float* dlv_pre, evae_stream_t s) {
  EVAE_REQUIRE(B >= 0 && zdim > 0, "reparam_logq_bwd_hardtanh: bad sizes");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(mu && logvar && eps && z && lv_pre && dmu && dlv_pre, "reparam_logq_bwd_hardtanh: null pointer");
  reparam_logq_bwd_ht_kernel<<<ELT_GRID((size_t)B * zdim), 256, 0, (hipStream_t)s>>>(mu, logvar, eps, z, dz, dz2, dlogq, lv_pre,
                                                                                      lo, hi, B, zdim, dmu, dlv_pre);
  return check_launch("reparam_logq_bwd_hardtanh");
}

extern "C" int evae_reparam_logq_bwd_hardtanh_tail(const float* mu, const float* logvar, const float* eps, const float* z,
                                                   const float* dz, const float* dz2, const float* dlogq, const float* lv_pre,
                                                   float lo, float hi, int B, int zdim, float* dmu, float* dlv_pre,
                                                   const float* logp, const float* RE, const float* logq, const float* beta_dev,
                                                   float beta_host, float* loss, float* KL, float* means, const float* sum_src,
                                                   int sum_n, float* sum_dst, evae_stream_t s) {
  EVAE_REQUIRE(B > 0 && zdim > 0 && sum_n >= 0, "reparam_logq_bwd_hardtanh_tail: bad sizes");
  EVAE_REQUIRE(mu && logvar && eps && z && lv_pre && dmu && dlv_pre, "reparam_logq_bwd_hardtanh_tail: null pointer");
  EVAE_REQUIRE(loss == nullptr || (logp && RE && logq && KL), "reparam_logq_bwd_hardtanh_tail: incomplete ELBO arguments");
  EVAE_REQUIRE(sum_dst == nullptr || sum_src, "reparam_logq_bwd_hardtanh_tail: incomplete sum arguments");
  const unsigned eb = (unsigned)(((size_t)B * zdim + 255) / 256);
  const unsigned tail = (loss || sum_dst) ? 1u : 0u;
  reparam_logq_bwd_ht_tail_kernel<<<eb + tail, 256, 0, (hipStream_t)s>>>(mu, logvar, eps, z, dz, dz2, dlogq, lv_pre, lo, hi, B, zdim, dmu,
                                                                         dlv_pre, (int)eb, logp, RE, logq, beta_dev, beta_host, loss, KL,
                                                                         means, sum_src, sum_n, sum_dst);
  return check_launch("reparam_logq_bwd_hardtanh_tail");
}

extern "C" int evae_bernoulli_unit_step(const float* x, const float* mean, int B, int D, const float* beta_dev, float beta_host,
                                        float* RE, float* cRE, float* cKL, float* neg_cKL, float* dpre, evae_stream_t s) {
  EVAE_REQUIRE(B > 0 && D > 0, "bernoulli_unit_step: bad sizes");
  EVAE_REQUIRE(x && mean && RE && cRE && cKL && neg_cKL && dpre, "bernoulli_unit_step: null pointer");
  bernoulli_unit_step_kernel<<<ROWS_GRID(B), LNT, 0, (hipStream_t)s>>>(x, mean, B, D, beta_dev, beta_host, RE, cRE, cKL, neg_cKL, dpre);
  return check_launch("bernoulli_unit_step");
}

extern "C" int evae_elbo_fwd(const float* RE, const float* logq, const float* logp, const float* beta_dev,
                             float beta_host, int B, float* loss, float* KL, float* means, evae_stream_t s) {
  EVAE_REQUIRE(B >= 0, "elbo_fwd: bad size");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(RE && logq && logp && loss && KL, "elbo_fwd: null pointer");
  elbo_fwd_kernel<<<1, 256, 0, (hipStream_t)s>>>(RE, logq, logp, nullptr, nullptr, beta_dev, beta_host, B, loss, KL, means);
  return check_launch("elbo_fwd");
}

extern "C" int evae_elbo2_fwd(const float* RE, const float* logq1, const float* logp1, const float* logq2, const float* logp2,
                              const float* beta_dev, float beta_host, int B, float* loss, float* KL, float* means,
                              evae_stream_t s) {
  EVAE_REQUIRE(B >= 0, "elbo2_fwd: bad size");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(RE && logq1 && logp1 && logq2 && logp2 && loss && KL, "elbo2_fwd: null pointer");
  elbo_fwd_kernel<<<1, 256, 0, (hipStream_t)s>>>(RE, logq1, logp1, logq2, logp2, beta_dev, beta_host, B, loss, KL, means);
  return check_launch("elbo2_fwd");
}

__global__ void step_stats_kernel(const float* loss, const float* re, const float* kl, float* step3, float* totals3) {
  if (threadIdx.x < 3) {
    const float v = threadIdx.x == 0 ? loss[0] : (threadIdx.x == 1 ? -re[0] : kl[0]);
    step3[threadIdx.x] = v;
    if (totals3) totals3[threadIdx.x] += v;
  }
}

extern "C" int evae_step_stats_add(const float* loss, const float* re, const float* kl, float* step3, float* totals3,
                                   evae_stream_t s) {
  EVAE_REQUIRE(loss && re && kl && step3, "step_stats_add: null pointer");
  step_stats_kernel<<<1, 64, 0, (hipStream_t)s>>>(loss, re, kl, step3, totals3);
  return check_launch("step_stats_add");
}

// The two scalar <-> row moves of the exemplar prior's one log-variance (models/BaseModel.py:25-26 prior_log_variance [1],
// :101 `center_log_variance[0, :]`): the row the prior kernels read = the value repeated zdim times, and the value's gradient =
// the sum of the row's.  One wave each (they used to be an ATen copy and an ATen reduction inside the captured step).
__global__ void broadcast_scalar_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
  const float v = src[0];
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = v;
}
__global__ void sum_small_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += x[i];          // fixed order per lane, fixed butterfly: deterministic
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = s;
}
extern "C" int evae_broadcast_scalar(const float* src, float* dst, int n, evae_stream_t s) {
  EVAE_REQUIRE(src && dst && n >= 0, "broadcast_scalar: bad arguments");
  if (n == 0) return EVAE_OK;
  broadcast_scalar_kernel<<<1, 64, 0, (hipStream_t)s>>>(src, dst, n);
  return check_launch("broadcast_scalar_kernel");
}
extern "C" int evae_sum_small(const float* x, int n, float* out, evae_stream_t s) {
  EVAE_REQUIRE(x && out && n >= 0, "sum_small: bad arguments");
  sum_small_kernel<<<1, 64, 0, (hipStream_t)s>>>(x, n, out);
  return check_launch("sum_small_kernel");
}

extern "C" int evae_elbo_bwd(const float* dloss, int n_dloss, const float* dRE, int n_dRE, const float* dKL,
                             int n_dKL, const float* beta_dev, float beta_host, int B, float* cRE, float* cKL,
                             float* neg_cKL, evae_stream_t s) {
  EVAE_REQUIRE(B >= 0, "elbo_bwd: bad size");
  if (B == 0) return EVAE_OK;
  EVAE_REQUIRE(cRE && cKL && neg_cKL, "elbo_bwd: null pointer");
  elbo_bwd_kernel<<<ELT_GRID((size_t)B), 256, 0, (hipStream_t)s>>>(dloss, n_dloss, dRE, n_dRE, dKL, n_dKL, beta_dev,
                                                                   beta_host, B, cRE, cKL, neg_cKL);
  return check_launch("elbo_bwd");
}
