// AdamNormGrad (utils/optimizer.py:32-80) as two multi-tensor launches:
//   1. per-tensor partial sums of squares (fixed block->chunk map, so the reduction order -- and the
//      result -- is deterministic),
//   2. per tensor: norm = sqrt(sum of partials); g = grad/(norm + 1e-7) (+ wd * p); Adam moments;
//      p -= step_size * m / (sqrt(v) + eps),  step_size = lr * sqrt(1 - b2^t) / (1 - b1^t).
#include "evae_common.h"

namespace evae {

constexpr int ANB = 128;       // partial-sum slots per tensor
constexpr int ACHUNK = 2048;   // elements per block: 256 threads x 2 float4 (tensors above ANB * ACHUNK take more)

// elements per block for a tensor: a multiple of 1024 (float4 x 256 threads), at most ANB blocks per tensor
__device__ __forceinline__ int64_t adam_chunk(int64_t numel) {
  int64_t c = (numel + ANB - 1) / ANB;
  c = (c + 1023) / 1024 * 1024;
  return c < ACHUNK ? ACHUNK : c;
}

__global__ __launch_bounds__(256) void adam_sumsq_kernel(const evae_adam_tensor_t* __restrict__ ts,
                                                         float* __restrict__ part /* [nt][ANB] */) {
  __shared__ double red[4];
  const evae_adam_tensor_t t = ts[blockIdx.y];
  const int64_t chunk = adam_chunk(t.numel);
  const int64_t beg = (int64_t)blockIdx.x * chunk;
  if (beg >= t.numel) return;                    // this tensor needs fewer blocks than the grid is wide
  int64_t end = beg + chunk;
  if (end > t.numel) end = t.numel;
  double s = 0.0;
  if ((reinterpret_cast<uintptr_t>(t.grad) & 15) == 0) {
    const int64_t nv = (end - beg) >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(t.grad + beg);   // beg is a multiple of 1024
    for (int64_t i = threadIdx.x; i < nv; i += 256) {
      const float4 g = g4[i];
      s += (double)g.x * (double)g.x + (double)g.y * (double)g.y + (double)g.z * (double)g.z + (double)g.w * (double)g.w;
    }
    for (int64_t i = beg + (nv << 2) + threadIdx.x; i < end; i += 256) {
      const float g = t.grad[i];
      s += (double)g * (double)g;
    }
  } else {
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
      const float g = t.grad[i];
      s += (double)g * (double)g;
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.y * ANB + blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}

__device__ __forceinline__ void adam_one(float gr, float& p, float& m, float& v, float inv, float step_size, float beta1,
                                         float omb1, float beta2, float omb2, float eps, float weight_decay) {
  float g = gr * inv;
  if (weight_decay != 0.f) g += weight_decay * p;
  m = m * beta1 + omb1 * g;
  v = v * beta2 + omb2 * g * g;
  p = p - step_size * (m / (sqrtf(v) + eps));
}

__global__ __launch_bounds__(256) void adam_step_kernel(const evae_adam_tensor_t* __restrict__ ts,
                                                        const float* __restrict__ part, float step_size_host,
                                                        const float* __restrict__ step_size_dev,
                                                        float beta1, float omb1, float beta2,
                                                        float omb2, float eps, float weight_decay,
                                                        const float* __restrict__ st_loss, const float* __restrict__ st_re,
                                                        const float* __restrict__ st_kl, float* __restrict__ st_step3,
                                                        float* __restrict__ st_tot3, int* __restrict__ st_toggle) {
  // the step's statistics (evae_step_stats_add) in this launch, the last of a captured training step: (loss, -RE, KL) of the
  // step and their running sums -- one launch less at the tail of every step
  if (st_step3 != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 3) {
    const float v = threadIdx.x == 0 ? st_loss[0] : (threadIdx.x == 1 ? -st_re[0] : st_kl[0]);
    st_step3[threadIdx.x] = v;
    if (st_tot3) st_tot3[threadIdx.x] += v;
    // ... and the parity of the control block's staging blocks (evae_batch_prologue_u8_step reads it at the head of the next step)
    if (st_toggle && threadIdx.x == 0) st_toggle[0] ^= 1;
  }
  const evae_adam_tensor_t t = ts[blockIdx.y];
  const int64_t chunk = adam_chunk(t.numel);
  const int64_t beg = (int64_t)blockIdx.x * chunk;
  if (beg >= t.numel) return;
  int64_t end = beg + chunk;
  if (end > t.numel) end = t.numel;
  const float step_size = step_size_dev ? step_size_dev[0] : step_size_host;
  const int nparts = (int)((t.numel + chunk - 1) / chunk);
  // the tensor's sum of squares from its <= ANB = 128 partials: two per lane and a wave butterfly -- the same order in every
  // wave of every block, so every block sees the same norm (a serial walk over the partials was a 115-long dependent chain in
  // front of 2048 elements of work)
  const int lane = threadIdx.x & 63;
  double tot = (lane < nparts ? (double)part[blockIdx.y * ANB + lane] : 0.0) +
               (lane + 64 < nparts ? (double)part[blockIdx.y * ANB + lane + 64] : 0.0);
  tot = wave_sum(tot);
  const float inv = 1.0f / ((float)sqrt(tot) + 1e-7f);
  const bool vec = ((reinterpret_cast<uintptr_t>(t.grad) | reinterpret_cast<uintptr_t>(t.param) |
                     reinterpret_cast<uintptr_t>(t.exp_avg) | reinterpret_cast<uintptr_t>(t.exp_avg_sq)) & 15) == 0;
  int64_t done = beg;
  if (vec) {
    const int64_t nv = (end - beg) >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(t.grad + beg);
    float4* p4 = reinterpret_cast<float4*>(t.param + beg);
    float4* m4 = reinterpret_cast<float4*>(t.exp_avg + beg);
    float4* v4 = reinterpret_cast<float4*>(t.exp_avg_sq + beg);
    for (int64_t i = threadIdx.x; i < nv; i += 256) {
      const float4 g = g4[i];
      float4 p = p4[i], m = m4[i], v = v4[i];
      adam_one(g.x, p.x, m.x, v.x, inv, step_size, beta1, omb1, beta2, omb2, eps, weight_decay);
      adam_one(g.y, p.y, m.y, v.y, inv, step_size, beta1, omb1, beta2, omb2, eps, weight_decay);
      adam_one(g.z, p.z, m.z, v.z, inv, step_size, beta1, omb1, beta2, omb2, eps, weight_decay);
      adam_one(g.w, p.w, m.w, v.w, inv, step_size, beta1, omb1, beta2, omb2, eps, weight_decay);
      m4[i] = m; v4[i] = v; p4[i] = p;
    }
    done = beg + (nv << 2);
  }
  for (int64_t i = done + threadIdx.x; i < end; i += 256) {
    float p = t.param[i], m = t.exp_avg[i], v = t.exp_avg_sq[i];
    adam_one(t.grad[i], p, m, v, inv, step_size, beta1, omb1, beta2, omb2, eps, weight_decay);
    t.exp_avg[i] = m; t.exp_avg_sq[i] = v; t.param[i] = p;
  }
}

}  // namespace evae

using namespace evae;

extern "C" size_t evae_adam_normgrad_workspace_bytes(int n_tensors) {
  return align_up((size_t)(n_tensors > 0 ? n_tensors : 1) * ANB * sizeof(float), 256);
}

static int adam_normgrad_core(const evae_adam_tensor_t* tensors, int n_tensors,
                              int64_t max_numel, int step, double lr, double beta1, double beta2,
                              double eps, double weight_decay, const float* step_size_dev,
                              void* ws, size_t ws_bytes, const float* st_loss, const float* st_re, const float* st_kl,
                              float* st_step3, float* st_tot3, int* st_toggle, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(n_tensors >= 0 && step >= 1, "adam_normgrad_step: bad arguments");
  if (n_tensors == 0) return EVAE_OK;
  EVAE_REQUIRE(tensors != nullptr, "adam_normgrad_step: null tensor table");
  if (ws == nullptr || ws_bytes < evae_adam_normgrad_workspace_bytes(n_tensors)) {
    set_error("adam_normgrad_step: workspace too small");
    return EVAE_EWORKSPACE;
  }
  const double bc1 = 1.0 - pow(beta1, step);
  const double bc2 = 1.0 - pow(beta2, step);
  const float step_size = (float)(lr * sqrt(bc2) / bc1);
  float* part = (float*)ws;
  // grid.x = an upper bound on the blocks any tensor needs (every tensor's chunk is >= ACHUNK and it never needs more
  // than ANB blocks); max_numel <= 0: unknown, assume the worst
  int gx = ANB;
  if (max_numel > 0 && (max_numel + ACHUNK - 1) / ACHUNK < ANB) gx = (int)((max_numel + ACHUNK - 1) / ACHUNK);
  adam_sumsq_kernel<<<dim3(gx, n_tensors), 256, 0, stream>>>(tensors, part);
  int rc = check_launch("adam_sumsq");
  if (rc) return rc;
  adam_step_kernel<<<dim3(gx, n_tensors), 256, 0, stream>>>(tensors, part, step_size, step_size_dev, (float)beta1, (float)(1.0 - beta1),
                                                           (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay,
                                                           st_loss, st_re, st_kl, st_step3, st_tot3, st_toggle);
  return check_launch("adam_step");
}

extern "C" int evae_adam_normgrad_step(const evae_adam_tensor_t* tensors, int n_tensors,
                                       int64_t max_numel, int step, double lr, double beta1, double beta2,
                                       double eps, double weight_decay, const float* step_size_dev,
                                       void* ws, size_t ws_bytes, evae_stream_t stream_) {
  return adam_normgrad_core(tensors, n_tensors, max_numel, step, lr, beta1, beta2, eps, weight_decay, step_size_dev, ws, ws_bytes,
                            nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream_);
}

extern "C" int evae_adam_normgrad_step_stats(const evae_adam_tensor_t* tensors, int n_tensors,
                                             int64_t max_numel, int step, double lr, double beta1, double beta2,
                                             double eps, double weight_decay, const float* step_size_dev,
                                             void* ws, size_t ws_bytes, const float* loss, const float* re, const float* kl,
                                             float* step3, float* totals3, int* toggle, evae_stream_t stream_) {
  EVAE_REQUIRE(n_tensors > 0 && loss && re && kl && step3, "adam_normgrad_step_stats: null pointer or empty tensor table");
  return adam_normgrad_core(tensors, n_tensors, max_numel, step, lr, beta1, beta2, eps, weight_decay, step_size_dev, ws, ws_bytes,
                            loss, re, kl, step3, totals3, toggle, stream_);
}
