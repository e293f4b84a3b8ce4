// AdamNormGrad (utils/optimizer.py:32-80) as two multi-tensor launches:
//   1. per-tensor partial sums of squares (fixed block->chunk map, so the reduction order -- and the
//      result -- is deterministic),
//   2. per tensor: norm = sqrt(sum of partials); g = grad/(norm + 1e-7) (+ wd * p); Adam moments;
//      p -= step_size * m / (sqrt(v) + eps),  step_size = lr * sqrt(1 - b2^t) / (1 - b1^t).
#include "evae_common.h"

namespace evae {

constexpr int ANB = 32;  // blocks per tensor

__global__ __launch_bounds__(256) void adam_sumsq_kernel(const evae_adam_tensor_t* __restrict__ ts,
                                                         float* __restrict__ part /* [nt][ANB] */) {
  __shared__ double red[4];
  const evae_adam_tensor_t t = ts[blockIdx.y];
  const int64_t chunk = (t.numel + ANB - 1) / ANB;
  const int64_t beg = (int64_t)blockIdx.x * chunk;
  int64_t end = beg + chunk;
  if (end > t.numel) end = t.numel;
  double s = 0.0;
  for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
    const float g = t.grad[i];
    s += (double)g * (double)g;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.y * ANB + blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adam_step_kernel(const evae_adam_tensor_t* __restrict__ ts,
                                                        const float* __restrict__ part, float step_size_host,
                                                        const float* __restrict__ step_size_dev,
                                                        float beta1, float omb1, float beta2,
                                                        float omb2, float eps, float weight_decay) {
  const evae_adam_tensor_t t = ts[blockIdx.y];
  const float step_size = step_size_dev ? step_size_dev[0] : step_size_host;
  double tot = 0.0;
  for (int i = 0; i < ANB; ++i) tot += (double)part[blockIdx.y * ANB + i];
  const float inv = 1.0f / ((float)sqrt(tot) + 1e-7f);
  const int64_t chunk = (t.numel + ANB - 1) / ANB;
  const int64_t beg = (int64_t)blockIdx.x * chunk;
  int64_t end = beg + chunk;
  if (end > t.numel) end = t.numel;
  for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
    float g = t.grad[i] * inv;
    float p = t.param[i];
    if (weight_decay != 0.f) g += weight_decay * p;
    const float m = t.exp_avg[i] * beta1 + omb1 * g;
    const float v = t.exp_avg_sq[i] * beta2 + omb2 * g * g;
    t.exp_avg[i] = m;
    t.exp_avg_sq[i] = v;
    t.param[i] = p - step_size * (m / (sqrtf(v) + eps));
  }
}

}  // namespace evae

using namespace evae;

extern "C" size_t evae_adam_normgrad_workspace_bytes(int n_tensors) {
  return align_up((size_t)(n_tensors > 0 ? n_tensors : 1) * ANB * sizeof(float), 256);
}

extern "C" int evae_adam_normgrad_step(const evae_adam_tensor_t* tensors, int n_tensors,
                                       int64_t max_numel, int step, double lr, double beta1, double beta2,
                                       double eps, double weight_decay, const float* step_size_dev,
                                       void* ws, size_t ws_bytes, evae_stream_t stream_) {
  (void)max_numel;
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
  adam_sumsq_kernel<<<dim3(ANB, n_tensors), 256, 0, stream>>>(tensors, part);
  int rc = check_launch("adam_sumsq");
  if (rc) return rc;
  adam_step_kernel<<<dim3(ANB, n_tensors), 256, 0, stream>>>(tensors, part, step_size, step_size_dev, (float)beta1, (float)(1.0 - beta1),
                                                           (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay);
  return check_launch("adam_step");
}
