// Weight normalisation of a SET of convolution filters in one launch: w_i = v_i * (g_i / ||v_i||), the norm over everything but the
// output channel (torch.nn.utils.weight_norm, dim 0 -- how reference models/fully_conv.py:18,41-58 wraps every convolution), and its
// gradient (dv_i, dg_i from dw_i).  fully_conv has 29 such filters per network pass; one tiny launch each (and one more in the
// backward pass) is 2 x 46 launches a training step.
#include "evae_common.h"

namespace evae {

constexpr int WN_MAX = 32;
struct WnSet {
  const float* v[WN_MAX];
  const float* g[WN_MAX];
  float* w[WN_MAX];               // forward
  const float* dw[WN_MAX];        // backward
  float* dv[WN_MAX];
  float* dg[WN_MAX];
  int row0[WN_MAX + 1];           // first block (= output channel) of filter i
  int cols[WN_MAX];
  int n;
};

// block sum in a fixed order: lane-strided partials, wave butterflies, the four wave sums in wave order.  In double: a filter row is
// <= a few thousand elements, and the correctly rounded norm is what the reference's v * (g / norm) sees to within its own last bit
// (a 24-block residual network amplifies a last-bit difference in every weight to 1e-4 in its gradients)
__device__ __forceinline__ double wn_block_sum(double a, double* red) {
  a = wave_sum(a);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

template <bool BWD>
__global__ __launch_bounds__(256) void weight_norm_set_kernel(const WnSet s) {
  __shared__ double red[4];
  int i = 0;
  while (i + 1 < s.n && (int)blockIdx.x >= s.row0[i + 1]) ++i;
  const int r = (int)blockIdx.x - s.row0[i], cols = s.cols[i];
  const float* v = s.v[i] + (size_t)r * cols;
  double ss = 0.0, dot = 0.0;
  const float* dw = BWD ? s.dw[i] + (size_t)r * cols : nullptr;
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float x = v[c];
    ss += (double)x * x;
    if (BWD) dot += (double)x * dw[c];
  }
  ss = wn_block_sum(ss, red);
  const float norm = (float)sqrt(ss), gv = s.g[i][r];
  if (!BWD) {
    const float sc = gv / norm;
    float* w = s.w[i] + (size_t)r * cols;
    for (int c = threadIdx.x; c < cols; c += 256) w[c] = v[c] * sc;
  } else {
    dot = wn_block_sum(dot, red);
    const float sc = gv / norm, k = (float)(dot / ss);
    float* dv = s.dv[i] + (size_t)r * cols;
    for (int c = threadIdx.x; c < cols; c += 256) dv[c] = sc * (dw[c] - v[c] * k);
    if (threadIdx.x == 0) s.dg[i][r] = (float)(dot / sqrt(ss));
  }
}

static int wn_fill(WnSet& s, int n, const void* const* v, const void* const* g, const int* rows, const int* cols) {
  s.n = n; s.row0[0] = 0;
  for (int i = 0; i < n; ++i) {
    EVAE_REQUIRE(v[i] && g[i] && rows[i] > 0 && cols[i] > 0, "weight_norm_set: null pointer / empty filter");
    s.v[i] = (const float*)v[i]; s.g[i] = (const float*)g[i]; s.cols[i] = cols[i]; s.row0[i + 1] = s.row0[i] + rows[i];
  }
  return EVAE_OK;
}

}  // namespace evae
using namespace evae;

extern "C" int evae_weight_norm_set_fwd(int n, const void* const* v, const void* const* g, void* const* w, const int* rows, const int* cols,
                                        evae_stream_t stream) {
  EVAE_REQUIRE(n >= 0 && n <= WN_MAX, "weight_norm_set_fwd: 0 .. 32 filters a call");
  if (n == 0) return EVAE_OK;
  EVAE_REQUIRE(v && g && w && rows && cols, "weight_norm_set_fwd: null pointer");
  WnSet s = {};
  int rc = wn_fill(s, n, v, g, rows, cols);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) { EVAE_REQUIRE(w[i], "weight_norm_set_fwd: null output"); s.w[i] = (float*)w[i]; }
  weight_norm_set_kernel<false><<<s.row0[n], 256, 0, (hipStream_t)stream>>>(s);
  return check_launch("weight_norm_set_fwd");
}

extern "C" int evae_weight_norm_set_bwd(int n, const void* const* v, const void* const* g, const void* const* dw, void* const* dv, void* const* dg,
                                        const int* rows, const int* cols, evae_stream_t stream) {
  EVAE_REQUIRE(n >= 0 && n <= WN_MAX, "weight_norm_set_bwd: 0 .. 32 filters a call");
  if (n == 0) return EVAE_OK;
  EVAE_REQUIRE(v && g && dw && dv && dg && rows && cols, "weight_norm_set_bwd: null pointer");
  WnSet s = {};
  int rc = wn_fill(s, n, v, g, rows, cols);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) {
    EVAE_REQUIRE(dw[i] && dv[i] && dg[i], "weight_norm_set_bwd: null pointer");
    s.dw[i] = (const float*)dw[i]; s.dv[i] = (float*)dv[i]; s.dg[i] = (float*)dg[i];
  }
  weight_norm_set_kernel<true><<<s.row0[n], 256, 0, (hipStream_t)stream>>>(s);
  return check_launch("weight_norm_set_bwd");
}
