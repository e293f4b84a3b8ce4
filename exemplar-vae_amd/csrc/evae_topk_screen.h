// fp32 screening + exact re-ranking path of evae_pairdist_topk (evae_topk_screen.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace evae {
// 0 when the screening path does not apply to this problem (the caller then runs the exact scan kernel)
size_t topk_screen_workspace_bytes(int B, int N, int zdim, int k);
// EVAE_OK / error code; *handled = 0 when the problem does not qualify (nothing was launched)
int topk_screen(const float* q, int B, const float* cache, int N, int zdim, int k, unsigned flags, int64_t index_base,
                int64_t* out_idx, float* out_val, void* ws, size_t ws_bytes, hipStream_t stream, int* handled);
}  // namespace evae
