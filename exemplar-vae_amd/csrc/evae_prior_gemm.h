// Exemplar prior at large latent sizes on the GEMM kernel family: interface between evae_prior.hip (entry points,
// VALU kernels) and evae_prior_gemm.hip (see there).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace evae {

struct PriorGemmFwdLayout {      // byte offsets into the caller's workspace
  size_t flag, mu, Zs, zn, Cs, cn, pm, ps, pn, total;
  int zp, ldp, tiles_m, stream;      // stream: the partial rows come from prior_x6_lse_kernel (one per split)
};
struct PriorGemmBwdLayout {
  size_t flag, mu, Zs, zn, Cs, cn, P, T, U, rs, dvp, ws_data, ws_weight, total;
  size_t ws_data_bytes, ws_weight_bytes;
  int zp, ldz, ldp, nblk;
};

bool prior_gemm_applies(int B, int C, int zdim);
// ns_valu: partial rows the direct-difference fallback (prior_fwd_kernel) writes into the same partial planes
PriorGemmFwdLayout prior_gemm_fwd_layout(int B, int C, int zdim, int ns_valu, bool stream = false);
// evaluator-sized unmasked calls at z <= 48 (IWAE): the streaming split-bf16 kernel of evae_prior_gemm.hip
bool prior_stream_applies(int B, int C, int zdim, bool masked);
// enqueues mean / prep / GEMM; the caller then enqueues the flag-gated fallback and prior_gemm_merge
int prior_gemm_fwd(const float* z, int B, const float* centres, int C, int zdim, const float* log_var, const int64_t* z_idx,
                   const int64_t* c_idx, float norm_limit, char* ws, const PriorGemmFwdLayout& L, hipStream_t stream);
void prior_gemm_merge(const char* ws, const PriorGemmFwdLayout& L, int B, int ns_valu, float* om, float* os, float* on,
                      hipStream_t stream);
PriorGemmBwdLayout prior_gemm_bwd_layout(int B, int C, int zdim);
int prior_gemm_bwd(const float* z, int B, const float* centres, int C, int zdim, const float* log_var, const int64_t* z_idx,
                   const int64_t* c_idx, const float* lse, const float* gout, float norm_limit, float* dz, float* dc,
                   float* dlogvar, char* ws, const PriorGemmBwdLayout& L, hipStream_t stream);

}  // namespace evae
