/*
 * evae_hip.h -- C ABI of libevae_hip.so, the MI355X (gfx950) hot path of Exemplar-VAE.
 *
 * The reference (sajadn/Exemplar-VAE) is pure Python and has no FFI: its hot path sits behind the
 * Python API of models/BaseModel.py, models/AbsModel.py, utils/distributions.py, utils/nn.py,
 * utils/knn_on_latent.py and utils/optimizer.py.  The drop-in therefore has two layers:
 *   (1) exemplar-vae_amd/{models,utils}/ -- the reference's module/class/function names,
 *   (2) this C ABI underneath, bound with ctypes (exemplar-vae_amd/evae/_lib.py; a reference
 *       maintainer's binding stub is shown in INTEGRATION.md).
 * Each entry point below cites the reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - C linkage, plain pointers and sizes; no C++/torch types cross the boundary.
 *   - Every pointer is a caller-owned DEVICE pointer (row-major, contiguous, fp32 unless stated;
 *     indices int64).  Nothing is retained after return.
 *   - Work is enqueued on `stream` (a hipStream_t passed as void*); no hidden synchronisation and no
 *     hidden allocation: scratch is passed in (`ws`, size from the matching *_workspace_bytes).
 *     All entry points are hipGraph-capturable.
 *   - Return value: 0 = ok, <0 = EVAE_E*; evae_last_error() gives a thread-local message.
 */
#ifndef EVAE_HIP_H
#define EVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVAE_OK 0
#define EVAE_EINVAL (-1)   /* bad argument (null pointer, negative size, unsupported k, ...) */
#define EVAE_EWORKSPACE (-2) /* workspace too small */
#define EVAE_ELAUNCH (-3)  /* HIP launch error */

#define EVAE_ABI_VERSION 1

typedef void* evae_stream_t; /* hipStream_t */

int evae_version(void);
const char* evae_last_error(void);
/* Host-side helper of the captured training step (no reference counterpart: utils/training.py:27-46 feeds every step from the host):
 * the double-buffered upload of the step's control block -- wait(up, ev_used); copy h_pinned -> d_stage on `up`; record(ev_up, up);
 * wait(step, ev_up); copy d_stage -> d_ctl on `step`; record(ev_used, step) -- as one call.  ev_*: hipEvent_t handles.
 * d_ctl = NULL: no device copy and no record(ev_used) -- the step's first launch takes the block from d_stage itself
 * (evae_batch_prologue_u8_step) and the caller records ev_used behind the step. */
int evae_ctl_upload(void* d_stage, const void* h_pinned, void* d_ctl, size_t bytes, evae_stream_t up, evae_stream_t step,
                    void* ev_used, void* ev_up);

/* Host-side (no device work): the duplicates among a step's exemplar draw.  The reference draws number_components indices WITH
 * replacement (models/BaseModel.py:245) and encodes every draw; an image drawn twice has the same encoding, the same prior
 * term and the same gradient twice.  draws [n_draws] -> rows [cap]: the distinct indices in first-occurrence order (tail padded
 * with rows[0]); inv [n_draws]: each draw's position in rows; rep [cap]: one draw of each distinct row (padding: 0); mult [cap]:
 * multiplicities as floats (padding: 0).  Returns the number of distinct rows, or -1 (more than cap / index out of range). */
int evae_host_dedup(const int64_t* draws, int n_draws, int64_t n_rows, int cap, int64_t* rows, int64_t* inv, int64_t* rep, float* mult);

/* ----------------------------------------------------------------------------------------------
 * Exemplar prior: fused all-pairs distance + leave-one-out mask + online log-sum-exp.
 * Replaces utils/distributions.py:12-25 (pairwise_distance, log_normal_diag_vectorized) and
 * models/BaseModel.py:98-109,124-125 (log_p_z_exemplar, the max/log-sum-exp of log_p_z) without
 * materialising the [B x C] matrix.
 *
 *   p_ij = -1/2 sum_k (log_var_k + log 2pi) - 1/2 sum_k (z_ik - c_jk)^2 / exp(log_var_k)
 *   masked (p_ij := -inf) where z_idx_i == c_idx_j  (both non-NULL = training & !no_mask)
 *
 * evae_prior_lse_fwd returns, for ONE shard of exemplars, per row i:
 *   out_max_i = max_j p_ij, out_sumexp_i = sum_j exp(p_ij - out_max_i), out_nmask_i = #masked.
 * An empty shard (C == 0) yields (-inf, 0, 0).  `out_prob` (optional, [B x C]) receives p_ij
 * (log_p_z(sum=False), BaseModel.py:126-127, before the `- log(denominator)` of :108).
 * evae_prior_merge combines R shard partials ([R x B] each, R >= 1) into
 *   out_logprior_i = LSE_i - log(c_total - sum_r nmask_ri)      (BaseModel.py:100,107-108,124-125)
 *   out_lse        = the forward -> backward TOKEN, 2 B floats: [0, B) the merged row maximum M_i = max_j p_ij, [B, 2 B) the log
 *                    of the normalised sum log sum_j exp(p_ij - M_i); LSE_i = their sum.  They travel apart because the backward
 *                    forms (p_ij - M_i) - log sum: p_ij - M_i is exact where the weight is not negligible (p_ij is the same
 *                    single-rounded cst - d2_ij/2 the forward took its maximum over), so exp(p_ij - LSE_i) stays a normalised
 *                    softmax at any magnitude of the log-density (|LSE| ~ 1e8 for an untrained fully_conv net, golden G21);
 *                    the rounded sum M + log sum would lose ulp(LSE) nats.  A caller with a plain LSE passes (LSE, 0).
 * which is the all-reduce of partial log-sum-exps of the sharded prior (gathered by the caller
 * with one RCCL all-gather; see exemplar-vae_amd/evae/shard.py).
 * Sizes: zdim <= 512.  zdim <= 64 (multiple of 4, no out_prob) runs the forward on the matrix cores in the expanded form
 * |z|^2 + |c|^2 - 2 z.c -- the reference's own formulation (there in fp64) --, zdim <= 56 the backward too; everything else
 * takes the direct-difference VALU kernels.  The fp32 expanded form is kept inside the 1e-5 bar for any input: latents are
 * centred per query tile (the distance is translation invariant) and a tile whose centred squared norms (sigma units)
 * still exceed a limit is recomputed as direct differences by the same launch (evae_prior_set_norm_limit; default 4096).
 * The backward recomputes w_ij = exp((p_ij - M_i) - log sum_i) from the token above.
 */
/* limit < 0 restores the default; 0 sends every tile through the direct-difference path (tests).  Process-wide. */
int evae_prior_set_norm_limit(float limit);
/* c_idx value that masks an exemplar slot for EVERY query (excluded from the sum, counted in nmask): the duplicate slots of
 * the fixed-size exemplar list of the approximate prior (evae_select_exemplars below), so that the denominator stays
 * #unique - #leave-one-out hits as in models/BaseModel.py:265-270 while every shape is static. */
#define EVAE_PRIOR_MASK_ALL (-3)
size_t evae_prior_lse_fwd_workspace_bytes(int B, int C, int zdim);
int evae_prior_lse_fwd(const float* z, int B, const float* centres, int C, int zdim,
                       const float* log_var /* [zdim] */,
                       const int64_t* z_idx /* [B] or NULL */, const int64_t* c_idx /* [C] or NULL */,
                       float* out_max /* [B] */, float* out_sumexp /* [B] */, float* out_nmask /* [B] */,
                       float* out_prob /* [B x C] or NULL */,
                       void* ws, size_t ws_bytes, evae_stream_t stream);
int evae_prior_merge(const float* max /* [R x B] */, const float* sumexp, const float* nmask,
                     int R, int B, float c_total,
                     float* out_logprior /* [B] */, float* out_lse /* token [2 B] or NULL */,
                     evae_stream_t stream);
/* The same forward, but the per-split partials of the launch stay in the workspace un-merged: three planes of
 * *plane_rows x B floats at ws, ws + plane_rows B, ws + 2 plane_rows B (max, sumexp, nmask), of which the first *nsplit
 * rows are valid (both are host integers, fixed by (B, C, zdim)).  For callers that merge them together with something
 * else -- evae_prior_elbo_fwd below.  zdim <= 64 only (larger latents run on the GEMM path and merge on the device). */
int evae_prior_lse_fwd_splits(const float* z, int B, const float* centres, int C, int zdim, const float* log_var,
                              const int64_t* z_idx, const int64_t* c_idx, void* ws, size_t ws_bytes,
                              int* nsplit, int* plane_rows, evae_stream_t stream);
/* Tail of a training step's forward in ONE launch (utils/training.py:33 -> models/BaseModel.py:65-77,124-125): merge R
 * partial rows (row stride ldp) per query -- the splits of evae_prior_lse_fwd_splits, or the all-gathered shard partials --,
 *   logp_i = LSE_i - log(c_total - sum_r nmask_ri),  KL_i = logq_i - logp_i,  loss_i = beta KL_i - RE_i,
 * and, when means != NULL, the batch means (loss, RE, KL).  beta from device memory when beta_dev != NULL. */
int evae_prior_elbo_fwd(const float* pmax, const float* psum, const float* pnmask, int R, int ldp, int B, float c_total,
                        const float* RE, const float* logq, const float* beta_dev, float beta_host,
                        float* logp /* [B] */, float* lse /* token [2 B] or NULL */, float* loss /* [B] */, float* KL /* [B] */,
                        float* means /* [3] or NULL */, evae_stream_t stream);
/* The same, plus the coefficient vectors of evae_elbo_bwd for "batch mean of the loss, upstream gradient 1" (a captured step's
 * loss.backward(ones)): cRE = -1/B, cKL = beta/B, neg_cKL = -beta/B -- the backward pass starts one launch later. */
int evae_prior_elbo_fwd_coef(const float* pmax, const float* psum, const float* pnmask, int R, int ldp, int B, float c_total,
                             const float* RE, const float* logq, const float* beta_dev, float beta_host, float* logp, float* lse,
                             float* loss, float* KL, float* means, float* cRE, float* cKL, float* neg_cKL, evae_stream_t stream);

/* evae_prior_elbo_fwd(_coef) as two launches for a step on two streams: the merge of the partial rows (token, logp and -- all
 * three or none -- the coefficient vectors) needs nothing of the reconstruction term and is all the prior's backward waits for;
 * the ELBO assembly (loss, KL, batch means; models/BaseModel.py:71-75,124-125) runs on the stream that produced RE. */
int evae_prior_merge_coef(const float* pmax, const float* psum, const float* pnmask, int R, int ldp, int B, float c_total,
                          const float* beta_dev, float beta_host, float* logp /* [B] */, float* lse /* token [2 B] */,
                          float* cRE, float* cKL, float* neg_cKL, evae_stream_t stream);
int evae_elbo_assemble(const float* logp, const float* RE, const float* logq, const float* beta_dev, float beta_host, int B,
                       float* loss /* [B] */, float* KL /* [B] */, float* means /* [3] or NULL */, evae_stream_t stream);

/* The prior of a CAPTURED training step (one device, B <= 128 queries, C <= 30 720 exemplars, zdim <= 56 and a multiple of 4) as one
 * launch + the dz / dlogvar reduction: forward partials, their merge (two levels of last-arriving blocks, fixed order) and the
 * backward of -beta / B * sum_i logp_i on the products each block still holds (csrc/evae_prior_train.h).  Replaces
 * evae_prior_lse_fwd_splits + evae_prior_merge_coef / evae_prior_elbo_fwd_coef + evae_prior_lse_bwd under the promise of
 * evae_prior_elbo_fwd_coef (batch mean of the loss, upstream gradient 1); same arithmetic per pair as those kernels' matrix-core
 * paths (models/BaseModel.py:98-128 and what autograd derives from it).  `state`: 256 bytes of device memory, zeroed ONCE by the
 * caller and handed to every call (inter-block counters and a generation word; word 10 counts blocks whose bounded wait ran out --
 * never, unless the grid was not co-resident).  phase 0: both launches; 1: the kernel; 2: the reduction (dz, dlogvar). */
int evae_prior_train_applies(int B, int C, int zdim);   /* also: the grid fits the CURRENT device's resident blocks (CUs x occupancy, with a margin) */
/* word 10 of `state`, read back and cleared (synchronises `stream`): > 0 = that many blocks gave up waiting since the last call --
 * gradients of some step are wrong and the caller must raise; < 0 = EVAE_E*.  Call where the host synchronises anyway. */
int evae_prior_train_gave_up(void* state, evae_stream_t stream);
size_t evae_prior_train_workspace_bytes(int B, int C, int zdim);
int evae_prior_train_step(const float* z, int B, const float* centres, int C, int zdim, const float* log_var,
                          const int64_t* z_idx, const int64_t* c_idx, float c_total, const float* beta_dev, float beta_host,
                          float* logp /* [B] */, float* token /* [2 B] */, float* cRE, float* cKL, float* neg_cKL,
                          float* dz /* [B x zdim] */, float* dcentres /* [C x zdim] */, float* dlogvar /* [zdim] */, void* state,
                          void* ws, size_t ws_bytes, int phase, evae_stream_t stream);
/* ... of a step that encoded each DISTINCT image of its draw once (the reference draws with replacement and encodes every draw,
 * models/BaseModel.py:243-254): `centres` = the n_rows distinct rows' encodings, exemplar j of the prior = row rows_inv[j] (j < C); the
 * per-draw centre gradients land in the scratch dc_draws [C x zdim] and the reduction launch folds them:
 * dcentres[u] = rows_mult[u] * dc_draws[rows_rep[u]] for u < n_rows (padding rows carry multiplicity 0).  Same two launches as
 * evae_prior_train_step -- no gather launch in front of the prior, none behind it (r06). */
int evae_prior_train_step_rows(const float* z, int B, const float* centres, int n_rows, const int64_t* rows_inv, const int64_t* rows_rep,
                               const float* rows_mult, int C, int zdim, const float* log_var, const int64_t* z_idx,
                               const int64_t* c_idx, float c_total, const float* beta_dev, float beta_host, float* logp, float* token,
                               float* cRE, float* cKL, float* neg_cKL, float* dz, float* dcentres /* [n_rows x zdim] */,
                               float* dc_draws /* [C x zdim] */, float* dlogvar, void* state, void* ws, size_t ws_bytes,
                               evae_stream_t stream);

/* Backward of sum_i grad_out_i * logprior_i through the prior (what autograd derives from
 * BaseModel.py:98-128 + distributions.py:12-25), by recomputation from the saved row LSE:
 *   w_ij = exp(p_ij - lse_i);  dz_i = sum_j g_i w_ij (c_j - z_i)/var;  dc_j = sum_i g_i w_ij (z_i - c_j)/var;
 *   dlogvar_k = sum_ij g_i w_ij (-1/2 + 1/2 (z_ik - c_jk)^2 / var_k).
 * `lse` is the GLOBAL out_lse of evae_prior_merge; dz and dlogvar are this shard's partial sums
 * (sum-all-reduce across shards), dcentres is complete for the shard's exemplars. */
size_t evae_prior_lse_bwd_workspace_bytes(int B, int C, int zdim);
int evae_prior_lse_bwd(const float* z, int B, const float* centres, int C, int zdim,
                       const float* log_var, const int64_t* z_idx, const int64_t* c_idx,
                       const float* lse /* token [2 B] of evae_prior_merge */, const float* grad_out /* [B] */,
                       float* dz /* [B x zdim] */, float* dcentres /* [C x zdim] */,
                       float* dlogvar /* [zdim] */,
                       void* ws, size_t ws_bytes, evae_stream_t stream);
/* The same in two halves: phase 1 = everything up to and including dcentres (complete on return of the launch), phase 2 = the
 * reduction of the per-split dz / dlogvar partials left in the workspace -> dz, dlogvar (a no-op on the paths that do all
 * their work in phase 1).  Lets a caller continue with dcentres on one stream while dz is finished on another. */
int evae_prior_lse_bwd_phased(const float* z, int B, const float* centres, int C, int zdim, const float* log_var,
                              const int64_t* z_idx, const int64_t* c_idx, const float* lse, const float* grad_out,
                              float* dz, float* dcentres, float* dlogvar, void* ws, size_t ws_bytes, int phase,
                              evae_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Distance + top-K.  Replaces pairwise_distance(z, sub_cache).topk(k, largest=False)
 * (models/BaseModel.py:263-264) and find_nearest_neighbors (utils/knn_on_latent.py:4-9: sqrt of the
 * direct-difference distance, topk(k=20, sorted=True)).
 * Distances are accumulated in fp64 and rounded once to fp32 (bit-compatible with the reference's
 * fp64 pairwise_distance); selection order is (value ascending, index ascending); k <= 64.
 * `index_base` is added to every returned index (shard offset).  out_idx/out_val are [B x k],
 * sorted nearest first.  out_val may be NULL.
 */
#define EVAE_TOPK_SQRT 1u /* apply sqrtf to the fp32 distance before comparing (knn_on_latent.py:8) */
size_t evae_pairdist_topk_workspace_bytes(int B, int N, int zdim, int k);
int evae_pairdist_topk(const float* q, int B, const float* cache, int N, int zdim, int k,
                       unsigned flags, int64_t index_base,
                       int64_t* out_idx, float* out_val,
                       void* ws, size_t ws_bytes, evae_stream_t stream);
/* Materialised [B x N] squared distances, fp64-accumulated, rounded once to fp32: the value
 * utils/distributions.py:12-18 returns.  Not on the hot path (kept for API completeness). */
int evae_pairwise_distance(const float* q, int B, const float* cache, int N, int zdim,
                           float* out /* [B x N] */, evae_stream_t stream);
/* Merge R candidate lists ([R x B x k], e.g. all-gathered shard results) into the global top-k. */
int evae_topk_merge(const float* val /* [R x B x k] */, const int64_t* idx, int R, int B, int k,
                    int64_t* out_idx /* [B x k] */, float* out_val /* [B x k] or NULL */,
                    evae_stream_t stream);
/* The exemplar selection of the approximate prior with static shapes (reference models/BaseModel.py:265-266:
 * `unique(nearest_indices)` then `exemplars_indices[nearest]`): pos [n] = the flattened top-k positions (0 <= pos < C) into the
 * candidate list cand_idx [C] of dataset rows.  sel_rows[i] = cand_idx[pos[i]] for every slot; c_idx[i] = sel_rows[i] for the
 * first slot that names a position and EVAE_PRIOR_MASK_ALL for its repeats.  n_unique (optional, device) receives the
 * number of distinct positions.  One block; n <= 16 384. */
int evae_select_exemplars(const int64_t* pos, int n, const int64_t* cand_idx, int C, int64_t* sel_rows, int64_t* c_idx,
                          int* n_unique, evae_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Dense layers on fp32 MFMA (v_mfma_f32_32x32x2_f32).  Replace utils/nn.py:29-69 (NonLinear,
 * GatedDense) and torch.nn.Linear as used by models/VAE.py:15-30, models/HVAE_2level.py:15-66.
 * Weights keep nn.Linear layout [out x in].  `rows` (optional int64 [M]) gathers the input rows
 * x[rows[m], :] inside the A-tile load: the exemplar gather of models/BaseModel.py:247 without
 * materialising the [C x D] copy.
 *
 * evae_gated_dense_fwd:  h = x Wh^T + bh ; s = sigmoid(x Wg^T + bg) ; out = h * s
 *   The backward needs only `out` and `s` (dg = dout*h*s*(1-s) = dout*out*(1-s)): save_s ([M x N], NULL for
 *   inference) is the one extra array to keep; save_h (optional, normally NULL) also returns h.
 * evae_linear_fwd:       y = act(x W^T + b), act in EVAE_ACT_*; `pre` (optional) saves the
 *   pre-activation needed by hardtanh's backward.
 * evae_gated_dense_bwd_input: given dout [M x N], the layer's forward output `out` (= h*s) and s:
 *   dh = dout*s ; dg = dout*out*(1-s)   (written to dh, dg: [M x N] each)
 * evae_dense_bwd_data:   dx = dy1 W1 (+ dy2 W2)         [M x K]   (dy [M x N], W [N x K])
 *   optionally fused with the gated-dense input derivative of the layer below (out_prev, s_prev
 *   non-NULL: that layer's forward output and gate): writes dh_prev/dg_prev instead of dx.
 * evae_dense_bwd_weight: dW = dy^T x (rows-gathered x allowed), db = column sums of dy.
 * Thin problems (the 100-row batch path, the small weight-gradient outputs) are split along the
 * contraction into `ws` partials and finished by a second kernel in a fixed order (deterministic);
 * the split is chosen by a wave-quantisation model of the 256-CU chip, see csrc/evae_gemm_core.h (make_plan).
 */
#define EVAE_ACT_NONE 0
#define EVAE_ACT_SIGMOID 1
#define EVAE_ACT_HARDTANH 2 /* fp32 GEMMs on the bf16 matrix pipe (csrc/evae_gemm_x6.h): operands are split in-kernel into three bf16 terms each and six
 * partial products are accumulated in fp32 -- fp32-GEMM accuracy at up to 16/6 of the fp32 matrix rate.  Applies to launches
 * whose operands are both contraction-contiguous (forward layers, channels-last convolutions) with at least `min_rows`
 * output rows.  enabled: 0 off, 1 on, < 0 unchanged; min_rows < 0 unchanged.  Defaults: on, 2048 (env EVAE_X6, EVAE_X6_MIN_ROWS). */
int evae_gemm_x6_configure(int enabled, int min_rows);
int evae_gemm_x6_applies(int M, int N, int gated);   /* 1 when a launch with M x N outputs takes the split-bf16 kernel */

/* clamp to [act_lo, act_hi] */

size_t evae_dense_fwd_workspace_bytes(int M, int K, int N, int gated);
int evae_gated_dense_fwd(const float* x, const int64_t* rows, int M, int K, int ldx,
                         const float* wh, const float* bh, const float* wg, const float* bg, int N,
                         float* out, float* save_h, float* save_s,
                         void* ws, size_t ws_bytes, evae_stream_t stream);
int evae_linear_fwd(const float* x, const int64_t* rows, int M, int K, int ldx,
                    const float* w, const float* b, int N, int act, float act_lo, float act_hi,
                    float* y, float* pre, void* ws, size_t ws_bytes, evae_stream_t stream);
/* The two heads of the encoder on the same input and the sample, for launches of few rows (models/VAE.py:24-26,
 * models/BaseModel.py:79-82, utils/distributions.py:28-33): z_mean = x wm^T + bm, lv_pre = x wl^T + bl, logvar = clamp(lv_pre,
 * lv_lo, lv_hi), z = z_mean + eps exp(logvar / 2), logq[m] = log N(z | z_mean, exp(logvar)) -- one split-K GEMM (both products)
 * and one finish launch instead of evae_linear_fwd x 2 + evae_reparam_logq_fwd (five launches).  lv_pre, logq may be NULL. */
size_t evae_heads_reparam_fwd_workspace_bytes(int M, int K, int Z);
int evae_heads_reparam_fwd(const float* x, int M, int K, int ldx, const float* wm, const float* bm, const float* wl,
                           const float* bl, int Z, float lv_lo, float lv_hi, const float* eps, float* z_mean, float* lv_pre,
                           float* logvar, float* z, float* logq, void* ws, size_t ws_bytes, evae_stream_t stream);
/* The same in the batch-sized case only (one launch; evae_heads_reparam_fwd_bcast_applies says whether M, K, Z, ldx are), and
 * dst[0 .. n) = src[0] by the same launch: evae_broadcast_scalar's work -- the exemplar prior's log-variance row
 * (models/BaseModel.py:101), which the step's prior launch reads behind this one -- without a graph node of its own. */
int evae_heads_reparam_fwd_bcast_applies(int M, int K, int Z, int ldx);
int evae_heads_reparam_fwd_bcast(const float* x, int M, int K, int ldx, const float* wm, const float* bm, const float* wl,
                                 const float* bl, int Z, float lv_lo, float lv_hi, const float* eps, float* z_mean, float* lv_pre,
                                 float* logvar, float* z, float* logq, const float* src, float* dst, int n, evae_stream_t stream);
/* The same two heads with the log-density of a GIVEN sample zq [M x Z] (models/AbsHModel.py:17-20 p(z1 | z2) and the
 * log_normal_diag(z1, p1_mu, p1_lv) of :99-100): logp[m] = log N(zq[m] | z_mean[m], exp(logvar[m])); workspace as above. */
int evae_heads_density_fwd(const float* x, int M, int K, int ldx, const float* wm, const float* bm, const float* wl,
                           const float* bl, int Z, float lv_lo, float lv_hi, const float* zq, float* z_mean, float* lv_pre,
                           float* logvar, float* logp, void* ws, size_t ws_bytes, evae_stream_t stream);
/* dy1/dy2: [M x N] with row stride ldy (so dh and dg may be the two halves of one [M x 2N] buffer);
 * output(s) [M x K] with row stride ldo; out_prev/s_prev are dense [M x K]. */
size_t evae_dense_bwd_data_workspace_bytes(int M, int N, int K, int npairs);
int evae_dense_bwd_data(const float* dy1, const float* w1, const float* dy2, const float* w2,
                        int M, int N, int ldy, int K,
                        const float* out_prev, const float* s_prev,
                        float* dx_or_dh, float* dg, int ldo,
                        void* ws, size_t ws_bytes, evae_stream_t stream);
/* The same with the transposed weights of the split-bf16 path handed in (wT [npairs][K][evae_dense_bwd_data_wt_ld(N)],
 * evae_dense_bwd_data_wt_bytes: they depend on the weights alone, so a training step prepares them in its head launch);
 * wT == NULL: evae_dense_bwd_data.  Launches that stay on the fp32 kernel ignore wT. */
size_t evae_dense_bwd_data_wt_bytes(int N, int K, int npairs);
int evae_dense_bwd_data_wt_ld(int N);
int evae_dense_bwd_data_wt(const float* dy1, const float* w1, const float* dy2, const float* w2, int M, int N, int ldy, int K,
                           const float* out_prev, const float* s_prev, float* dx_or_dh, float* dg, int ldo, const float* wT,
                           void* ws, size_t ws_bytes, evae_stream_t stream);
/* dw [N x K] = dy^T x with dy [M x N] (row stride ldy), x [* x K] (row stride ldx, optional row gather);
 * db [N] = column sums of dy.  Passing the [M x 2N] buffer [dh | dg] yields [dWh ; dWg] in one launch. */
size_t evae_dense_bwd_weight_workspace_bytes(int M, int N, int K);
int evae_dense_bwd_weight(const float* dy, int M, int N, int ldy, const float* x, const int64_t* rows,
                          int K, int ldx, float* dw /* [N x K] */, float* db /* [N] or NULL */,
                          int accumulate, void* ws, size_t ws_bytes, evae_stream_t stream);
/* The same in two halves for callers that schedule by hand: phase 1 = the split-K GEMM into the workspace partials,
 * phase 2 = the finish launch (fixed-order sum of the partial planes -> dw, db), possibly on another stream once phase 1
 * has completed; the workspace must not be reused in between. */
int evae_dense_bwd_weight_phased(const float* dy, int M, int N, int ldy, const float* x, const int64_t* rows,
                                 int K, int ldx, float* dw, float* db, int accumulate, void* ws, size_t ws_bytes,
                                 int phase, evae_stream_t stream);
/* ----------------------------------------------------------------------------------------------
 * First encoder layer on a uint8-resident image store (the data side of models/BaseModel.py:243-248: the C exemplar images
 * of a step are rows of dataset.tensors[0]; utils/load_data/base_load_data.py:39-40 makes those pixels k/255).  The store keeps
 * the byte k (value = k * x_scale): a quarter of the HBM footprint and gather traffic of fp32 rows.  Bytes are exact in bf16 and
 * an fp32 weight is exactly the sum of three bf16 terms, so the layer runs as three exact-product bf16 MFMAs per element with
 * fp32 accumulation -- fp32-GEMM accuracy at 16/3 of the fp32 matrix rate (csrc/evae_dense_u8.hip).
 *   evae_dense_u8_prepare    both weight banks [N x K] fp32 -> the bf16 terms in the kernel's tile order (once per weight update)
 *   evae_gated_dense_fwd_u8  out[m] = (x[rows[m]] Wh^T * x_scale + bh) * sigmoid(x[rows[m]] Wg^T * x_scale + bg); save_s as above
 * Rows of the store must start 16-byte aligned (ldx % 16 == 0) and the allocation must extend 32 bytes past the last row. */
int evae_dense_u8_supported(int K, long long ldx);
size_t evae_dense_u8_prepared_bytes(int N, int K);
int evae_dense_u8_prepare(const float* wh, const float* wg, int N, int K, void* prepared, size_t prepared_bytes,
                          evae_stream_t stream);
int evae_gated_dense_fwd_u8(const unsigned char* x, const int64_t* rows, int M, int K, long long ldx, float x_scale,
                            const void* prepared, const float* bh, const float* bg, int N,
                            float* out /* [M x N] */, float* save_s /* [M x N] or NULL */, evae_stream_t stream);
/* Weight gradient of that layer: dw [N x K] = x_scale * dy^T x(rows), db [N] = column sums of dy (dy [M x N], row stride ldy:
 * the merged [dh | dg] buffer).  Same arithmetic: the byte rows are exact in bf16, dy is split exactly into three bf16 terms;
 * pre-passes transpose the gathered bytes and lay dy^T out in tile order, the product runs split along the batch rows into
 * workspace partials that a last launch sums in a fixed order. */
size_t evae_dense_bwd_weight_u8_workspace_bytes(int M, int N, int K);
/* dy may be NULL when the producer has already written its tile images into the workspace: evae_dense_bwd_weight_u8_images
 * gives their byte offset inside `ws` and the number of 32-row K-slabs per column tile; evae_dense_bwd_data_img is the data
 * gradient of the layer above with exactly that output -- (dh, dg) = the gate derivative of the layer below applied to
 * dy1 W1 (+ dy2 W2), for rows [m_base, m_base + M) of the merged [dh | dg] buffer (m_base a multiple of 8; the rows between
 * m_base + M and the next multiple of 8 are written as zeros, so the launch that ends off a multiple of 8 goes last).  wT /
 * ws as evae_dense_bwd_data_wt / evae_dense_bwd_data (the split-bf16 kernel's transposed weights; ws of
 * evae_dense_bwd_data_workspace_bytes(M, N, K, pairs)).  Image bytes that no row / column of the problem maps to (the tail of
 * the last slab, columns beyond 2K) must be zero: zero the workspace once after allocating it. */
int evae_dense_bwd_weight_u8_images(int M, int N, int K, size_t* offset, int* nslab);
int evae_dense_bwd_data_img(const float* dy1, const float* w1, const float* dy2, const float* w2, int M, int N, int ldy, int K,
                            const float* out_prev, const float* s_prev, void* img, int img_nslab, int m_base,
                            const float* wT /* or NULL */, void* ws, size_t ws_bytes, evae_stream_t stream);
int evae_dense_bwd_weight_u8(const float* dy, int M, int N, long long ldy, const unsigned char* x, const int64_t* rows,
                             int K, long long ldx, float x_scale, float* dw /* [N x K] */, float* db /* [N] or NULL */,
                             void* ws, size_t ws_bytes, evae_stream_t stream);
/* The same in two calls that may go to different streams (the caller orders them with an event): phase 1 = the pre-passes
 * into the workspace, phase 2 = the product and its finish; or phase 3 = the byte gather-transpose alone (needs x / rows only,
 * dy and dw may be NULL: a training step issues it during its forward pass), phase 4 = everything else.  phase | 16 (also
 * 0 | 16): no finish launch -- the partial planes stay in ws for evae_dense_bwd_weight_finish_group. */
int evae_dense_bwd_weight_u8_phased(const float* dy, int M, int N, long long ldy, const unsigned char* x, const int64_t* rows,
                                    int K, long long ldx, float x_scale, float* dw, float* db, void* ws, size_t ws_bytes,
                                    int phase, evae_stream_t stream);
/* dh, dg: [M x N] with row stride ldo (the two halves of one [M x 2N] buffer when ldo = 2N) */
/* ONE finish launch for the split-K weight gradients of a training step (r03): the GEMMs were issued without their own finish
 * -- evae_dense_bwd_weight_phased(phase 1) for fp32 rows, evae_dense_bwd_weight_u8_phased(phase | 16) for the byte layer -- and
 * leave partial planes in their workspaces; this call sums them (fixed order, same arithmetic as the separate finish launches)
 * into dw / db.  A job repeats the sizes, row strides and workspace of its GEMM call; byte_rows != 0: the byte layer (x_scale as there; at
 * most one such job); at most three fp32 jobs (no row gather, no accumulation). */
typedef struct { int byte_rows; int M, N, K, ldy, ldx; float x_scale; float* dw; float* db; void* ws; size_t ws_bytes; } evae_wgrad_finish_job_t;
int evae_dense_bwd_weight_finish_group(const evae_wgrad_finish_job_t* jobs, int njobs, evae_stream_t stream);
/* Several thin weight gradients in ONE launch (the batch rows' leaf layers of a training step: reference utils/nn.py:44-69
 * backward of the decoder's GatedDense layers and of the log-variance head, utils/training.py:39): every job is
 * dw [N x K] = dy^T x (+ db [N] = column sums of dy, NULL to skip) over M <= 128 contraction rows, no row gather; accumulate != 0:
 * added to what dw / db hold (the second application of a layer both row sets run through: autograd's sum, formed in place),
 * pointers 16-byte aligned and N, K, ldy, ldx multiples of 4; at most 6 jobs.  EVAE_EINVAL (nothing launched) when a job does not
 * qualify -- issue them with evae_dense_bwd_weight then. */
typedef struct { const float* dy; const float* x; float* dw; float* db; int M, N, K, ldy, ldx, accumulate; } evae_wgrad_job_t;
int evae_dense_bwd_weight_group(const evae_wgrad_job_t* jobs, int njobs, evae_stream_t stream);
int evae_gated_dense_bwd_input(const float* dout, const float* out, const float* s, int M, int N,
                               float* dh, float* dg, int ldo, evae_stream_t stream);
/* the same with a row stride on dout (ldd >= N): the upstream gradient of one half of a concatenation (models/AbsHModel.py:26,37
 * torch.cat of two gated layers' outputs) is read where it lies */
int evae_gated_dense_bwd_input_ld(const float* dout, int ldd, const float* out, const float* s, int M, int N,
                                  float* dh, float* dg, int ldo, evae_stream_t stream);
/* a gated layer's whole backward with respect to its input (reference utils/nn.py:62-68 under autograd): [dh | dg] = (dout s,
 * dout (h s)(1 - s)) into dpre [M x 2N] (row stride ldp) for the weight gradient, dx [M x K] = dh Wh + dg Wg.  One launch for
 * batch-sized row counts (the gate derivative formed in the operand load), evae_gated_dense_bwd_input_ld + evae_dense_bwd_data
 * otherwise -- the same bits either way; ws as for evae_dense_bwd_data(M, N, K, 2 pairs). */
int evae_gated_dense_bwd(const float* dout, int ldd, const float* out, const float* s, int M, int N, const float* wh,
                         const float* wg, int K, float* dpre, int ldp, float* dx, int ldo, void* ws, size_t ws_bytes,
                         evae_stream_t stream);
/* dst [n x z] = scale[r] * src[idx[r]] (scale NULL: 1), z a multiple of 4: the bridge between the DISTINCT exemplar rows a captured
 * step encodes and the draws (with replacement, models/BaseModel.py:245) its prior sees -- see evae_host_dedup */
int evae_gather_rows(const float* src, const int64_t* idx, const float* scale, int n, int z, float* dst, evae_stream_t stream);
int evae_act_bwd(const float* dy, const float* y_or_pre, size_t n, int act, float act_lo, float act_hi,
                 float* dpre, evae_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Convolutions as implicit GEMMs on fp32 MFMA (LDS-staged im2col, never materialised).  Replace
 * utils/nn.py:72-114 (GatedConv2d, Conv2d) and the nn.Conv2d layers of models/convHVAE_2level.py:13-92 and
 * models/fully_conv.py:12-81, forward and backward.  Tensors are NCHW fp32 contiguous, filters keep the
 * nn.Conv2d layout [Co x C x KH x KW]; square stride/padding, dilation 1, groups 1.
 *   evae_conv2d_fwd:        out = act(conv(x, wh) + bh)                       (wg == NULL)
 *                           out = act(conv(x, wh) + bh) * sigmoid(conv(x, wg) + bg), saving h and s
 *   evae_conv2d_bwd_data:   dx = conv_transpose(dyh, wh) (+ conv_transpose(dyg, wg))
 *   evae_conv2d_bwd_weight: dw [(1|2) Co x C*KH*KW] = dy^T im2col(x) for dyh (and dyg, stacked), db [(1|2) Co]
 * `what` in the workspace query: 0 forward, 1 data gradient, 2 weight gradient.
 */
typedef struct {
  int N, C, H, W;        /* input [N x C x H x W] */
  int Co, KH, KW, stride, pad;
} evae_conv_desc_t;
size_t evae_conv2d_workspace_bytes(const evae_conv_desc_t* d, int what, int gated);
int evae_conv2d_fwd(const float* x, const evae_conv_desc_t* d, const float* wh, const float* bh,
                    const float* wg, const float* bg, int act, float act_lo, float act_hi,
                    float* out, float* save_h, float* save_s, void* ws, size_t ws_bytes,
                    evae_stream_t stream);
int evae_conv2d_bwd_data(const float* dyh, const float* wh, const float* dyg, const float* wg,
                         const evae_conv_desc_t* d, float* dx, void* ws, size_t ws_bytes,
                         evae_stream_t stream);
int evae_conv2d_bwd_weight(const float* dyh, const float* dyg, const float* x, const evae_conv_desc_t* d,
                           float* dw, float* db, void* ws, size_t ws_bytes, evae_stream_t stream);
/* The same three operations on channels-last activations ([N][H][W][C] storage, i.e. torch.channels_last of the
 * logical NCHW tensor), run as instances of the dense GEMM kernel: with the contraction ordered (kh, kw, c) a
 * 32-wide K-slab is 32 channels of ONE filter tap, so the im2col gather is a per-slab scalar offset plus one
 * validity bit per pixel.  Filters and their gradients keep the nn.Conv2d layout [Co][C][KH][KW].
 * `evae_conv2d_cl_supported(d, what, gated)` (what: 0 forward, 1 data gradient, 2 weight gradient) tells whether
 * the geometry qualifies (forward: C % 4 == 0 and C >= 16, or a thin first layer; data gradient: C % 4 == 0; weight
 * gradient: C % 4 == 0 and (1|2)Co % 4 == 0; tensors below 2 GiB); otherwise use the NCHW entry points above.
 * dy of the backward calls is ONE buffer [N][OH][OW][ldy], ctot = Co or 2 Co: channels [0, Co) hold dh, [Co, 2 Co) dg
 * (gated layers; evae_gated_dense_bwd_input writes exactly that with ldo = ldy).  ldy = evae_conv2d_cl_dy_stride(ctot)
 * = ctot when ctot % 4 == 0 (a plain layer's upstream gradient is used as it is), else ctot rounded up to a multiple
 * of 32 with ZERO padding channels. */
int evae_conv2d_cl_supported(const evae_conv_desc_t* d, int what, int gated);
int evae_conv2d_cl_dy_stride(int ctot);
size_t evae_conv2d_cl_workspace_bytes(const evae_conv_desc_t* d, int what, int gated);
int evae_conv2d_cl_fwd(const float* x, const evae_conv_desc_t* d, const float* wh, const float* bh,
                       const float* wg, const float* bg, int act, float act_lo, float act_hi,
                       float* out, float* save_h, float* save_s, void* ws, size_t ws_bytes,
                       evae_stream_t stream);
int evae_conv2d_cl_bwd_data(const float* dy, const float* wh, const float* wg, const evae_conv_desc_t* d,
                            float* dx, void* ws, size_t ws_bytes, evae_stream_t stream);
int evae_conv2d_cl_bwd_weight(const float* dy, const float* x, const evae_conv_desc_t* d, int gated,
                              float* dw, float* db, void* ws, size_t ws_bytes, evae_stream_t stream);
/* Residual blocks of models/fully_conv.py:13-23, y = x + conv(ELU(x)) (weight-normed 3x3, same shape in and out), on the
 * channels-last kernels with the elementwise work in the epilogues:
 *   evae_elu_fwd               a = ELU(x)
 *   evae_conv2d_cl_fwd_res     y = conv(a, w) + b + residual            (residual = x)
 *   evae_conv2d_cl_bwd_data_res dx = residual + ELU'(x) * conv_transpose(dy, w), ELU'(x) = (a > 0 ? 1 : a + 1)   (residual = dy)
 * Channel counts with C % 4 == 0, C != 32; the weight gradient is evae_conv2d_cl_bwd_weight on (dy, a). */
int evae_elu_fwd(const float* x, size_t n, float* out, evae_stream_t stream);
int evae_conv2d_cl_res_supported(const evae_conv_desc_t* d);   /* 1 when the three *_res / weight-gradient calls of a block take d */
int evae_conv2d_cl_fwd_res(const float* a, const evae_conv_desc_t* d, const float* w, const float* b, const float* residual,
                           float* out, void* ws, size_t ws_bytes, evae_stream_t stream);
int evae_conv2d_cl_bwd_data_res(const float* dy, const float* w, const evae_conv_desc_t* d, const float* residual,
                                const float* elu_out, float* dx, void* ws, size_t ws_bytes, evae_stream_t stream);

/* Window convolutions over pre-split pixel images (csrc/evae_conv_win.h): the gated layers BETWEEN two layers of a convolutional
 * encoder stack (reference utils/nn.py:72-97, models/convHVAE_2level.py:21-46), whose activations then never exist as fp32
 * tensors.  A "pixel image" holds an activation [pixels x channels] (channels % 16 == 0) as three bf16 planes per 16 pixels x 16
 * channels (6 bytes per element, evae_cw_image_bytes): the exact three-term split of every fp32 value, written by the producing
 * layer's epilogue, read by the next layer's forward, by the data gradient (the merged [dh | dg] image) and by the weight gradient.
 * Row order of an image: natural (n, y, x), or PARITY-PLANAR -- the four stride-2 parity classes of an image as four sub-images
 * -- when the tensor's consumer has stride 2.  fp32 side tensors (gates, optional copies) are always natural channels-last.
 * The forward contraction keeps the window of input pixels of a block's taps resident in LDS (one copy per channel group, not per
 * tap); arithmetic: six bf16 partial products per fp32 product, fp32 accumulate (the fp32 bar, as evae_gated_dense_fwd's x6 path).
 *   evae_cw_supported(d, what): what = 0 forward (C % 16 == 0, Co % 32 == 0, odd square filter with pad = (K - 1) / 2, stride 1 | 2,
 *     square even grid), 1 data gradient (C == 32 or C % 64 == 0), 2 weight gradient (stride 1, C % 32 == 0, 3 x 3 | 5 x 5, 2 Co <= 128).
 *   evae_cw_pack_image: fp32 channels-last [N][H][W][C] -> image (planar bit 0: rows parity-planar; bit 1: the image of ELU(x)) -- the entry of a stack.
 *   evae_cw_fwd_gated: out = (conv(x, wh) + bh) * sigmoid(conv(x, wg) + bg); x image rows natural (stride 1) | planar (stride 2);
 *     oimg (rows planar when out_planar), out_s = the gate fp32 [N OH OW][Co], out_f = optional fp32 copy of out.
 *   evae_cw_bwd_data_gate: v = conv_transpose([dh | dg], [wh | wg]) and, in the epilogue, the gate derivative of the layer BELOW
 *     (its output image eimg -- rows planar when d->stride == 2 -- and gate e_s): [dh' | dg'] = [v s' | v out' (1 - s')] as an image
 *     in eimg's row order (oimg, 2 C channels) and / or fp32 natural [N H W][2 C] (out_f).
 *   evae_cw_gate_bwd_image: the same gate derivative for an upstream gradient v that is an fp32 tensor (exit of a stack).
 *   evae_cw_bwd_weight: dw [2 Co][C][K][K], db [2 Co] from the merged-gradient image and the input image (contraction over pixels
 *     through the LDS transpose read; partial planes per block, added in block order: deterministic). */
size_t evae_cw_image_bytes(long long rows, int channels);
int evae_cw_supported(const evae_conv_desc_t* d, int what);
size_t evae_cw_workspace_bytes(const evae_conv_desc_t* d, int what);
int evae_cw_pack_image(const float* x, int N, int H, int W, int C, int planar, void* img, evae_stream_t stream);
int evae_cw_fwd_gated(const void* ximg, const evae_conv_desc_t* d, const float* wh, const float* bh, const float* wg, const float* bg,
                      void* oimg, int out_planar, float* out_s, float* out_f, void* ws, size_t ws_bytes, evae_stream_t stream);
int evae_cw_bwd_data_gate(const void* dyimg, int dy_planar, const evae_conv_desc_t* d, const float* wh, const float* wg,
                          const void* eimg, const float* e_s, void* oimg, float* out_f, void* ws, size_t ws_bytes,
                          evae_stream_t stream);
int evae_cw_gate_bwd_image(const float* v, const void* eimg, int planar, const float* s, int N, int H, int W, int C, void* oimg,
                           float* out_f, evae_stream_t stream);
int evae_cw_bwd_weight(const void* dyimg, int dy_planar, const void* ximg, const evae_conv_desc_t* d, float* dw, float* db,
                       void* ws, size_t ws_bytes, evae_stream_t stream);
/* Residual blocks y = x + conv(ELU(x), w) + b of models/fully_conv.py:13-23 (C == Co <= 128, C % 16 == 0, stride 1, 'same' padding) on
 * pixel images: a block's convolution operand is the image of ELU(x), which the block before writes in its epilogue and from which
 * the backward derives ELU'(x) = (a > 0 ? 1 : a + 1).  evae_cw_pack_image's `planar` bit 1 asks for the image of ELU(x) (entry of a run).
 *   evae_cw_res_fwd: y -> out_f (fp32 [N H W][C]) and / or oimg = image of ELU(y).
 *   evae_cw_res_bwd_data: dx = dy + ELU'(x) conv_transpose(dy, w): dyimg + dy_f (the same gradient, fp32), aimg -> dx_f and / or dximg.
 *   evae_cw_bwd_weight_plain: dw [Co][C][K][K], db [Co] from dy's image and the input image (the image of ELU(x)).
 * Workspace: evae_cw_workspace_bytes(d, 5) for the two convolutions, (d, 7) for the weight gradient. */
int evae_cw_res_supported(const evae_conv_desc_t* d);
int evae_cw_res_fwd(const void* aimg, const evae_conv_desc_t* d, const float* w, const float* b, const float* x, float* out_f, void* oimg,
                    void* ws, size_t ws_bytes, evae_stream_t stream);
/* evae_cw_res_pack_filters: the filter images of a run of n <= 16 same-shaped blocks in one launch (forward images behind fwd_imgs,
 * data-gradient images behind bwd_imgs, evae_cw_workspace_bytes(d, 5) apart; either may be NULL); evae_cw_res_fwd / _bwd_data then
 * take an image as `ws` with w == NULL. */
int evae_cw_res_pack_filters(const evae_conv_desc_t* d, int n, const void* const* w, void* fwd_imgs, void* bwd_imgs, evae_stream_t stream);
/* A whole run of n <= 16 blocks per call (same launches, n - 1 fewer trips through the host language per direction):
 *   evae_cw_res_run_fwd: block k reads aimg0 / oimg[k - 1] and x0 / out_f[k - 1], writes out_f[k], oimg[k] (oimg[n - 1] may be NULL);
 *   evae_cw_res_run_bwd: blocks n - 1 .. 0: dw[k], db[k] (may be NULL) from the gradient image entering block k (dyimg_top / dximg[k + 1])
 *     and aimgs[k] = the image of ELU(x_k); dx_f[k], dximg[k] (dximg[0] may be NULL); ws: evae_cw_workspace_bytes(d, 7) bytes.
 *   fimgs / bimgs: the filter images of evae_cw_res_pack_filters. */
int evae_cw_res_run_fwd(const evae_conv_desc_t* d, int n, const void* fimgs, const void* const* bias, const void* aimg0, const float* x0,
                        void* const* out_f, void* const* oimg, evae_stream_t stream);
int evae_cw_res_run_bwd(const evae_conv_desc_t* d, int n, const void* bimgs, const void* const* aimgs, const void* dyimg_top, const float* dy_top,
                        void* const* dx_f, void* const* dximg, void* const* dw, void* const* db, void* ws, size_t ws_bytes, evae_stream_t stream);
int evae_cw_res_bwd_data(const void* dyimg, const evae_conv_desc_t* d, const float* w, const void* aimg, const float* dy_f, float* dx_f,
                         void* dximg, void* ws, size_t ws_bytes, evae_stream_t stream);
int evae_cw_bwd_weight_plain(const void* dyimg, const void* ximg, const evae_conv_desc_t* d, float* dw, float* db, void* ws,
                             size_t ws_bytes, evae_stream_t stream);
/* The FIRST layer of a stack (one input channel = the data; GatedConv2d(1, 32, 7, 1, 3) of models/convHVAE_2level.py:21-27): a
 * contraction over <= 49 taps is bound by the bytes it writes, so it runs in exact fp32 (v_mfma_f32_32x32x2_f32) from an fp32
 * window of the input in LDS (no patch matrix) and leaves as the next layer's pixel image + gate.  evae_cw_supported(d, 3 | 4):
 * forward | weight gradient (C == 1, stride 1, 'same' padding, Co % 32 == 0; weight gradient 2 Co <= 64).
 *   evae_cw_first_fwd: x fp32 [N][H][W] -> oimg (rows planar when out_planar), out_s [N H W][Co], optional fp32 copy out_f.
 *   evae_cw_first_bwd_weight: dw [2 Co][K K], db [2 Co] from the merged fp32 gradient dy [N H W][2 Co] (evae_cw_bwd_data_gate's
 *     out_f) and x; partial planes per block added in block order (deterministic). */
int evae_cw_first_fwd(const float* x, const evae_conv_desc_t* d, const float* wh, const float* bh, const float* wg, const float* bg,
                      void* oimg, int out_planar, float* out_s, float* out_f, evae_stream_t stream);
size_t evae_cw_first_workspace_bytes(void);
int evae_cw_first_bwd_weight(const float* dy, const float* x, const evae_conv_desc_t* d, float* dw, float* db, void* ws,
                             size_t ws_bytes, evae_stream_t stream);

/* Plain 3 x 3 convolutions ('same' padding, stride 1 or 2) on pixel images: fully_conv's weight-normed convolutions outside its residual
 * runs and its output head (reference models/fully_conv.py:41-58).  d->C / d->Co are the REAL channel counts (any: 3 -> 48, 48 -> 3);
 * an image carries them rounded up to 16 with zeros above.  evae_cw_plain_supported(d, what): 0 forward, 1 data gradient (C % 8 == 0),
 * 2 weight gradient (= evae_cw_bwd_weight_plain on a descriptor with both counts rounded up to 16; slice dw / db).
 *   evae_cw_pack_image_ex: fp32 -> image for x with Cx <= C real channels, rows of ldx floats (channels-last) or contiguous NCHW planes
 *     (nchw), optionally times ELU'(pre) given aux = ELU(pre) (channels-last rows of lda floats); flags as evae_cw_pack_image.
 *   evae_cw_plain_fwd: y = conv(x, w) + b; act bit 0: y = ELU(y), bit 1: the image holds ELU(y) (a residual run reads it);
 *     out_f [N OH OW][ldo] (ldo >= Co rounded up to 8, zeros above Co) and / or oimg.
 *   evae_cw_plain_bwd_data: dx_f [N H W][ldx] (ldx >= C rounded up to 8: whole 8-channel pieces, zeros above C) and / or dximg
 *     (C % 16 == 0) from the image of dy (rows planar when dy_planar).
 *   evae_cw_pack_image_ex flag bit 2: x is the half-resolution tensor [N][H / 2][W / 2] -- nn.Upsample(scale_factor=2) in front of the
 *     convolution (models/fully_conv.py:50,54) happens in the pack; evae_cw_upsample2_bwd: its gradient, dx [N][H/2][W/2][C] = the sum
 *     of the four dy [N][H][W][ld] pixels. */
int evae_cw_upsample2_bwd(const float* dy, long long ld, int N, int H, int W, int C, float* dx, evae_stream_t stream);
int evae_cw_plain_supported(const evae_conv_desc_t* d, int what);
size_t evae_cw_plain_workspace_bytes(const evae_conv_desc_t* d, int what);
int evae_cw_pack_image_ex(const float* x, long long ldx, int Cx, int nchw, const float* aux, long long lda, int N, int H, int W, int C,
                          int flags, void* img, evae_stream_t stream);
int evae_cw_plain_fwd(const void* ximg, const evae_conv_desc_t* d, const float* w, const float* b, int act, float* out_f, int ldo,
                      void* oimg, int out_planar, void* ws, size_t ws_bytes, evae_stream_t stream);
int evae_cw_plain_bwd_data(const void* dyimg, int dy_planar, const evae_conv_desc_t* d, const float* w, float* dx_f, int ldx, void* dximg,
                           void* ws, size_t ws_bytes, evae_stream_t stream);

/* Weight normalisation of a SET of filters in one launch (torch.nn.utils.weight_norm over dim 0, the wrapper of every convolution
 * of reference models/fully_conv.py:18,41-58): w_i [rows_i][cols_i] = v_i * (g_i / ||v_i row||), n <= 32 filters a call;
 * _bwd: dv_i, dg_i [rows_i] from dw_i.  Sums in a fixed order (deterministic). */
int evae_weight_norm_set_fwd(int n, const void* const* v, const void* const* g, void* const* w, const int* rows, const int* cols,
                             evae_stream_t stream);
int evae_weight_norm_set_bwd(int n, const void* const* v, const void* const* g, const void* const* dw, void* const* dv, void* const* dg,
                             const int* rows, const int* cols, evae_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Latent sampling and log-densities on [B x zdim] / [B x D] rows.
 * evae_reparam_logq: z = mu + eps*exp(logvar/2) (models/BaseModel.py:79-82, eps supplied by the
 *   caller's RNG) and log q(z|x) = log_normal_diag(z, mu, logvar) (utils/distributions.py:28-33).
 * evae_bernoulli_ll: sum_d x log p + (1-x) log(1-p), p = clamp(mean, 1e-5, 1-1e-5)
 *   (utils/distributions.py:44-51); backward gives d/dmean.
 */
int evae_reparam_logq_fwd(const float* mu, const float* logvar, const float* eps, int B, int zdim,
                          float* z, float* logq, evae_stream_t stream);
int evae_reparam_logq_bwd(const float* mu, const float* logvar, const float* eps, const float* z,
                          const float* dz, const float* dlogq, int B, int zdim,
                          float* dmu, float* dlogvar, evae_stream_t stream);
/* The same with a second upstream gradient of z (dz + dz2, either may be NULL) and the Hardtanh(lo, hi) of the
 * log-variance head (models/VAE.py:25-26) folded in: dlv_pre is the gradient of the head's pre-activation lv_pre. */
int evae_reparam_logq_bwd_hardtanh(const float* mu, const float* logvar, const float* eps, const float* z,
                                   const float* dz, const float* dz2, const float* dlogq, const float* lv_pre,
                                   float lo, float hi, int B, int zdim, float* dmu, float* dlv_pre,
                                   evae_stream_t stream);
/* ... and, in one more block of the same launch, the two single-block launches that sat beside it on the batch rows' chain of a
 * captured step: evae_elbo_assemble (logp, RE, logq -> loss, KL, means; loss = NULL: not wanted) and evae_sum_small
 * (sum_dst[0] = sum of sum_src[0 .. sum_n); sum_dst = NULL: not wanted).  Same arithmetic and summation order as the three. */
int evae_reparam_logq_bwd_hardtanh_tail(const float* mu, const float* logvar, const float* eps, const float* z, const float* dz,
                                        const float* dz2, const float* dlogq, const float* lv_pre, float lo, float hi, int B,
                                        int zdim, float* dmu, float* dlv_pre, const float* logp, const float* RE,
                                        const float* logq, const float* beta_dev, float beta_host, float* loss, float* KL,
                                        float* means, const float* sum_src, int sum_n, float* sum_dst, evae_stream_t stream);
int evae_log_normal_diag_fwd(const float* x, const float* mu, const float* logvar, int B, int zdim,
                             float* out, evae_stream_t stream);
int evae_log_normal_diag_bwd(const float* x, const float* mu, const float* logvar, const float* dout,
                             int B, int zdim, float* dx, float* dmu, float* dlogvar,
                             evae_stream_t stream);
/* ELBO assembly (models/BaseModel.py:71-75): KL = logq - logp, loss = beta*KL - RE, optional batch means
 * means[3] = (mean loss, mean RE, mean KL).  beta is read from `beta_dev` when non-NULL (graph-captured
 * steps), else `beta_host`.  evae_elbo_bwd turns upstream gradients of (loss, RE, KL) -- each NULL, a scalar
 * (n = 1: gradient of the batch mean) or a [B] vector -- into per-row coefficients cRE = d/dRE, cKL = d/dKL. */
int evae_elbo_fwd(const float* RE, const float* logq, const float* logp, const float* beta_dev,
                  float beta_host, int B, float* loss, float* KL, float* means, evae_stream_t stream);
/* Two latent layers (models/AbsHModel.py:88-106): KL = (logq1 - logp1) + (logq2 - logp2), in that grouping; the rest as above
 * (evae_elbo_bwd's cKL is the gradient of both logq, neg_cKL of both logp). */
int evae_elbo2_fwd(const float* RE, const float* logq1, const float* logp1, const float* logq2, const float* logp2,
                   const float* beta_dev, float beta_host, int B, float* loss, float* KL, float* means, evae_stream_t stream);
int evae_elbo_bwd(const float* dloss, int n_dloss, const float* dRE, int n_dRE, const float* dKL, int n_dKL,
                  const float* beta_dev, float beta_host, int B, float* cRE, float* cKL, float* neg_cKL,
                  evae_stream_t stream);
/* Running epoch statistics on the device (utils/training.py:41-46 keeps train_loss / train_re / train_kl as
 * host floats read back every step): step3 = (loss, -re, kl) of this step, totals3 += step3.  One launch. */
int evae_step_stats_add(const float* loss, const float* re, const float* kl, float* step3, float* totals3,
                        evae_stream_t stream);
/* ----------------------------------------------------------------------------------------------
 * Pre-split bf16 operand images ("p6", csrc/evae_p6_image.h, csrc/evae_gemm_p6.h): the fp32 GEMMs of the exemplar rows'
 * chain -- utils/nn.py:44-69 GatedDense forward, its data gradient and its weight gradient, at models/BaseModel.py:243-248's
 * C = 25 000 rows -- on the bf16 matrix pipe with fp32 accuracy (every element = three round-to-nearest bf16 terms, six of
 * the nine partial products, fp32 accumulation) and NO splitting inside the GEMMs: the producer of an operand writes it once
 * as an image of its TRANSPOSE (rows = the tensor's columns, k = its batch rows), which is the weight gradient's operand as
 * it stands and which the forward / data-gradient kernels read through the LDS transpose read.
 *   image geometry  evae_p6_nks(K) k-steps for a contraction of K; evae_p6_nks_rows(M) for one along M batch rows;
 *                   evae_p6_image_bytes(rows, nks).  A buffer is zero-filled ONCE (padding rows / k are never written).
 *   producers       evae_gated_dense_fwd_timg / evae_gated_dense_fwd_u8_timg (a gated layer's output), evae_dense_bwd_data_timg
 *                   (the gate-fused data gradient's [dh | dg]); weights: evae_p6_pack_rows (forward: [h | g] pair order),
 *                   evae_p6_pack_cols (data gradient: W^T with the banks stacked along the contraction); evae_p6_fill_row
 *                   (the all-ones row behind x's columns that carries the bias gradient).
 *   consumers       evae_gated_dense_fwd_p6t, evae_dense_bwd_data_p6t, evae_dense_bwd_weight_p6.
 * t_row0 / t_kbase: first image row / first k index (a multiple of 8) this launch's columns / rows map to: the exemplar rows
 * and the batch rows of a step write disjoint k ranges of the same images. */
/* Layers over a batch-sized number of rows (and hidden-width layers up to max_rows rows, default 4096 / EVAE_THIN_ROWS) run as ONE
 * launch instead of a split-K GEMM + finish (csrc/evae_thin.h; evae_gated_dense_fwd, evae_linear_fwd, evae_dense_bwd_data*,
 * evae_heads_reparam_fwd / _density_fwd dispatch there).  max_rows < 0: query.  Returns the value in force. */
int evae_thin_configure(int max_rows);
int evae_p6_nks(int K);
int evae_p6_nks_rows(int M);
size_t evae_p6_image_bytes(int rows, int nks);
int evae_gemm_p6_applies(int M, int N, int gated);
int evae_p6_pack_rows(const float* x, const float* x2 /* gated: bank g */, int R, int K, long long ld, int gated, void* img,
                      size_t img_bytes, evae_stream_t stream);
int evae_p6_pack_cols(const float* x, const float* x2 /* or NULL */, int Kd, int R, long long ld, int ones_row /* or -1 */,
                      int nks, void* img, size_t img_bytes, evae_stream_t stream);
int evae_p6_fill_row(void* img, int nks, int row, float value, int k_begin, int k_end, evae_stream_t stream);
int evae_gated_dense_fwd_timg(const float* x, const int64_t* rows, int M, int K, int ldx, const float* wh, const float* bh,
                              const float* wg, const float* bg, int N, float* out, float* save_h, float* save_s,
                              void* timg, int t_nks, int t_row0, int t_kbase, void* ws, size_t ws_bytes, evae_stream_t stream);
int evae_gated_dense_fwd_u8_timg(const unsigned char* x, const int64_t* rows, int M, int K, long long ldx, float x_scale,
                                 const void* prepared, const float* bh, const float* bg, int N, float* out, float* save_s,
                                 void* timg, int t_nks, int t_row0, int t_kbase, evae_stream_t stream);
/* as evae_dense_bwd_data_wt with the gate epilogue (out_prev, s_prev != NULL); dx_or_dh == NULL: image only */
int evae_dense_bwd_data_timg(const float* dy1, const float* w1, const float* dy2, const float* w2, int M, int N, int ldy, int K,
                             const float* out_prev, const float* s_prev, float* dx_or_dh, float* dg, int ldo,
                             const float* wT /* or NULL */, void* timg, int t_nks, int t_row0, int t_kbase, void* ws,
                             size_t ws_bytes, evae_stream_t stream);
int evae_gated_dense_fwd_p6t(const void* xT_img, int x_nks, int M, int K, const void* w_img, const float* bh, const float* bg,
                             int N, float* out /* [M x N] */, float* save_s /* [M x N] or NULL */, evae_stream_t stream);
/* u8_img != NULL: (dh, dg) as the tile images of evae_dense_bwd_weight_u8 (as evae_dense_bwd_data_img); else fp32 dh / dg */
int evae_dense_bwd_data_p6t(const void* dyT_img, int dy_nks, int M, int N, const void* wT_img, int K, const float* out_prev,
                            const float* s_prev, float* dh, float* dg, int ldo, void* u8_img, int u8_nslab, int u8_mbase,
                            evae_stream_t stream);
size_t evae_dense_bwd_weight_p6_workspace_bytes(int nks, int N, int K);
int evae_dense_bwd_weight_p6(const void* dyT_img, const void* xT_img, int nks, int N, int K, float* dw /* [N x K] */,
                             float* db /* [N] or NULL: then x^T's image carries the ones row K */, void* ws, size_t ws_bytes,
                             evae_stream_t stream);
/* The exemplar prior's ONE log-variance (models/BaseModel.py:25-26) as the row the prior kernels read (the reference's
 * `center_log_variance[0, :]`, :101) and back: dst[0..n) = src[0]; out[0] = sum of x[0..n) in a fixed order.  One wave each. */
int evae_broadcast_scalar(const float* src, float* dst, int n, evae_stream_t stream);
int evae_sum_small(const float* x, int n, float* out, evae_stream_t stream);
/* Head of a training step in one launch (utils/training.py:27-31 + models/BaseModel.py:79-81): gather the batch rows
 * idx[b] of the device-resident dataset, binarise them (x = 1 with probability data, the `torch.bernoulli(data)` of
 * dynamic binarisation) or copy them (binarize = 0), and draw eps ~ N(0, 1) [B x zdim] (eps_out may be NULL).
 * Randomness is counter-based (Philox4x32-10): seed_ctr is a DEVICE array {seed, step counter}; the same (seed, counter)
 * always yields the same draws, so a replayed hipGraph advances by updating the counter in place. */
int evae_batch_prologue(const float* data, int64_t ldd, const int64_t* idx, int B, int D, int binarize,
                        const int64_t* seed_ctr /* device [2] */, float* x_out, int64_t ldx,
                        float* eps_out /* [B x zdim] or NULL */, int zdim, evae_stream_t stream);
/* The same on a uint8-resident store (pixel = byte / x_div, e.g. 255): writes the fp32 batch x_out AND its bytes into `stage`
 * (the staging rows of the store: 255 / 0 when binarised), where evae_gated_dense_fwd_u8 / evae_dense_bwd_weight_u8 gather them. */
int evae_batch_prologue_u8(const unsigned char* data, int64_t ldd, const int64_t* idx, int B, int D, int binarize,
                           const int64_t* seed_ctr, float x_div, float* x_out, int64_t ldx, unsigned char* stage, int64_t lds,
                           float* eps_out, int zdim, evae_stream_t stream);
/* The same, and in the same launch the weight split of evae_dense_u8_prepare (wh, wg [N x K] -> prepared): the head of a
 * training step on the byte store is one launch (exemplar-vae_amd/evae/graph.py); replaces utils/training.py:27-31 +
 * models/BaseModel.py:79-81 as evae_batch_prologue does. */
/* jobs (host array, at most two): weight transpositions done by further blocks of the same launch -- dst[p][k][n] = w_p[n][k]
 * with row stride ldt = evae_dense_bwd_data_wt_ld(N), the buffer evae_dense_bwd_data_wt takes: the backward pass of the step
 * then has no transposition launches. */
typedef struct { const float* w1; const float* w2; float* dst; int N, K, ldt; } evae_wt_job_t;
int evae_batch_prologue_u8_prepare(const unsigned char* data, int64_t ldd, const int64_t* idx, int B, int D, int binarize,
                                   const int64_t* seed_ctr, float x_div, float* x_out, int64_t ldx, unsigned char* stage,
                                   int64_t lds, float* eps_out, int zdim, const float* wh, const float* wg, int N, int K,
                                   void* prepared, size_t prepared_bytes, const evae_wt_job_t* jobs, int njobs,
                                   evae_stream_t stream);
/* ... and the hand-over of a captured step's control block in the same launch (exemplar-vae_amd/evae/graph.py): the host
 * uploads step n's block into staging block n & 1 (evae_ctl_upload with d_ctl = NULL); the launch reads the batch indices
 * (idx_word) and the generator's (seed, counter) pair (seed_word; 8-byte words from the start of a block) from the staging block
 * that the parity word state[0] names (0 / 1) and copies that block into `ctl`, which every later launch of the step reads.
 * The launch does not write the parity: the step's last launch flips it (evae_adam_normgrad_step_stats' toggle) or the caller
 * sets it.  bytes: a multiple of 16.  wh / wg / prepared may be NULL (no weight split). */
typedef struct { const void* stage0; const void* stage1; void* ctl; size_t bytes; const int* state; size_t idx_word, seed_word; } evae_ctl_job_t;
/* packs (host array, at most two): weight images of the pre-split GEMMs built by further blocks of the same launch --
 * cols == 0: evae_p6_pack_rows(x, x2, R, K, ld, gated = flag, img, img_bytes); cols != 0: evae_p6_pack_cols(x, x2, Kd = K, R, ld,
 * ones_row = flag, nks, img, img_bytes) -- the same images, bit for bit. */
typedef struct { const float* x; const float* x2; void* img; size_t img_bytes; long long ld; int cols, R, K, flag, nks; } evae_p6_pack_job_t;
int evae_batch_prologue_u8_step(const unsigned char* data, int64_t ldd, int B, int D, int binarize, float x_div, float* x_out,
                                int64_t ldx, unsigned char* stage, int64_t lds, float* eps_out, int zdim, const float* wh,
                                const float* wg, int N, int K, void* prepared, size_t prepared_bytes, const evae_wt_job_t* jobs,
                                int njobs, const evae_ctl_job_t* ctl, const evae_p6_pack_job_t* packs, int npacks,
                                evae_stream_t stream);
/* evae_log_normal_diag_bwd with the Hardtanh(lo, hi) of a log-variance head folded in: dlv_pre is the gradient of its pre-activation */
int evae_log_normal_diag_bwd_hardtanh(const float* x, const float* mu, const float* logvar, const float* lv_pre, float lo, float hi,
                                      const float* dout, int B, int zdim, float* dx, float* dmu, float* dlv_pre, evae_stream_t stream);
int evae_bernoulli_ll_fwd(const float* x, const float* mean, int B, int D, float* out,
                          evae_stream_t stream);
int evae_bernoulli_ll_bwd(const float* x, const float* mean, const float* dout, int B, int D,
                          float* dmean, evae_stream_t stream);
/* 256-bin discretised logistic log-likelihood summed over dim 1 (utils/distributions.py:54-66: continuous inputs with
 * use_logit = False, the reconstruction term of config 5): out_i = sum_k log(sigmoid(xs + 1/(256 s)) - sigmoid(xs) + 1e-7),
 * xs = (floor(256 x)/256 - mean)/s, s = exp(logvar).  logvar is [B x D], or ONE device value when lv_scalar != 0
 * (models/fully_conv.py's decoder_logstd, models/AbsModel.py:35-37).  Backward of sum_i dout_i out_i: dmean [B x D] (or NULL),
 * dlogvar [B x D] -- or [1] for a scalar log-variance, reduced in a fixed order through ws_rows [B] -- (or NULL). */
int evae_log_logistic256_fwd(const float* x, const float* mean, const float* logvar, int lv_scalar, int B, int D,
                             float* out /* [B] */, evae_stream_t stream);
int evae_log_logistic256_bwd(const float* x, const float* mean, const float* logvar, int lv_scalar,
                             const float* dout /* [B] */, int B, int D, float* dmean, float* dlogvar,
                             float* ws_rows /* [B], scalar log-variance only */, evae_stream_t stream);
/* d/dpre when mean = sigmoid(pre) (the p_x_mean head of models/BaseModel.py:28-29): the two steps in one launch */
int evae_bernoulli_sigmoid_bwd(const float* x, const float* mean, const float* dout, int B, int D,
                               float* dpre, evae_stream_t stream);
/* The reconstruction term of a step whose backward is loss.backward(ones) on the batch means (utils/training.py:33-36), one
 * launch for three: RE [B] (evae_bernoulli_ll_fwd), the coefficient vectors of evae_elbo_bwd for a unit upstream gradient of the
 * mean loss (cRE = -1/B, cKL = beta/B, neg_cKL = -beta/B; beta from device memory when beta_dev != NULL) and
 * dpre = evae_bernoulli_sigmoid_bwd(x, mean, cRE) -- the same arithmetic, element by element. */
int evae_bernoulli_unit_step(const float* x, const float* mean, int B, int D, const float* beta_dev, float beta_host, float* RE,
                             float* cRE, float* cKL, float* neg_cKL, float* dpre, evae_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * AdamNormGrad (utils/optimizer.py:32-80): g <- g/(||g||_2 + 1e-7) per tensor, then Adam with eps
 * added after the sqrt and a bias-corrected step size.  One launch pair updates ALL tensors:
 * `ptrs` is a device array of n_tensors records {param, grad, exp_avg, exp_avg_sq, numel}.
 * The bias-corrected step size lr*sqrt(1-b2^t)/(1-b1^t) is computed on the host from `step`, or -- when
 * the step is replayed from a hipGraph, where kernel arguments are frozen -- read from the device scalar
 * `step_size_dev`, which the caller updates before each replay.
 */
typedef struct {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t numel;
} evae_adam_tensor_t;
size_t evae_adam_normgrad_workspace_bytes(int n_tensors);
int evae_adam_normgrad_step(const evae_adam_tensor_t* tensors /* device */, int n_tensors,
                            int64_t max_numel, int step, double lr, double beta1, double beta2, double eps,
                            double weight_decay, const float* step_size_dev /* device scalar or NULL */,
                            void* ws, size_t ws_bytes, evae_stream_t stream);
/* The same, and in its last launch the statistics of evae_step_stats_add (step3 = (loss, -re, kl), totals3 += step3; totals3
 * may be NULL): the tail of a captured training step is one launch shorter (utils/training.py:41-46).  toggle (may be NULL): a
 * device word that launch XORs with 1 -- the parity of the control block's staging blocks (evae_batch_prologue_u8_step). */
int evae_adam_normgrad_step_stats(const evae_adam_tensor_t* tensors /* device */, int n_tensors, int64_t max_numel, int step,
                                  double lr, double beta1, double beta2, double eps, double weight_decay,
                                  const float* step_size_dev, void* ws, size_t ws_bytes, const float* loss, const float* re,
                                  const float* kl, float* step3, float* totals3, int* toggle, evae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EVAE_HIP_H */
