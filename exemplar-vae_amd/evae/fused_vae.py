"""One-node autograd implementation of the `vae` exact-prior training loss (the body of
models/BaseModel.py:65-77 + AbsModel.py:13-19,44-49 + BaseModel.py:243-248 in the reference).

Same arithmetic as the modular path (utils.nn.GatedDense, evae.ops.PriorLogP, ...), arranged for the
hardware:

* the B batch rows ride along the C exemplar rows through the encoder: the binarised batch sits in staging
  rows behind the HBM-resident dataset (the captured step gathers it there itself, evae/graph.py) and ONE
  row-gathered GEMM per encoder layer serves all C + B rows (forward and weight gradient), so the 100-row batch
  path costs three thin decoder layers instead of a second pass over every encoder layer;
* every GatedDense backward uses the merged [dh | dg] buffer (one weight-gradient GEMM per layer) and
  the gate derivative of the layer below is applied in the epilogue of the data-gradient GEMM;
* one autograd node instead of ~25: 61 kernel launches per step and no per-op autograd bookkeeping;
* two streams scheduled by hand: the exemplar-prior chain (with its collectives when sharded) and the big
  GEMMs on the main stream, the decoder chain of the batch rows and every weight gradient nobody waits for on
  the side stream (DESIGN.md section 4 for what was measured to arrive at this order of issue).

With torch.distributed active and shard=True the exemplar rows are this rank's shard: the per-row
partials (max, sumexp, nmask) are all-gathered and merged, dz / dlogvar are sum-all-reduced and
dcentres is scaled by the world size (see evae/shard.py for why)."""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib, ops, shard

ACT_NONE, ACT_SIGMOID, ACT_HARDTANH = ops.ACT_NONE, ops.ACT_SIGMOID, ops.ACT_HARDTANH

# generation of the staging rows behind a dataset buffer (keyed by its address): a second fused forward overwrites the rows a
# pending backward would gather its batch from -- the backward checks that its forward was the last one to stage
UNIT_UPSTREAM = [False]   # set by evae/graph.py around its step: the only backward is loss.backward(ones) on the batch mean
WT_DONE = {}       # (wm, w2h, w2g) pointers -> (wT of the head, wT of layer 2) the step's head launch just wrote
P6_LAST = [None]   # (w2h ptr, w2g ptr, H, forward image, data-gradient image) of the last step that took the pre-split layer 2
P6_DONE = {}       # (w2h ptr, w2g ptr) -> True when the step's head launch just built both images (evae/graph.py)
PREP_DONE = {}     # prepared-weights buffer -> (w1h, w1g) pointers it was just filled from, by the step's head launch
_STAGE_GEN = {}
_XT_GEN = {}       # workspace pointer -> generation of the transposed byte rows the forward pass left there
_ZEROED = {}      # workspace name -> (address, shape) it was last zero-filled for
# r06: element-wise launches of the batch rows' chain merged into their neighbours (a replayed graph node costs the host 3.6 us):
# bit 0 the log-variance row's broadcast rides in the heads' launch; bit 1 RE + the unit coefficients + the sigmoid head's gradient
# are one launch (evae_bernoulli_unit_step); bits 2 / 3 the ELBO's assembly / the log-variance gradient's sum ride in the
# reparameterisation's backward (evae_reparam_logq_bwd_hardtanh_tail): five nodes less per step, bit-equal results.  Un-profiled
# (tools/ab_node_merge.sh, profiles/r06_ab/node_merge.txt): C = 200 0.199 -> 0.189 ms, c1 0.217 -> 0.208, c2a 0.315 -> 0.310; at c2
# (GPU-bound) bits 0, 1, 3 are neutral and bit 2 costs 5 us -- the assembly then sits ON the chain the layer-2 weight gradient waits
# for instead of in front of it --, so it is merged in host-bound steps only.
def node_merge(Cl, B):
    e = os.environ.get("EVAE_NODE_MERGE")
    if e is not None:
        return int(e)
    return 15 if Cl + B <= 8192 else 11


ONE_STREAM = [os.environ.get("EVAE_ONE_STREAM", "0") == "1"]      # experiment: the whole step on the caller's stream
_APPROX_ROWS = {}  # (device, slots, batch, dataset rows) -> the approximate-prior step's gather list (static tail)

# schedule switches (bit mask; tools/chain_bench.sh sweeps them): 1 = the prior's dz' / dlogvar reduction on the side stream,
# 2 = the finish launch of encoder layer 2's weight gradient on the side stream, 4 = byte store: the layer-2 data gradient writes
# the three-term bf16 tile images of [dh | dg]^T itself (evae_dense_bwd_data_img) instead of an fp32 buffer + transposing pre-pass:
# measured 0.7741 vs 0.7754 ms -- the 38 us pre-pass goes, but the epilogue's 8-byte scattered stores cost the GEMM +15 us and the
# weight-gradient GEMM reads the images +11 us slower; kept as an option, off.  1 and 2 OFF: measured at config 2 (r02), 0 ->
# 0.975 ms/step, 1 -> 1.002, 2 -> 1.085, 3 -> 1.076 -- a launch that runs beside a CU-filling GEMM costs that GEMM more than
# the launch saves on the main stream.  8 = byte store: the two bandwidth-bound pre-passes of the first layer's weight gradient
# (evae_dense_bwd_weight_u8_phased) on the side stream beside layer 2's weight-gradient GEMM: 0.772 / 0.800 ms vs 0.744 -- the same
# lesson once more, off.
SCHED = int(os.environ.get("EVAE_SCHED", "0"))
# byte store: encoder layer 1's (dh, dg) leave the layer-2 data gradient as the bf16 tile images of its weight gradient
# (evae_dense_bwd_data_img; r03: 16-byte stores after a lane-pair exchange, on the split-bf16 kernel) -- no fp32 [Mp x 2H]
# buffer, no 36-us split / transposition pre-pass.  EVAE_IMG_DGRAD=0: the fp32 buffer + pre-pass
FINISH_GROUP_HEAD = os.environ.get("EVAE_FINISH_GROUP_HEAD", "0") == "1"   # ... the mean head's too (needs the side stream joined first)
# One finish launch for the step's split-K weight gradients (evae_dense_bwd_weight_finish_group) instead of one behind each GEMM.
# Measured r03 and OFF: launches that follow each other on ONE stream of the replayed graph start back to back (gap 0.0 us in
# profiles/r03_step_timeline_C25000.txt; the ~6 us gaps sit at cross-stream joins), so the merged launch saves nothing -- it ran
# 23.4 us against 8.7 + 10.1 for the two it replaces (c2 0.6446-0.6468 vs 0.641-0.645 ms); with the mean head's finish in it as
# well the side stream has to be joined earlier: c2 +10 us, C = 200 +17 us.
FINISH_GROUP = os.environ.get("EVAE_FINISH_GROUP", "0") == "1"
# the mean head's weight gradient (a 29-us streaming launch alone) right behind the batch rows' reparameterisation backward instead of
# at the end of the side stream's chain: it then runs beside the HBM-bound head data gradient, not beside layer 2's data gradient
HEADW_EARLY = int(os.environ.get("EVAE_HEADW_EARLY", "0"))   # r06: OFF -- with the head gradient in front of them the batch rows' thin data gradients (which the main stream's layer-2 weight gradient waits for) finished 20 us after layer 2's data gradient: 0.553 -> 0.542 ms
ELBO_SPLIT = os.environ.get("EVAE_ELBO_SPLIT", "0") != "0"         # captured step: merge on the prior's stream, ELBO assembly beside it
# (r04 A/B, profiles/r04_ab: the 7-block merge launch takes 10.7 us against the one-block merge + ELBO's 11.6 and the join it
#  saves comes back as a 9-us gap in front of the prior's backward: c2 0.616-0.621 -> 0.623-0.629 ms.  Off; what replaced it:)
# captured step on one device: the prior's forward partials, their merge and its backward as ONE launch in the forward pass
# (csrc/evae_prior_train.h), the ELBO assembly on the side stream
PRIOR_TRAIN = os.environ.get("EVAE_PRIOR_TRAIN", "1") != "0"
GROUP_LEAVES = os.environ.get("EVAE_GROUP_LEAVES", "1") != "0"      # the batch rows' four leaf weight gradients as one launch
IMG_DGRAD = os.environ.get("EVAE_IMG_DGRAD", "1") != "0" or bool(SCHED & 4)
THIN_ROWS = 1024     # batch rows up to here take the fp32 split-K kernel for the first layer even on the byte store
# byte store, exact prior, a machine-filling exemplar count: encoder layer 2's three GEMMs over pre-split bf16 operand images
# (csrc/evae_gemm_p6.h) -- layer 1's output and the heads' (dh2, dg2) leave their producers' epilogues as the images of their
# transposes, the weights are split once per step; no fp32 operand is re-split inside a GEMM.  EVAE_P6=0: the split-bf16 /
# fp32 kernels of r03
P6 = os.environ.get("EVAE_P6", "1") != "0"
_P6_READY = {}     # image buffer pointer -> (shape key) it was zero-filled (and its ones row written) for
_ONES = {}         # device -> a one-element tensor holding 1.0 (the unit upstream gradient of a captured step)

PARAM_ORDER = [
    "prior_log_variance",
    "p_x_mean.linear.weight", "p_x_mean.linear.bias",
    "q_z_layers.0.h.weight", "q_z_layers.0.h.bias", "q_z_layers.0.g.weight", "q_z_layers.0.g.bias",
    "q_z_layers.1.h.weight", "q_z_layers.1.h.bias", "q_z_layers.1.g.weight", "q_z_layers.1.g.bias",
    "q_z_mean.weight", "q_z_mean.bias",
    "q_z_logvar.linear.weight", "q_z_logvar.linear.bias",
    "p_x_layers.0.h.weight", "p_x_layers.0.h.bias", "p_x_layers.0.g.weight", "p_x_layers.0.g.bias",
    "p_x_layers.1.h.weight", "p_x_layers.1.h.bias", "p_x_layers.1.g.weight", "p_x_layers.1.g.bias",
]


# The duplicates among the exemplar draw, set by the captured step's runner around its forward pass (evae/graph.py): None, or
# (draws [C] int64, inv [C] int64, rep [Cl] int64, mult [Cl] fp32) with ex_idx = the Cl DISTINCT rows (padded, multiplicity 0).  The
# encoder then runs over the distinct rows only; the prior sees all C draws -- centres gathered through inv, leave-one-out mask on
# the draws' indices -- and a distinct row's head gradient is mult x the gradient of ONE of its draws (duplicates have identical
# centres, hence identical prior gradients).  Same loss, same gradients as encoding every draw (reference models/BaseModel.py:243-254).
DEDUP = [None]


def _vp(v):
    if v is None:
        return None
    return C.c_void_p(v if isinstance(v, int) else v.data_ptr())


class _K:
    """Thin raw launchers over the C ABI (no autograd, caller-owned outputs)."""

    def __init__(self, device, stream=None, suffix=""):
        """Launches go to `stream` (default: the current one) and take their split-K workspaces from buffers named with
        `suffix`, so that two launchers on two streams never share scratch memory."""
        self.lib = _lib.load()
        self.dev = device
        self.st = ops._stream() if stream is None else C.c_void_p(stream.cuda_stream)
        self.sfx = suffix

    def ws(self, name, nbytes):
        return ops._workspace(name + self.sfx, nbytes, self.dev)

    _side = {}

    def side_stream(self):
        """Second stream for the decoder chain of the batch rows: it only meets the exemplar-prior chain at the ELBO (forward)
        and at dz (backward), so the two chains of small launches run side by side, captured as parallel branches of the
        step's hipGraph.  Collectives of the sharded prior stay on the main stream."""
        if ONE_STREAM[0]:
            return torch.cuda.current_stream(self.dev)
        key = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        st = _K._side.get(key)
        if st is None:
            st = _K._side[key] = torch.cuda.Stream(device=self.dev, priority=int(os.environ.get("EVAE_SIDE_PRIORITY", "-1")))
        return st

    def gated_fwd(self, x, rows, M, K, ldx, wh, bh, wg, bg, N, out, h, s):
        nb = self.lib.evae_dense_fwd_workspace_bytes(M, K, N, 1)
        w = self.ws("fwd", nb)
        ex_, pipe_ = ops.gemm_pipe(M, N, True, 2.0 * M * K * 2 * N) if rows is None else (None, "fp32-mfma")
        ops.probed("gated_dense_fwd M=%d K=%d N=%d%s" % (M, K, N, " (row gather)" if rows is not None else ""), 2.0 * M * K * 2 * N,
                   lambda: _lib.check(self.lib.evae_gated_dense_fwd(_vp(x), _vp(rows), M, K, ldx, _vp(wh), _vp(bh), _vp(wg), _vp(bg),
                                                                    N, _vp(out), _vp(h), _vp(s), _vp(w), w.numel(), self.st),
                                      "gated_fwd"), executed=ex_, pipe=pipe_)

    def linear_fwd(self, x, M, K, ldx, w_, b, N, act, lo, hi, y, pre):
        nb = self.lib.evae_dense_fwd_workspace_bytes(M, K, N, 0)
        w = self.ws("fwd", nb)
        _lib.check(self.lib.evae_linear_fwd(_vp(x), None, M, K, ldx, _vp(w_), _vp(b), N, act, lo, hi, _vp(y), _vp(pre),
                                            _vp(w), w.numel(), self.st), "linear_fwd")

    def bwd_data(self, dy1, w1, dy2, w2, M, N, ldy, K, out_prev, s_prev, out, dg, ldo, wT=None):
        """wT: the transposed weights prepared by the step's head launch (evae_dense_bwd_data_wt), or None"""
        nb = self.lib.evae_dense_bwd_data_workspace_bytes(M, N, K, 2 if dy2 is not None else 1)
        w = self.ws("dgrad", nb)
        fl_ = 2.0 * M * N * K * (2 if dy2 is not None else 1)
        ex_, pipe_ = ops.gemm_pipe(M, K, False, fl_) if (N % 4 == 0 and ldy % 4 == 0) else (None, "fp32-mfma")
        ops.probed("dense_bwd_data M=%d N=%d%s K=%d%s" % (M, N, "+%d" % N if dy2 is not None else "", K,
                                                          " (gate-backward epilogue)" if out_prev is not None else ""),
                   fl_,
                   lambda: _lib.check(self.lib.evae_dense_bwd_data_wt(_vp(dy1), _vp(w1), _vp(dy2), _vp(w2), M, N, ldy, K, _vp(out_prev),
                                                                      _vp(s_prev), _vp(out), _vp(dg), ldo, _vp(wT), _vp(w), w.numel(),
                                                                      self.st),
                                      "bwd_data"), executed=ex_, pipe=pipe_)

    def bwd_weight(self, dy, M, N, ldy, x, rows, K, ldx, dw, db, phase=0, ws_name="wgrad", finish_on=None):
        """phase 1 / 2: the split-K GEMM and its finish as separate calls (finish_on = the launcher whose stream runs it)"""
        nb = self.lib.evae_dense_bwd_weight_workspace_bytes(M, N, K)
        w = ops._workspace(ws_name + self.sfx, nb, self.dev)
        if phase == 0:
            # (a narrow output over many rows -- the heads' [40 x 300] over all C + B rows -- streams its operands once: an HBM entry)
            ops.probed("dense_bwd_weight M=%d N=%d K=%d (+db, split-K GEMM + finish)" % (M, N, K), 2.0 * M * N * K,
                       hbm_bytes=(4.0 * M * (N + K)) if (N <= 64 and M >= 2048 and rows is None) else None, fn=
                       lambda: _lib.check(self.lib.evae_dense_bwd_weight(_vp(dy), M, N, ldy, _vp(x), _vp(rows), K, ldx, _vp(dw),
                                                                         _vp(db), 0, _vp(w), w.numel(), self.st), "bwd_weight"))
        else:
            st = self.st if finish_on is None else finish_on.st
            call = lambda: _lib.check(self.lib.evae_dense_bwd_weight_phased(_vp(dy), M, N, ldy, _vp(x), _vp(rows), K, ldx, _vp(dw),
                                                                            _vp(db), 0, _vp(w), w.numel(), phase, st), "bwd_weight_phased")
            if phase == 1:
                ops.probed("dense_bwd_weight M=%d N=%d K=%d (+db, split-K GEMM; its planes are summed by the step's last grouped launch)"
                           % (M, N, K), 2.0 * M * N * K, call)
            else:
                call()


class VaeExactLoss(torch.autograd.Function):
    """forward(x [B x D] binarised batch, x_idx [B], data_ext [(N + pad) x D] resident dataset with staging
    rows, n_data, ex_idx_local [Cl] int64 (device), c_total, eps [B x z], beta, sharded, *params (PARAM_ORDER))
    -> (loss [B], RE [B], KL [B])."""

    @staticmethod
    def forward(ctx, x, x_idx, data_ext, n_data, ex_idx, c_total, eps, beta, sharded, no_mask, average, rows_ext, staged,
                approx_cache, approx_k, *params):
        (plv, wp, bp, w1h, b1h, w1g, b1g, w2h, b2h, w2g, b2g, wm, bm, wl, bl,
         d1h, e1h, d1g, e1g, d2h, e2h, d2g, e2g) = params
        dev = x.device
        k = _K(dev)
        lib = k.lib
        B, D = x.shape
        # approximate prior (reference models/BaseModel.py:256-271): ex_idx holds the CANDIDATE draw; the exemplar rows are the
        # B * k static slots picked by the top-K over the cached latents of the candidates (repeats masked, evae_select_exemplars)
        approx = approx_cache is not None
        Cl = B * int(approx_k) if approx else ex_idx.numel()
        if approx:
            c_total = Cl
        Mp = Cl + B
        H = w1h.shape[0]
        Z = wm.shape[0]
        f32 = dict(device=dev, dtype=torch.float32)
        x = x.contiguous()
        # stage the batch behind the dataset; one gather list for exemplars + batch.  data_ext is the fp32 copy of the
        # dataset or its uint8 store (models/BaseModel.py::resident_u8): then the first layer runs on the byte kernels
        u8 = data_ext.dtype == torch.uint8
        stage = data_ext[n_data:n_data + B]
        gen = _STAGE_GEN[data_ext.data_ptr()] = _STAGE_GEN.get(data_ext.data_ptr(), 0) + 1
        if u8:
            if not staged:                        # the captured step (evae/graph.py) writes the bytes itself
                stage.copy_(torch.round(x * 255.0))
        elif x.data_ptr() != stage.data_ptr():    # the captured step gathers the batch there itself
            stage.copy_(x)
        # rows_ext: caller-kept [Cl + B] gather list whose head IS ex_idx and whose tail already names the staging rows
        if approx:
            # head: filled behind the top-K below; the tail (the staging rows) never changes -- one buffer per geometry, kept
            # (a second forward before this one's backward is refused by the staging-row generation check, which covers it too)
            rkey = (dev, Cl, B, int(n_data))
            rows = _APPROX_ROWS.get(rkey)
            if rows is None:
                rows = _APPROX_ROWS[rkey] = torch.empty(Cl + B, dtype=torch.int64, device=dev)
                rows[Cl:] = torch.arange(n_data, n_data + B, device=dev)
        elif rows_ext is not None and rows_ext.numel() == Cl + B and rows_ext.data_ptr() == ex_idx.data_ptr():
            rows = rows_ext
        else:
            rows = torch.cat((ex_idx, torch.arange(n_data, n_data + B, device=dev)))
        ldd = data_ext.stride(0)
        main = torch.cuda.current_stream()
        side = k.side_stream()
        kd = _K(dev, stream=side, suffix="_side")        # launcher of the decoder chain
        side.wait_stream(main)
        # ---- encoder.  The Cl exemplar rows on the main stream (the big GEMMs); the B batch rows as thin launches on the
        #      side stream, writing the tail rows of the same activation buffers (the weight gradients later run over all
        #      Cl + B rows in one launch per layer).  The whole batch-row path -- encoder, heads, sampling, decoder,
        #      reconstruction term -- thereby runs BESIDE the exemplar encoder instead of behind it; the two streams meet
        #      at the prior (which needs z) and at the ELBO.
        # a gated layer keeps its output and its gate s for the backward (dg = dout * out * (1 - s)); h is never stored
        A1 = torch.empty((Mp, H), **f32); s1 = torch.empty_like(A1)
        A2 = torch.empty((Mp, H), **f32); s2 = torch.empty_like(A2)
        mean_all = torch.empty((Mp, Z), **f32)
        centres = mean_all[:Cl]
        z_mean = mean_all[Cl:]
        A2b = A2[Cl:]
        logvar = torch.empty((B, Z), **f32); lv_pre = torch.empty_like(logvar)
        z = torch.empty((B, Z), **f32); logq = torch.empty(B, **f32)
        D1 = torch.empty((B, H), **f32); sd1 = torch.empty_like(D1)
        D2 = torch.empty((B, H), **f32); sd2 = torch.empty_like(D2)
        xmean = torch.empty((B, D), **f32)
        RE = torch.empty(B, **f32)
        offb = 4 * Cl                                    # byte offset of the batch rows, per float of row width
        p6 = bool(u8 and P6 and IMG_DGRAD and not approx and Cl > 0 and Cl % 8 == 0 and B <= THIN_ROWS and not (SCHED & 10)
                  and lib.evae_gemm_p6_applies(Cl, H, 1) and lib.evae_gemm_x6_applies(Cl, H, 0))
        if p6:
            # images: h1^T (+ the all-ones row behind its H rows: the bias gradient of layer 2), [dh2 | dg2]^T, both with k along
            # all Cl + B rows; layer 2's weights in forward ([h | g] pairs) and data-gradient (W^T, banks stacked) form
            nks_m = lib.evae_p6_nks_rows(Mp)
            t_h1 = k.ws("p6_h1", lib.evae_p6_image_bytes(H + 1, nks_m))
            t_dq2 = k.ws("p6_dq2", lib.evae_p6_image_bytes(2 * H, nks_m))
            w2_img = k.ws("p6_w2", lib.evae_p6_image_bytes((H + 63) // 64 * 128, lib.evae_p6_nks(H)))
            w2t_img = k.ws("p6_w2t", lib.evae_p6_image_bytes(H, lib.evae_p6_nks(2 * H)))
            for t_, rows_ in ((t_h1, H + 1), (t_dq2, 2 * H)):
                key_ = (t_.numel(), Mp, H)
                if _P6_READY.get(t_.data_ptr()) != key_:     # padding rows / k are never written: zero once per buffer and shape
                    t_.zero_()
                    if rows_ == H + 1:
                        _lib.check(lib.evae_p6_fill_row(_vp(t_), nks_m, H, 1.0, 0, Mp, k.st), "p6_fill_row")
                    _P6_READY[t_.data_ptr()] = key_
        if u8:
            # both first-layer launches read the weights as three bf16 terms in tile order: split once, in front of the fork
            prep = k.ws("u8prep", lib.evae_dense_u8_prepared_bytes(H, D))
            if PREP_DONE.pop(prep.data_ptr(), None) != (w1h.data_ptr(), w1g.data_ptr()):
                # (the captured step's head launch -- evae/graph.py -- does this split beside its batch prologue and says so)
                _lib.check(lib.evae_dense_u8_prepare(_vp(w1h), _vp(w1g), H, D, _vp(prep), prep.numel(), k.st), "u8_prepare")
            side.wait_stream(main)

            def l1_fwd(kk, rows_ptr, M, o):
                fl = 2.0 * M * D * 2 * H
                if p6:        # ... and the output as h1^T's image, rows o / 4 .. of its k range
                    ops.probed("gated_dense_fwd_u8 M=%d K=%d N=%d (uint8 rows, three bf16 terms; output + its pre-split image)" % (M, D, H), fl,
                               lambda: _lib.check(lib.evae_gated_dense_fwd_u8_timg(
                                   _vp(data_ext), _vp(rows_ptr), M, D, ldd, 1.0 / 255.0, _vp(prep), _vp(b1h), _vp(b1g), H,
                                   _vp(A1.data_ptr() + o * H), _vp(s1.data_ptr() + o * H), _vp(t_h1), nks_m, 0, o // 4, kk.st),
                                   "gated_fwd_u8_timg"), executed=3 * fl, pipe="bf16-mfma")
                    return
                ops.probed("gated_dense_fwd_u8 M=%d K=%d N=%d (uint8 rows, three bf16 terms)" % (M, D, H), fl,
                           lambda: _lib.check(lib.evae_gated_dense_fwd_u8(_vp(data_ext), _vp(rows_ptr), M, D, ldd, 1.0 / 255.0, _vp(prep),
                                                                          _vp(b1h), _vp(b1g), H, _vp(A1.data_ptr() + o * H),
                                                                          _vp(s1.data_ptr() + o * H), kk.st), "gated_fwd_u8"),
                           executed=3 * fl, pipe="bf16-mfma")
        else:
            def l1_fwd(kk, rows_ptr, M, o):
                kk.gated_fwd(data_ext, rows_ptr, M, D, ldd, w1h, b1h, w1g, b1g, H, A1.data_ptr() + o * H, None,
                             s1.data_ptr() + o * H)
        if Cl > 0 and not approx:
            l1_fwd(k, rows, Cl, 0)
        xt_early = bool(u8 and not approx and not (SCHED & 64))
        xt_late = xt_early and not (SCHED & 128)
        def xt_gather():
            # the byte layer's weight gradient wants the gathered rows transposed (pixel-major): that needs the gather list
            # only, so it runs in the forward pass on the side stream instead of in the backward's chain of launches -- r03:
            # BEHIND the batch-row chain (it then runs beside the prior / ELBO / prior-backward launches, when the machine is
            # nearly idle) instead of in front of it (where it delayed that chain by its 28 us and shared the exemplar
            # encoder's bandwidth: the side stream, not the main one, was what the prior waited for)
            wq_ = k.ws("wgrad_u8_fused", lib.evae_dense_bwd_weight_u8_workspace_bytes(Mp, 2 * H, D))
            gen_ = _XT_GEN[wq_.data_ptr()] = _XT_GEN.get(wq_.data_ptr(), 0) + 1
            _lib.check(lib.evae_dense_bwd_weight_u8_phased(None, Mp, 2 * H, 2 * H, _vp(data_ext), _vp(rows), D, ldd, 1.0 / 255.0,
                                                           None, None, _vp(wq_), wq_.numel(), 3, kd.st), "bwd_weight_u8(gather)")
            return wq_, gen_
        lv_row = torch.empty(Z, **f32)                     # the prior's log-variance row
        beta_dev = beta if torch.is_tensor(beta) else None
        beta_host = 0.0 if beta_dev is not None else float(beta)
        dd = DEDUP[0]
        if dd is not None and (approx or dd[2].numel() != Cl or dd[3].numel() != Cl):
            raise _lib.EvaeError("fused step: the duplicate tables do not belong to this exemplar set")
        Cp = dd[0].numel() if dd is not None else Cl          # exemplars the prior sees (all draws; this rank's when sharded)
        prior_train = bool(PRIOR_TRAIN and UNIT_UPSTREAM[0] and average and not sharded and Cl > 0 and not ONE_STREAM[0]
                           and ops.prior_train_applies(B, Cp, Z))
        coef = None
        with torch.cuda.stream(side):
            # (on the side stream, in front of everything: with the byte gather moved behind the batch-row chain the MAIN stream's
            # head GEMM is what the prior waits for, and this 5-us launch sat between the two)
            thin_heads = B <= THIN_ROWS and not (SCHED & 1024)
            bc_in_heads = bool((node_merge(Cl, B) & 1) and thin_heads and lib.evae_heads_reparam_fwd_bcast_applies(B, H, Z, H))
            if not bc_in_heads:
                _lib.check(lib.evae_broadcast_scalar(_vp(plv.detach()), _vp(lv_row), Z, kd.st), "broadcast_scalar")
            w2_ready = None
            if p6:
                # layer 2's weights as images (they change every step); the main stream meets them behind the first layer --
                # unless the captured step's head launch built them (evae/graph.py: two launches and one join less)
                P6_LAST[0] = (w2h.data_ptr(), w2g.data_ptr(), H, w2_img, w2t_img)
                if not P6_DONE.pop((w2h.data_ptr(), w2g.data_ptr()), False):
                    _lib.check(lib.evae_p6_pack_rows(_vp(w2h), _vp(w2g), H, H, H, 1, _vp(w2_img), w2_img.numel(), kd.st), "p6_pack_rows")
                    _lib.check(lib.evae_p6_pack_cols(_vp(w2h), _vp(w2g), H, H, H, -1, lib.evae_p6_nks(2 * H), _vp(w2t_img),
                                                     w2t_img.numel(), kd.st), "p6_pack_cols")
                    w2_ready = torch.cuda.Event(); w2_ready.record()
            if xt_early and not xt_late:
                wq, xt_gen = xt_gather()
            if p6:
                # the batch rows' first layer on the fp32 split-K kernel, its finish also writing rows Cl .. of h1^T's image
                nbf = lib.evae_dense_fwd_workspace_bytes(B, D, H, 1)
                wf = kd.ws("fwd", nbf)
                _lib.check(lib.evae_gated_dense_fwd_timg(_vp(x), None, B, D, x.stride(0), _vp(w1h), _vp(b1h), _vp(w1g), _vp(b1g), H,
                                                         _vp(A1.data_ptr() + offb * H), None, _vp(s1.data_ptr() + offb * H), _vp(t_h1),
                                                         nks_m, 0, Cl, _vp(wf), wf.numel(), kd.st), "gated_fwd_timg")
            elif u8 and B <= THIN_ROWS and not (SCHED & 32):
                # a thin launch of the byte kernel walks its 25 K-slabs on five blocks (29 us alone, 66 us beside the exemplar
                # GEMM); the batch is here as fp32 too (x = byte / 255), and the fp32 kernel splits K over the machine
                kd.gated_fwd(x, None, B, D, x.stride(0), w1h, b1h, w1g, b1g, H, A1.data_ptr() + offb * H, None,
                             s1.data_ptr() + offb * H)
            else:
                l1_fwd(kd, rows.data_ptr() + 8 * Cl, B, offb)
            kd.gated_fwd(A1.data_ptr() + offb * H, None, B, H, H, w2h, b2h, w2g, b2g, H,
                         A2.data_ptr() + offb * H, None, s2.data_ptr() + offb * H)
            if thin_heads:
                # both heads (one gated-style split-K GEMM: bank h = mean, bank g = log-variance) and the sample in two
                # launches instead of five
                if bc_in_heads:
                    _lib.check(lib.evae_heads_reparam_fwd_bcast(_vp(A2b), B, H, H, _vp(wm), _vp(bm), _vp(wl), _vp(bl), Z, -6.0, 2.0,
                                                                _vp(eps), _vp(z_mean), _vp(lv_pre), _vp(logvar), _vp(z), _vp(logq),
                                                                _vp(plv.detach()), _vp(lv_row), Z, kd.st), "heads_reparam_fwd_bcast")
                else:
                    wh_ = kd.ws("heads", lib.evae_heads_reparam_fwd_workspace_bytes(B, H, Z))
                    _lib.check(lib.evae_heads_reparam_fwd(_vp(A2b), B, H, H, _vp(wm), _vp(bm), _vp(wl), _vp(bl), Z, -6.0, 2.0, _vp(eps),
                                                          _vp(z_mean), _vp(lv_pre), _vp(logvar), _vp(z), _vp(logq), _vp(wh_),
                                                          wh_.numel(), kd.st), "heads_reparam_fwd")
                if approx:
                    zm_ready = torch.cuda.Event(); zm_ready.record()
            else:
                kd.linear_fwd(A2b, B, H, H, wm, bm, Z, ACT_NONE, 0.0, 0.0, z_mean, None)
                if approx:
                    zm_ready = torch.cuda.Event(); zm_ready.record()
                kd.linear_fwd(A2b, B, H, H, wl, bl, Z, ACT_HARDTANH, -6.0, 2.0, logvar, lv_pre)
                # ---- sample, decode, reconstruct
                _lib.check(lib.evae_reparam_logq_fwd(_vp(z_mean), _vp(logvar), _vp(eps), B, Z, _vp(z), _vp(logq), kd.st), "reparam")
            z_ready = torch.cuda.Event(); z_ready.record()
            kd.gated_fwd(z, None, B, Z, Z, d1h, e1h, d1g, e1g, H, D1, None, sd1)
            kd.gated_fwd(D1, None, B, H, H, d2h, e2h, d2g, e2g, H, D2, None, sd2)
            kd.linear_fwd(D2, B, H, H, wp, bp, D, ACT_SIGMOID, 0.0, 0.0, xmean, None)
            dpx_fwd = None
            if prior_train and (node_merge(Cl, B) & 2):
                # RE, the step's coefficient vectors and the sigmoid head's gradient in one launch (the backward pass skips its own)
                coef = (torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, **f32))
                dpx_fwd = torch.empty((B, D), **f32)
                _lib.check(lib.evae_bernoulli_unit_step(_vp(x), _vp(xmean), B, D, _vp(beta_dev), beta_host, _vp(RE), _vp(coef[0]),
                                                        _vp(coef[1]), _vp(coef[2]), _vp(dpx_fwd), kd.st), "bernoulli_unit_step")
            else:
                _lib.check(lib.evae_bernoulli_ll_fwd(_vp(x), _vp(xmean), B, D, _vp(RE), kd.st), "bernoulli")
            re_ready = torch.cuda.Event(); re_ready.record()
            if prior_train and dpx_fwd is None:
                # the step's coefficient vectors (-1/B, beta/B, -beta/B: evae_elbo_bwd of a unit upstream gradient on the batch mean)
                # HERE, on the stream whose backward chain reads them: that chain then waits for nothing of the prior's launch
                coef = (torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, **f32))
                one = _ONES.get(dev)
                if one is None:
                    one = _ONES[dev] = torch.ones(1, **f32)
                _lib.check(lib.evae_elbo_bwd(_vp(one), 1, None, 0, None, 0, _vp(beta_dev), beta_host, B, _vp(coef[0]), _vp(coef[1]),
                                             _vp(coef[2]), kd.st), "elbo_bwd(unit)")
            if xt_late:
                wq, xt_gen = xt_gather()
        ci_sel = None
        if approx:
            # the batch's own cache rows refreshed with its means, top-k of every batch row among the candidates' cached
            # latents, the B * k slots re-encoded below; the rest of the batch-row path keeps running on the side stream
            main.wait_event(zm_ready)
            xi = ops._i64(x_idx)
            approx_cache.index_copy_(0, xi, z_mean)
            sub_cache = approx_cache.index_select(0, ex_idx)
            nearest, _ = ops.pairdist_topk(z_mean, sub_cache, int(approx_k), want_val=False)
            sel_rows, ci_sel = ops.select_exemplars(nearest.view(-1), ex_idx, out_rows=rows[:Cl])     # straight into the gather list
            l1_fwd(k, rows, Cl, 0)
        if Cl > 0:
            if p6:
                if w2_ready is not None:
                    main.wait_event(w2_ready)
                fl2 = 2.0 * Cl * H * 2 * H
                ops.probed("gated_dense_fwd M=%d K=%d N=%d (pre-split bf16 images)" % (Cl, H, H), fl2,
                           lambda: _lib.check(lib.evae_gated_dense_fwd_p6t(_vp(t_h1), nks_m, Cl, H, _vp(w2_img), _vp(b2h), _vp(b2g), H,
                                                                           _vp(A2), _vp(s2), k.st), "gated_fwd_p6t"),
                           executed=6 * fl2, pipe="bf16-mfma")
            else:
                k.gated_fwd(A1, None, Cl, H, H, w2h, b2h, w2g, b2g, H, A2, None, s2)
            k.linear_fwd(A2, Cl, H, H, wm, bm, Z, ACT_NONE, 0.0, 0.0, mean_all, None)
        if approx:
            approx_cache.index_copy_(0, sel_rows, centres)       # repeats of a row carry identical encodings
        main.wait_event(z_ready)                           # (also orders lv_row, written at the head of the side stream)
        # the centres the prior sees: one per DRAW.  With duplicate tables outside the one-launch prior (eager steps, sharded steps:
        # r06) they are gathered from the distinct rows' encodings here and the prior's kernels run over them as over any exemplar set
        centres_p = centres
        if dd is not None and not prior_train:
            centres_p = torch.empty((Cp, Z), **f32)
            _lib.check(lib.evae_gather_rows(_vp(centres), _vp(dd[1]), None, Cp, Z, _vp(centres_p), k.st), "gather_rows(centres)")
        # ---- exemplar prior (leave-one-out mask in training unless no_mask; with its collectives when sharded) on the
        #      main stream ...
        zi = None if no_mask else ops._i64(x_idx)
        ci = None if no_mask else (ci_sel if approx else ops._i64(dd[0] if dd is not None else ex_idx))
        logp = torch.empty(B, **f32); lse = torch.empty((2, B), **f32)      # lse: the (max, log sum) token of the merge
        z_all, zi_all = z, zi
        prior_done = None
        if prior_train:
            # forward AND backward of the prior here: token, log p, dcentres (rows [0, Cl) of the head-gradient buffer the backward
            # pass carries on with), dz' and dlogvar'.  ev_pre: what the side stream's backward chain waits for instead of the
            # main stream's latest launch (it needs nothing of the prior before dz')
            dmean_all = torch.empty((Mp, Z), **f32)
            packed = torch.empty(B * Z + Z, **f32)
            ev_pre = torch.cuda.Event(); ev_pre.record()
            keep_alive = None
            if dd is not None:
                # the prior over all C draws: centres of the draws gathered from the distinct rows' encodings; a distinct row's
                # gradient = multiplicity x the gradient of one of its draws
                dc_x = torch.empty((Cp, Z), **f32)
                keep_alive = [dc_x]
                if os.environ.get("EVAE_PRIOR_ROWS", "1") != "0":
                    # r06: the kernel reads draw j's centre through inv[j] and the reduction launch folds the per-draw gradients onto the
                    # distinct rows: the two gather launches around the prior are gone (2 x ~5 us of the critical path)
                    ops.prior_train_step_rows(z, centres, (dd[1], dd[2], dd[3]), lv_row, zi, ci, float(c_total),
                                              beta_dev if beta_dev is not None else beta_host,
                                              out=(logp, lse, None, packed[:B * Z].view(B, Z), dmean_all[:Cl], packed[B * Z:]), dc_draws=dc_x,
                                              stream=k.st)
                else:
                    centres_x = torch.empty((Cp, Z), **f32)
                    keep_alive.append(centres_x)
                    _lib.check(lib.evae_gather_rows(_vp(centres), _vp(dd[1]), None, Cp, Z, _vp(centres_x), k.st), "gather_rows(centres)")
                    ops.prior_train_step(z, centres_x, lv_row, zi, ci, float(c_total), beta_dev if beta_dev is not None else beta_host,
                                         out=(logp, lse, None, packed[:B * Z].view(B, Z), dc_x, packed[B * Z:]), stream=k.st)
                    _lib.check(lib.evae_gather_rows(_vp(dc_x), _vp(dd[2]), _vp(dd[3]), Cl, Z, _vp(dmean_all), k.st), "gather_rows(dcentres)")
            else:
                ops.prior_train_step(z, centres, lv_row, zi, ci, float(c_total), beta_dev if beta_dev is not None else beta_host,
                                     out=(logp, lse, None, packed[:B * Z].view(B, Z), dmean_all[:Cl], packed[B * Z:]), stream=k.st)
            # (the scratch of this launch rides along: the side stream's backward chain starts at ev_pre -- BESIDE the prior's launch -- and
            #  its buffers are allocated while the main stream is current; a scratch tensor freed at the end of this forward would be
            #  handed to them while the prior still reads / writes it.  r06: an allocation-order change made exactly that happen)
            prior_done = (dmean_all, packed, ev_pre, keep_alive)
        if sharded == 2:
            # data-parallel batches over sharded exemplars: every rank scores the queries of ALL ranks against its
            # shard (same pair count as B queries against all C), the partials go back to their owners
            z_all = shard._all_gather_flat(z).reshape(-1, Z)
            zi_all = None if zi is None else shard._all_gather_flat(zi.contiguous()).reshape(-1)
            m, s, n, _ = ops.prior_lse_fwd(z_all, centres_p, lv_row, zi_all, ci)
            m, s, n = shard.gather_partials(m, s, n)                  # [R x R*B] each
            r0 = dist.get_rank() * B
            m, s, n = (t[:, r0:r0 + B].contiguous() for t in (m, s, n))
        elif sharded:
            m, s, n, _ = ops.prior_lse_fwd(z, centres_p, lv_row, zi, ci)
            m, s, n = shard.gather_partials(m, s, n)
        if sharded:
            R, ldp = m.shape[0], m.shape[1]
            pm, ps, pn = m, s, n
        elif prior_train:
            pass
        else:
            # one device: the per-split partials stay un-merged in the workspace
            nb = lib.evae_prior_lse_fwd_workspace_bytes(B, Cp, Z)
            w = k.ws("prior_fwd", nb)
            ns, prow = C.c_int(0), C.c_int(0)
            _lib.check(lib.evae_prior_lse_fwd_splits(_vp(z), B, _vp(centres_p), Cp, Z, _vp(lv_row), _vp(zi), _vp(ci), _vp(w),
                                                     w.numel(), C.byref(ns), C.byref(prow), k.st), "prior_lse_fwd_splits")
            R, ldp = ns.value, B
            pm = w.data_ptr(); ps = pm + 4 * prow.value * B; pn = ps + 4 * prow.value * B
        # captured step (unit upstream promise) on one device: merge on the main stream, ELBO assembly on the side stream
        elbo_split = bool(ELBO_SPLIT and UNIT_UPSTREAM[0] and average and not sharded and xt_late and not ONE_STREAM[0])
        if prior_train or elbo_split:
            pass                             # the main stream does not meet the reconstruction term at all
        elif xt_late:
            main.wait_event(re_ready)        # (the side stream carries on with the gather; the backward pass joins it)
        else:
            main.wait_stream(side)
        # ---- merge of the partial log-sum-exps (splits of this device, or the gathered shards) + ELBO assembly (+ batch
        #      means) in ONE launch
        loss = torch.empty(B, **f32); KL = torch.empty(B, **f32)
        means = torch.empty(3, **f32) if average else None
        if prior_train:
            pass          # loss / KL / means: assembled on the side stream by the backward pass, behind its wait for dz'
        elif elbo_split:
            # r04: the merge (token, logp, coefficients) is all the prior's backward waits for: it follows the partials on the main
            # stream without meeting the side stream (a join and the single-block merge + ELBO launch less on the critical
            # path); loss / KL / means are assembled on the side stream, which the backward pass joins before anyone reads them
            coef = (torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, **f32))
            _lib.check(lib.evae_prior_merge_coef(_vp(pm), _vp(ps), _vp(pn), R, ldp, B, float(c_total), _vp(beta_dev), beta_host,
                                                 _vp(logp), _vp(lse), _vp(coef[0]), _vp(coef[1]), _vp(coef[2]), k.st), "prior_merge_coef")
            merged = torch.cuda.Event(); merged.record()
            with torch.cuda.stream(side):
                side.wait_event(merged)
                _lib.check(lib.evae_elbo_assemble(_vp(logp), _vp(RE), _vp(logq), _vp(beta_dev), beta_host, B, _vp(loss), _vp(KL),
                                                  _vp(means), kd.st), "elbo_assemble")
        elif UNIT_UPSTREAM[0] and average and not sharded:
            # the caller (evae/graph.py) promises loss.backward(ones) on the batch mean and nothing else: the backward pass's
            # coefficient vectors are then known here (-1/B, beta/B, -beta/B) and its elbo_bwd launch is not needed
            coef = (torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, **f32))
            _lib.check(lib.evae_prior_elbo_fwd_coef(_vp(pm), _vp(ps), _vp(pn), R, ldp, B, float(c_total), _vp(RE), _vp(logq),
                                                    _vp(beta_dev), beta_host, _vp(logp), _vp(lse), _vp(loss), _vp(KL), _vp(means),
                                                    _vp(coef[0]), _vp(coef[1]), _vp(coef[2]), k.st), "prior_elbo_fwd_coef")
        else:
            _lib.check(lib.evae_prior_elbo_fwd(_vp(pm), _vp(ps), _vp(pn), R, ldp, B, float(c_total), _vp(RE), _vp(logq),
                                               _vp(beta_dev), beta_host, _vp(logp), _vp(lse), _vp(loss), _vp(KL), _vp(means),
                                               k.st), "prior_elbo_fwd")
        ctx.coef = coef
        ctx.dpx = dpx_fwd
        # (kept until the backward pass has joined the side stream: the assembly reads / writes them there, behind launches of
        #  the main stream that would otherwise be free to take their memory)
        ctx.elbo_keep = (RE, logq, logp, loss, KL, means) if (elbo_split or prior_train) else None
        ctx.prior_done = prior_done
        ctx.dd = (dd, centres_p) if (dd is not None and not prior_train) else None
        ctx.wt = WT_DONE.pop((wm.data_ptr(), w2h.data_ptr(), w2g.data_ptr()), None)
        ctx.set_materialize_grads(False)       # unused outputs (RE, KL) then arrive as None, not as zero-filled tensors
        ctx.k_dev = dev
        ctx.dims = (B, D, H, Z, Cl, Mp, ldd, beta, int(sharded))
        ctx.dp = (z_all, zi_all)
        ctx.stage_gen = gen
        ctx.xt = (wq.data_ptr(), xt_gen) if xt_early else None
        ctx.p6 = (t_h1, t_dq2, w2t_img, nks_m) if p6 else None
        ctx.bufs = (x, rows, data_ext, A1, s1, A2, s2, mean_all, logvar, lv_pre, z, D1, sd1, D2, sd2,
                    xmean, lv_row, zi, ci, lse, eps)
        ctx.save_for_backward(*params)
        if average:
            l, r, kl = means.unbind(0)
            return l, r, kl
        return loss, RE, KL

    @staticmethod
    def backward(ctx, dloss, dRE, dKL):
        params = ctx.saved_tensors
        (plv, wp, bp, w1h, b1h, w1g, b1g, w2h, b2h, w2g, b2g, wm, bm, wl, bl,
         d1h, e1h, d1g, e1g, d2h, e2h, d2g, e2g) = params
        (x, rows, data_ext, A1, s1, A2, s2, mean_all, logvar, lv_pre, z, D1, sd1, D2, sd2,
         xmean, lv_row, zi, ci, lse, eps) = ctx.bufs
        B, D, H, Z, Cl, Mp, ldd, beta, sharded = ctx.dims
        if _STAGE_GEN.get(data_ext.data_ptr()) != ctx.stage_gen:
            raise _lib.EvaeError("fused vae step: another fused forward re-used the staging rows of this dataset before this "
                                 "backward ran (gradient accumulation over two batches): set model._use_fused = False for it")
        dev = ctx.k_dev
        k = _K(dev)
        lib = k.lib
        f32 = dict(device=dev, dtype=torch.float32)
        # every parameter gradient is a view of one buffer (slots padded to 256 B), see shard.register_flat_grads
        slots = (("plv", 1), ("wp", D * H), ("bp", D), ("w1", 2 * H * D), ("b1", 2 * H), ("w2", 2 * H * H), ("b2", 2 * H),
                 ("wm", Z * H), ("bm", Z), ("wl", Z * H), ("bl", Z), ("d1", 2 * H * Z), ("e1", 2 * H), ("d2", 2 * H * H),
                 ("e2", 2 * H))
        offs, tot = {}, 0
        for name, n in slots:
            offs[name] = (tot, n)
            tot += (n + 63) // 64 * 64
        gflat = torch.empty(tot, **f32)          # the padding between slots is never read (only summed by the all-reduce)

        def gslot(name, *shape):
            o, n = offs[name]
            return gflat[o:o + n].view(*shape)
        if sharded == 1 and os.environ.get("EVAE_REDUCE_ALL", "0") != "1":
            # replicated batch: only the encoder's gradients (slots w1 .. bm, contiguous) differ between the ranks
            shard.register_flat_grads(gflat, offs["w1"][0], offs["bm"][0] + (offs["bm"][1] + 63) // 64 * 64)
        elif sharded:
            shard.register_flat_grads(gflat)
        # upstream gradients are per-row vectors (average=False) or scalars of the batch means (average=True)
        beta_dev = beta if torch.is_tensor(beta) else None
        if ctx.coef is not None and dRE is None and dKL is None and dloss is not None and dloss.numel() == 1:
            cRE, cKL, gp = ctx.coef              # written by the forward pass under the caller's unit-upstream promise
        else:
            cRE = torch.empty(B, **f32); cKL = torch.empty(B, **f32); gp = torch.empty(B, **f32)
            gl = None if dloss is None else dloss.contiguous()
            gr = None if dRE is None else dRE.contiguous()
            gk = None if dKL is None else dKL.contiguous()
            _lib.check(lib.evae_elbo_bwd(_vp(gl), 0 if gl is None else gl.numel(), _vp(gr), 0 if gr is None else gr.numel(),
                                         _vp(gk), 0 if gk is None else gk.numel(), _vp(beta_dev),
                                         0.0 if beta_dev is not None else float(beta), B, _vp(cRE), _vp(cKL), _vp(gp), k.st),
                       "elbo_bwd")
        # ---- two chains from here to the weight gradients (the order of issue below is the order the captured graph starts
        #      things in, and it matters: a chain of small launches crawls next to a GEMM that fills every CU):
        #   main stream: prior term d(-cKL * logp) (+ its collectives when sharded) -> dcentres, which land in the
        #                head-gradient buffer -> data gradients of the head and of encoder layer 2 for the Cl exemplar rows
        #                (none of it needs the decoder);
        #   side stream: reconstruction term through the decoder -> dz (+ the prior's dz') -> the same data gradients for
        #                the B batch rows, then the B-row weight gradients of decoder / log-variance head, which are leaves.
        #   They meet at the weight gradients, which run over all Cl + B rows in one launch per layer.
        #   dz' and dlogvar' share one packed buffer so that the sharded case all-reduces it in place.
        main = torch.cuda.current_stream()
        side = k.side_stream()
        kd = _K(dev, stream=side, suffix="_side")
        prior_done = ctx.prior_done is not None and ctx.coef is not None and cRE is ctx.coef[0]
        dmean_all = ctx.prior_done[0] if prior_done else torch.empty((Mp, Z), **f32)
        dq2 = torch.empty((Mp, 2 * H), **f32)                              # [dh2 | dg2] for all C + B rows
        # byte store: encoder layer 1's (dh, dg) leave the layer-2 data gradient as the bf16 tile images its weight gradient
        # reads (no fp32 [Mp x 2H] buffer, no transposing pre-pass); needs row blocks in aligned fours
        # (only where the exemplar rows' data gradient fills the machine: a thin launch pays more for the un-split image
        # epilogue than the pre-pass it saves -- C = 200: 0.291 -> 0.301 ms)
        img_mode = (data_ext.dtype == torch.uint8 and Cl % 8 == 0 and IMG_DGRAD and bool(lib.evae_gemm_x6_applies(Cl, H, 0)))
        p6 = ctx.p6 is not None
        if p6:
            t_h1, t_dq2, w2t_img, nks_m = ctx.p6
            assert img_mode
        dq1 = None if img_mode else torch.empty((Mp, 2 * H), **f32)
        if img_mode:
            nb_w1 = lib.evae_dense_bwd_weight_u8_workspace_bytes(Mp, 2 * H, D)
            ws_w1 = k.ws("wgrad_u8_fused", nb_w1)
            off_img, nslab_img = C.c_size_t(0), C.c_int(0)
            _lib.check(lib.evae_dense_bwd_weight_u8_images(Mp, 2 * H, D, C.byref(off_img), C.byref(nslab_img)), "u8_images")
            img_ptr = ws_w1.data_ptr() + off_img.value
            key = (ws_w1.data_ptr(), Mp, H, D)
            if _ZEROED.get("wgrad_u8") != key:       # image bytes no row / column maps to must be zero: once per buffer and shape
                # (the IMAGE region only: the transposed byte rows the forward pass left in front of it stay)
                nimg = ((2 * H + 127) // 128) * nslab_img.value * 3 * 128 * 32 * 2
                ws_w1[off_img.value:off_img.value + nimg].zero_()
                _ZEROED["wgrad_u8"] = key

            def l2_dgrad(kk, M, ob, m_base):
                if p6 and kk is k:
                    # the exemplar rows: [dh2 | dg2]^T's image x W2^T's image on the bf16 pipe, (dh1, dg1) as the byte layer's images
                    fl_ = 2.0 * M * 2 * H * H
                    ops.probed("dense_bwd_data M=%d N=%d+%d K=%d (pre-split bf16 images; gate-backward epilogue -> bf16 tile images)"
                               % (M, H, H, H), fl_,
                               lambda: _lib.check(lib.evae_dense_bwd_data_p6t(_vp(t_dq2), nks_m, M, 2 * H, _vp(w2t_img), H,
                                                                              _vp(A1.data_ptr() + ob * H), _vp(s1.data_ptr() + ob * H),
                                                                              None, None, 0, _vp(img_ptr), nslab_img.value, m_base, kk.st),
                                                  "bwd_data_p6t"), executed=6 * fl_, pipe="bf16-mfma")
                    return
                nbd = lib.evae_dense_bwd_data_workspace_bytes(M, H, H, 2)
                wd = kk.ws("dgrad", nbd)
                wT = None if (ctx.wt is None or kk is not k) else ctx.wt[1]
                fl_ = 2.0 * M * 2 * H * H
                ex_, pipe_ = ops.gemm_pipe(M, H, False, fl_)
                ops.probed("dense_bwd_data M=%d N=%d+%d K=%d (gate-backward epilogue -> bf16 tile images)" % (M, H, H, H), fl_,
                           lambda: _lib.check(lib.evae_dense_bwd_data_img(
                               _vp(dq2.data_ptr() + ob * 2 * H), _vp(w2h), _vp(dq2.data_ptr() + ob * 2 * H + 4 * H), _vp(w2g), M, H,
                               2 * H, H, _vp(A1.data_ptr() + ob * H), _vp(s1.data_ptr() + ob * H), _vp(img_ptr), nslab_img.value,
                               m_base, _vp(wT), _vp(wd), wd.numel(), kk.st), "bwd_data_img"), executed=ex_, pipe=pipe_)
        else:
            def l2_dgrad(kk, M, ob, m_base):
                kk.bwd_data(dq2.data_ptr() + ob * 2 * H, w2h, dq2.data_ptr() + ob * 2 * H + 4 * H, w2g, M, H, 2 * H, H,
                            A1.data_ptr() + ob * H, s1.data_ptr() + ob * H, dq1.data_ptr() + ob * 2 * H,
                            dq1.data_ptr() + ob * 2 * H + 4 * H, 2 * H,
                            wT=None if (ctx.wt is None or kk is not k) else ctx.wt[1])
        dpx_done = ctx.dpx is not None and ctx.coef is not None and cRE is ctx.coef[0]     # (evae_bernoulli_unit_step, forward pass)
        dpx = ctx.dpx if dpx_done else torch.empty((B, D), **f32)
        dp2 = torch.empty((B, 2 * H), **f32)                               # [dh | dg] of decoder layer 2
        dp1 = torch.empty((B, 2 * H), **f32)
        dz = torch.empty((B, Z), **f32)
        dlvp = torch.empty((B, Z), **f32)
        centres = mean_all[:Cl]
        z_mean = mean_all[Cl:]
        off = 4 * Cl
        prior_finish = None
        # duplicate tables outside the one-launch prior: the prior's backward runs over the per-draw centres and a distinct row's
        # gradient is multiplicity x one of its draws' (evae_gather_rows with a scale), written where the head's data gradient reads
        dd_, centres_p = ctx.dd if ctx.dd is not None else (None, centres)
        Cp = centres_p.shape[0]
        dc_out = torch.empty((Cp, Z), **f32) if dd_ is not None else dmean_all

        def dc_fold():
            if dd_ is not None:
                _lib.check(lib.evae_gather_rows(_vp(dc_out), _vp(dd_[2]), _vp(dd_[3]), Cl, Z, _vp(dmean_all), k.st), "gather_rows(dcentres)")
        if prior_done:
            side.wait_event(ctx.prior_done[2])       # (not the prior's launch, which the forward pass put on the main stream)
        else:
            side.wait_stream(main)
        if sharded == 2:
            # data-parallel batches: the shard-side backward runs over the queries of all ranks (their lse and upstream
            # coefficients are gathered first); dz partials are summed over the shards and every rank keeps its rows.
            # dcentres and dlogvar are complete sums over all queries: the mean all-reduce of the parameter gradients
            # in AdamNormGrad.step then yields the gradient of the global-batch mean loss, no rescaling needed.
            z_all, zi_all = ctx.dp
            RB = z_all.shape[0]
            lg = shard._all_gather_flat(torch.cat((lse.reshape(2, B), gp.reshape(1, B))))            # [R x 3 x B]
            lse_all = torch.stack((lg[:, 0].reshape(-1), lg[:, 1].reshape(-1))).contiguous()       # token of all R B queries
            gp_all = lg[:, 2].reshape(-1).contiguous()
            dz_all = torch.empty((RB, Z), **f32); dlv = torch.empty(Z, **f32)
            nb = lib.evae_prior_lse_bwd_workspace_bytes(RB, Cp, Z)
            w = k.ws("prior_bwd", nb)
            _lib.check(lib.evae_prior_lse_bwd(_vp(z_all), RB, _vp(centres_p), Cp, Z, _vp(lv_row), _vp(zi_all), _vp(ci), _vp(lse_all),
                                              _vp(gp_all), _vp(dz_all), _vp(dc_out), _vp(dlv), _vp(w), w.numel(), k.st),
                       "prior_bwd")
            dc_fold()
            dist.all_reduce(dz_all, op=dist.ReduceOp.SUM)
            r0 = dist.get_rank() * B
            dzp = dz_all[r0:r0 + B]
        elif prior_done:
            packed = ctx.prior_done[1]           # the forward pass ran the prior's backward with it (evae_prior_train_step)
            dzp = packed[:B * Z].view(B, Z); dlv = packed[B * Z:]
        else:
            assert ctx.prior_done is None, "fused vae step: the captured form's backward was called with another upstream gradient"
            packed = torch.empty(B * Z + Z, **f32)
            dzp = packed[:B * Z].view(B, Z); dlv = packed[B * Z:]
            nb = lib.evae_prior_lse_bwd_workspace_bytes(B, Cp, Z)
            w = k.ws("prior_bwd", nb)
            pb_args = (_vp(z), B, _vp(centres_p), Cp, Z, _vp(lv_row), _vp(zi), _vp(ci), _vp(lse), _vp(gp), _vp(dzp),
                       _vp(dc_out), _vp(dlv), _vp(w), w.numel())
            if sharded == 1 or dd_ is not None or not (SCHED & 1):
                _lib.check(lib.evae_prior_lse_bwd(*pb_args, k.st), "prior_bwd")
                dc_fold()
                if sharded == 1:
                    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
                    if Cl > 0:
                        dmean_all[:Cl].mul_(float(dist.get_world_size()))
            else:
                # one device: the main stream only needs dcentres (phase 1); the reduction of the dz' / dlogvar partials
                # (phase 2) is issued on the side stream, in front of its only consumer
                _lib.check(lib.evae_prior_lse_bwd_phased(*pb_args, 1, k.st), "prior_bwd(1)")
                prior_finish = lambda: _lib.check(lib.evae_prior_lse_bwd_phased(*pb_args, 2, kd.st), "prior_bwd(2)")
        dz_ready = torch.cuda.Event(); dz_ready.record()
        if Cl > 0:
            if p6:
                # (dh2, dg2) of the exemplar rows leave as [dh2 | dg2]^T's image only: their two consumers read images
                nbh = lib.evae_dense_bwd_data_workspace_bytes(Cl, Z, H, 1)
                wh_ = k.ws("dgrad", nbh)
                flh = 2.0 * Cl * Z * H
                exh, pih = ops.gemm_pipe(Cl, H, False, flh)
                # a K = 40 product with a [Cl x 2H] output: it streams -- reads dmean, A2 and s2, writes 6 bytes per element of [dh | dg]
                ops.probed("dense_bwd_data M=%d N=%d K=%d (gate-backward epilogue -> pre-split image)" % (Cl, Z, H), flh,
                           hbm_bytes=4.0 * Cl * (Z + 2 * H) + 6.0 * Cl * 2 * H, fn=
                           lambda: _lib.check(lib.evae_dense_bwd_data_timg(_vp(dmean_all), _vp(wm), None, None, Cl, Z, Z, H, _vp(A2), _vp(s2),
                                                                           None, None, 2 * H, _vp(None if ctx.wt is None else ctx.wt[0]),
                                                                           _vp(t_dq2), nks_m, 0, 0, _vp(wh_), wh_.numel(), k.st),
                                              "bwd_data_timg"))
            else:
                k.bwd_data(dmean_all, wm, None, None, Cl, Z, Z, H, A2, s2, dq2, dq2.data_ptr() + 4 * H, 2 * H,
                           wT=None if ctx.wt is None else ctx.wt[0])
            l2_dgrad(k, Cl, 0, 0)
        batch_rows_done = torch.cuda.Event()
        g_plv = gslot("plv", 1)
        g_wp = gslot("wp", D, H); g_bp = gslot("bp", D)
        g_d2 = gslot("d2", 2 * H, H); g_e2 = gslot("e2", 2 * H)
        g_d1 = gslot("d1", 2 * H, Z); g_e1 = gslot("e1", 2 * H)
        g_wl = gslot("wl", Z, H); g_bl = gslot("bl", Z)
        g_wm = gslot("wm", Z, H); g_bm = gslot("bm", Z)
        fin_group = FINISH_GROUP and data_ext.dtype == torch.uint8 and not (SCHED & 10) and Mp > 128 and not p6
        headw_early = bool(HEADW_EARLY and not fin_group and Cl > 0)
        with torch.cuda.stream(side):
            # through the Bernoulli log-likelihood and the sigmoid head at once, then down the decoder
            if not dpx_done:
                _lib.check(lib.evae_bernoulli_sigmoid_bwd(_vp(x), _vp(xmean), _vp(cRE), B, D, _vp(dpx), kd.st), "bernoulli_sigmoid_bwd")
            kd.bwd_data(dpx, wp, None, None, B, D, D, H, D2, sd2, dp2, dp2.data_ptr() + 4 * H, 2 * H)
            kd.bwd_data(dp2, d2h, dp2.data_ptr() + 4 * H, d2g, B, H, 2 * H, H, D1, sd1, dp1, dp1.data_ptr() + 4 * H, 2 * H)
            kd.bwd_data(dp1, d1h, dp1.data_ptr() + 4 * H, d1g, B, H, 2 * H, Z, None, None, dz, None, Z)
            side.wait_event(dz_ready)
            if prior_finish is not None:
                prior_finish()
            NODE_MERGE = node_merge(Cl, B)
            if NODE_MERGE & 12:
                # reparameterisation + log q (+ the prior's dz', + the Hardtanh of the log-variance head), and in one more block of
                # the launch the ELBO's assembly (bit 2) and the sum of the prior's log-variance gradient row (bit 3)
                RE_, logq_, logp_, loss_, KL_, means_ = ctx.elbo_keep if prior_done else (None,) * 6
                if prior_done and not (NODE_MERGE & 4):
                    _lib.check(lib.evae_elbo_assemble(_vp(logp_), _vp(RE_), _vp(logq_), _vp(beta_dev), 0.0 if beta_dev is not None else float(beta),
                                                      B, _vp(loss_), _vp(KL_), _vp(means_), kd.st), "elbo_assemble")
                    loss_ = None
                sum_here = bool(NODE_MERGE & 8)
                _lib.check(lib.evae_reparam_logq_bwd_hardtanh_tail(
                    _vp(z_mean), _vp(logvar), _vp(eps), _vp(z), _vp(dz), _vp(dzp), _vp(cKL), _vp(lv_pre), -6.0, 2.0, B, Z,
                    _vp(dmean_all.data_ptr() + off * Z), _vp(dlvp), _vp(logp_), _vp(RE_), _vp(logq_), _vp(beta_dev),
                    0.0 if beta_dev is not None else float(beta), _vp(loss_), _vp(KL_), _vp(means_),
                    _vp(dlv) if sum_here else None, Z, _vp(g_plv) if sum_here else None, kd.st), "reparam_bwd_tail")
            else:
                if prior_done:
                    RE_, logq_, logp_, loss_, KL_, means_ = ctx.elbo_keep
                    _lib.check(lib.evae_elbo_assemble(_vp(logp_), _vp(RE_), _vp(logq_), _vp(beta_dev), 0.0 if beta_dev is not None else float(beta),
                                                      B, _vp(loss_), _vp(KL_), _vp(means_), kd.st), "elbo_assemble")
                # reparameterisation + log q (+ the prior's dz', + the Hardtanh of the log-variance head): one launch
                _lib.check(lib.evae_reparam_logq_bwd_hardtanh(_vp(z_mean), _vp(logvar), _vp(eps), _vp(z), _vp(dz), _vp(dzp), _vp(cKL),
                                                              _vp(lv_pre), -6.0, 2.0, B, Z, _vp(dmean_all.data_ptr() + off * Z),
                                                              _vp(dlvp), kd.st), "reparam_bwd")
            if headw_early:
                kd.bwd_weight(dmean_all, Mp, Z, Z, A2, None, H, H, g_wm, g_bm)       # mean head, all C + B rows
            # head and encoder layer 2, batch rows
            if p6:
                # (fp32 rows for the batch rows' own layer-2 data gradient, and rows Cl .. of the image for the weight gradient)
                nbh = lib.evae_dense_bwd_data_workspace_bytes(B, Z, H, 2)
                wh2 = kd.ws("dgrad", nbh)
                _lib.check(lib.evae_dense_bwd_data_timg(_vp(dmean_all.data_ptr() + off * Z), _vp(wm), _vp(dlvp), _vp(wl), B, Z, Z, H,
                                                        _vp(A2.data_ptr() + off * H), _vp(s2.data_ptr() + off * H),
                                                        _vp(dq2.data_ptr() + off * 2 * H), _vp(dq2.data_ptr() + off * 2 * H + 4 * H), 2 * H,
                                                        None, _vp(t_dq2), nks_m, 0, Cl, _vp(wh2), wh2.numel(), kd.st), "bwd_data_timg(batch)")
            else:
                kd.bwd_data(dmean_all.data_ptr() + off * Z, wm, dlvp, wl, B, Z, Z, H, A2.data_ptr() + off * H,
                            s2.data_ptr() + off * H, dq2.data_ptr() + off * 2 * H, dq2.data_ptr() + off * 2 * H + 4 * H, 2 * H)
            dq2_rows_done = torch.cuda.Event(); dq2_rows_done.record()      # all layer 2's weight gradient needs of the batch rows
            l2_dgrad(kd, B, off, Cl)
            batch_rows_done.record()

        # ONE finish launch for the three split-K weight gradients of the step (encoder layer 1 on the byte store, layer 2, mean
        # head) at the very end, instead of one behind each GEMM: a dependent launch less on the main stream's chain (fin_group)

        def leaves():     # nobody waits for them before the optimizer
            with torch.cuda.stream(side):
                # mean head, all C + B rows (its finish joins the step's grouped finish launch when that is on)
                if fin_group and FINISH_GROUP_HEAD:
                    kd.bwd_weight(dmean_all, Mp, Z, Z, A2, None, H, H, g_wm, g_bm, phase=1, ws_name="wgradh")
                elif not headw_early:
                    kd.bwd_weight(dmean_all, Mp, Z, Z, A2, None, H, H, g_wm, g_bm)
                # the four leaf layers whose contraction is the B batch rows: ONE grouped launch (evae_dense_bwd_weight_group;
                # r02: four launches of 8-9 us each at the end of the side stream's chain)
                jobs = ((dpx, B, D, D, D2, H, H, g_wp, g_bp), (dp2, B, 2 * H, 2 * H, D1, H, H, g_d2, g_e2),
                        (dp1, B, 2 * H, 2 * H, z, Z, Z, g_d1, g_e1),
                        (dlvp, B, Z, Z, A2.data_ptr() + off * H, H, H, g_wl, g_bl))
                grouped = B <= 128 and GROUP_LEAVES and all(n_ % 4 == 0 and k_ % 4 == 0 for _, _, n_, _, _, k_, _, _, _ in jobs)
                if grouped:
                    arr = (_lib.WgradJob * len(jobs))()
                    for i_, (dy_, m_, n_, ldy_, x_, k_, ldx_, dw_, db_) in enumerate(jobs):
                        arr[i_].dy = dy_.data_ptr(); arr[i_].x = x_ if isinstance(x_, int) else x_.data_ptr()
                        arr[i_].dw = dw_.data_ptr(); arr[i_].db = db_.data_ptr()
                        arr[i_].M, arr[i_].N, arr[i_].K, arr[i_].ldy, arr[i_].ldx = m_, n_, k_, ldy_, ldx_
                    _lib.check(lib.evae_dense_bwd_weight_group(C.cast(arr, C.c_void_p), len(jobs), kd.st), "bwd_weight_group")
                else:
                    for dy_, m_, n_, ldy_, x_, k_, ldx_, dw_, db_ in jobs:
                        kd.bwd_weight(dy_, m_, n_, ldy_, x_, None, k_, ldx_, dw_, db_)
                if not (NODE_MERGE & 8):
                    _lib.check(lib.evae_sum_small(_vp(dlv), Z, _vp(g_plv), kd.st), "sum_small")
        leaves()      # (issued HERE: captured after the main stream's weight gradients instead, the same launches replay at
        #                0.82-0.94 ms for C = 200 and 1.2 ms at c2 -- this runtime's graph replay is very sensitive to where a
        #                branch's nodes sit relative to the other branch's, r03 measurement)
        # Layer 2's weight gradient starts as soon as the batch rows' dq2 exist, without waiting for their layer-1 (dh, dg) (the
        # side stream's last thin data gradient, stretched to ~40 us beside the exemplar rows' GEMM).  r03: with the leaf
        # gradients as four launches this was 0.668 -> 0.684-0.694 ms (they then started behind layer 2's CU-filling launch and
        # ended after the main stream); with the leaves grouped into one launch it is 0.662-0.664 -> 0.658 ms.  SCHED & 256: off.
        split_wait = bool(SCHED & 256) and not (SCHED & 10)    # r04: off -- the thin batch-row chain ends long before layer 2's data gradient; one join less (-4 us)
        main.wait_event(dq2_rows_done if split_wait else batch_rows_done)
        # ---- weight gradients of the two encoder layers over all C + B rows
        #      (layer 2's finish launch runs on the side stream, beside layer 1's GEMM instead of in front of it)
        g_w2 = gslot("w2", 2 * H, H); g_b2 = gslot("b2", 2 * H)
        w2_args = (dq2, Mp, 2 * H, 2 * H, A1, None, H, H, g_w2, g_b2)
        g_w1 = gslot("w1", 2 * H, D); g_b1 = gslot("b1", 2 * H)
        def w1_grad(no_finish=False):
            if data_ext.dtype == torch.uint8:
                nb = lib.evae_dense_bwd_weight_u8_workspace_bytes(Mp, 2 * H, D)
                w = k.ws("wgrad_u8_fused", nb)
                fl = 2.0 * Mp * 2 * H * D
                # the forward pass left the transposed byte rows in the workspace (unless another step used it since)
                have_xt = ctx.xt is not None and ctx.xt == (w.data_ptr(), _XT_GEN.get(w.data_ptr()))
                ops.probed("dense_bwd_weight_u8 M=%d N=%d K=%d (uint8 rows, three bf16 terms; %s)"
                           % (Mp, 2 * H, D, "GEMM; planes summed by the step's last grouped launch" if no_finish else "pre-passes + GEMM + finish"),
                           fl, lambda: _lib.check(lib.evae_dense_bwd_weight_u8_phased(_vp(dq1), Mp, 2 * H, 2 * H, _vp(data_ext), _vp(rows), D,
                                                                                      ldd, 1.0 / 255.0, _vp(g_w1), _vp(g_b1), _vp(w),
                                                                                      w.numel(), (4 if have_xt else 0) | (16 if no_finish else 0), k.st)
                                              if (have_xt or no_finish) else
                                              lib.evae_dense_bwd_weight_u8(_vp(dq1), Mp, 2 * H, 2 * H, _vp(data_ext), _vp(rows), D,
                                                                           ldd, 1.0 / 255.0, _vp(g_w1), _vp(g_b1), _vp(w),
                                                                           w.numel(), k.st), "bwd_weight_u8"),
                           executed=3 * fl, pipe="bf16-mfma")
            else:
                k.bwd_weight(dq1, Mp, 2 * H, 2 * H, data_ext, rows, D, ldd, g_w1, g_b1)
        if SCHED & 2:
            k.bwd_weight(*w2_args, phase=1, ws_name="wgrad2")
            w2_done = torch.cuda.Event(); w2_done.record()
            w1_grad()
            with torch.cuda.stream(side):
                side.wait_event(w2_done)
                k.bwd_weight(*w2_args, phase=2, ws_name="wgrad2", finish_on=kd)
        elif (SCHED & 8) and data_ext.dtype == torch.uint8:
            # the bandwidth-bound pre-passes of the byte layer's weight gradient (gather-transpose of the rows, split and
            # transposition of dy) on the side stream, beside layer 2's matrix-bound weight gradient
            nb = lib.evae_dense_bwd_weight_u8_workspace_bytes(Mp, 2 * H, D)
            w = k.ws("wgrad_u8_fused", nb)

            def w1_phase(phase, st):
                _lib.check(lib.evae_dense_bwd_weight_u8_phased(_vp(dq1), Mp, 2 * H, 2 * H, _vp(data_ext), _vp(rows), D, ldd, 1.0 / 255.0,
                                                               _vp(g_w1), _vp(g_b1), _vp(w), w.numel(), phase, st), "bwd_weight_u8_phased")
            dq1_ready = torch.cuda.Event(); dq1_ready.record()
            with torch.cuda.stream(side):
                side.wait_event(dq1_ready)
                w1_phase(1, kd.st)
                pre_done = torch.cuda.Event(); pre_done.record()
            k.bwd_weight(*w2_args)
            main.wait_event(pre_done)
            w1_phase(2, k.st)
        elif fin_group:
            k.bwd_weight(*w2_args, phase=1, ws_name="wgrad2")
            if split_wait:
                main.wait_event(batch_rows_done)
            w1_grad(no_finish=True)
            if FINISH_GROUP_HEAD:
                main.wait_stream(side)      # (the mean head's GEMM and the grouped leaves)
            arr = (_lib.WgradFinishJob * 3)()
            w_u8 = k.ws("wgrad_u8_fused", lib.evae_dense_bwd_weight_u8_workspace_bytes(Mp, 2 * H, D))
            w_2 = ops._workspace("wgrad2" + k.sfx, lib.evae_dense_bwd_weight_workspace_bytes(Mp, 2 * H, H), dev)
            w_h = ops._workspace("wgradh" + kd.sfx, lib.evae_dense_bwd_weight_workspace_bytes(Mp, Z, H), dev)
            fjobs = ((1, Mp, 2 * H, D, 2 * H, ldd, 1.0 / 255.0, g_w1, g_b1, w_u8), (0, Mp, 2 * H, H, 2 * H, H, 1.0, g_w2, g_b2, w_2),
                     (0, Mp, Z, H, Z, H, 1.0, g_wm, g_bm, w_h))[:3 if FINISH_GROUP_HEAD else 2]
            for i_, (byte_, m_, n_, k_, ldy_, ldx_, xs_, dw_, db_, ws_) in enumerate(fjobs):
                arr[i_].byte_rows, arr[i_].M, arr[i_].N, arr[i_].K, arr[i_].ldy, arr[i_].ldx = byte_, m_, n_, k_, ldy_, ldx_
                arr[i_].x_scale = xs_; arr[i_].dw = dw_.data_ptr(); arr[i_].db = db_.data_ptr()
                arr[i_].ws = ws_.data_ptr(); arr[i_].ws_bytes = ws_.numel()
            _lib.check(lib.evae_dense_bwd_weight_finish_group(C.cast(arr, C.c_void_p), len(fjobs), k.st), "bwd_weight_finish_group")
        elif p6:
            # layer 2's weight gradient (+ bias gradient through h1^T's ones row) from the two images, on the bf16 pipe
            nbw = lib.evae_dense_bwd_weight_p6_workspace_bytes(nks_m, 2 * H, H)
            ww = k.ws("wgrad_p6", nbw)
            flw = 2.0 * Mp * 2 * H * H
            ops.probed("dense_bwd_weight M=%d N=%d K=%d (+db; pre-split bf16 images, split-K GEMM + finish)" % (Mp, 2 * H, H), flw,
                       lambda: _lib.check(lib.evae_dense_bwd_weight_p6(_vp(t_dq2), _vp(t_h1), nks_m, 2 * H, H, _vp(g_w2), _vp(g_b2), _vp(ww),
                                                                       ww.numel(), k.st), "bwd_weight_p6"), executed=6 * flw, pipe="bf16-mfma")
            if split_wait:
                main.wait_event(batch_rows_done)
            w1_grad()
        else:
            k.bwd_weight(*w2_args)
            if split_wait:
                main.wait_event(batch_rows_done)
            w1_grad()
        main.wait_stream(side)
        ctx.bufs = None
        ctx.elbo_keep = None
        ctx.prior_done = None
        ctx.dd = None
        grads = (g_plv, g_wp, g_bp, g_w1[:H], g_b1[:H], g_w1[H:], g_b1[H:], g_w2[:H], g_b2[:H], g_w2[H:], g_b2[H:],
                 g_wm, g_bm, g_wl, g_bl, g_d1[:H], g_e1[:H], g_d1[H:], g_e1[H:], g_d2[:H], g_e2[:H], g_d2[H:], g_e2[H:])
        return (None,) * 15 + grads
