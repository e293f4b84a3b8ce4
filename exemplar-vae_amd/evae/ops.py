"""Host-side operators over the C ABI: raw launches on the current HIP stream plus the
torch.autograd.Function wrappers the reference-named modules (utils/nn.py, utils/distributions.py,
models/BaseModel.py) are built from.  PyTorch here is device memory, streams and autograd plumbing;
all arithmetic on [rows x features] data happens in libevae_hip.so.  No CPU fallback."""
import ctypes as C
import os

import torch

from . import _lib

ACT_NONE, ACT_SIGMOID, ACT_HARDTANH = 0, 1, 2
TOPK_SQRT = 1

_ws = {}
_ws_retired = []   # replaced buffers stay allocated: a captured hipGraph (evae/graph.py) may hold their addresses
PROBE = None   # bench.py sets {"records": []}: (name, start event, end event, algorithmic flops, executed flops, pipe) per big launch


def probed(name, flops, fn, executed=None, pipe="fp32-mfma", min_flops=2e9, hbm_bytes=None):
    """Run fn(); while bench.py's roofline probe is on, bracket it with a HIP event pair on the current stream (launches
    below min_flops are not worth an event pair).  `executed`: flops issued to the matrix pipe when they differ from the
    algorithmic count (three bf16 terms per product on the uint8 path).  hbm_bytes (pipe="hbm"): the ALGORITHMIC bytes of a
    streaming launch -- what it must read and write once -- recorded in place of the flop counts (>= 16 MB to be worth a pair)."""
    if PROBE is None or (flops < PROBE.get("min_flops", min_flops) if hbm_bytes is None else hbm_bytes < 16e6):
        return fn()
    if hbm_bytes is not None:
        flops, executed, pipe = hbm_bytes, hbm_bytes, "hbm"
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    PROBE["records"].append((name, e0, e1, float(flops), float(executed if executed is not None else flops), pipe))
    return r


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.EvaeError(
                "evae ops run on MI355X only (got a %s tensor); there is no CPU fallback" % t.device)


_SIDE_STREAM_OBJS = []     # ... the stream objects themselves
_SIDE_STREAMS = {}         # raw handle -> tag of streams registered as "side" streams: launches issued there get their own workspaces


def register_side_stream(stream, tag="side"):
    """Two launches that share a named workspace must not run at the same time.  A model that issues part of its step on a
    second stream (evae/fused_vae.py, models/AbsHModel.py) registers that stream here: every workspace requested while it is
    the current stream -- also from autograd's backward nodes, which run on the stream of their forward -- is a separate buffer."""
    _SIDE_STREAMS[int(stream.cuda_stream)] = tag
    if all(s_.cuda_stream != stream.cuda_stream for s_ in _SIDE_STREAM_OBJS):
        _SIDE_STREAM_OBJS.append(stream)


_MODEL_SIDE = {}


def model_side_stream(device):
    """The registered second stream of the modular two-stream training paths (one per device, made on first use)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _MODEL_SIDE.get(key)
    if st is None:
        st = _MODEL_SIDE[key] = torch.cuda.Stream(device=device)
        register_side_stream(st)
    return st


_MODEL_LEAF = {}
_LEAF_ACTIVE = [None]


def model_leaf_stream(device):
    """Third stream of the modular two-stream training paths: the weight gradients of the thin layers (leaves of the backward
    pass: nothing but the optimizer reads them) run there, off the chain of data gradients."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _MODEL_LEAF.get(key)
    if st is None:
        st = _MODEL_LEAF[key] = torch.cuda.Stream(device=device)
        register_side_stream(st, "leaf")
    return st


class leaf_branch:
    """with ops.leaf_branch(stream): layers built inside (utils.nn.GatedDense) put their weight-gradient node on `stream`
    (gated_dense_split below); stream None: no effect."""

    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        self.prev = _LEAF_ACTIVE[0]
        _LEAF_ACTIVE[0] = self.stream

    def __exit__(self, *a):
        _LEAF_ACTIVE[0] = self.prev


def active_leaf_stream():
    return _LEAF_ACTIVE[0]


def _workspace(name, nbytes, device):
    if _SIDE_STREAMS:
        tag = _SIDE_STREAMS.get(int(torch.cuda.current_stream(device).cuda_stream))
        if tag is not None:
            name = name + "@" + tag
    key = (name, device.index if device.index is not None else torch.cuda.current_device())
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            # Never free a workspace: launches captured into a hipGraph have its raw address baked in, and a later,
            # larger request under the same name (evaluate_loss scores against all N_train exemplars) must not turn
            # every replay into a write to freed memory.  Growth is geometric, so the retired total stays < the live size.
            _ws_retired.append(buf)
            nbytes = max(int(nbytes), 2 * buf.numel())
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _i64(t):
    if t is None:
        return None
    if t.dtype != torch.int64:
        t = t.long()
    t = t.reshape(-1)
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------
# exemplar prior
# ------------------------------------------------------------------------------------------------
def prior_lse_fwd(z, centres, log_var, z_idx=None, c_idx=None, want_prob=False):
    """One shard's partials: (max [B], sumexp [B], nmask [B], prob [B x C] or None)."""
    lib = _lib.load()
    _need_cuda(z, centres, log_var, z_idx, c_idx)
    z, centres, log_var = _f32(z), _f32(centres), _f32(log_var).reshape(-1)
    B, zd = z.shape
    Cn = centres.shape[0]
    z_idx, c_idx = _i64(z_idx), _i64(c_idx)
    if z_idx is not None and c_idx is not None:
        assert z_idx.numel() == B and c_idx.numel() == Cn
    else:
        z_idx = c_idx = None
    m = torch.empty(B, device=z.device); s = torch.empty_like(m); n = torch.empty_like(m)
    prob = torch.empty((B, Cn), device=z.device) if want_prob else None
    nb = lib.evae_prior_lse_fwd_workspace_bytes(B, Cn, zd)
    ws = _workspace("prior_fwd", nb, z.device)
    _lib.check(lib.evae_prior_lse_fwd(_p(z), B, _p(centres), Cn, zd, _p(log_var), _p(z_idx), _p(c_idx),
                                      _p(m), _p(s), _p(n), _p(prob), _p(ws), ws.numel(), _stream()),
               "evae_prior_lse_fwd")
    return m, s, n, prob


def gemm_x6_configure(enabled=-1, min_rows=-1):
    """Policy of the split-bf16 fp32 GEMM kernel (include/evae_hip.h: evae_gemm_x6_configure)."""
    _lib.check(_lib.load().evae_gemm_x6_configure(int(enabled), int(min_rows)), "evae_gemm_x6_configure")


def thin_configure(max_rows=-1):
    """Row-count limit of the one-launch thin layer kernels (include/evae_hip.h: evae_thin_configure); returns the value in force."""
    return int(_lib.load().evae_thin_configure(int(max_rows)))


def gemm_pipe(M, N, gated, flops):
    """(executed flops, pipe label) of an fp32 GEMM launch with M x N outputs for the roofline probe: six bf16 products per
    fp32 product when it takes the split-bf16 kernel."""
    if _lib.load().evae_gemm_x6_applies(int(M), int(N), int(bool(gated))):
        return 6.0 * flops, "bf16-mfma"
    return flops, "fp32-mfma"


def prior_set_norm_limit(limit):
    """Largest centred squared norm (sigma units) of a query tile the matrix-core prior kernels still evaluate in the
    expanded form; above it they switch to direct differences.  Negative restores the default, 0 forces the direct path."""
    _lib.check(_lib.load().evae_prior_set_norm_limit(float(limit)), "evae_prior_set_norm_limit")


def prior_merge(m, s, n, c_total, out=None):
    """[R x B] shard partials -> (logprior [B], token [2 x B]); `out` = caller-allocated (logprior, token).  The token is what
    the backward takes in place of the row log-sum-exp: row 0 = the row maximum, row 1 = the log of the normalised sum, kept
    apart (lse = token.sum(0)) so that exp(p_ij - lse_i) stays exact at any magnitude of the log-density
    (csrc/evae_prior.hip::prior_merge_kernel)."""
    lib = _lib.load()
    _need_cuda(m, s, n)
    m, s, n = _f32(m), _f32(s), _f32(n)
    if m.dim() == 1:
        m, s, n = m[None], s[None], n[None]
    R, B = m.shape
    lp, lse = out if out is not None else (torch.empty(B, device=m.device), torch.empty((2, B), device=m.device))
    assert lse.numel() == 2 * B and lse.is_contiguous()
    _lib.check(lib.evae_prior_merge(_p(m), _p(s), _p(n), R, B, float(c_total), _p(lp), _p(lse), _stream()),
               "evae_prior_merge")
    return lp, lse


_PT_STATE = {}


def prior_train_state(device):
    """The 256-byte inter-block state of evae_prior_train_step: one per device and stream tag, zeroed when it is made and never again"""
    tag = _SIDE_STREAMS.get(int(torch.cuda.current_stream(device).cuda_stream), "") if _SIDE_STREAMS else ""
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag)
    st = _PT_STATE.get(key)
    if st is None:
        st = _PT_STATE[key] = torch.zeros(64, dtype=torch.int32, device=device)
    return st


def prior_train_check():
    """Raise if any block of any one-launch prior (evae_prior_train_step) gave up waiting for the other blocks since the last
    check: its grid was not co-resident (another tenant of the device, a partition smaller than the query said) and some step's
    dz / dcentres / dlogvar were formed with a stale token.  Reads back: call it where the host synchronises anyway (end of an
    epoch, the replica check, the end of a timed loop) -- never inside a capture."""
    lib = _lib.load()
    for key, st in list(_PT_STATE.items()):
        n = lib.evae_prior_train_gave_up(_p(st), torch.cuda.current_stream(st.device).cuda_stream)
        if n < 0:
            _lib.check(n, "evae_prior_train_gave_up")
        if n > 0:
            raise RuntimeError("evae: %d block(s) of the one-launch exemplar prior gave up waiting for the rest of their grid (device "
                               "%s): gradients of at least one step are wrong.  The grid was not co-resident -- set "
                               "EVAE_PRIOR_TRAIN=0 to use the three-launch prior on this device" % (n, key[0]))


def prior_train_applies(B, Cn, zd):
    return bool(_lib.load().evae_prior_train_applies(int(B), int(Cn), int(zd)))


def prior_train_step_rows(z, centres, rows, log_var, z_idx, c_idx, c_total, beta, out, dc_draws, stream=None):
    """prior_train_step for a step that encoded each distinct image once: centres [n_rows x z], rows = (inv [C], rep [n_rows],
    mult [n_rows]); out = (logp, token, coef, dz, dcentres [n_rows x z], dlogvar), dc_draws [C x z] scratch."""
    lib = _lib.load()
    inv, rep, mult = rows
    B, zd = z.shape
    Cn = inv.numel()
    logp, token, coef, dz, dc, dlv = out
    beta_dev = beta if torch.is_tensor(beta) else None
    ws = _workspace("prior_train", lib.evae_prior_train_workspace_bytes(B, Cn, zd), z.device)
    st = prior_train_state(z.device)
    c3 = coef if coef is not None else (None, None, None)
    _lib.check(lib.evae_prior_train_step_rows(_p(z), B, _p(centres), centres.shape[0], _p(inv), _p(rep), _p(mult), Cn, zd, _p(log_var),
                                              _p(_i64(z_idx)), _p(_i64(c_idx)), float(c_total), _p(beta_dev),
                                              0.0 if beta_dev is not None else float(beta), _p(logp), _p(token), _p(c3[0]), _p(c3[1]),
                                              _p(c3[2]), _p(dz), _p(dc), _p(dc_draws), _p(dlv), _p(st), _p(ws), ws.numel(),
                                              _stream() if stream is None else stream), "evae_prior_train_step_rows")
    return out


def prior_train_step(z, centres, log_var, z_idx, c_idx, c_total, beta, want_coef=True, out=None, phase=0, stream=None):
    """The exemplar prior of a captured training step in one launch (+ the dz / dlogvar reduction): returns
    (logp [B], token [2 x B], (cRE, cKL, neg_cKL) or None, dz [B x z], dcentres [C x z], dlogvar [z]) for the loss
    mean_i(beta (logq_i - logp_i) - RE_i) with upstream gradient 1 (csrc/evae_prior_train.h).  `beta`: float or device scalar.
    `out` = caller-allocated (logp, token, coef, dz, dcentres, dlogvar)."""
    lib = _lib.load()
    _need_cuda(z, centres, log_var, z_idx, c_idx)
    z, centres, log_var = _f32(z), _f32(centres), _f32(log_var).reshape(-1)
    B, zd = z.shape
    Cn = centres.shape[0]
    z_idx, c_idx = _i64(z_idx), _i64(c_idx)
    if z_idx is None or c_idx is None:
        z_idx = c_idx = None
    dev = z.device
    if out is None:
        f = dict(device=dev, dtype=torch.float32)
        coef = (torch.empty(B, **f), torch.empty(B, **f), torch.empty(B, **f)) if want_coef else None
        out = (torch.empty(B, **f), torch.empty((2, B), **f), coef, torch.empty((B, zd), **f), torch.empty((Cn, zd), **f), torch.empty(zd, **f))
    logp, token, coef, dz, dc, dlv = out
    beta_dev = beta if torch.is_tensor(beta) else None
    ws = _workspace("prior_train", lib.evae_prior_train_workspace_bytes(B, Cn, zd), dev)
    st = prior_train_state(dev)
    c3 = coef if coef is not None else (None, None, None)
    _lib.check(lib.evae_prior_train_step(_p(z), B, _p(centres), Cn, zd, _p(log_var), _p(z_idx), _p(c_idx), float(c_total), _p(beta_dev),
                                         0.0 if beta_dev is not None else float(beta), _p(logp), _p(token), _p(c3[0]), _p(c3[1]), _p(c3[2]),
                                         _p(dz), _p(dc), _p(dlv), _p(st), _p(ws), ws.numel(), int(phase),
                                         _stream() if stream is None else stream), "evae_prior_train_step")
    return out


def prior_lse_bwd(z, centres, log_var, z_idx, c_idx, lse, grad_out):
    lib = _lib.load()
    _need_cuda(z, centres, log_var, lse, grad_out)
    if lse.numel() == z.shape[0] and lse.numel() > 0:      # a plain row log-sum-exp: token (lse, 0)
        lse = torch.stack((lse.reshape(-1).float(), torch.zeros_like(lse.reshape(-1), dtype=torch.float32)))
    assert lse.numel() == 2 * z.shape[0], "prior_lse_bwd: lse is the [2 x B] token of prior_merge (or a plain [B] log-sum-exp)"
    z, centres, log_var = _f32(z), _f32(centres), _f32(log_var).reshape(-1)
    lse, grad_out = _f32(lse), _f32(grad_out)
    B, zd = z.shape
    Cn = centres.shape[0]
    z_idx, c_idx = _i64(z_idx), _i64(c_idx)
    if z_idx is None or c_idx is None:
        z_idx = c_idx = None
    dz = torch.empty_like(z); dc = torch.empty_like(centres); dlv = torch.empty(zd, device=z.device)
    nb = lib.evae_prior_lse_bwd_workspace_bytes(B, Cn, zd)
    ws = _workspace("prior_bwd", nb, z.device)
    _lib.check(lib.evae_prior_lse_bwd(_p(z), B, _p(centres), Cn, zd, _p(log_var), _p(z_idx), _p(c_idx),
                                      _p(lse), _p(grad_out), _p(dz), _p(dc), _p(dlv), _p(ws), ws.numel(),
                                      _stream()), "evae_prior_lse_bwd")
    return dz, dc, dlv


class PriorLogP(torch.autograd.Function):
    """log p(z_i) under the exemplar mixture of ONE device's exemplars (models/BaseModel.py:98-128).
    forward(z, centres, log_var_row [zdim], z_idx|None, c_idx|None) -> logprior [B]."""

    @staticmethod
    def forward(ctx, z, centres, log_var_row, z_idx, c_idx):
        m, s, n, _ = prior_lse_fwd(z, centres, log_var_row, z_idx, c_idx)
        lp, lse = prior_merge(m, s, n, centres.shape[0])
        ctx.save_for_backward(z, centres, log_var_row, lse)
        ctx.idx = (z_idx, c_idx)
        return lp

    @staticmethod
    def backward(ctx, g):
        z, centres, log_var_row, lse = ctx.saved_tensors
        z_idx, c_idx = ctx.idx
        dz, dc, dlv = prior_lse_bwd(z, centres, log_var_row, z_idx, c_idx, lse, g.contiguous())
        return dz, dc, dlv.reshape(log_var_row.shape), None, None


# The captured step's promise (evae/graph.py): the loss is mean_i(beta KL_i - RE_i) with log p(z_i) entering KL_i with coefficient
# -1, and what follows is loss.backward(ones) -- so d loss / d log p(z_i) = -beta / B is known in the forward pass.  The runner
# puts its beta (device scalar) here around calculate_loss + backward; None: no promise.
STEP_BETA = [None]


class PriorLogPTrain(torch.autograd.Function):
    """PriorLogP under STEP_BETA's promise: forward partials, merge and backward in ONE launch (evae_prior_train_step,
    csrc/evae_prior_train.h); backward() hands out the gradients the forward pass computed (EVAE_PRIOR_TRAIN_CHECK=1: after
    checking the upstream gradient against -beta / B)."""

    @staticmethod
    def forward(ctx, z, centres, log_var_row, z_idx, c_idx, beta):
        logp, _tok, _c, dz, dc, dlv = prior_train_step(z.detach(), centres.detach(), log_var_row.detach(), z_idx, c_idx,
                                                       centres.shape[0], beta, want_coef=False)
        ctx.grads = (dz, dc, dlv.reshape(log_var_row.shape))
        ctx.beta = beta
        return logp

    @staticmethod
    def backward(ctx, g):
        dz, dc, dlv = ctx.grads
        if os.environ.get("EVAE_PRIOR_TRAIN_CHECK") == "1" and not torch.cuda.is_current_stream_capturing():
            want = -(ctx.beta if torch.is_tensor(ctx.beta) else torch.tensor(float(ctx.beta), device=g.device)) / g.numel()
            assert torch.allclose(g, want.expand_as(g), rtol=1e-6, atol=0.0), "PriorLogPTrain: the upstream gradient is not -beta / B"
        ctx.grads = None
        return dz, dc, dlv, None, None, None


def prior_logp(z, centres, log_var_row, z_idx, c_idx):
    """log p(z) [B] of ONE device's exemplars, differentiable: PriorLogP, or its one-launch training form under STEP_BETA"""
    beta = STEP_BETA[0]
    if (beta is not None and torch.is_grad_enabled() and z.is_cuda and z.dim() == 2 and centres.dim() == 2
            and z.dtype == torch.float32 and centres.dtype == torch.float32 and z.is_contiguous() and centres.is_contiguous()
            and (z.data_ptr() | centres.data_ptr()) % 16 == 0 and prior_train_applies(z.shape[0], centres.shape[0], z.shape[1])):
        return PriorLogPTrain.apply(z, centres, log_var_row, z_idx, c_idx, beta)
    return PriorLogP.apply(z, centres, log_var_row, z_idx, c_idx)


# ------------------------------------------------------------------------------------------------
# distance + top-K
# ------------------------------------------------------------------------------------------------
def pairdist_topk(q, cache, k, sqrt=False, index_base=0, want_val=True):
    lib = _lib.load()
    _need_cuda(q, cache)
    q, cache = _f32(q), _f32(cache)
    B, zd = q.shape
    N = cache.shape[0]
    idx = torch.empty((B, k), dtype=torch.int64, device=q.device)
    val = torch.empty((B, k), device=q.device) if want_val else None
    nb = lib.evae_pairdist_topk_workspace_bytes(B, N, zd, k)
    ws = _workspace("topk", nb, q.device)
    _lib.check(lib.evae_pairdist_topk(_p(q), B, _p(cache), N, zd, k, TOPK_SQRT if sqrt else 0,
                                      int(index_base), _p(idx), _p(val), _p(ws), ws.numel(), _stream()),
               "evae_pairdist_topk")
    return idx, val


def pairwise_distance(q, cache):
    """[B x N] fp32 squared distances (fp64-accumulated)."""
    lib = _lib.load()
    _need_cuda(q, cache)
    q, cache = _f32(q), _f32(cache)
    out = torch.empty((q.shape[0], cache.shape[0]), device=q.device)
    _lib.check(lib.evae_pairwise_distance(_p(q), q.shape[0], _p(cache), cache.shape[0], q.shape[1], _p(out),
                                          _stream()), "evae_pairwise_distance")
    return out


class PairwiseDistance(torch.autograd.Function):
    """utils/distributions.py:12-18 with its gradient: D_ij = |q_i - c_j|^2,
    dq = 2 (rowsum(G) q - G c), dc = 2 (colsum(G) c - G^T q), the two products on the dense GEMM kernels."""

    @staticmethod
    def forward(ctx, q, cache):
        ctx.save_for_backward(q, cache)
        return pairwise_distance(q.detach(), cache.detach())

    @staticmethod
    def backward(ctx, G):
        q, cache = ctx.saved_tensors
        lib = _lib.load()
        qf, cf, G = _f32(q.detach()), _f32(cache.detach()), _f32(G)
        B, zd = qf.shape
        Cn = cf.shape[0]
        dq = dc = None
        if ctx.needs_input_grad[0]:
            Gc = _bwd_data(G.data_ptr(), cf, None, None, B, Cn, Cn, G.device)          # G c  [B x z]
            dq = 2.0 * (G.sum(dim=1, keepdim=True) * qf - Gc)
        if ctx.needs_input_grad[1]:
            Gtq, colsum = _bwd_weight(G, qf, None, zd)                                    # G^T q [C x z], colsum(G) [C]
            dc = 2.0 * (colsum.unsqueeze(1) * cf - Gtq)
        return dq, dc


PRIOR_MASK_ALL = -3      # EVAE_PRIOR_MASK_ALL: a c_idx entry that masks its exemplar slot for every query
SELECT_EXEMPLARS_MAX = 16384   # evae_select_exemplars keeps a call's positions in one block's LDS (csrc/evae_topk.hip)


def select_exemplars(pos, cand_idx, want_count=False, out_rows=None):
    """Static-shape form of `unique` + gather (models/BaseModel.py:265-266): pos [n] top-k positions into cand_idx [C].
    -> (sel_rows [n] dataset rows of every slot, c_idx [n]: the row for the first slot naming a position, PRIOR_MASK_ALL
    for its repeats).  out_rows: a contiguous int64 [n] tensor to take sel_rows (the head of a step's gather list)."""
    lib = _lib.load()
    _need_cuda(pos, cand_idx)
    pos, cand_idx = _i64(pos), _i64(cand_idx)
    n = pos.numel()
    if out_rows is not None:
        assert out_rows.dtype == torch.int64 and out_rows.numel() == n and out_rows.is_contiguous()
    sel = out_rows if out_rows is not None else torch.empty(n, dtype=torch.int64, device=pos.device)
    cidx = torch.empty(n, dtype=torch.int64, device=pos.device)
    cnt = torch.empty(1, dtype=torch.int32, device=pos.device) if want_count else None
    _lib.check(lib.evae_select_exemplars(_p(pos), n, _p(cand_idx), cand_idx.numel(), _p(sel), _p(cidx), _p(cnt), _stream()),
               "evae_select_exemplars")
    return (sel, cidx, cnt) if want_count else (sel, cidx)


def topk_merge(val, idx):
    """[R x B x k] candidate lists -> global ([B x k] idx, [B x k] val)."""
    lib = _lib.load()
    _need_cuda(val, idx)
    val = _f32(val)
    idx = idx.long().contiguous()
    R, B, k = val.shape
    oi = torch.empty((B, k), dtype=torch.int64, device=val.device)
    ov = torch.empty((B, k), device=val.device)
    _lib.check(lib.evae_topk_merge(_p(val), _p(idx), R, B, k, _p(oi), _p(ov), _stream()), "evae_topk_merge")
    return oi, ov


# ------------------------------------------------------------------------------------------------
# dense layers
# ------------------------------------------------------------------------------------------------
def _vp(ptr):
    return C.c_void_p(ptr)


# Weight gradients of thin layers (a contraction over <= 128 batch rows) are leaves of the backward pass: nothing but the optimizer
# reads them, yet each one is a launch on the chain of data gradients (8-15 us apiece in the 2-level model's step, a dozen of them).
# Inside `with ops.deferred_wgrads():` (the captured step and utils.training put it around loss.backward()) such a gradient is only
# ALLOCATED where autograd asks for it and the products are formed behind the backward pass, six jobs per grouped launch
# (evae_dense_bwd_weight_group), on the stream the layer's backward ran on.  Only weights that ONE layer application uses per pass
# qualify (a weight met twice -- q(z2 | x) runs over the exemplar rows and over the batch, through whichever Functions -- has its two
# gradients added by autograd as soon as both exist): the scope walks the loss's autograd graph once and keeps the leaves that
# exactly one edge leads to; a deferred weight that turns up a second time after all is flushed at once with every stream made
# to wait for it.  EVAE_DEFER_WGRAD=0: off.
_DEFER = [None]
_DEFER_ON = os.environ.get("EVAE_DEFER_WGRAD", "1") != "0"


def _flush_wgrads(jobs, everyone_waits=False):
    """launch the pending jobs, grouped by the stream their layer's backward ran on; the current stream (everyone_waits: every
    registered stream) then waits for those launches"""
    lib = _lib.load()
    cur = torch.cuda.current_stream()
    by_stream = {}
    for j in jobs:
        by_stream.setdefault(j[8].cuda_stream, (j[8], []))[1].append(j)
    for st, js in by_stream.values():
        for i in range(0, len(js), 6):
            part = js[i:i + 6]
            arr = (_lib.WgradJob * len(part))()
            for a, (dy, M, N, x, K, dw, db, _key, _st, made_ev) in zip(arr, part):
                a.dy = dy.data_ptr(); a.x = x.data_ptr(); a.dw = dw.data_ptr(); a.db = db.data_ptr()
                a.M, a.N, a.K, a.ldy, a.ldx = M, N, K, dy.stride(0), x.stride(0)
                if made_ev is not None:            # dw += ...: behind the launch that wrote the other application's gradient
                    a.accumulate = 1
                    st.wait_event(made_ev)
            _lib.check(lib.evae_dense_bwd_weight_group(C.cast(arr, C.c_void_p), len(part), C.c_void_p(st.cuda_stream)),
                       "evae_dense_bwd_weight_group(deferred)")
        waiters = [cur]
        if everyone_waits:
            waiters += [s_ for s_ in _SIDE_STREAM_OBJS if s_.cuda_stream != cur.cuda_stream]
        for w_ in waiters:
            if w_.cuda_stream != st.cuda_stream:
                w_.wait_stream(st)


def _single_use_leaves(root):
    """(single, stealable): data pointers of the leaf tensors that exactly ONE edge of root's autograd graph leads to (one layer
    application per pass), and of the leaves whose AccumulateGrad will INSTALL the tensor it is handed instead of adding it to an
    existing one -- .grad is None and no tensor hooks.  Deferring (or accumulating in place) hands autograd a buffer that is
    filled later: a leaf with a gradient already there would get `grad += <unfilled buffer>` at once (zero_grad(set_to_none=
    False), gradient accumulation, a second backward), so only stealable leaves qualify; the others are computed on the spot."""
    counts = {}
    stealable = set()
    fn0 = getattr(root, "grad_fn", None)
    if fn0 is None:
        return set(), set()
    seen, stack = {fn0}, [fn0]
    while stack:
        n = stack.pop()
        for nf, _ in n.next_functions:
            if nf is None:
                continue
            v = getattr(nf, "variable", None)
            if v is not None:                      # AccumulateGrad of a leaf
                counts[v.data_ptr()] = counts.get(v.data_ptr(), 0) + 1
                if v.grad is None and not getattr(v, "_backward_hooks", None):
                    stealable.add(v.data_ptr())
            elif nf not in seen:
                seen.add(nf); stack.append(nf)
    return {k for k, c in counts.items() if c == 1 and k in stealable}, stealable


class deferred_wgrads:
    """with ops.deferred_wgrads(loss): loss.backward() -- `loss`: the tensor whose graph is about to be walked backwards (None: the
    scope defers nothing)"""

    def __init__(self, root=None):
        self.root = root

    def __enter__(self):
        self.prev = _DEFER[0]
        if _DEFER_ON and self.root is not None:
            single, stealable = _single_use_leaves(self.root)
            _DEFER[0] = {"jobs": [], "uses": {}, "made": {}, "single": single, "stealable": stealable}
        else:
            _DEFER[0] = None
        self.root = None
        return self

    def __exit__(self, et, ev, tb):
        cur, _DEFER[0] = _DEFER[0], self.prev
        if cur is not None and et is None and cur["jobs"]:
            _flush_wgrads(cur["jobs"])
        return False


def _try_defer(dy, x, K, key, dw=None, db=None):
    """(dw, db) views for autograd with the product left to the scope's grouped launches, or None (not deferrable: compute now)"""
    cur = _DEFER[0]
    if cur is None or key is None:
        return None
    M, N = dy.shape
    uses = cur["uses"][key] = cur["uses"].get(key, 0) + 1
    if uses > 1 and any(j[7] == key for j in cur["jobs"]):
        # a deferred weight used again in this pass: autograd is about to add the two gradients
        _flush_wgrads(cur["jobs"], everyone_waits=True); cur["jobs"] = []
        return None
    thin = (0 < M <= 128 and N % 4 == 0 and K % 4 == 0
            and dy.dtype == torch.float32 and x.dtype == torch.float32 and dy.stride(1) == 1 and x.stride(1) == 1
            and dy.stride(0) % 4 == 0 and x.stride(0) % 4 == 0 and (dy.data_ptr() | x.data_ptr()) % 16 == 0)
    made = cur["made"].get(key)
    if uses == 2 and made is not None and thin and _ACC_WGRAD and made[0].shape == (N, K) and made[1] is not None:
        # the layer's OTHER application (q(z2 | .) over the exemplar rows: a big GEMM, already launched) made this weight's
        # gradient in this pass: the batch rows' product is ADDED to it by a grouped launch behind the backward pass, and this
        # application hands autograd nothing -- the sum autograd would form with an element-wise launch per tensor (ten per step
        # of the 2-level model, four of them behind the last weight gradient, in front of the optimizer) is formed in place
        cur["jobs"].append((dy, M, N, x, K, made[0], made[1], key, torch.cuda.current_stream(), made[2]))
        return _ACCUMULATED
    if not (uses == 1 and key in cur["single"] and thin):
        return None
    if dw is None:
        dw = torch.empty((N, K), device=dy.device); db = torch.empty(N, device=dy.device)
    # (autograd gets VIEWS of dw / db: AccumulateGrad installs a gradient it is handed without copying only while nobody
    #  else holds that tensor object -- handed the buffers the job keeps, it would clone them, still unfilled; and a view
    #  made here holds its base, so it is the job that keeps the bases)
    cur["jobs"].append((dy, M, N, x, K, dw, db, key, torch.cuda.current_stream(), None))
    return dw.view(N, K), db.view(N)


_ACCUMULATED = (None, None)          # _bwd_weight's answer when the product joins a gradient made earlier in the pass
_ACC_WGRAD = os.environ.get("EVAE_ACC_WGRAD", "1") != "0"


def _note_made(key, dw, db):
    """A weight gradient computed NOW (not deferred) inside a deferred_wgrads scope: remembered under the weight's identity, with
    an event behind its launches, so that a second application of the layer can add its product to it (see _try_defer).  Returns
    the views to hand to autograd (the scope keeps the bases: handed the bases themselves, AccumulateGrad would clone them)."""
    cur = _DEFER[0]
    if (cur is None or key is None or not _ACC_WGRAD or db is None or key in cur["made"] or cur["uses"].get(key, 0) != 1
            or key not in cur["stealable"]):
        return dw, db
    ev = torch.cuda.Event(); ev.record()
    cur["made"][key] = (dw, db, ev)
    return dw.view(dw.shape), db.view(db.shape)


def _bwd_weight(dy, x, rows, K, want_db=True, key=None):
    """dw [N x K] = dy^T x(rows), db [N]; dy may be the combined [M x 2N] buffer [dh | dg].  `key`: identity of the weight
    (its data pointer) for the deferred form above."""
    lib = _lib.load()
    M, N = dy.shape
    dw = torch.empty((N, K), device=dy.device)
    db = torch.empty(N, device=dy.device) if want_db else None
    if rows is None and want_db and key is not None:
        d = _try_defer(dy, x, K, key, dw, db)
        if d is not None:
            return d
    elif key is not None and _DEFER[0] is not None:
        _DEFER[0]["uses"][key] = _DEFER[0]["uses"].get(key, 0) + 1
    nb = lib.evae_dense_bwd_weight_workspace_bytes(M, N, K)
    ws = _workspace("wgrad", nb, dy.device)
    _lib.check(lib.evae_dense_bwd_weight(_p(dy), M, N, dy.stride(0), _p(x), _p(rows), K, x.stride(0), _p(dw),
                                         _p(db), 0, _p(ws), ws.numel(), _stream()), "evae_dense_bwd_weight")
    return _note_made(key, dw, db)


def _bwd_data(dy1_ptr, w1, dy2_ptr, w2, M, N, ldy, device, out_prev=None, s_prev=None, out=None, dg_ptr=None, ldo=None):
    """dx [M x K] = dy1 W1 (+ dy2 W2); with out_prev/s_prev (forward output and gate of the layer below) the gate derivative of the layer below is applied
    in the epilogue and (dh, dg) are written instead (optionally into the halves of one [M x 2K] buffer)."""
    lib = _lib.load()
    K = w1.shape[1]
    if out is None:
        out = torch.empty((M, K), device=device)
        ldo = K
    np_ = 2 if dy2_ptr is not None else 1
    nb = lib.evae_dense_bwd_data_workspace_bytes(M, N, K, np_)
    ws = _workspace("dgrad", nb, device)
    out_ptr = out if isinstance(out, int) else out.data_ptr()
    _lib.check(lib.evae_dense_bwd_data(_vp(dy1_ptr), _p(w1), None if dy2_ptr is None else _vp(dy2_ptr), _p(w2),
                                       M, N, ldy, K, _p(out_prev), _p(s_prev), _vp(out_ptr),
                                       None if dg_ptr is None else _vp(dg_ptr), ldo, _p(ws), ws.numel(), _stream()),
               "evae_dense_bwd_data")
    return out


def _gated_bwd(dout, gout, s, wh, wg, dpre):
    """A gated layer's backward wrt its input (reference utils/nn.py:62-68 under autograd): fills dpre = [dh | dg] for the weight
    gradient and returns dx = dh Wh + dg Wg (evae_gated_dense_bwd: one launch for batch-sized row counts)."""
    lib = _lib.load()
    M, N = gout.shape
    K = wh.shape[1]
    dx = torch.empty((M, K), device=gout.device)
    ws = _workspace("dgrad", lib.evae_dense_bwd_data_workspace_bytes(M, N, K, 2), gout.device)
    _lib.check(lib.evae_gated_dense_bwd(_p(dout), dout.stride(0), _p(gout), _p(s), M, N, _p(wh), _p(wg), K, _p(dpre), 2 * N,
                                        _p(dx), K, _p(ws), ws.numel(), _stream()), "evae_gated_dense_bwd")
    return dx


def _rows_x(x, rows):
    """Validate the (x, rows) pair: x is [R x K] with unit inner stride; rows gathers M rows of it."""
    x = x if x.dtype == torch.float32 else x.float()
    if x.dim() != 2:
        x = x.reshape(x.shape[0], -1)
    if x.stride(1) != 1:
        x = x.contiguous()
    rows = _i64(rows)
    M = x.shape[0] if rows is None else rows.numel()
    return x, rows, M


class ExpandRowsFn(torch.autograd.Function):
    """Encodings of the DISTINCT exemplar rows [U x z] -> encodings of every draw [C x z] (out[j] = src[inv[j]]), for a captured
    step that encodes each distinct image once (the reference draws with replacement, models/BaseModel.py:245, and encodes every
    draw).  Backward: a distinct row's gradient = its multiplicity x the gradient of ONE of its draws (rep) -- duplicates have
    identical encodings, hence identical gradients from the prior; padding rows have multiplicity 0.  evae_gather_rows both ways."""

    @staticmethod
    def forward(ctx, src, inv, rep, mult):
        lib = _lib.load()
        src = _f32(src)
        if not src.is_contiguous():
            src = src.contiguous()
        n, z = inv.numel(), src.shape[1]
        out = torch.empty((n, z), device=src.device)
        _lib.check(lib.evae_gather_rows(_p(src), _p(inv), None, n, z, _p(out), _stream()), "evae_gather_rows")
        ctx.save_for_backward(rep, mult)
        ctx.u = src.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        rep, mult = ctx.saved_tensors
        g = _f32(g)
        if not g.is_contiguous():
            g = g.contiguous()
        out = torch.empty((ctx.u, g.shape[1]), device=g.device)
        _lib.check(lib.evae_gather_rows(_p(g), _p(rep), _p(mult), ctx.u, g.shape[1], _p(out), _stream()), "evae_gather_rows(bwd)")
        return out, None, None, None


def expand_rows(src, inv, rep, mult):
    return ExpandRowsFn.apply(src, inv, rep, mult)


class GatedDenseFn(torch.autograd.Function):
    """utils/nn.py:44-69 (activation None, gate on): h(x) * sigmoid(g(x)), optional row gather."""

    @staticmethod
    def forward(ctx, x, rows, wh, bh, wg, bg):
        lib = _lib.load()
        _need_cuda(x, rows, wh, wg)
        x, rows, M = _rows_x(x, rows)
        wh, wg = _f32(wh), _f32(wg)
        N, K = wh.shape
        out = torch.empty((M, N), device=x.device)
        need_grad = any(ctx.needs_input_grad)
        s = torch.empty_like(out) if need_grad else None     # the backward needs out and s only (dg = dout*out*(1-s))
        nb = lib.evae_dense_fwd_workspace_bytes(M, K, N, 1)
        ws = _workspace("fwd", nb, x.device)
        ex_, pipe_ = gemm_pipe(M, N, True, 2.0 * M * K * 2 * N) if rows is None else (None, "fp32-mfma")
        probed("gated_dense_fwd M=%d K=%d N=%d%s" % (M, K, N, " (row gather)" if rows is not None else ""), 2.0 * M * K * 2 * N,
               lambda: _lib.check(lib.evae_gated_dense_fwd(_p(x), _p(rows), M, K, x.stride(0), _p(wh), _p(bh), _p(wg), _p(bg),
                                                           N, _p(out), None, _p(s), _p(ws), ws.numel(), _stream()),
                                  "evae_gated_dense_fwd"), executed=ex_, pipe=pipe_)
        if need_grad:
            ctx.save_for_backward(x, rows, wh, wg, out, s)
        ctx.has_bias = (bh is not None, bg is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, rows, wh, wg, gout, s = ctx.saved_tensors
        M, N = gout.shape
        if not (dout.dtype == torch.float32 and dout.dim() == 2 and dout.stride(1) == 1 and dout.stride(0) >= N):
            dout = _f32(dout)           # (a column block of a wider gradient -- half of a torch.cat's -- is read where it lies)
        K = wh.shape[1]
        dpre = torch.empty((M, 2 * N), device=gout.device)          # [dh | dg]: one buffer, one weight-grad GEMM
        base = dpre.data_ptr()
        dx = None
        if ctx.needs_input_grad[0]:
            if rows is not None:
                raise _lib.EvaeError("gradient wrt a row-gathered input is not supported")
            dx = _gated_bwd(dout, gout, s, wh, wg, dpre)           # gate derivative + data gradient (one launch when batch-sized)
        else:
            _lib.check(lib.evae_gated_dense_bwd_input_ld(_p(dout), dout.stride(0), _p(gout), _p(s), M, N, _vp(base), _vp(base + 4 * N),
                                                         2 * N, _stream()), "evae_gated_dense_bwd_input_ld")
        dw, db = _bwd_weight(dpre, x, rows, K, key=wh.data_ptr())  # [dWh ; dWg], [dbh ; dbg]
        if dw is None:             # added, behind the backward pass, to the gradient the layer's other application made (_try_defer)
            return (dx, None, None, None, None, None)
        return (dx, None, dw[:N], (db[:N] if ctx.has_bias[0] else None), dw[N:],
                (db[N:] if ctx.has_bias[1] else None))


class GatedDenseDataFn(torch.autograd.Function):
    """GatedDenseFn without the parameters' gradients: (x, wh, bh, wg, bg) -> (out, s), backward = dx only."""

    @staticmethod
    def forward(ctx, x, wh, bh, wg, bg):
        lib = _lib.load()
        _need_cuda(x, wh, wg)
        x, _, M = _rows_x(x, None)
        wh, wg = _f32(wh), _f32(wg)
        N, K = wh.shape
        out = torch.empty((M, N), device=x.device); s = torch.empty_like(out)
        ws = _workspace("fwd", lib.evae_dense_fwd_workspace_bytes(M, K, N, 1), x.device)
        _lib.check(lib.evae_gated_dense_fwd(_p(x), None, M, K, x.stride(0), _p(wh), _p(bh), _p(wg), _p(bg), N, _p(out), None, _p(s),
                                            _p(ws), ws.numel(), _stream()), "evae_gated_dense_fwd")
        ctx.save_for_backward(wh, wg, out, s)
        ctx.mark_non_differentiable(s)
        ctx.set_materialize_grads(False)
        return out, s

    @staticmethod
    def backward(ctx, dout, _ds):
        lib = _lib.load()
        wh, wg, gout, s = ctx.saved_tensors
        if dout is None or not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        M, N = gout.shape
        if not (dout.dtype == torch.float32 and dout.dim() == 2 and dout.stride(1) == 1 and dout.stride(0) >= N):
            dout = _f32(dout)
        dpre = torch.empty((M, 2 * N), device=gout.device)
        return _gated_bwd(dout, gout, s, wh, wg, dpre), None, None, None, None


class GatedDenseParamFn(torch.autograd.Function):
    """The parameters' half of a gated layer as a node of its own: forward is an alias of the layer's output (no launch); the
    backward turns the output's gradient into (dWh, dbh, dWg, dbg).  Built under a third stream, autograd runs it there, beside
    the chain of data gradients instead of inside it."""

    @staticmethod
    def forward(ctx, out, s, x, wh, bh, wg, bg):
        ctx.save_for_backward(out, s, x)
        ctx.K = wh.shape[1]
        ctx.has_bias = (bh is not None, bg is not None)
        ctx.set_materialize_grads(False)
        return out.view_as(out)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        if dout is None:
            return (None,) * 7
        gout, s, x = ctx.saved_tensors
        M, N = gout.shape
        if not (dout.dtype == torch.float32 and dout.dim() == 2 and dout.stride(1) == 1 and dout.stride(0) >= N):
            dout = _f32(dout)
        st = torch.cuda.current_stream()
        for t in (gout, s, x, dout):       # made on another stream's pool, read here
            t.record_stream(st)
        dpre = torch.empty((M, 2 * N), device=gout.device)
        base = dpre.data_ptr()
        _lib.check(lib.evae_gated_dense_bwd_input_ld(_p(dout), dout.stride(0), _p(gout), _p(s), M, N, _vp(base), _vp(base + 4 * N),
                                                     2 * N, _stream()), "evae_gated_dense_bwd_input_ld")
        dw, db = _bwd_weight(dpre, x, None, ctx.K)
        return (None, None, None, dw[:N], (db[:N] if ctx.has_bias[0] else None), dw[N:], (db[N:] if ctx.has_bias[1] else None))


class _MergeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, token):
        return a.view_as(a)

    @staticmethod
    def backward(ctx, g):
        return g, g


def gated_dense_split(x, wh, bh, wg, bg, leaf):
    """gated_dense with the parameters' gradient node on the stream `leaf` (same values, same gradients)"""
    x2, _, _ = _rows_x(x, None)
    det = lambda t: None if t is None else t.detach()
    out, s = GatedDenseDataFn.apply(x2, det(wh), det(bh), det(wg), det(bg))
    with torch.cuda.stream(leaf):
        token = GatedDenseParamFn.apply(out.detach(), s, x2.detach(), wh, bh, wg, bg)
    return _MergeFn.apply(out, token)


def u8_prepare(wh, wg, out=None):
    """fp32 weight banks [N x K] -> the three-term bf16 tile images of the uint8 first-layer kernels"""
    lib = _lib.load()
    _need_cuda(wh, wg)
    wh, wg = _f32(wh.detach()), _f32(wg.detach())
    N, K = wh.shape
    nb = lib.evae_dense_u8_prepared_bytes(N, K)
    if out is None or out.numel() < nb:
        out = torch.empty(nb, dtype=torch.uint8, device=wh.device)
    _lib.check(lib.evae_dense_u8_prepare(_p(wh), _p(wg), N, K, _p(out), out.numel(), _stream()), "evae_dense_u8_prepare")
    return out


def gated_dense_fwd_u8(x_u8, rows, x_scale, prepared, bh, bg, N, out=None, save_s=None):
    """forward only: out [M x N] for the M gathered rows of the byte store x_u8 [R x K]"""
    lib = _lib.load()
    _need_cuda(x_u8, rows, prepared)
    assert x_u8.dtype == torch.uint8 and x_u8.dim() == 2 and x_u8.stride(1) == 1
    rows = _i64(rows)
    M, K = rows.numel(), x_u8.shape[1]
    if out is None:
        out = torch.empty((M, N), device=x_u8.device)
    _lib.check(lib.evae_gated_dense_fwd_u8(_p(x_u8), _p(rows), M, K, x_u8.stride(0), float(x_scale), _p(prepared), _p(bh),
                                           _p(bg), N, _p(out), _p(save_s), _stream()), "evae_gated_dense_fwd_u8")
    return out


def dense_bwd_weight_u8(dy, x_u8, rows, x_scale, dw=None, db=None, ws_name="wgrad_u8"):
    """dw [N x K] = x_scale * dy^T x_u8[rows], db [N] = column sums of dy (dy [M x N], any row stride)"""
    lib = _lib.load()
    _need_cuda(dy, x_u8, rows)
    assert dy.dtype == torch.float32 and dy.stride(1) == 1 and x_u8.dtype == torch.uint8
    rows = _i64(rows)
    M, N = dy.shape
    K = x_u8.shape[1]
    if dw is None:
        dw = torch.empty((N, K), device=dy.device)
    if db is None:
        db = torch.empty(N, device=dy.device)
    nb = lib.evae_dense_bwd_weight_u8_workspace_bytes(M, N, K)
    ws = _workspace(ws_name, nb, dy.device)
    _lib.check(lib.evae_dense_bwd_weight_u8(_p(dy), M, N, dy.stride(0), _p(x_u8), _p(rows), K, x_u8.stride(0), float(x_scale),
                                            _p(dw), _p(db), _p(ws), ws.numel(), _stream()), "evae_dense_bwd_weight_u8")
    return dw, db


class LinearFn(torch.autograd.Function):
    """act(x W^T + b): torch.nn.Linear / utils/nn.py:29-41 NonLinear."""

    @staticmethod
    def forward(ctx, x, rows, w, b, act, lo, hi):
        lib = _lib.load()
        _need_cuda(x, rows, w)
        x, rows, M = _rows_x(x, rows)
        w = _f32(w)
        N, K = w.shape
        y = torch.empty((M, N), device=x.device)
        need_grad = any(ctx.needs_input_grad)
        pre = torch.empty_like(y) if (need_grad and act == ACT_HARDTANH) else None
        nb = lib.evae_dense_fwd_workspace_bytes(M, K, N, 0)
        ws = _workspace("fwd", nb, x.device)
        _lib.check(lib.evae_linear_fwd(_p(x), _p(rows), M, K, x.stride(0), _p(w), _p(b), N, act, float(lo),
                                       float(hi), _p(y), _p(pre), _p(ws), ws.numel(), _stream()), "evae_linear_fwd")
        if need_grad:
            ctx.save_for_backward(x, rows, w, pre if pre is not None else y)
        ctx.act = (act, float(lo), float(hi))
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, rows, w, aux = ctx.saved_tensors
        act, lo, hi = ctx.act
        dy = _f32(dy)
        if act != ACT_NONE:
            dpre = torch.empty_like(dy)
            _lib.check(lib.evae_act_bwd(_p(dy), _p(aux), dy.numel(), act, lo, hi, _p(dpre), _stream()),
                       "evae_act_bwd")
        else:
            dpre = dy
        dw, db = _bwd_weight(dpre, x, rows, w.shape[1], key=w.data_ptr())
        dx = None
        if ctx.needs_input_grad[0]:
            if rows is not None:
                raise _lib.EvaeError("gradient wrt a row-gathered input is not supported")
            dx = _bwd_data(dpre.data_ptr(), w, None, None, dpre.shape[0], dpre.shape[1], dpre.stride(0), dpre.device)
        return dx, None, dw, (db if ctx.has_bias else None), None, None, None


class GatedDenseU8Fn(torch.autograd.Function):
    """GatedDense over M gathered rows of the uint8-resident image store (pixel = byte * x_scale): the first encoder layer of
    the exemplar rows on the modular autograd path (hvae_2level, or vae outside the fused node) -- same byte kernels as the
    fused step (csrc/evae_dense_u8.hip: bytes exact in bf16, weights / dy as three bf16 terms, fp32-GEMM accuracy)."""

    @staticmethod
    def forward(ctx, x_u8, rows, x_scale, wh, bh, wg, bg):
        _need_cuda(x_u8, rows, wh, wg)
        rows = _i64(rows)
        wh, wg = _f32(wh), _f32(wg)
        N, K = wh.shape
        M = rows.numel()
        prep = u8_prepare(wh.detach(), wg.detach(), out=_workspace("u8prep_mod%d" % N, _lib.load().evae_dense_u8_prepared_bytes(N, K),
                                                                   x_u8.device))
        need_grad = any(ctx.needs_input_grad)
        out = torch.empty((M, N), device=x_u8.device)
        s = torch.empty_like(out) if need_grad else None
        fl = 2.0 * M * K * 2 * N
        probed("gated_dense_fwd_u8 M=%d K=%d N=%d (uint8 rows, three bf16 terms)" % (M, K, N), fl,
               lambda: gated_dense_fwd_u8(x_u8, rows, x_scale, prep, bh, bg, N, out=out, save_s=s), executed=3 * fl, pipe="bf16-mfma")
        if need_grad:
            ctx.save_for_backward(x_u8, rows, out, s)
        ctx.wkey = wh.data_ptr()
        ctx.x_scale = float(x_scale)
        ctx.has_bias = (bh is not None, bg is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x_u8, rows, gout, s = ctx.saved_tensors
        dout = _f32(dout)
        M, N = gout.shape
        K = x_u8.shape[1]
        dpre = torch.empty((M, 2 * N), device=gout.device)          # [dh | dg]: one buffer, one weight-grad GEMM
        base = dpre.data_ptr()
        _lib.check(lib.evae_gated_dense_bwd_input(_p(dout), _p(gout), _p(s), M, N, _vp(base), _vp(base + 4 * N),
                                                  2 * N, _stream()), "evae_gated_dense_bwd_input")
        fl = 2.0 * M * 2 * N * K
        dw, db = probed("dense_bwd_weight_u8 M=%d N=%d K=%d (uint8 rows, three bf16 terms; pre-passes + GEMM + finish)" % (M, 2 * N, K),
                        fl, lambda: dense_bwd_weight_u8(dpre, x_u8, rows, ctx.x_scale, ws_name="wgrad_u8_mod"), executed=3 * fl,
                        pipe="bf16-mfma")
        if _DEFER[0] is not None:          # (the batch rows' application of this layer may add its product to this one: _try_defer)
            _DEFER[0]["uses"][ctx.wkey] = _DEFER[0]["uses"].get(ctx.wkey, 0) + 1
            dw, db = _note_made(ctx.wkey, dw, db)
        return (None, None, None, dw[:N], (db[:N] if ctx.has_bias[0] else None), dw[N:], (db[N:] if ctx.has_bias[1] else None))


def gated_dense(x, wh, bh, wg, bg, rows=None, x_scale=None):
    """x fp32 [R x K] (optionally row-gathered), or the uint8 image store with x_scale (pixel = byte * x_scale; rows required)"""
    if x.dtype == torch.uint8:
        if rows is None or x_scale is None:
            raise _lib.EvaeError("gated_dense on the uint8 store needs the gather list and x_scale")
        return GatedDenseU8Fn.apply(x, rows, float(x_scale), wh, bh, wg, bg)
    return GatedDenseFn.apply(x, rows, wh, bh, wg, bg)


def linear(x, w, b, act=ACT_NONE, lo=0.0, hi=0.0, rows=None):
    return LinearFn.apply(x, rows, w, b, act, lo, hi)


# ------------------------------------------------------------------------------------------------
# convolutions (implicit GEMM)
# ------------------------------------------------------------------------------------------------
def _conv_desc(x, w, stride, pad):
    N, Cc, H, W = x.shape
    Co, Ci, KH, KW = w.shape
    assert Ci == Cc, "groups != 1 is not supported"
    d = _lib.ConvDesc(N, Cc, H, W, Co, KH, KW, int(stride), int(pad))
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    return d, OH, OW


def _int1(v):
    if isinstance(v, (tuple, list)):
        assert all(e == v[0] for e in v), "only square stride / padding"
        return int(v[0])
    return int(v)


CL = torch.channels_last


def _cl(t):
    """channels-last storage ([N][H][W][C]) of a logical NCHW tensor"""
    return t if t.is_contiguous(memory_format=CL) else t.contiguous(memory_format=CL)


class Conv2dFn(torch.autograd.Function):
    """act(conv2d(x, wh) + bh) [* sigmoid(conv2d(x, wg) + bg)]  -- utils/nn.py:72-114, nn.Conv2d.

    Two kernel families behind one interface (include/evae_hip.h): channels-last GEMM instances when the geometry
    qualifies (input channels a multiple of 32: every layer but the first of the reference's conv stacks), the NCHW
    implicit-GEMM kernels otherwise.  Tensors keep their logical NCHW shape; outputs of the channels-last family
    are torch.channels_last tensors, so a stack of qualifying layers never changes layout."""

    @staticmethod
    def forward(ctx, x, wh, bh, wg, bg, stride, pad, act, lo, hi):
        lib = _lib.load()
        _need_cuda(x, wh, wg)
        x = x.float(); wh = _f32(wh); wg = None if wg is None else _f32(wg)
        d, OH, OW = _conv_desc(x, wh, stride, pad)
        gated = wg is not None
        need_grad = any(ctx.needs_input_grad)
        cl = bool(lib.evae_conv2d_cl_supported(C.byref(d), 0, int(gated)))
        if cl:
            x = _cl(x)
            fmt = dict(device=x.device, memory_format=CL)
        else:
            x = x.contiguous()
            fmt = dict(device=x.device)
        out = torch.empty((d.N, d.Co, OH, OW), **fmt)
        s = torch.empty((d.N, d.Co, OH, OW), **fmt) if (gated and need_grad) else None   # out and s suffice for the backward
        pre = torch.empty((d.N, d.Co, OH, OW), **fmt) if (not gated and need_grad and act == ACT_HARDTANH) else None
        if cl:
            if os.environ.get("EVAE_CONV_TRACE"):
                print("conv_cl_fwd N=%d C=%d H=%d W=%d Co=%d k=%d s=%d p=%d gated=%d grad=%d" % (d.N, d.C, d.H, d.W, d.Co, d.KH, d.stride, d.pad, gated, need_grad), flush=True)
            nb = lib.evae_conv2d_cl_workspace_bytes(C.byref(d), 0, int(gated))
            ws = _workspace("conv", nb, x.device)
            _lib.check(lib.evae_conv2d_cl_fwd(_p(x), C.byref(d), _p(wh), _p(bh), _p(wg), _p(bg), act, float(lo), float(hi),
                                              _p(out), _p(None if gated else pre), _p(s), _p(ws), ws.numel(), _stream()),
                       "evae_conv2d_cl_fwd")
        else:
            nb = lib.evae_conv2d_workspace_bytes(C.byref(d), 0, int(gated))
            ws = _workspace("conv", nb, x.device)
            _lib.check(lib.evae_conv2d_fwd(_p(x), C.byref(d), _p(wh), _p(bh), _p(wg), _p(bg), act, float(lo), float(hi),
                                           _p(out), _p(None if gated else pre), _p(s), _p(ws), ws.numel(), _stream()),
                       "evae_conv2d_fwd")
        if need_grad:
            ctx.save_for_backward(x, wh, wg, s, pre if pre is not None else out)
        ctx.cfg = (d, gated, act, float(lo), float(hi), bh is not None, bg is not None, cl)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, wh, wg, s, aux = ctx.saved_tensors
        d, gated, act, lo, hi, has_bh, has_bg, cl = ctx.cfg
        dev = dout.device
        shp = wh.shape
        ctot = d.Co * (2 if gated else 1)
        K = d.C * d.KH * d.KW
        OH, OW = dout.shape[2], dout.shape[3]
        # channels-last family for the gradients when this layer ran it forward (saved tensors are channels-last)
        # or when the weight gradient qualifies anyway (then x is re-laid out once)
        cl_w = bool(lib.evae_conv2d_cl_supported(C.byref(d), 2, int(gated))) and cl
        cl_d = bool(lib.evae_conv2d_cl_supported(C.byref(d), 1, int(gated))) and cl
        if cl:
            dout = _cl(dout.float())
        else:
            dout = _f32(dout)
        n = dout.numel()
        if cl:
            # [dh | dg | zero padding] per pixel in ONE buffer: logical [N, ldy, OH, OW], channels-last
            ldy = lib.evae_conv2d_cl_dy_stride(ctot)
            if ldy == ctot:
                dy = torch.empty((d.N, ldy, OH, OW), device=dev, memory_format=CL)
            else:
                dy = torch.empty((d.N, ldy, OH, OW), device=dev, memory_format=CL).zero_()
            if gated:
                if act != ACT_NONE:
                    raise _lib.EvaeError("gated conv with an activation on h: compose it from two plain convs")
                # rows = pixels, columns = Co channels: dh to columns [0, Co), dg to [Co, 2 Co)
                _lib.check(lib.evae_gated_dense_bwd_input(_p(dout), _p(aux), _p(s), n // d.Co, d.Co, _p(dy),
                                                          C.c_void_p(dy.data_ptr() + 4 * d.Co), ldy, _stream()),
                           "evae_gated_dense_bwd_input")
            elif ldy == ctot and act != ACT_NONE:
                _lib.check(lib.evae_act_bwd(_p(dout), _p(aux), n, act, lo, hi, _p(dy), _stream()), "evae_act_bwd")
            elif ldy == ctot:
                dy = dout
            else:
                src = dout
                if act != ACT_NONE:
                    src = torch.empty_like(dout)
                    _lib.check(lib.evae_act_bwd(_p(dout), _p(aux), n, act, lo, hi, _p(src), _stream()), "evae_act_bwd")
                dy[:, :ctot].copy_(src)
            dw = torch.empty((ctot, K), device=dev); db = torch.empty(ctot, device=dev)
            if cl_w:
                nb = lib.evae_conv2d_cl_workspace_bytes(C.byref(d), 2, int(gated))
                ws = _workspace("conv", nb, dev)
                _lib.check(lib.evae_conv2d_cl_bwd_weight(_p(dy), _p(x), C.byref(d), int(gated), _p(dw), _p(db), _p(ws),
                                                         ws.numel(), _stream()), "evae_conv2d_cl_bwd_weight")
            dx = None
            if ctx.needs_input_grad[0] and cl_d:
                dx = torch.empty(x.shape, device=dev, memory_format=CL)
                nb = lib.evae_conv2d_cl_workspace_bytes(C.byref(d), 1, int(gated))
                ws = _workspace("conv", nb, dev)
                _lib.check(lib.evae_conv2d_cl_bwd_data(_p(dy), _p(wh), _p(wg), C.byref(d), _p(dx), _p(ws), ws.numel(),
                                                       _stream()), "evae_conv2d_cl_bwd_data")
            if (not cl_w) or (ctx.needs_input_grad[0] and not cl_d):
                # the rest through the NCHW kernels (e.g. the data gradient of a 6-channel output layer)
                dhn = dy[:, :d.Co].contiguous()
                dgn = dy[:, d.Co:ctot].contiguous() if gated else None
                xn = x.contiguous()
                if not cl_w:
                    nb = lib.evae_conv2d_workspace_bytes(C.byref(d), 2, int(gated))
                    ws = _workspace("conv", nb, dev)
                    _lib.check(lib.evae_conv2d_bwd_weight(_p(dhn), _p(dgn), _p(xn), C.byref(d), _p(dw), _p(db), _p(ws),
                                                          ws.numel(), _stream()), "evae_conv2d_bwd_weight")
                if ctx.needs_input_grad[0] and not cl_d:
                    dx = torch.empty(xn.shape, device=dev)
                    nb = lib.evae_conv2d_workspace_bytes(C.byref(d), 1, int(gated))
                    ws = _workspace("conv", nb, dev)
                    _lib.check(lib.evae_conv2d_bwd_data(_p(dhn), _p(wh), _p(dgn), _p(wg), C.byref(d), _p(dx), _p(ws),
                                                        ws.numel(), _stream()), "evae_conv2d_bwd_data")
        else:
            if gated:
                if act != ACT_NONE:
                    raise _lib.EvaeError("gated conv with an activation on h: compose it from two plain convs")
                dh = torch.empty_like(dout); dg = torch.empty_like(dout)
                _lib.check(lib.evae_gated_dense_bwd_input(_p(dout), _p(aux), _p(s), 1, n, _p(dh), _p(dg), n, _stream()),
                           "evae_gated_dense_bwd_input")
            else:
                dg = None
                if act != ACT_NONE:
                    dh = torch.empty_like(dout)
                    _lib.check(lib.evae_act_bwd(_p(dout), _p(aux), n, act, lo, hi, _p(dh), _stream()), "evae_act_bwd")
                else:
                    dh = dout
            dw = torch.empty((ctot, K), device=dev); db = torch.empty(ctot, device=dev)
            nb = lib.evae_conv2d_workspace_bytes(C.byref(d), 2, int(gated))
            ws = _workspace("conv", nb, dev)
            _lib.check(lib.evae_conv2d_bwd_weight(_p(dh), _p(dg), _p(x), C.byref(d), _p(dw), _p(db), _p(ws), ws.numel(),
                                                  _stream()), "evae_conv2d_bwd_weight")
            dx = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                nb = lib.evae_conv2d_workspace_bytes(C.byref(d), 1, int(gated))
                ws = _workspace("conv", nb, dev)
                _lib.check(lib.evae_conv2d_bwd_data(_p(dh), _p(wh), _p(dg), _p(wg), C.byref(d), _p(dx), _p(ws), ws.numel(),
                                                    _stream()), "evae_conv2d_bwd_data")
        gwh = dw[:d.Co].reshape(shp)
        gwg = dw[d.Co:].reshape(shp) if gated else None
        return (dx, gwh, db[:d.Co] if has_bh else None, gwg, (db[d.Co:] if (gated and has_bg) else None),
                None, None, None, None, None)


class ResBlockFn(torch.autograd.Function):
    """x + conv2d(ELU(x), w) + b -- the residual block of models/fully_conv.py:13-23 on the channels-last kernels: ELU is one
    elementwise launch (its output is the convolution's operand AND what the backward needs), the residual add rides in the
    convolution's epilogue, and the block's data gradient dy + ELU'(x) * conv_transpose(dy, w) is ONE launch."""

    @staticmethod
    def forward(ctx, x, w, b):
        lib = _lib.load()
        _need_cuda(x, w)
        x = _cl(x.float()); w = _f32(w)
        d, OH, OW = _conv_desc(x, w, 1, (w.shape[2] - 1) // 2)
        a = torch.empty_like(x, memory_format=CL)
        _lib.check(lib.evae_elu_fwd(_p(x), x.numel(), _p(a), _stream()), "evae_elu_fwd")
        out = torch.empty_like(x, memory_format=CL)
        ws = _workspace("conv", lib.evae_conv2d_cl_workspace_bytes(C.byref(d), 0, 0), x.device)
        _lib.check(lib.evae_conv2d_cl_fwd_res(_p(a), C.byref(d), _p(w), _p(b), _p(x), _p(out), _p(ws), ws.numel(), _stream()),
                   "evae_conv2d_cl_fwd_res")
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(a, w)
        ctx.cfg = (d, b is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        a, w = ctx.saved_tensors
        d, has_b = ctx.cfg
        dev = dout.device
        dy = _cl(dout.float())
        K = d.C * d.KH * d.KW
        dw = db = None
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            dw = torch.empty((d.Co, K), device=dev); db = torch.empty(d.Co, device=dev)
            ws = _workspace("conv", lib.evae_conv2d_cl_workspace_bytes(C.byref(d), 2, 0), dev)
            _lib.check(lib.evae_conv2d_cl_bwd_weight(_p(dy), _p(a), C.byref(d), 0, _p(dw), _p(db), _p(ws), ws.numel(), _stream()),
                       "evae_conv2d_cl_bwd_weight")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(a, memory_format=CL)
            ws = _workspace("conv", lib.evae_conv2d_cl_workspace_bytes(C.byref(d), 1, 0), dev)
            _lib.check(lib.evae_conv2d_cl_bwd_data_res(_p(dy), _p(w), C.byref(d), _p(dy), _p(a), _p(dx), _p(ws), ws.numel(),
                                                       _stream()), "evae_conv2d_cl_bwd_data_res")
        return dx, (None if dw is None else dw.reshape(w.shape)), (db if has_b else None)


def res_block_supported(x, w, stride, padding):
    """ELU -> 3x3 'same' convolution -> + x on the fused path?  The geometry rules live in ONE place, the library's
    evae_conv2d_cl_res_supported (shape kept, stride 1, channel counts / workspace / images-per-pass limits of the
    channels-last kernels); anything it refuses runs as x + f(x) on the plain layers."""
    if not (x.is_cuda and x.dim() == 4 and w.dim() == 4 and w.shape[1] == x.shape[1]):
        return False
    d, _, _ = _conv_desc(x, w, _int1(stride), _int1(padding))
    return bool(_lib.load().evae_conv2d_cl_res_supported(C.byref(d)))


def res_block(x, w, b):
    return ResBlockFn.apply(x, w, b)


def conv2d(x, w, b, stride=1, padding=0, act=ACT_NONE, lo=0.0, hi=0.0):
    return Conv2dFn.apply(x, w, b, None, None, _int1(stride), _int1(padding), act, lo, hi)


def gated_conv2d(x, wh, bh, wg, bg, stride=1, padding=0):
    return Conv2dFn.apply(x, wh, bh, wg, bg, _int1(stride), _int1(padding), ACT_NONE, 0.0, 0.0)


# ------------------------------------------------------------------------------------------------
# a stack of gated convolutions on pre-split pixel images (csrc/evae_conv_win.h)
# ------------------------------------------------------------------------------------------------
CONV_STACK_MIN_IMAGES = int(os.environ.get("EVAE_CONV_STACK_MIN", "1024"))     # below: the layer-by-layer path (batch rows)
CONV_STACK_ON = os.environ.get("EVAE_CONV_STACK", "1") != "0"


def _pad32(c):
    return (int(c) + 31) // 32 * 32


def conv_stack_depth(x_shape, layers):
    """How many leading layers of a stack of gated convolutions `layers` = [(wh, stride, pad), ...] the image pipeline takes:
    layer 0 on the channels-last kernels (an input that is data: no gradient), layers 1 .. b - 1 on the window kernels with their
    activations as pixel images.  0 = not worth it / not supported."""
    lib = _lib.load()
    N, Cc, H, W = x_shape
    if len(layers) < 2:
        return 0
    # ok_inner[i]: layer i can be an INNER layer of the run (its output feeds the next layer as an image: Co % 16 == 0, real width);
    # ok_last[i]: it can be the LAST TAKEN layer -- GatedConvStackFn zero-pads that one's filters to a multiple of 32 output
    # channels whenever Co % 32 != 0 (a 6-channel layer as 32: five times the matrix work of a layer that has 1 % of the stack's), so
    # the padded geometry is what has to be supported.  Every layer >= 1 needs a weight-gradient path: the window kernel, or
    # the channels-last one (the truncated-depth case checked the unpadded width, and no weight-gradient fallback)
    ok_inner, ok_last = [], []
    for i, (wh, st, pd) in enumerate(layers):
        Co, Ci, KH, KW = wh.shape
        if Ci != Cc:
            break

        def supported(co):
            d = _lib.ConvDesc(N, Cc, H, W, co, KH, KW, int(st), int(pd))
            if i == 0:
                return bool(lib.evae_conv2d_cl_supported(C.byref(d), 0, 1) and lib.evae_conv2d_cl_supported(C.byref(d), 2, 1))
            return bool(lib.evae_cw_supported(C.byref(d), 0) and lib.evae_cw_supported(C.byref(d), 1)
                        and (lib.evae_cw_supported(C.byref(d), 2) or lib.evae_conv2d_cl_supported(C.byref(d), 2, 1)))
        ok_inner.append(Co % 16 == 0 and supported(Co))
        ok_last.append(i >= 1 and (supported(Co) if Co % 32 == 0 else supported(_pad32(Co))))
        if not ok_inner[-1]:
            break
        Cc, H, W = Co, (H + 2 * pd - KH) // st + 1, (W + 2 * pd - KW) // st + 1
    # the longest prefix whose inner layers are all fine and whose last layer can close the run
    for b in range(len(ok_last), 1, -1):
        if ok_last[b - 1] and all(ok_inner[:b - 1]):
            return b
    return 0


class GatedConvStackFn(torch.autograd.Function):
    """Layers 0 .. b - 1 of a stack of gated convolutions act(h(x)) * sigmoid(g(x)) (reference utils/nn.py:72-97, as
    models/convHVAE_2level.py:21-46 chains them) with every activation between two layers as a pre-split pixel image
    (include/evae_hip.h, evae_cw_*): a layer's epilogue writes the image the next layer's window loads read; in the backward pass a
    layer's data gradient applies the gate derivative of the layer below in its epilogue and writes the merged [dh | dg] image
    that layer's own gradients read -- no elementwise pass, no patch matrix, no fp32 activation except where a kernel outside
    the family still wants one.  Layer 0 reads the data itself (evae_cw_first_*: exact fp32 from a window of the input in LDS, no
    patch matrix).  x is data (no gradient).  args: x, n, cfg = ((stride, pad), ...), then wh, bh, wg, bg per layer."""

    @staticmethod
    def forward(ctx, x, cfg, *params):
        lib = _lib.load()
        b = len(cfg)
        L = [list(params[4 * i:4 * i + 4]) for i in range(b)]
        _need_cuda(x, *[t for t in params if t is not None])
        co_real = [int(L[i][0].shape[0]) for i in range(b)]
        if b >= 2 and co_real[b - 1] % 32:
            # last layer of the module: output channels zero-padded to a multiple of 32 (conv_stack_depth)
            padc = _pad32(co_real[b - 1]) - co_real[b - 1]
            wh_, bh_, wg_, bg_ = L[b - 1]
            zw = wh_.new_zeros((padc,) + tuple(wh_.shape[1:]))
            L[b - 1] = [torch.cat((_f32(wh_), zw)), None if bh_ is None else torch.cat((bh_.float(), bh_.new_zeros(padc))),
                        torch.cat((_f32(wg_), zw)), None if bg_ is None else torch.cat((bg_.float(), bg_.new_zeros(padc)))]
        x = _cl(x.float())
        dev = x.device
        need_grad = any(ctx.needs_input_grad[2:])
        N = x.shape[0]
        ds, shp = [], []
        Cc, H, W = x.shape[1], x.shape[2], x.shape[3]
        for i in range(b):
            Co, Ci, KH, KW = L[i][0].shape
            st, pd = cfg[i]
            ds.append(_lib.ConvDesc(N, Cc, H, W, Co, KH, KW, int(st), int(pd)))
            Cc, H, W = Co, (H + 2 * pd - KH) // st + 1, (W + 2 * pd - KW) // st + 1
            shp.append((Co, H, W))
        wg_cw = [i >= 1 and bool(lib.evae_cw_supported(C.byref(ds[i]), 2)) for i in range(b)]     # weight gradient on the window kernels
        planar = [i + 1 < b and cfg[i + 1][0] == 2 for i in range(b)]                                # row order of layer i's output image
        fmt = dict(device=dev, memory_format=CL)

        def image(rows, ch):
            return torch.empty(int(lib.evae_cw_image_bytes(rows, ch)), dtype=torch.uint8, device=dev)

        # layer 0 on the data: the first-layer kernel (fp32 window in LDS -> image + gate), or the channels-last kernels + a packing pass
        d0 = ds[0]
        Co0, H0, W0 = shp[0]
        wh0, bh0, wg0, bg0 = L[0]
        first_fwd = bool(lib.evae_cw_supported(C.byref(d0), 3))
        first_wg = bool(lib.evae_cw_supported(C.byref(d0), 4))
        need_out0 = need_grad and not wg_cw[1]
        imgs = [image(N * H0 * W0, Co0)]
        s0 = torch.empty((N, Co0, H0, W0), **fmt) if (need_grad or not first_fwd) else None
        out0 = torch.empty((N, Co0, H0, W0), **fmt) if (need_out0 or not first_fwd) else None
        if first_fwd:
            _lib.check(lib.evae_cw_first_fwd(_p(x), C.byref(d0), _p(_f32(wh0)), _p(bh0), _p(_f32(wg0)), _p(bg0), _p(imgs[0]), int(planar[0]),
                                             _p(s0), _p(out0), _stream()), "evae_cw_first_fwd")
        else:
            ws = _workspace("conv", lib.evae_conv2d_cl_workspace_bytes(C.byref(d0), 0, 1), dev)
            _lib.check(lib.evae_conv2d_cl_fwd(_p(x), C.byref(d0), _p(_f32(wh0)), _p(bh0), _p(_f32(wg0)), _p(bg0), ACT_NONE, 0.0, 0.0,
                                              _p(out0), None, _p(s0), _p(ws), ws.numel(), _stream()), "evae_conv2d_cl_fwd")
            _lib.check(lib.evae_cw_pack_image(_p(out0), N, H0, W0, Co0, int(planar[0]), _p(imgs[0]), _stream()), "evae_cw_pack_image")
        outf = [out0 if need_out0 else None]         # fp32 copies: the input of a layer whose weight gradient runs outside the family
        gates = [s0 if need_grad else None]
        for i in range(1, b):
            Co, Hh, Ww = shp[i]
            wh, bh, wg, bg = L[i]
            oimg = image(N * Hh * Ww, Co)
            last = i == b - 1
            s = torch.empty((N, Co, Hh, Ww), **fmt) if need_grad else None
            of = torch.empty((N, Co, Hh, Ww), **fmt) if (last or (need_grad and not wg_cw[i + 1])) else None
            ws = _workspace("cw", lib.evae_cw_workspace_bytes(C.byref(ds[i]), 0), dev)
            _lib.check(lib.evae_cw_fwd_gated(_p(imgs[i - 1]), C.byref(ds[i]), _p(_f32(wh)), _p(bh), _p(_f32(wg)), _p(bg), _p(oimg), int(planar[i]),
                                             _p(s), _p(of), _p(ws), ws.numel(), _stream()), "evae_cw_fwd_gated")
            imgs.append(oimg); gates.append(s); outf.append(of)
            if not need_grad:
                imgs[i - 1] = None
        out = outf[b - 1]
        if need_grad:
            ctx.save_for_backward(x, *[t for t in params if t is not None])
            ctx.keep = (imgs, gates, outf[:b - 1], ds, shp, planar, wg_cw, [[t is not None for t in params[4 * i:4 * i + 4]] for i in range(b)], first_wg,
                        co_real)
        return out if co_real[b - 1] == shp[b - 1][0] else out[:, :co_real[b - 1]]

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        imgs, gates, outf, ds, shp, planar, wg_cw, has, first_wg, co_real = ctx.keep
        saved = list(ctx.saved_tensors)
        x = saved.pop(0)
        b = len(ds)
        L = []
        for i in range(b):
            L.append([saved.pop(0) if h else None for h in has[i]])
        dev = dout.device
        N = x.shape[0]
        if co_real[b - 1] != shp[b - 1][0]:       # the padded last layer: its missing channels have no upstream gradient
            full = torch.empty((N, shp[b - 1][0], dout.shape[2], dout.shape[3]), device=dev, memory_format=CL).zero_()
            full[:, :co_real[b - 1]] = dout
            dout = full
            wlast = L[b - 1]
            padc = shp[b - 1][0] - co_real[b - 1]
            zw = wlast[0].new_zeros((padc,) + tuple(wlast[0].shape[1:]))
            L[b - 1] = [torch.cat((_f32(wlast[0]), zw)), wlast[1], torch.cat((_f32(wlast[2]), zw)), wlast[3]]
        dout = _cl(dout.float())
        grads = [None] * (4 * b)

        def image(rows, ch):
            return torch.empty(int(lib.evae_cw_image_bytes(rows, ch)), dtype=torch.uint8, device=dev)

        def put(i, dw, db):
            Co, Cr = shp[i][0], co_real[i]
            wshape = (Cr,) + tuple(L[i][0].shape[1:])
            grads[4 * i] = dw[:Cr].reshape(wshape)
            grads[4 * i + 2] = dw[Co:Co + Cr].reshape(wshape)
            if has[i][1]:
                grads[4 * i + 1] = db[:Cr]
            if has[i][3]:
                grads[4 * i + 3] = db[Co:Co + Cr]

        # exit of the stack: gate derivative of the last layer from the fp32 upstream gradient
        Co, Hh, Ww = shp[b - 1]
        dyimg = image(N * Hh * Ww, 2 * Co)
        dyf = torch.empty((N, 2 * Co, Hh, Ww), device=dev, memory_format=CL) if not wg_cw[b - 1] else None
        _lib.check(lib.evae_cw_gate_bwd_image(_p(dout), _p(imgs[b - 1]), 0, _p(gates[b - 1]), N, Hh, Ww, Co, _p(dyimg), _p(dyf), _stream()),
                   "evae_cw_gate_bwd_image")
        imgs[b - 1] = None; gates[b - 1] = None
        for i in range(b - 1, 0, -1):
            d = ds[i]
            Co = shp[i][0]
            K = d.C * d.KH * d.KW
            dw = torch.empty((2 * Co, K), device=dev); db = torch.empty(2 * Co, device=dev)
            if wg_cw[i]:
                ws = _workspace("cw", lib.evae_cw_workspace_bytes(C.byref(d), 2), dev)
                _lib.check(lib.evae_cw_bwd_weight(_p(dyimg), int(planar[i]), _p(imgs[i - 1]), C.byref(d), _p(dw), _p(db), _p(ws), ws.numel(),
                                                  _stream()), "evae_cw_bwd_weight")
            else:
                ws = _workspace("conv", lib.evae_conv2d_cl_workspace_bytes(C.byref(d), 2, 1), dev)
                _lib.check(lib.evae_conv2d_cl_bwd_weight(_p(dyf), _p(outf[i - 1]), C.byref(d), 1, _p(dw), _p(db), _p(ws), ws.numel(), _stream()),
                           "evae_conv2d_cl_bwd_weight")
            put(i, dw, db)
            # data gradient + gate derivative of layer i - 1
            Cp, Hp, Wp = shp[i - 1]
            want_img = i - 1 >= 1
            want_f = (i - 1 == 0) or not wg_cw[i - 1]
            nimg = image(N * Hp * Wp, 2 * Cp) if want_img else None
            nf = torch.empty((N, 2 * Cp, Hp, Wp), device=dev, memory_format=CL) if want_f else None
            ws = _workspace("cw", lib.evae_cw_workspace_bytes(C.byref(d), 1), dev)
            _lib.check(lib.evae_cw_bwd_data_gate(_p(dyimg), int(planar[i]), C.byref(d), _p(_f32(L[i][0])), _p(_f32(L[i][2])), _p(imgs[i - 1]),
                                                 _p(gates[i - 1]), _p(nimg), _p(nf), _p(ws), ws.numel(), _stream()), "evae_cw_bwd_data_gate")
            imgs[i - 1] = None; gates[i - 1] = None
            if i - 1 < len(outf):
                pass
            dyimg, dyf = nimg, nf
        # layer 0: weight gradient on the channels-last kernels from the merged fp32 gradient
        d0 = ds[0]
        Co0 = shp[0][0]
        K0 = d0.C * d0.KH * d0.KW
        dw = torch.empty((2 * Co0, K0), device=dev); db = torch.empty(2 * Co0, device=dev)
        if first_wg:
            ws = _workspace("cw", lib.evae_cw_first_workspace_bytes(), dev)
            _lib.check(lib.evae_cw_first_bwd_weight(_p(dyf), _p(x), C.byref(d0), _p(dw), _p(db), _p(ws), ws.numel(), _stream()),
                       "evae_cw_first_bwd_weight")
        else:
            ws = _workspace("conv", lib.evae_conv2d_cl_workspace_bytes(C.byref(d0), 2, 1), dev)
            _lib.check(lib.evae_conv2d_cl_bwd_weight(_p(dyf), _p(x), C.byref(d0), 1, _p(dw), _p(db), _p(ws), ws.numel(), _stream()),
                       "evae_conv2d_cl_bwd_weight")
        put(0, dw, db)
        ctx.keep = None
        return (None, None) + tuple(grads)


def _cup(c, m):
    return (c + m - 1) // m * m


def plain_conv_supported(x, w, stride, padding, need_dx=None, upsample=False):
    """A 3 x 3 'same' convolution (stride 1 or 2, any channel counts <= 128) on the window kernels?  (enough pixels to fill the machine;
    upsample: x is the half-resolution tensor in front of nn.Upsample(scale_factor=2))"""
    if not (CONV_STACK_ON and x.is_cuda and x.dim() == 4 and w.dim() == 4):
        return False
    N, Cc, H, W = x.shape
    if upsample:
        H, W = 2 * H, 2 * W
    Co, Ci, KH, KW = w.shape
    st = stride[0] if isinstance(stride, (tuple, list)) else stride
    pd = padding[0] if isinstance(padding, (tuple, list)) else padding
    if Ci != Cc or KH != 3 or KW != 3 or pd != 1 or st not in (1, 2) or N * H * W < RES_STACK_MIN_PIXELS // 4:
        return False
    lib = _lib.load()
    d = _lib.ConvDesc(N, Cc, H, W, Co, 3, 3, st, 1)
    need_dx = x.requires_grad if need_dx is None else need_dx
    grads = torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)
    ok = bool(lib.evae_cw_plain_supported(C.byref(d), 0))
    if ok and grads:
        ok = bool(lib.evae_cw_plain_supported(C.byref(d), 2)) and (not need_dx or bool(lib.evae_cw_plain_supported(C.byref(d), 1)))
    return ok


class PlainConvFn(torch.autograd.Function):
    """y = conv2d(x, w, b) (3 x 3, 'same', stride 1 or 2), optionally ELU(y), on the window kernels over pixel images: the weight-normed
    convolutions outside fully_conv's residual runs and its 3-channel output head (reference models/fully_conv.py:41-58).  The input is
    split into its image once (parity-planar rows for stride 2) and kept for the weight gradient; the gradient arrives in whatever
    layout autograd hands it (NCHW from the likelihood) and is packed -- times ELU'(y) when the ELU was fused -- into the image both
    gradients read."""

    @staticmethod
    def forward(ctx, x, w, b, stride, elu, upsample=False):
        lib = _lib.load()
        w = _f32(w)
        _need_cuda(x, w)
        x = x.float()
        nchw = 1 if (x.is_contiguous() and not x.is_contiguous(memory_format=CL)) else 0       # (the data arrives as NCHW planes)
        if not nchw:
            x = _cl(x)
        dev = x.device
        N, Cc, H, W = x.shape
        if upsample:                                   # nn.Upsample(scale_factor=2) in front of the convolution: done by the pack
            H, W = 2 * H, 2 * W
        Co = w.shape[0]
        st = int(stride)
        OH, OW = H // st, W // st
        d = _lib.ConvDesc(N, Cc, H, W, Co, 3, 3, st, 1)
        Cp, Cop, Co8 = _cup(Cc, 16), _cup(Co, 16), _cup(Co, 8)
        ximg = torch.empty(int(lib.evae_cw_image_bytes(N * H * W, Cp)), dtype=torch.uint8, device=dev)
        _lib.check(lib.evae_cw_pack_image_ex(_p(x), Cc, Cc, nchw, None, 0, N, H, W, Cp, (1 if st == 2 else 0) | (4 if upsample else 0), _p(ximg),
                                             _stream()), "evae_cw_pack_image_ex")
        out = torch.empty((N, OH, OW, Co8), device=dev)
        ws = _workspace("cw", lib.evae_cw_plain_workspace_bytes(C.byref(d), 0), dev)
        _lib.check(lib.evae_cw_plain_fwd(_p(ximg), C.byref(d), _p(w), _p(b), 1 if elu else 0, _p(out), Co8, None, 0, _p(ws), ws.numel(), _stream()),
                   "evae_cw_plain_fwd")
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(w)
            ctx.keep = (ximg, d, out if elu else None, b is not None, (Cp, Cop, Co8), bool(upsample))
        return out.permute(0, 3, 1, 2)[:, :Co]

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (w,) = ctx.saved_tensors
        ximg, d, act, has_b, (Cp, Cop, Co8), upsample = ctx.keep
        ctx.keep = None
        dev = dy.device
        N, Cc, H, W, Co, st = d.N, d.C, d.H, d.W, d.Co, d.stride
        OH, OW = H // st, W // st
        dy = dy.float()
        if dy.is_contiguous():
            nchw, ldy = 1, 0
        elif dy.permute(0, 2, 3, 1).is_contiguous():
            nchw, ldy = 0, Co
        else:
            dy, nchw, ldy = dy.contiguous(), 1, 0
        dyimg = torch.empty(int(lib.evae_cw_image_bytes(N * OH * OW, Cop)), dtype=torch.uint8, device=dev)
        _lib.check(lib.evae_cw_pack_image_ex(_p(dy), ldy, Co, nchw, _p(act), Co8, N, OH, OW, Cop, 0, _p(dyimg), _stream()), "evae_cw_pack_image_ex(dy)")
        gw = gb = gx = None
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            r = _lib.ConvDesc(N, Cp, H, W, Cop, 3, 3, st, 1)
            dwp = torch.empty((Cop, Cp, 3, 3), device=dev); dbp = torch.empty(Cop, device=dev)
            ws = _workspace("cw", lib.evae_cw_plain_workspace_bytes(C.byref(d), 2), dev)
            _lib.check(lib.evae_cw_bwd_weight_plain(_p(dyimg), _p(ximg), C.byref(r), _p(dwp), _p(dbp), _p(ws), ws.numel(), _stream()),
                       "evae_cw_bwd_weight_plain")
            gw = dwp[:Co, :Cc].contiguous() if (Cop != Co or Cp != Cc) else dwp
            gb = dbp[:Co] if has_b else None
        if ctx.needs_input_grad[0]:
            C8 = _cup(Cc, 8)
            dx = torch.empty((N, H, W, C8), device=dev)
            ws = _workspace("cw", lib.evae_cw_plain_workspace_bytes(C.byref(d), 1), dev)
            _lib.check(lib.evae_cw_plain_bwd_data(_p(dyimg), 0, C.byref(d), _p(w), _p(dx), C8, None, _p(ws), ws.numel(), _stream()),
                       "evae_cw_plain_bwd_data")
            if upsample:
                lo = torch.empty((N, H // 2, W // 2, C8), device=dev)
                _lib.check(lib.evae_cw_upsample2_bwd(_p(dx), C8, N, H, W, C8, _p(lo), _stream()), "evae_cw_upsample2_bwd")
                dx = lo
            gx = dx.permute(0, 3, 1, 2)[:, :Cc]
        return gx, gw, gb, None, None, None


def plain_conv(x, w, b, stride=1, elu=False, upsample=False):
    return PlainConvFn.apply(x, w, b, int(stride[0] if isinstance(stride, (tuple, list)) else stride), bool(elu), bool(upsample))


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


class WeightNormSetFn(torch.autograd.Function):
    """w_i = v_i * (g_i / ||v_i||) (norm over all but dim 0: torch.nn.utils.weight_norm as reference models/fully_conv.py:18,41-58
    wraps its convolutions) for a SET of filters in ONE launch, and one more for all their (dv_i, dg_i).  args: v_0, g_0, v_1, g_1, ..."""

    @staticmethod
    def forward(ctx, *vg):
        lib = _lib.load()
        vs = [_f32(t) for t in vg[0::2]]
        gs = [_f32(t) for t in vg[1::2]]
        _need_cuda(*vs, *gs)
        ws = [torch.empty_like(v) for v in vs]
        rows = (C.c_int * len(vs))(*[v.shape[0] for v in vs])
        cols = (C.c_int * len(vs))(*[v.numel() // v.shape[0] for v in vs])
        for k in range(0, len(vs), 32):
            n = min(32, len(vs) - k)
            _lib.check(lib.evae_weight_norm_set_fwd(n, _ptr_array(vs[k:k + n]), _ptr_array(gs[k:k + n]), _ptr_array(ws[k:k + n]),
                                                    C.byref(rows, 4 * k), C.byref(cols, 4 * k), _stream()), "evae_weight_norm_set_fwd")
        ctx.save_for_backward(*vs, *gs)
        return tuple(ws)

    @staticmethod
    def backward(ctx, *dws):
        lib = _lib.load()
        saved = ctx.saved_tensors
        n_all = len(saved) // 2
        vs, gs = list(saved[:n_all]), list(saved[n_all:])
        dws = [torch.zeros_like(v) if d is None else _f32(d) for d, v in zip(dws, vs)]
        dvs = [torch.empty_like(v) for v in vs]
        dgs = [torch.empty_like(g) for g in gs]
        rows = (C.c_int * n_all)(*[v.shape[0] for v in vs])
        cols = (C.c_int * n_all)(*[v.numel() // v.shape[0] for v in vs])
        for k in range(0, n_all, 32):
            n = min(32, n_all - k)
            _lib.check(lib.evae_weight_norm_set_bwd(n, _ptr_array(vs[k:k + n]), _ptr_array(gs[k:k + n]), _ptr_array(dws[k:k + n]),
                                                    _ptr_array(dvs[k:k + n]), _ptr_array(dgs[k:k + n]), C.byref(rows, 4 * k), C.byref(cols, 4 * k),
                                                    _stream()), "evae_weight_norm_set_bwd")
        out = []
        for dv, dg in zip(dvs, dgs):
            out += [dv, dg]
        return tuple(out)


def weight_norm_set(pairs):
    """pairs = [(v, g), ...] -> [w, ...]   (one launch; autograd through to every v and g)"""
    flat = []
    for v, g in pairs:
        flat += [v, g]
    return list(WeightNormSetFn.apply(*flat))


RES_STACK_MIN_PIXELS = int(os.environ.get("EVAE_RES_STACK_MIN_PIXELS", "16384"))


def res_stack_supported(x, weights):
    """A run of residual blocks x + conv(ELU(x)) (models/fully_conv.py:13-23) on the window kernels?  (same channel count in and
    out, stride 1, enough pixels to fill the machine)"""
    if not (CONV_STACK_ON and x.is_cuda and x.dim() == 4 and 1 <= len(weights) <= 16):      # (evae_cw_res_run_*: at most 16 blocks a call)
        return False
    N, Cc, H, W = x.shape
    if N * H * W < RES_STACK_MIN_PIXELS:
        return False
    lib = _lib.load()
    for w in weights:
        Co, Ci, KH, KW = w.shape
        if Ci != Cc or Co != Cc:
            return False
        d = _lib.ConvDesc(N, Cc, H, W, Co, KH, KW, 1, (KH - 1) // 2)
        if not lib.evae_cw_res_supported(C.byref(d)):
            return False
    return True


class ResStackFn(torch.autograd.Function):
    """A run of residual blocks x_{k+1} = x_k + conv(ELU(x_k), w_k) + b_k (reference models/fully_conv.py:13-23) on pixel images: every
    block's convolution reads the image of ELU(x_k) that the block before wrote in its epilogue (the entry packs it once), adds bias
    and residual in its own; in the backward pass one launch per block forms dx_k = dx_{k+1} + ELU'(x_k) conv_transpose(dx_{k+1}, w_k)
    -- ELU' from the saved image -- and writes it as fp32 and as the image the next block's two gradients read.  No ELU launch, no
    fp32 activation kept for the backward.  The filter images of the whole run are packed in one launch and each direction is ONE
    call into the library (evae_cw_res_run_fwd / _bwd): an eager fully_conv step waits for its host.  args: x, then (w, b) per block."""

    @staticmethod
    def forward(ctx, x, *params):
        lib = _lib.load()
        nb = len(params) // 2
        ws_ = [_f32(params[2 * k]) for k in range(nb)]
        bs_ = [params[2 * k + 1] for k in range(nb)]
        _need_cuda(x, *ws_)
        x = _cl(x.float())
        dev = x.device
        N, Cc, H, W = x.shape
        need_grad = any(ctx.needs_input_grad)
        st = _stream()
        d = _lib.ConvDesc(N, Cc, H, W, Cc, ws_[0].shape[2], ws_[0].shape[3], 1, (ws_[0].shape[2] - 1) // 2)
        assert nb <= 16 and all(tuple(w.shape) == tuple(ws_[0].shape) for w in ws_), "a run: <= 16 blocks of one shape"
        ib = int(lib.evae_cw_image_bytes(N * H * W, Cc))
        fb = int(lib.evae_cw_workspace_bytes(C.byref(d), 5))
        imgs = torch.empty((nb, ib), dtype=torch.uint8, device=dev)           # images of ELU(x_0) .. ELU(x_{nb-1})
        # x_1 .. x_nb, channels-last: block k reads x_k and writes x_{k+1}, only the last is returned -- two scratch buffers take turns for
        # the ones in between (a 6-block run at 10 000 rows of cache_z kept 2.9 GB alive through the returned view)
        y_last = torch.empty((N, H, W, Cc), device=dev)
        y_tmp = torch.empty((min(2, nb - 1), N, H, W, Cc), device=dev) if nb > 1 else None
        ys = [y_last if k == nb - 1 else y_tmp[k & 1] for k in range(nb)]
        filt = torch.empty(((2 if need_grad else 1), nb, fb), dtype=torch.uint8, device=dev)
        _lib.check(lib.evae_cw_pack_image(_p(x), N, H, W, Cc, 2, _p(imgs[0]), st), "evae_cw_pack_image(ELU)")
        _lib.check(lib.evae_cw_res_pack_filters(C.byref(d), nb, _ptr_array(ws_), _p(filt[0]), _p(filt[1]) if need_grad else None, st),
                   "evae_cw_res_pack_filters")
        VP = C.c_void_p * nb
        bias = VP(*[None if b is None else b.data_ptr() for b in bs_])
        outf = VP(*[ys[k].data_ptr() for k in range(nb)])
        oimg = VP(*[imgs[k + 1].data_ptr() if k + 1 < nb else None for k in range(nb)])
        _lib.check(lib.evae_cw_res_run_fwd(C.byref(d), nb, _p(filt[0]), bias, _p(imgs[0]), _p(x), outf, oimg, st), "evae_cw_res_run_fwd")
        if need_grad:
            ctx.save_for_backward(*ws_)
            ctx.keep = (imgs, d, [b is not None for b in bs_], filt[1], bs_)
        return y_last.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        imgs, d, has_b, bimg, bs_ = ctx.keep
        ctx.keep = None
        ws_ = list(ctx.saved_tensors)
        nb = len(ws_)
        dev = dout.device
        dy = _cl(dout.float())
        N, Cc, H, W = dy.shape
        st = _stream()
        ib = imgs.shape[1]
        # image of the incoming gradient, and of dx_k (k >= 1): block k reads dx_{k+1} (fp32 and image) and writes dx_k -- two of each
        # take turns, dx_0 (the one returned) has its own fp32 buffer and no image
        dtop = torch.empty((ib,), dtype=torch.uint8, device=dev)
        _lib.check(lib.evae_cw_pack_image(_p(dy), N, H, W, Cc, 0, _p(dtop), st), "evae_cw_pack_image")
        dimg2 = torch.empty((min(2, nb - 1), ib), dtype=torch.uint8, device=dev) if nb > 1 else None
        dimgs = [None if k == 0 else dimg2[(k - 1) & 1] for k in range(nb)]
        dx0 = torch.empty((N, H, W, Cc), device=dev)
        dx_tmp = torch.empty((min(2, nb - 1), N, H, W, Cc), device=dev) if nb > 1 else None
        dxs = [dx0 if k == 0 else dx_tmp[(k - 1) & 1] for k in range(nb)]
        dws = torch.empty((nb,) + tuple(ws_[0].shape), device=dev)
        dbs = torch.empty((nb, Cc), device=dev)
        ws = _workspace("cw", lib.evae_cw_workspace_bytes(C.byref(d), 7), dev)
        VP = C.c_void_p * nb
        aim = VP(*[imgs[k].data_ptr() for k in range(nb)])
        dxf = VP(*[dxs[k].data_ptr() for k in range(nb)])
        dxi = VP(*[dimgs[k].data_ptr() if k >= 1 else None for k in range(nb)])
        dimg_top = dtop
        dwp = VP(*[dws[k].data_ptr() for k in range(nb)])
        dbp = VP(*[dbs[k].data_ptr() for k in range(nb)])
        _lib.check(lib.evae_cw_res_run_bwd(C.byref(d), nb, _p(bimg), aim, _p(dimg_top), _p(dy), dxf, dxi, dwp, dbp, _p(ws), ws.numel(), st),
                   "evae_cw_res_run_bwd")
        grads = []
        for k in range(nb):
            grads += [dws[k], dbs[k] if has_b[k] else None]
        return (dx0.permute(0, 3, 1, 2) if ctx.needs_input_grad[0] else None,) + tuple(grads)


def res_stack(x, blocks):
    """blocks = [(w, b), ...] -> x after the run of residual blocks"""
    flat = []
    for w, b in blocks:
        flat += [w, b]
    return ResStackFn.apply(x, *flat)


def res_window_probe(N, Cc, H, K=3, seed=0):
    """Launch closures {fwd, dgrad, wgrad} of ONE residual block x + conv(ELU(x)) (Cc channels, H x H) on the window kernels, on
    random images prepared once (bench.py's c5 roofline timing, tools/kernel_probe.py)."""
    lib = _lib.load()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(seed)
    d = _lib.ConvDesc(N, Cc, H, H, Cc, K, K, 1, (K - 1) // 2)
    x = torch.randn((N, H, H, Cc), device=dev, generator=g)
    w = torch.randn((Cc, Cc, K, K), device=dev, generator=g) * 0.03; b = torch.zeros(Cc, device=dev)
    img = lambda: torch.empty(int(lib.evae_cw_image_bytes(N * H * H, Cc)), dtype=torch.uint8, device=dev)
    a, o, dyimg, dximg = img(), img(), img(), img()
    _lib.check(lib.evae_cw_pack_image(_p(x), N, H, H, Cc, 2, _p(a), _stream()), "evae_cw_pack_image")
    dy = torch.randn((N, H, H, Cc), device=dev, generator=g) * 0.1
    _lib.check(lib.evae_cw_pack_image(_p(dy), N, H, H, Cc, 0, _p(dyimg), _stream()), "evae_cw_pack_image")
    y = torch.empty_like(x); dx = torch.empty_like(x)
    dw = torch.empty((Cc, Cc * K * K), device=dev); db = torch.empty(Cc, device=dev)
    w5 = _workspace("cw", lib.evae_cw_workspace_bytes(C.byref(d), 5), dev)
    w7 = _workspace("cw_wgrad", lib.evae_cw_workspace_bytes(C.byref(d), 7), dev)
    return {"desc": d,
            "fwd": lambda: _lib.check(lib.evae_cw_res_fwd(_p(a), C.byref(d), _p(w), _p(b), _p(x), _p(y), _p(o), _p(w5), w5.numel(), _stream()), "evae_cw_res_fwd"),
            "dgrad": lambda: _lib.check(lib.evae_cw_res_bwd_data(_p(dyimg), C.byref(d), _p(w), _p(a), _p(dy), _p(dx), _p(dximg), _p(w5), w5.numel(),
                                                                 _stream()), "evae_cw_res_bwd_data"),
            "wgrad": lambda: _lib.check(lib.evae_cw_bwd_weight_plain(_p(dyimg), _p(a), C.byref(d), _p(dw), _p(db), _p(w7), w7.numel(), _stream()),
                                        "evae_cw_bwd_weight_plain")}


def conv_window_probe(N, Cc, H, Co, K, stride, out_planar=False, seed=0):
    """Launch closures {fwd, dgrad, wgrad} of ONE gated layer (Cc -> Co, K x K, H x H input) on the window kernels, on random
    images prepared once (bench.py's roofline timing, tools/kernel_probe.py's PMC passes): the launches a stack issues for it."""
    lib = _lib.load()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(seed)
    pad, OH = (K - 1) // 2, H // stride
    d = _lib.ConvDesc(N, Cc, H, H, Co, K, K, stride, pad)
    x = torch.randn((N, H, H, Cc), device=dev, generator=g)
    wh = torch.randn((Co, Cc, K, K), device=dev, generator=g) * 0.03; wg = torch.randn((Co, Cc, K, K), device=dev, generator=g) * 0.03
    b = torch.zeros(Co, device=dev)
    img = lambda rows, ch: torch.empty(int(lib.evae_cw_image_bytes(rows, ch)), dtype=torch.uint8, device=dev)
    ximg = img(N * H * H, Cc)
    _lib.check(lib.evae_cw_pack_image(_p(x), N, H, H, Cc, int(stride == 2), _p(ximg), _stream()), "evae_cw_pack_image")
    oimg = img(N * OH * OH, Co); s = torch.empty((N, OH, OH, Co), device=dev)
    dy = torch.randn((N, OH, OH, 2 * Co), device=dev, generator=g) * 0.1
    dyimg = img(N * OH * OH, 2 * Co)
    _lib.check(lib.evae_cw_pack_image(_p(dy), N, OH, OH, 2 * Co, int(out_planar), _p(dyimg), _stream()), "evae_cw_pack_image")
    es = torch.rand((N, H, H, Cc), device=dev, generator=g)
    dimg = img(N * H * H, 2 * Cc)
    dw = torch.empty((2 * Co, Cc * K * K), device=dev); db = torch.empty(2 * Co, device=dev)
    del x, dy
    ws = {w: _workspace("cw", lib.evae_cw_workspace_bytes(C.byref(d), w), dev) for w in (0, 1, 2) if lib.evae_cw_supported(C.byref(d), w)}
    out = {"desc": d}
    if 0 in ws:
        out["fwd"] = lambda: _lib.check(lib.evae_cw_fwd_gated(_p(ximg), C.byref(d), _p(wh), _p(b), _p(wg), _p(b), _p(oimg), int(out_planar), _p(s),
                                                              None, _p(ws[0]), ws[0].numel(), _stream()), "evae_cw_fwd_gated")
    if 1 in ws:
        out["dgrad"] = lambda: _lib.check(lib.evae_cw_bwd_data_gate(_p(dyimg), int(out_planar), C.byref(d), _p(wh), _p(wg), _p(ximg), _p(es), _p(dimg),
                                                                    None, _p(ws[1]), ws[1].numel(), _stream()), "evae_cw_bwd_data_gate")
    if 2 in ws:
        out["wgrad"] = lambda: _lib.check(lib.evae_cw_bwd_weight(_p(dyimg), int(out_planar), _p(ximg), C.byref(d), _p(dw), _p(db), _p(ws[2]),
                                                                 ws[2].numel(), _stream()), "evae_cw_bwd_weight")
    return out


def gated_conv_stack(x, layers):
    """layers = [(wh, bh, wg, bg, stride, pad), ...] -> output of the last one (logical NCHW, channels-last storage)"""
    cfg = tuple((_int1(l[4]), _int1(l[5])) for l in layers)
    flat = []
    for l in layers:
        flat += [l[0], l[1], l[2], l[3]]
    return GatedConvStackFn.apply(x, cfg, *flat)


# ------------------------------------------------------------------------------------------------
# latent sampling / log-densities
# ------------------------------------------------------------------------------------------------
class ReparamLogQ(torch.autograd.Function):
    """(mu, logvar, eps) -> (z = mu + eps*exp(logvar/2), log q(z|x))."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        lib = _lib.load()
        _need_cuda(mu, logvar, eps)
        mu, logvar, eps = _f32(mu), _f32(logvar), _f32(eps)
        B, zd = mu.shape
        z = torch.empty_like(mu); logq = torch.empty(B, device=mu.device)
        _lib.check(lib.evae_reparam_logq_fwd(_p(mu), _p(logvar), _p(eps), B, zd, _p(z), _p(logq), _stream()),
                   "evae_reparam_logq_fwd")
        ctx.save_for_backward(mu, logvar, eps, z)
        return z, logq

    @staticmethod
    def backward(ctx, dz, dlogq):
        lib = _lib.load()
        mu, logvar, eps, z = ctx.saved_tensors
        B, zd = mu.shape
        dz = None if dz is None else _f32(dz)
        dlogq = None if dlogq is None else _f32(dlogq)
        dmu = torch.empty_like(mu); dlv = torch.empty_like(mu)
        _lib.check(lib.evae_reparam_logq_bwd(_p(mu), _p(logvar), _p(eps), _p(z), _p(dz), _p(dlogq), B, zd,
                                             _p(dmu), _p(dlv), _stream()), "evae_reparam_logq_bwd")
        return dmu, dlv, None


class HeadsReparamFn(torch.autograd.Function):
    """(h, wm, bm, wl, bl, eps) -> (z, z_mean, logvar, log q(z | .)): the mean head, the Hardtanh(lo, hi) log-variance head, the
    sample and its log-density (reference models/VAE.py:24-26 / AbsHModel.py:22-29, BaseModel.py:79-82, utils/distributions.py:
    28-33) as evae_heads_reparam_fwd's two launches (separately: 2 linear layers x 2, the sample, the density = 6), and a
    backward of four (evae_reparam_logq_bwd_hardtanh, one data gradient over both heads, one grouped weight gradient)
    where the separate Functions take thirteen."""

    @staticmethod
    def forward(ctx, h, wm, bm, wl, bl, eps, lo, hi):
        lib = _lib.load()
        _need_cuda(h, wm, wl, eps)
        h, wm, wl, eps = _f32(h), _f32(wm), _f32(wl), _f32(eps)
        if h.stride(1) != 1:
            h = h.contiguous()
        M, K = h.shape
        Z = wm.shape[0]
        assert wl.shape == wm.shape and wm.shape[1] == K and eps.shape == (M, Z) and eps.is_contiguous()
        z_mean = torch.empty((M, Z), device=h.device); lv_pre = torch.empty_like(z_mean); logvar = torch.empty_like(z_mean)
        z = torch.empty_like(z_mean); logq = torch.empty(M, device=h.device)
        ws = _workspace("heads", lib.evae_heads_reparam_fwd_workspace_bytes(M, K, Z), h.device)
        _lib.check(lib.evae_heads_reparam_fwd(_p(h), M, K, h.stride(0), _p(wm), _p(bm), _p(wl), _p(bl), Z, float(lo), float(hi),
                                              _p(eps), _p(z_mean), _p(lv_pre), _p(logvar), _p(z), _p(logq), _p(ws), ws.numel(),
                                              _stream()), "evae_heads_reparam_fwd")
        ctx.save_for_backward(h, wm, wl, z_mean, lv_pre, logvar, eps, z)
        ctx.set_materialize_grads(False)           # unused outputs (the moments, on the training path) arrive as None, not as zeros
        ctx.clamp = (float(lo), float(hi))
        ctx.has_bias = (bm is not None, bl is not None)
        return z, z_mean, logvar, logq

    @staticmethod
    def backward(ctx, dz, dmean, dlogvar, dlogq):
        lib = _lib.load()
        h, wm, wl, z_mean, lv_pre, logvar, eps, z = ctx.saved_tensors
        lo, hi = ctx.clamp
        M, K = h.shape
        Z = wm.shape[0]
        if dz is None and dmean is None and dlogvar is None and dlogq is None:
            return (None,) * 8
        dz = None if dz is None else _f32(dz)
        dlogq = None if dlogq is None else _f32(dlogq)
        dmu = torch.empty_like(z_mean); dlvp = torch.empty_like(z_mean)
        _lib.check(lib.evae_reparam_logq_bwd_hardtanh(_p(z_mean), _p(logvar), _p(eps), _p(z), _p(dz), None, _p(dlogq), _p(lv_pre),
                                                      lo, hi, M, Z, _p(dmu), _p(dlvp), _stream()), "evae_reparam_logq_bwd_hardtanh")
        if dmean is not None:                       # a consumer of the moments themselves (none on the training path)
            dmu = dmu + dmean
        if dlogvar is not None:
            dlvp = dlvp + dlogvar * ((lv_pre > lo) & (lv_pre < hi))
        return _heads_bwd(h, wm, wl, dmu, dlvp, ctx.needs_input_grad[0], ctx.has_bias) + (None, None, None)


def heads_reparam(h, wm, bm, wl, bl, eps, lo, hi):
    return HeadsReparamFn.apply(h, wm, bm, wl, bl, eps, lo, hi)


def _heads_bwd(h, wm, wl, dmu, dlvp, want_dh, has_bias):
    """what the two heads' backward shares: dh over both heads in one data gradient, both weight gradients in one grouped launch"""
    lib = _lib.load()
    M, K = h.shape
    Z = wm.shape[0]
    dh = _bwd_data(dmu.data_ptr(), wm, dlvp.data_ptr(), wl, M, Z, Z, h.device) if want_dh else None
    if M <= 128 and Z % 4 == 0 and K % 4 == 0 and h.stride(0) % 4 == 0:
        # each head on its own: deferred behind the backward pass when its weight allows (ops.deferred_wgrads; the mean head of
        # q(z2 | x) also encodes the exemplar rows and does not), what is left as one grouped launch here
        res, now = {}, []
        for name, dy_, w_ in (("m", dmu, wm), ("l", dlvp, wl)):
            d = _try_defer(dy_, h, K, w_.data_ptr())
            if d is not None:
                res[name] = d
            else:
                dw_ = torch.empty((Z, K), device=h.device); db_ = torch.empty(Z, device=h.device)
                res[name] = (dw_, db_); now.append((dy_, dw_, db_))
        if now:
            arr = (_lib.WgradJob * len(now))()
            for j, (dy_, dw_, db_) in enumerate(now):
                arr[j].dy = dy_.data_ptr(); arr[j].x = h.data_ptr(); arr[j].dw = dw_.data_ptr(); arr[j].db = db_.data_ptr()
                arr[j].M, arr[j].N, arr[j].K, arr[j].ldy, arr[j].ldx = M, Z, K, dy_.stride(0), h.stride(0)
            _lib.check(lib.evae_dense_bwd_weight_group(C.cast(arr, C.c_void_p), len(now), _stream()), "evae_dense_bwd_weight_group")
        (dwm, dbm), (dwl, dbl) = res["m"], res["l"]
    else:
        dwm, dbm = _bwd_weight(dmu, h, None, K)
        dwl, dbl = _bwd_weight(dlvp, h, None, K)
    return dh, dwm, (dbm if has_bias[0] else None), dwl, (dbl if has_bias[1] else None)


class HeadsDensityFn(torch.autograd.Function):
    """(h, wm, bm, wl, bl, zq) -> (z_mean, logvar, log N(zq | z_mean, exp(logvar))): the conditional p(z1 | z2) of the 2-level models
    (reference models/AbsHModel.py:17-20) with the density term of kl_loss (:99-100) -- evae_heads_density_fwd's two launches
    forward (two linear layers x 2 + the density = 5 separately), evae_log_normal_diag_bwd_hardtanh + one data gradient + one
    grouped weight gradient backward (10 separately)."""

    @staticmethod
    def forward(ctx, h, wm, bm, wl, bl, zq, lo, hi):
        lib = _lib.load()
        _need_cuda(h, wm, wl, zq)
        h, wm, wl, zq = _f32(h), _f32(wm), _f32(wl), _f32(zq)
        M, K = h.shape
        Z = wm.shape[0]
        assert wl.shape == wm.shape and wm.shape[1] == K and zq.shape == (M, Z)
        z_mean = torch.empty((M, Z), device=h.device); lv_pre = torch.empty_like(z_mean); logvar = torch.empty_like(z_mean)
        logp = torch.empty(M, device=h.device)
        ws = _workspace("heads", lib.evae_heads_reparam_fwd_workspace_bytes(M, K, Z), h.device)
        _lib.check(lib.evae_heads_density_fwd(_p(h), M, K, h.stride(0), _p(wm), _p(bm), _p(wl), _p(bl), Z, float(lo), float(hi), _p(zq),
                                              _p(z_mean), _p(lv_pre), _p(logvar), _p(logp), _p(ws), ws.numel(), _stream()),
                   "evae_heads_density_fwd")
        ctx.save_for_backward(h, wm, wl, z_mean, lv_pre, logvar, zq)
        ctx.set_materialize_grads(False)
        ctx.clamp = (float(lo), float(hi))
        ctx.has_bias = (bm is not None, bl is not None)
        return z_mean, logvar, logp

    @staticmethod
    def backward(ctx, dmean, dlogvar, dlogp):
        lib = _lib.load()
        if dmean is None and dlogvar is None and dlogp is None:
            return (None,) * 8
        h, wm, wl, z_mean, lv_pre, logvar, zq = ctx.saved_tensors
        lo, hi = ctx.clamp
        M, Z = z_mean.shape
        dz = None
        if dlogp is not None:
            dmu = torch.empty_like(z_mean); dlvp = torch.empty_like(z_mean)
            dz = torch.empty_like(z_mean) if ctx.needs_input_grad[5] else None
            _lib.check(lib.evae_log_normal_diag_bwd_hardtanh(_p(zq), _p(z_mean), _p(logvar), _p(lv_pre), lo, hi, _p(_f32(dlogp)), M, Z,
                                                             _p(dz), _p(dmu), _p(dlvp), _stream()), "evae_log_normal_diag_bwd_hardtanh")
        else:
            dmu = torch.zeros_like(z_mean); dlvp = torch.zeros_like(z_mean)
        if dmean is not None:
            dmu = dmu + dmean
        if dlogvar is not None:
            dlvp = dlvp + dlogvar * ((lv_pre > lo) & (lv_pre < hi))
        dh, dwm, dbm, dwl, dbl = _heads_bwd(h, wm, wl, dmu, dlvp, ctx.needs_input_grad[0], ctx.has_bias)
        return dh, dwm, dbm, dwl, dbl, dz, None, None


def heads_density(h, wm, bm, wl, bl, zq, lo, hi):
    return HeadsDensityFn.apply(h, wm, bm, wl, bl, zq, lo, hi)


class ElboFn(torch.autograd.Function):
    """(RE, log q1, log p1[, log q2, log p2], beta) -> KL = (log q1 - log p1) [+ (log q2 - log p2)], loss = beta KL - RE, and with
    `average` their batch means (reference models/BaseModel.py:71-77, AbsHModel.py:88-106): one launch each way (evae_elbo_fwd /
    evae_elbo2_fwd, evae_elbo_bwd) where the tensor expressions take eight and ten.  Returns (loss, KL) per row, or the three
    means (loss, RE, KL) as 0-d tensors.  beta: a number, or a one-element device tensor (captured steps); no gradient."""

    @staticmethod
    def forward(ctx, RE, lq1, lp1, lq2, lp2, beta, average):
        lib = _lib.load()
        _need_cuda(RE, lq1, lp1)
        RE, lq1, lp1 = _f32(RE).contiguous(), _f32(lq1).contiguous(), _f32(lp1).contiguous()
        two = lq2 is not None
        if two:
            lq2, lp2 = _f32(lq2).contiguous(), _f32(lp2).contiguous()
        B = RE.numel()
        assert RE.dim() == 1 and all(t is None or t.shape == RE.shape for t in (lq1, lp1, lq2, lp2))
        beta_dev = beta if torch.is_tensor(beta) else None
        beta_host = 0.0 if beta_dev is not None else float(beta)
        loss = torch.empty(B, device=RE.device); KL = torch.empty_like(loss)
        means = torch.empty(3, device=RE.device) if average else None
        if two:
            _lib.check(lib.evae_elbo2_fwd(_p(RE), _p(lq1), _p(lp1), _p(lq2), _p(lp2), _p(beta_dev), beta_host, B, _p(loss), _p(KL),
                                          _p(means), _stream()), "evae_elbo2_fwd")
        else:
            _lib.check(lib.evae_elbo_fwd(_p(RE), _p(lq1), _p(lp1), _p(beta_dev), beta_host, B, _p(loss), _p(KL), _p(means),
                                         _stream()), "evae_elbo_fwd")
        ctx.meta = (B, two, bool(average), beta_host)
        ctx.set_materialize_grads(False)
        ctx.beta_dev = beta_dev
        if average:
            return means[0], means[1], means[2]
        return loss, KL

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.load()
        B, two, average, beta_host = ctx.meta
        if average:
            gl, gr, gk = (None if g is None else _f32(g).contiguous() for g in grads)
        else:
            gl, gk = (None if g is None else _f32(g).contiguous() for g in grads)
            gr = None
        if gl is None and gr is None and gk is None:
            return (None,) * 7
        dev = next(g for g in (gl, gr, gk) if g is not None).device
        coef = torch.empty((3, B), device=dev)
        n = lambda g: 0 if g is None else g.numel()
        _lib.check(lib.evae_elbo_bwd(_p(gl), n(gl), _p(gr), n(gr), _p(gk), n(gk), _p(ctx.beta_dev), beta_host, B, _p(coef[0]),
                                     _p(coef[1]), _p(coef[2]), _stream()), "evae_elbo_bwd")
        need = ctx.needs_input_grad
        pick = lambda i, row: coef[row] if need[i] else None
        return (pick(0, 0), pick(1, 1), pick(2, 2), pick(3, 1) if two else None, pick(4, 2) if two else None, None, None)


def elbo(RE, logq, logp, beta, average, logq2=None, logp2=None):
    """-> (loss, RE, KL) of models/BaseModel.py:71-77, per row or as batch means"""
    if average:
        return ElboFn.apply(RE, logq, logp, logq2, logp2, beta, True)
    loss, KL = ElboFn.apply(RE, logq, logp, logq2, logp2, beta, False)
    return loss, RE, KL


class LogNormalDiag(torch.autograd.Function):
    """utils/distributions.py:28-33 summed over dim=1 for [B x z] inputs."""

    @staticmethod
    def forward(ctx, x, mu, logvar):
        lib = _lib.load()
        _need_cuda(x, mu, logvar)
        x, mu, logvar = _f32(x), _f32(mu), _f32(logvar)
        B, zd = x.shape
        out = torch.empty(B, device=x.device)
        _lib.check(lib.evae_log_normal_diag_fwd(_p(x), _p(mu), _p(logvar), B, zd, _p(out), _stream()),
                   "evae_log_normal_diag_fwd")
        ctx.save_for_backward(x, mu, logvar)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, mu, logvar = ctx.saved_tensors
        B, zd = x.shape
        g = _f32(g)
        need = ctx.needs_input_grad
        dx = torch.empty_like(x) if need[0] else None
        dmu = torch.empty_like(x) if need[1] else None
        dlv = torch.empty_like(x) if need[2] else None
        _lib.check(lib.evae_log_normal_diag_bwd(_p(x), _p(mu), _p(logvar), _p(g), B, zd, _p(dx), _p(dmu), _p(dlv),
                                                _stream()), "evae_log_normal_diag_bwd")
        return dx, dmu, dlv


class BernoulliLL(torch.autograd.Function):
    """utils/distributions.py:44-51 summed over dim=1: x [B x D] (no grad), mean [B x D]."""

    @staticmethod
    def forward(ctx, x, mean):
        lib = _lib.load()
        _need_cuda(x, mean)
        x, mean = _f32(x), _f32(mean)
        B, D = mean.shape
        out = torch.empty(B, device=mean.device)
        _lib.check(lib.evae_bernoulli_ll_fwd(_p(x), _p(mean), B, D, _p(out), _stream()), "evae_bernoulli_ll_fwd")
        ctx.save_for_backward(x, mean)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, mean = ctx.saved_tensors
        B, D = mean.shape
        g = _f32(g)
        dmean = torch.empty_like(mean)
        _lib.check(lib.evae_bernoulli_ll_bwd(_p(x), _p(mean), _p(g), B, D, _p(dmean), _stream()),
                   "evae_bernoulli_ll_bwd")
        return None, dmean


class LogLogistic256(torch.autograd.Function):
    """utils/distributions.py:54-66 summed over dim=1: x [B x D] (no grad), mean [B x D], logvar [B x D] or one value."""

    @staticmethod
    def forward(ctx, x, mean, logvar):
        lib = _lib.load()
        _need_cuda(x, mean, logvar)
        x, mean, logvar = _f32(x), _f32(mean), _f32(logvar)
        B, D = mean.shape
        scalar = logvar.numel() == 1
        assert scalar or logvar.shape == mean.shape
        out = torch.empty(B, device=mean.device)
        _lib.check(lib.evae_log_logistic256_fwd(_p(x), _p(mean), _p(logvar), int(scalar), B, D, _p(out), _stream()),
                   "evae_log_logistic256_fwd")
        ctx.save_for_backward(x, mean, logvar)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, mean, logvar = ctx.saved_tensors
        B, D = mean.shape
        scalar = logvar.numel() == 1
        g = _f32(g)
        dmean = torch.empty_like(mean) if ctx.needs_input_grad[1] else None
        dlv = torch.empty_like(logvar) if ctx.needs_input_grad[2] else None
        rows = torch.empty(B, device=mean.device) if (scalar and dlv is not None) else None
        _lib.check(lib.evae_log_logistic256_bwd(_p(x), _p(mean), _p(logvar), int(scalar), _p(g), B, D, _p(dmean), _p(dlv),
                                                _p(rows), _stream()), "evae_log_logistic256_bwd")
        return None, dmean, dlv


# ------------------------------------------------------------------------------------------------
# AdamNormGrad
# ------------------------------------------------------------------------------------------------
def adam_step_size(step, lr, beta1, beta2):
    """lr * sqrt(1 - beta2^t) / (1 - beta1^t)  (reference utils/optimizer.py:74-76)."""
    return lr * (1.0 - beta2 ** step) ** 0.5 / (1.0 - beta1 ** step)


def step_stats_add(loss, re, kl, step3, totals3=None):
    """step3 <- (loss, -re, kl); totals3 += step3 (the per-epoch sums of utils/training.py:41-46, kept on the device)."""
    _need_cuda(loss, re, kl, step3, totals3)
    _lib.check(_lib.load().evae_step_stats_add(_p(loss), _p(re), _p(kl), _p(step3), _p(totals3), _stream()), "step_stats_add")
    return step3


def batch_prologue(data, idx, binarize, seed_ctr, x_out, eps_out=None):
    """Head of a training step in one launch: x_out[b] = (binarised) data[idx[b]], eps_out ~ N(0, 1).
    seed_ctr: device int64 [2] = (seed, step counter) of the counter-based generator."""
    lib = _lib.load()
    _need_cuda(data, idx, seed_ctr, x_out)
    B, D = x_out.shape
    assert data.dtype == torch.float32 and x_out.dtype == torch.float32 and data.stride(1) == 1 and x_out.stride(1) == 1
    assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() == B and seed_ctr.dtype == torch.int64
    zd = 0
    if eps_out is not None:
        assert eps_out.is_contiguous() and eps_out.dtype == torch.float32 and eps_out.shape[0] == B
        zd = eps_out.shape[1]
    _lib.check(lib.evae_batch_prologue(_p(data), data.stride(0), _p(idx), B, D, 1 if binarize else 0, _p(seed_ctr),
                                       _p(x_out), x_out.stride(0), _p(eps_out), zd, _stream()), "evae_batch_prologue")
    return x_out, eps_out


def batch_prologue_u8(data_u8, idx, binarize, seed_ctr, x_div, x_out, stage_u8, eps_out=None, prepare=None, ctl_job=None, packs=None):
    """batch_prologue on the uint8-resident store: x_out [B x D] fp32 and the batch's bytes into stage_u8 [B x D].
    prepare = (wh, wg, out): the same launch also splits the first layer's weights (u8_prepare) into `out`.
    ctl_job = (stage0, stage1, ctl, state, idx_word, seed_word): the same launch hands over a captured step's control block
    (evae_batch_prologue_u8_step): idx / seed_ctr are then read from the staging block the device-side parity names.
    packs (with ctl_job only) = [(cols, x, x2 or None, R, K, ld, flag, nks, img)]: weight images of the pre-split GEMMs built by the
    same launch (evae_p6_pack_rows / _cols)."""
    lib = _lib.load()
    _need_cuda(data_u8, idx, seed_ctr, x_out, stage_u8)
    B, D = x_out.shape
    assert data_u8.dtype == torch.uint8 and stage_u8.dtype == torch.uint8 and data_u8.stride(1) == 1 and stage_u8.stride(1) == 1
    assert x_out.dtype == torch.float32 and x_out.stride(1) == 1 and tuple(stage_u8.shape) == (B, D)
    assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() == B and seed_ctr.dtype == torch.int64
    zd = 0
    if eps_out is not None:
        assert eps_out.is_contiguous() and eps_out.dtype == torch.float32 and eps_out.shape[0] == B
        zd = eps_out.shape[1]
    wh = wg = out = None
    jobs, arr = [], None
    if prepare is not None:
        wh, wg, out = prepare[:3]
        jobs = prepare[3] if len(prepare) > 3 else []        # [(w1, w2 or None, dst)]: transposed weights for the data gradients
        _need_cuda(wh, wg, out)
        assert wh.dtype == torch.float32 and wh.is_contiguous() and wg.is_contiguous() and wh.shape == wg.shape
        arr = (_lib.WtJob * max(len(jobs), 1))()
        for i, (w1, w2, dst) in enumerate(jobs):
            assert w1.is_contiguous() and (w2 is None or (w2.is_contiguous() and w2.shape == w1.shape))
            arr[i].w1 = w1.data_ptr(); arr[i].w2 = None if w2 is None else w2.data_ptr(); arr[i].dst = dst.data_ptr()
            arr[i].N, arr[i].K = w1.shape
            arr[i].ldt = lib.evae_dense_bwd_data_wt_ld(w1.shape[0])
            assert dst.numel() * dst.element_size() >= lib.evae_dense_bwd_data_wt_bytes(w1.shape[0], w1.shape[1], 1 if w2 is None else 2)
    N, K = (wh.shape if wh is not None else (0, 0))
    if ctl_job is not None:
        s0, s1, ctl, state, idx_word, seed_word = ctl_job
        _need_cuda(s0, s1, ctl, state)
        assert s0.dtype == s1.dtype == ctl.dtype == torch.int64 and s0.numel() == s1.numel() == ctl.numel() and ctl.numel() % 2 == 0
        assert state.dtype == torch.int32 and state.numel() >= 2 and s0.is_contiguous() and s1.is_contiguous() and ctl.is_contiguous()
        cj = _lib.CtlJob(s0.data_ptr(), s1.data_ptr(), ctl.data_ptr(), ctl.numel() * 8, state.data_ptr(), int(idx_word), int(seed_word))
        packs = packs or []
        parr = (_lib.P6PackJob * max(len(packs), 1))()
        for i, (cols, px, px2, pr, pk, pld, pflag, pnks, pimg) in enumerate(packs):
            _need_cuda(px, px2, pimg)
            assert px.dtype == torch.float32 and px.is_contiguous() and (px2 is None or px2.is_contiguous())
            parr[i].x = px.data_ptr(); parr[i].x2 = None if px2 is None else px2.data_ptr(); parr[i].img = pimg.data_ptr()
            parr[i].img_bytes = pimg.numel() * pimg.element_size(); parr[i].ld = int(pld); parr[i].cols = 1 if cols else 0
            parr[i].R, parr[i].K, parr[i].flag, parr[i].nks = int(pr), int(pk), int(pflag), int(pnks)
        _lib.check(lib.evae_batch_prologue_u8_step(_p(data_u8), data_u8.stride(0), B, D, 1 if binarize else 0, float(x_div), _p(x_out),
                                                   x_out.stride(0), _p(stage_u8), stage_u8.stride(0), _p(eps_out), zd, _p(wh), _p(wg),
                                                   N, K, _p(out), out.numel() if out is not None else 0,
                                                   C.cast(arr, C.c_void_p) if jobs else None, len(jobs), C.addressof(cj),
                                                   C.cast(parr, C.c_void_p) if packs else None, len(packs), _stream()),
                   "evae_batch_prologue_u8_step")
        return x_out, eps_out
    assert not packs, "batch_prologue_u8: image jobs ride with the control block's hand-over only"
    if prepare is not None:
        _lib.check(lib.evae_batch_prologue_u8_prepare(_p(data_u8), data_u8.stride(0), _p(idx), B, D, 1 if binarize else 0,
                                                      _p(seed_ctr), float(x_div), _p(x_out), x_out.stride(0), _p(stage_u8),
                                                      stage_u8.stride(0), _p(eps_out), zd, _p(wh), _p(wg), N, K,
                                                      _p(out), out.numel(), C.cast(arr, C.c_void_p) if jobs else None, len(jobs),
                                                      _stream()), "evae_batch_prologue_u8_prepare")
        return x_out, eps_out
    _lib.check(lib.evae_batch_prologue_u8(_p(data_u8), data_u8.stride(0), _p(idx), B, D, 1 if binarize else 0, _p(seed_ctr),
                                          float(x_div), _p(x_out), x_out.stride(0), _p(stage_u8), stage_u8.stride(0),
                                          _p(eps_out), zd, _stream()), "evae_batch_prologue_u8")
    return x_out, eps_out


def adam_flush_tables(table_caches):
    """Upload the pointer tables that adam_normgrad_step prepared while a hipGraph was being captured (call after the
    capture has ended and before the first replay)."""
    for tc in table_caches:
        if tc.get("pending"):
            tc["table"].copy_(tc["pinned"])
            tc["pending"] = False


def adam_normgrad_step(params, grads, exp_avgs, exp_avg_sqs, step, lr, beta1, beta2, eps, weight_decay,
                       table_cache=None, step_size_dev=None, stats=None):
    """One multi-tensor AdamNormGrad update (utils/optimizer.py:32-80).  `table_cache` (a dict) lets the
    caller reuse the device pointer table while the tensor addresses stay the same.  `stats` = (loss, re, kl, step3, totals3[, toggle]):
    the update's last launch also does step_stats_add's work (the captured training step: one launch less at its tail)."""
    lib = _lib.load()
    n = len(params)
    if n == 0:
        return
    dev = params[0].device
    _need_cuda(*params)
    key = tuple(t.data_ptr() for ts in (params, grads, exp_avgs, exp_avg_sqs) for t in ts)
    if table_cache is None:
        table_cache = {}
    nbytes = C.sizeof(_lib.AdamTensor) * n
    if "table" not in table_cache:
        # device table + a pinned staging buffer, both allocated eagerly (allocation is illegal while a
        # hipGraph is being captured; the copies below are not)
        table_cache["table"] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        table_cache["pinned"] = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        table_cache["key"] = None
    table = table_cache["table"]
    if table_cache["key"] != key:
        arr = (_lib.AdamTensor * n)()
        for i in range(n):
            assert grads[i].is_contiguous() and params[i].is_contiguous()
            arr[i].param = params[i].data_ptr(); arr[i].grad = grads[i].data_ptr()
            arr[i].exp_avg = exp_avgs[i].data_ptr(); arr[i].exp_avg_sq = exp_avg_sqs[i].data_ptr()
            arr[i].numel = params[i].numel()
        if torch.cuda.is_current_stream_capturing():
            # the captured launches read the table at replay time and its content (the addresses of this capture's
            # buffers) never changes afterwards: keep the bytes in the pinned buffer and let the caller upload them ONCE
            # after the capture (adam_flush_tables) instead of capturing a memcpy node that every replay would pay for
            C.memmove(table_cache["pinned"].data_ptr(), C.addressof(arr), nbytes)
            table_cache["pending"] = True
        else:
            # eager steps: autograd hands out fresh gradient tensors every step, so the table changes every step.  A copy
            # from pageable memory would hold the host until the stream has drained (the whole backward pass); a small
            # ring of pinned staging buffers keeps the upload asynchronous, and a slot is reused only after its copy ran.
            ring = table_cache.setdefault("ring", [])
            slot = table_cache.get("ring_next", 0)
            if len(ring) <= slot:
                ring.append([torch.empty(nbytes, dtype=torch.uint8).pin_memory(), None])
            pinned, ev = ring[slot]
            if ev is not None:
                ev.synchronize()
            C.memmove(pinned.data_ptr(), C.addressof(arr), nbytes)
            table.copy_(pinned, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            ring[slot][1] = ev
            table_cache["ring_next"] = (slot + 1) % 4
        table_cache["key"] = key
    nb = lib.evae_adam_normgrad_workspace_bytes(n)
    ws = _workspace("adam", nb, dev)
    if stats is not None:
        loss, re, kl, step3, totals3 = stats[:5]
        toggle = stats[5] if len(stats) > 5 else None       # int32 device word the launch XORs with 1 (evae/graph.py's hand-over)
        assert toggle is None or (toggle.dtype == torch.int32 and toggle.is_cuda)
        _lib.check(lib.evae_adam_normgrad_step_stats(_p(table), n, max(p.numel() for p in params), int(step), float(lr),
                                                     float(beta1), float(beta2), float(eps), float(weight_decay),
                                                     _p(step_size_dev), _p(ws), ws.numel(), _p(loss), _p(re), _p(kl), _p(step3),
                                                     _p(totals3), _p(toggle), _stream()), "evae_adam_normgrad_step_stats")
        return
    _lib.check(lib.evae_adam_normgrad_step(_p(table), n, max(p.numel() for p in params), int(step), float(lr),
                                           float(beta1), float(beta2), float(eps), float(weight_decay),
                                           _p(step_size_dev), _p(ws), ws.numel(), _stream()),
               "evae_adam_normgrad_step")
