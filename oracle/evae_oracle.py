"""CPU oracle for the Exemplar-VAE hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This file restates, in plain numpy, the arithmetic of the reference's hot path
(sajadn/Exemplar-VAE).  It is the checker for the HIP kernels: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
Nothing under `exemplar-vae_amd/` imports it, and the product path raises when
the HIP library is missing instead of falling back to this file.

Pinning: every function below is checked against golden vectors produced by
importing the real reference in the build container
(`tools/gen_goldens.py` -> `tests/golden/*.npz`, test: `tests/test_oracle_golden.py`).
The reference has no tests of its own, so those goldens are
the pin.

Each function cites the reference file:line it follows (paths relative to the
reference repository root).  Array dtype: functions compute in the dtype of
their inputs (float32 reproduces the reference; float64 gives a tighter
reference for gradient checks) except `pairwise_distance`, which like the
reference always accumulates in float64 and rounds once to float32.
"""
from __future__ import annotations

import math

import numpy as np

LOG_2_PI = math.log(2 * math.pi)          # utils/distributions.py:9
MIN_EPSILON = 1e-5                        # utils/distributions.py:6
MAX_EPSILON = 1.0 - 1e-5                  # utils/distributions.py:7


# --------------------------------------------------------------------------------------
# a1/a2  distances and Gaussian log-densities        (utils/distributions.py:12-41)
# --------------------------------------------------------------------------------------
def pairwise_distance(z, means):
    """utils/distributions.py:12-18: ||z_i||^2 + ||m_j||^2 - 2 z_i.m_j in fp64, cast to fp32."""
    z64 = np.asarray(z, dtype=np.float64)
    m64 = np.asarray(means, dtype=np.float64)
    d1 = (z64 ** 2).sum(axis=1)[:, None]
    d2 = (m64 ** 2).sum(axis=1)[None, :]
    d3 = z64 @ m64.T
    return (d1 + d2 + -2.0 * d3).astype(np.float32)


def log_normal_diag_vectorized(x, mean, log_var):
    """utils/distributions.py:21-25. `log_var` is [1 x z]; returns (log_normal [B x C], pair_dist)."""
    dt = x.dtype
    log_var = np.asarray(log_var, dtype=dt).reshape(1, -1)
    log_var_sqrt = np.exp(log_var * dt.type(0.5))
    pair_dist = pairwise_distance(x / log_var_sqrt, mean / log_var_sqrt).astype(dt)
    const = dt.type(-0.5) * np.sum(log_var + dt.type(LOG_2_PI), axis=1)      # shape [1]
    return const - dt.type(0.5) * pair_dist, pair_dist


def log_normal_diag(x, mean, log_var, axis=1):
    """utils/distributions.py:28-33 (sum over `dim`)."""
    dt = x.dtype
    ln = dt.type(-0.5) * (log_var + dt.type(LOG_2_PI) + (x - mean) ** 2 / np.exp(log_var))
    return ln.sum(axis=axis)


def log_normal_standard(x, axis=1):
    """utils/distributions.py:36-41."""
    dt = x.dtype
    ln = dt.type(-0.5) * x ** 2 - dt.type(0.5 * LOG_2_PI) * np.ones_like(x)
    return ln.sum(axis=axis)


def log_bernoulli(x, mean, axis=1):
    """utils/distributions.py:44-51: probabilities clamped to [1e-5, 1-1e-5]."""
    dt = mean.dtype
    probs = np.clip(mean, dt.type(MIN_EPSILON), dt.type(MAX_EPSILON))
    lb = x * np.log(probs) + (dt.type(1.0) - x) * np.log(dt.type(1.0) - probs)
    return lb.sum(axis=axis)


def sigmoid(x):
    return x.dtype.type(1.0) / (x.dtype.type(1.0) + np.exp(-x))


def log_logistic_256(x, mean, logvar, axis=1):
    """utils/distributions.py:54-66: 256-bin discretised logistic."""
    dt = mean.dtype
    bin_size = dt.type(1.0 / 256.0)
    scale = np.exp(logvar)
    xs = (np.floor(x / bin_size) * bin_size - mean) / scale
    cdf_plus = sigmoid(xs + bin_size / scale)
    cdf_minus = sigmoid(xs)
    return np.log(cdf_plus - cdf_minus + dt.type(1e-7)).sum(axis=axis)


# --------------------------------------------------------------------------------------
# a3/a4  exemplar prior                              (models/BaseModel.py:98-128)
# --------------------------------------------------------------------------------------
def log_p_z_exemplar(z, z_indices, centers, center_log_variance, center_indices, test, no_mask=False):
    """models/BaseModel.py:98-109.  `center_log_variance` is [C x z]; only row 0 is used (:101)."""
    dt = z.dtype
    denominator = np.full((len(z),), float(len(centers)), dtype=dt)
    lv = np.asarray(center_log_variance)[0:1, :]
    prob, _ = log_normal_diag_vectorized(z, centers, lv)
    prob = prob.copy()
    if test is False and no_mask is False:
        mask = np.asarray(z_indices).reshape(-1, 1) == np.asarray(center_indices).reshape(1, -1)
        prob[mask] = -np.inf
        denominator = denominator - mask.sum(axis=1).astype(dt)
    with np.errstate(divide="ignore"):
        prob -= np.log(denominator)[:, None]
    return prob


def logsumexp_rows(prob):
    """models/BaseModel.py:124-125: max + log sum exp(prob - max)."""
    pmax = prob.max(axis=1)
    return pmax + np.log(np.exp(prob - pmax[:, None]).sum(axis=1))


def log_p_z(z, z_indices, centers, center_log_variance, center_indices, test, no_mask=False, sum=True):
    """models/BaseModel.py:111-128 for prior == 'exemplar_prior'."""
    prob = log_p_z_exemplar(z, z_indices, centers, center_log_variance, center_indices, test, no_mask)
    return logsumexp_rows(prob) if sum else prob


# -- shard form: what one exemplar shard contributes, and how shards merge ----------
def prior_partials(z, z_indices, centers, log_var_row, center_indices, masked):
    """Per-row (max, sum exp(p - max), #masked) of the *un-normalised* log-density
    p_ij = log N(z_i | c_j, diag exp(log_var)) over one shard of exemplars.
    Same arithmetic as log_p_z_exemplar minus the `- log(denominator)` term, which
    needs the global count and is applied by `prior_merge`."""
    dt = z.dtype
    B = len(z)
    if len(centers) == 0:
        return (np.full(B, -np.inf, dt), np.zeros(B, dt), np.zeros(B, dt))
    prob, _ = log_normal_diag_vectorized(z, centers, np.asarray(log_var_row).reshape(1, -1))
    prob = prob.copy()
    nmask = np.zeros(B, dt)
    if masked:
        mask = np.asarray(z_indices).reshape(-1, 1) == np.asarray(center_indices).reshape(1, -1)
        prob[mask] = -np.inf
        nmask = mask.sum(axis=1).astype(dt)
    m = prob.max(axis=1)
    msafe = np.where(np.isfinite(m), m, dt.type(0))
    s = np.exp(prob - msafe[:, None]).sum(axis=1)
    return m, s.astype(dt), nmask


def prior_merge(ms, ss, nmasks, c_total):
    """Merge R shard partials ([R x B] each) into log p(z_i): the partial log-sum-exp all-reduce."""
    ms = np.asarray(ms); ss = np.asarray(ss); nmasks = np.asarray(nmasks)
    dt = ms.dtype
    M = ms.max(axis=0)
    Msafe = np.where(np.isfinite(M), M, dt.type(0))
    with np.errstate(invalid="ignore"):
        w = np.where(np.isfinite(ms), np.exp(ms - Msafe[None, :]), dt.type(0))
    tot = (ss * w).sum(axis=0)
    denom = dt.type(c_total) - nmasks.sum(axis=0)
    with np.errstate(divide="ignore"):
        return M + np.log(tot) - np.log(denom)


def prior_grads(z, z_indices, centers, log_var_row, center_indices, masked, grad_out):
    """Analytic gradient of sum_i grad_out_i * log p(z_i) wrt z, centres and the
    per-dimension log-variance row (what autograd yields through BaseModel.py:98-128)."""
    dt = z.dtype
    lv = np.asarray(log_var_row, dtype=dt).reshape(1, -1)
    prob = log_p_z_exemplar(z, z_indices, centers, np.broadcast_to(lv, (1, lv.shape[1])),
                            center_indices, test=not masked)
    lse = logsumexp_rows(prob)
    w = np.exp(prob - lse[:, None])                 # softmax weights, 0 where masked
    gw = w * np.asarray(grad_out, dtype=dt)[:, None]  # [B x C]
    inv_var = np.exp(-lv)                           # [1 x z]
    # d p_ij / d z_i = -(z_i - c_j) / var
    gsum_rows = gw.sum(axis=1)[:, None]
    dz = -(z * gsum_rows - gw @ centers) * inv_var
    gsum_cols = gw.sum(axis=0)[:, None]
    dc = (gw.T @ z - centers * gsum_cols) * inv_var
    # d p_ij / d lv_k = -1/2 + 1/2 (z_ik - c_jk)^2 / var_k
    # sum_ij gw_ij (z_ik - c_jk)^2 expanded into three small GEMMs, in fp64 (the [B x C x z] broadcast
    # of the direct form would be 400 MB at B=100, C=25 000)
    gw64, z64, c64 = gw.astype(np.float64), z.astype(np.float64), centers.astype(np.float64)
    sq = ((gw64.sum(axis=1)[:, None] * z64 ** 2).sum(axis=0) - 2.0 * (z64 * (gw64 @ c64)).sum(axis=0)
          + (gw64.sum(axis=0)[:, None] * c64 ** 2).sum(axis=0)).astype(dt)                    # [z]
    dlv = dt.type(-0.5) * gw.sum() + dt.type(0.5) * sq * inv_var[0]
    return dz.astype(dt), dc.astype(dt), dlv.astype(dt), lse


# --------------------------------------------------------------------------------------
# a6/a20  distance + top-K                           (models/BaseModel.py:263-264, utils/knn_on_latent.py:4-9)
# --------------------------------------------------------------------------------------
def topk_smallest(values, k):
    """k smallest per row, ordered by (value ascending, index ascending).  torch.topk's
    tie order is unspecified; this is the build's definition and
    equals the reference on tie-free rows."""
    values = np.asarray(values)
    idx = np.broadcast_to(np.arange(values.shape[1]), values.shape)
    order = np.lexsort((idx, values), axis=1)[:, :k]
    return np.take_along_axis(values, order, axis=1), order.astype(np.int64)


def nearest_exemplars_topk(z, sub_cache, k):
    """models/BaseModel.py:263-264: pairwise_distance(z, sub_cache).topk(k, largest=False)."""
    return topk_smallest(pairwise_distance(z, sub_cache), k)


def find_nearest_neighbors(z_val, z_train, k=20, chunk=2048):
    """utils/knn_on_latent.py:4-9: sqrt(sum_d (a-b)^2) in fp32 (direct difference), topk(k=20,
    largest=False, sorted=True) -> indices, nearest first."""
    z_val = np.asarray(z_val, dtype=np.float32)
    z_train = np.asarray(z_train, dtype=np.float32)
    dist = np.empty((len(z_val), len(z_train)), dtype=np.float32)
    for s in range(0, len(z_train), chunk):
        d = (z_val[:, None, :] - z_train[None, s:s + chunk, :]) ** 2
        dist[:, s:s + chunk] = np.sqrt(d.sum(axis=2, dtype=np.float32))
    return topk_smallest(dist, k)[1]


def pairdist_direct_f64(z, means):
    """sum_d (a-b)^2 accumulated in fp64, rounded once to fp32: the exact-arithmetic form the HIP
    top-K kernel uses; equals pairwise_distance() to the last fp32 bit except on measure-zero
    rounding boundaries (SURVEY.md section 7 probe)."""
    z64 = np.asarray(z, dtype=np.float64)
    m64 = np.asarray(means, dtype=np.float64)
    out = np.empty((len(z64), len(m64)), dtype=np.float32)
    for i in range(len(z64)):
        out[i] = ((m64 - z64[i]) ** 2).sum(axis=1)
    return out


# --------------------------------------------------------------------------------------
# a17  dense layers                                  (utils/nn.py:29-69)
# --------------------------------------------------------------------------------------
def linear(x, w, b):
    y = x @ w.T
    return y + b if b is not None else y


def hardtanh(x, lo, hi):
    return np.clip(x, x.dtype.type(lo), x.dtype.type(hi))


def gated_dense(x, wh, bh, wg, bg):
    """utils/nn.py:44-69 with activation=None, no_attention=False: h(x) * sigmoid(g(x))."""
    h = linear(x, wh, bh)
    s = sigmoid(linear(x, wg, bg))
    return h * s, (h, s)


def gated_dense_bwd(x, wh, wg, saved, dout, need_dx=True):
    h, s = saved
    dh = dout * s
    dg = dout * h * s * (s.dtype.type(1.0) - s)
    grads = {"wh": dh.T @ x, "bh": dh.sum(axis=0), "wg": dg.T @ x, "bg": dg.sum(axis=0)}
    dx = dh @ wh + dg @ wg if need_dx else None
    return dx, grads


# --------------------------------------------------------------------------------------
# a8-a13  the `vae` model: forward, loss, backward    (models/VAE.py, AbsModel.py, BaseModel.py)
# --------------------------------------------------------------------------------------
VAE_PARAM_NAMES = [
    "prior_log_variance",
    "p_x_mean.linear.weight", "p_x_mean.linear.bias",
    "q_z_layers.0.h.weight", "q_z_layers.0.h.bias", "q_z_layers.0.g.weight", "q_z_layers.0.g.bias",
    "q_z_layers.1.h.weight", "q_z_layers.1.h.bias", "q_z_layers.1.g.weight", "q_z_layers.1.g.bias",
    "q_z_mean.weight", "q_z_mean.bias",
    "q_z_logvar.linear.weight", "q_z_logvar.linear.bias",
    "p_x_layers.0.h.weight", "p_x_layers.0.h.bias", "p_x_layers.0.g.weight", "p_x_layers.0.g.bias",
    "p_x_layers.1.h.weight", "p_x_layers.1.h.bias", "p_x_layers.1.g.weight", "p_x_layers.1.g.bias",
]


def vae_init_params(rs, D=784, H=300, Z=40, dtype=np.float32):
    """He-normal weights (utils/nn.py:12-14), small biases, from a numpy RandomState: the
    'identical weights' both the reference (in gen_goldens) and the build load."""
    shapes = {
        "prior_log_variance": (1,),
        "p_x_mean.linear.weight": (D, H), "p_x_mean.linear.bias": (D,),
        "q_z_layers.0.h.weight": (H, D), "q_z_layers.0.h.bias": (H,),
        "q_z_layers.0.g.weight": (H, D), "q_z_layers.0.g.bias": (H,),
        "q_z_layers.1.h.weight": (H, H), "q_z_layers.1.h.bias": (H,),
        "q_z_layers.1.g.weight": (H, H), "q_z_layers.1.g.bias": (H,),
        "q_z_mean.weight": (Z, H), "q_z_mean.bias": (Z,),
        "q_z_logvar.linear.weight": (Z, H), "q_z_logvar.linear.bias": (Z,),
        "p_x_layers.0.h.weight": (H, Z), "p_x_layers.0.h.bias": (H,),
        "p_x_layers.0.g.weight": (H, Z), "p_x_layers.0.g.bias": (H,),
        "p_x_layers.1.h.weight": (H, H), "p_x_layers.1.h.bias": (H,),
        "p_x_layers.1.g.weight": (H, H), "p_x_layers.1.g.bias": (H,),
    }
    p = {}
    for name in VAE_PARAM_NAMES:
        shp = shapes[name]
        if name == "prior_log_variance":
            p[name] = np.asarray([-1.2], dtype=dtype)
        elif len(shp) == 2:
            p[name] = (rs.standard_normal(shp) * math.sqrt(2.0 / shp[1])).astype(dtype)
        else:
            p[name] = (rs.standard_normal(shp) * 0.05).astype(dtype)
    return p


def vae_q_z_layers(p, x):
    a1, s1 = gated_dense(x, p["q_z_layers.0.h.weight"], p["q_z_layers.0.h.bias"],
                         p["q_z_layers.0.g.weight"], p["q_z_layers.0.g.bias"])
    a2, s2 = gated_dense(a1, p["q_z_layers.1.h.weight"], p["q_z_layers.1.h.bias"],
                         p["q_z_layers.1.g.weight"], p["q_z_layers.1.g.bias"])
    return a2, (x, a1, s1, s2)


def vae_q_z(p, x, prior=False):
    """models/BaseModel.py:205-221 for model_name='vae'."""
    a2, saved = vae_q_z_layers(p, x)
    mean = linear(a2, p["q_z_mean.weight"], p["q_z_mean.bias"])
    if prior:
        logvar = p["prior_log_variance"] * np.ones((x.shape[0], mean.shape[1]), dtype=x.dtype)
        pre = None
    else:
        pre = linear(a2, p["q_z_logvar.linear.weight"], p["q_z_logvar.linear.bias"])
        logvar = hardtanh(pre, -6.0, 2.0)
    return mean, logvar, (saved, a2, pre)


def vae_p_x(p, z):
    """models/AbsModel.py:31-42, binary input."""
    d1, t1 = gated_dense(z, p["p_x_layers.0.h.weight"], p["p_x_layers.0.h.bias"],
                         p["p_x_layers.0.g.weight"], p["p_x_layers.0.g.bias"])
    d2, t2 = gated_dense(d1, p["p_x_layers.1.h.weight"], p["p_x_layers.1.h.bias"],
                         p["p_x_layers.1.g.weight"], p["p_x_layers.1.g.bias"])
    x_mean = sigmoid(linear(d2, p["p_x_mean.linear.weight"], p["p_x_mean.linear.bias"]))
    return x_mean, (z, d1, t1, d2, t2)


def vae_calculate_loss(p, x, x_indices, eps, exemplars, beta=1.0, average=False, training=True,
                       no_mask=False):
    """models/BaseModel.py:65-77 + AbsModel.py:13-19,44-49 for prior='exemplar_prior', binary input.

    `exemplars` is either ('images', ex_images [C x D], ex_indices [C])  -- exact-prior training,
    get_exemplar_set :243-248, centres encoded here WITH gradient --
    or ('embedding', centres [C x z], logvar [C x z], indices [C])       -- evaluation / cached.
    `eps` replaces the device RNG of reparameterize (:79-82).  Returns a dict with loss/RE/KL and
    everything the backward needs."""
    dt = x.dtype
    z_mean, z_logvar, enc_saved = vae_q_z(p, x)
    z_q = eps * np.exp(z_logvar * dt.type(0.5)) + z_mean
    x_mean, dec_saved = vae_p_x(p, z_q)
    RE = log_bernoulli(x, x_mean, axis=1)
    if exemplars[0] == "images":
        _, ex_x, ex_idx = exemplars
        centres, c_logvar, ex_saved = vae_q_z(p, ex_x, prior=True)
    else:
        _, centres, c_logvar, ex_idx = exemplars
        ex_saved = None
    test = not training
    prob = log_p_z_exemplar(z_q, x_indices, centres, c_logvar, ex_idx, test, no_mask)
    log_p = logsumexp_rows(prob)
    log_q = log_normal_diag(z_q, z_mean, z_logvar, axis=1)
    KL = -(log_p - log_q)
    loss = -RE + dt.type(beta) * KL
    out = dict(loss=loss, RE=RE, KL=KL, z_q=z_q, z_mean=z_mean, z_logvar=z_logvar, x_mean=x_mean,
               centres=centres, log_p=log_p, log_q=log_q,
               _saved=(enc_saved, dec_saved, ex_saved, prob, c_logvar, ex_idx))
    if average:
        out["loss"], out["RE"], out["KL"] = loss.mean(), RE.mean(), KL.mean()
    return out


def _encoder_bwd(p, saved_all, d_mean, d_logvar_pre, grads, scale_into=None):
    """Backward through q_z_mean / q_z_logvar heads and the two GatedDense encoder layers."""
    (x, a1, s1, s2), a2, _ = saved_all

    def acc(name, g):
        grads[name] = grads.get(name, 0) + g

    da2 = 0
    if d_mean is not None:
        acc("q_z_mean.weight", d_mean.T @ a2); acc("q_z_mean.bias", d_mean.sum(axis=0))
        da2 = da2 + d_mean @ p["q_z_mean.weight"]
    if d_logvar_pre is not None:
        acc("q_z_logvar.linear.weight", d_logvar_pre.T @ a2)
        acc("q_z_logvar.linear.bias", d_logvar_pre.sum(axis=0))
        da2 = da2 + d_logvar_pre @ p["q_z_logvar.linear.weight"]
    da1, g2 = gated_dense_bwd(a1, p["q_z_layers.1.h.weight"], p["q_z_layers.1.g.weight"], s2, da2)
    for k, v in g2.items():
        acc("q_z_layers.1.%s.%s" % (k[1], "weight" if k[0] == "w" else "bias"), v)
    _, g1 = gated_dense_bwd(x, p["q_z_layers.0.h.weight"], p["q_z_layers.0.g.weight"], s1, da1,
                            need_dx=False)
    for k, v in g1.items():
        acc("q_z_layers.0.%s.%s" % (k[1], "weight" if k[0] == "w" else "bias"), v)


def vae_loss_backward(p, x, x_indices, eps, fwd, beta=1.0, training=True, no_mask=False):
    """Gradient of mean(loss) (calculate_loss(..., average=True) then .backward(),
    utils/training.py:37-38) wrt every parameter; returns {name: grad}."""
    dt = x.dtype
    B = x.shape[0]
    enc_saved, dec_saved, ex_saved, prob, c_logvar, ex_idx = fwd["_saved"]
    z_q, z_mean, z_logvar, x_mean = fwd["z_q"], fwd["z_mean"], fwd["z_logvar"], fwd["x_mean"]
    grads = {}
    gl = dt.type(1.0 / B)                       # d mean(loss) / d loss_i
    # loss_i = -RE_i + beta * (log_q_i - log_p_i)
    # ---- reconstruction term: d(-RE)/d x_mean ------------------------------------------
    probs = np.clip(x_mean, dt.type(MIN_EPSILON), dt.type(MAX_EPSILON))
    inside = (x_mean >= dt.type(MIN_EPSILON)) & (x_mean <= dt.type(MAX_EPSILON))
    dxm = -gl * (x / probs - (dt.type(1) - x) / (dt.type(1) - probs)) * inside
    dpre = dxm * x_mean * (dt.type(1) - x_mean)  # through sigmoid
    zq_in, d1, t1, d2, t2 = dec_saved
    grads["p_x_mean.linear.weight"] = dpre.T @ d2
    grads["p_x_mean.linear.bias"] = dpre.sum(axis=0)
    dd2 = dpre @ p["p_x_mean.linear.weight"]
    dd1, g = gated_dense_bwd(d1, p["p_x_layers.1.h.weight"], p["p_x_layers.1.g.weight"], t2, dd2)
    for k, v in g.items():
        grads["p_x_layers.1.%s.%s" % (k[1], "weight" if k[0] == "w" else "bias")] = v
    dz, g = gated_dense_bwd(zq_in, p["p_x_layers.0.h.weight"], p["p_x_layers.0.g.weight"], t1, dd1)
    for k, v in g.items():
        grads["p_x_layers.0.%s.%s" % (k[1], "weight" if k[0] == "w" else "bias")] = v
    # ---- log q(z|x) term: + beta * log_q ------------------------------------------------
    gq = gl * dt.type(beta)
    var = np.exp(z_logvar)
    diff = z_q - z_mean
    dz = dz + gq * (-(diff / var))
    dmean = gq * (diff / var)
    dlogvar = gq * dt.type(-0.5) * (dt.type(1) - diff ** 2 / var)
    # ---- log p(z) term: - beta * log_p --------------------------------------------------
    masked = training and not no_mask
    lv_row = np.asarray(c_logvar)[0]
    centres = fwd["centres"]
    dzp, dc, dlv, _ = prior_grads(z_q, x_indices, centres, lv_row, ex_idx, masked,
                                  np.full(B, -gq, dtype=dt))
    dz = dz + dzp
    # ---- reparameterisation z = eps * exp(logvar/2) + mean ------------------------------
    dmean = dmean + dz
    dlogvar = dlogvar + dz * eps * np.exp(z_logvar * dt.type(0.5)) * dt.type(0.5)
    pre = enc_saved[2]
    dlogvar_pre = dlogvar * ((pre > dt.type(-6.0)) & (pre < dt.type(2.0)))
    _encoder_bwd(p, enc_saved, dmean, dlogvar_pre, grads)
    # ---- exemplar encoder (exact prior, models/BaseModel.py:247) ------------------------
    if ex_saved is not None:
        _encoder_bwd(p, ex_saved, dc, None, grads)
        # log-variance row 0 = prior_log_variance * ones  (BaseModel.py:213-214, :101)
        grads["prior_log_variance"] = np.asarray([dlv.sum()], dtype=dt)
    return grads


# --------------------------------------------------------------------------------------
# a24  AdamNormGrad                                   (utils/optimizer.py:32-80)
# --------------------------------------------------------------------------------------
def adam_normgrad_step(param, grad, exp_avg, exp_avg_sq, step, lr=5e-4, beta1=0.9, beta2=0.999,
                       eps=1e-8, weight_decay=0.0):
    """One AdamNormGrad update of one tensor; `step` is the 1-based step count after increment."""
    dt = param.dtype
    g = grad / (np.sqrt((grad.astype(np.float64) ** 2).sum()).astype(dt) + dt.type(1e-7))
    if weight_decay != 0:
        g = g + dt.type(weight_decay) * param
    exp_avg = exp_avg * dt.type(beta1) + dt.type(1 - beta1) * g
    exp_avg_sq = exp_avg_sq * dt.type(beta2) + dt.type(1 - beta2) * g * g
    denom = np.sqrt(exp_avg_sq) + dt.type(eps)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr * math.sqrt(bc2) / bc1
    param = param - dt.type(step_size) * exp_avg / denom
    return param, exp_avg, exp_avg_sq


def vae_train_step(p, opt_state, x, x_indices, eps, ex_x, ex_idx, beta, lr=5e-4):
    """The body of utils/training.py:27-40 for one batch (exact prior): loss, backward, optimizer.
    `opt_state` = {name: (exp_avg, exp_avg_sq)}, plus 'step'."""
    fwd = vae_calculate_loss(p, x, x_indices, eps, ("images", ex_x, ex_idx), beta=beta, average=True)
    grads = vae_loss_backward(p, x, x_indices, eps, fwd, beta=beta)
    opt_state["step"] = opt_state.get("step", 0) + 1
    for name in VAE_PARAM_NAMES:
        if name not in grads:
            continue
        m, v = opt_state.get(name, (np.zeros_like(p[name]), np.zeros_like(p[name])))
        p[name], m, v = adam_normgrad_step(p[name], grads[name].astype(p[name].dtype), m, v,
                                           opt_state["step"], lr=lr)
        opt_state[name] = (m, v)
    return fwd, grads
