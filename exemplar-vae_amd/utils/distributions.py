"""Reference-named distance / log-density functions (reference utils/distributions.py:12-66) on the
HIP kernels.  Signatures, argument meaning and return shapes follow the reference."""
import math

import torch

from evae import ops

min_epsilon = 1e-5
max_epsilon = 1. - 1e-5
log_2_pi = math.log(2 * math.pi)


def pairwise_distance(z, means):
    """[B x C] squared L2 distances, fp64-accumulated and rounded once to fp32 like the reference (:12-18).
    Differentiable like the reference's (evae.ops.PairwiseDistance); the training path itself never materialises the
    matrix -- it differentiates through the fused prior kernel (evae.ops.PriorLogP)."""
    return ops.PairwiseDistance.apply(z, means)


def log_normal_diag_vectorized(x, mean, log_var):
    """(:21-25) log N(x_i | mean_j, diag exp(log_var)) for all pairs; `log_var` is [1 x z].
    Returns (log_normal [B x C], pair_dist [B x C]).  With a gradient requested it is the reference's own composition
    (scale by 1/sigma, distance, affine map) on the differentiable distance; without, one fused kernel pass."""
    if torch.is_grad_enabled() and (x.requires_grad or mean.requires_grad or log_var.requires_grad):
        sd = log_var.mul(0.5).exp()
        pair_dist = pairwise_distance(x / sd, mean / sd)
        return -0.5 * torch.sum(log_var + log_2_pi, dim=1) - 0.5 * pair_dist, pair_dist
    lv = log_var.reshape(-1)
    _, _, _, prob = ops.prior_lse_fwd(x.detach(), mean.detach(), lv.detach(), want_prob=True)
    sd = lv.detach().mul(0.5).exp()
    return prob, ops.pairwise_distance(x.detach() / sd, mean.detach() / sd)


def _rows(t, like=None):
    if t.dim() == 2:
        return t
    return t.reshape(t.shape[0], -1)


def log_normal_diag(x, mean, log_var, average=False, dim=None):
    """(:28-33).  The [B x z] / dim=1 / sum case runs as one fused row kernel; other shapes fall back
    to the same formula composed from elementwise ops."""
    if (not average) and dim == 1 and x.dim() == 2 and x.shape == mean.shape == log_var.shape and x.is_cuda:
        return ops.LogNormalDiag.apply(x, mean, log_var)
    log_normal = -0.5 * (log_var + log_2_pi + torch.pow(x - mean, 2) / torch.exp(log_var))
    return torch.mean(log_normal, dim) if average else torch.sum(log_normal, dim)


def log_normal_standard(x, average=False, dim=None):
    """(:36-41)."""
    if (not average) and dim == 1 and x.dim() == 2 and x.is_cuda:
        return ops.LogNormalDiag.apply(x, torch.zeros_like(x), torch.zeros_like(x))
    log_normal = -0.5 * torch.pow(x, 2) - 0.5 * log_2_pi * x.new_ones(size=x.shape)
    return torch.mean(log_normal, dim) if average else torch.sum(log_normal, dim)


def log_bernoulli(x, mean, average=False, dim=None):
    """(:44-51) with the [1e-5, 1-1e-5] clamp."""
    if (not average) and dim == 1 and mean.dim() == 2 and x.shape == mean.shape and mean.is_cuda:
        return ops.BernoulliLL.apply(x, mean)
    probs = torch.clamp(mean, min=min_epsilon, max=max_epsilon)
    lb = x * torch.log(probs) + (1. - x) * torch.log(1. - probs)
    return torch.mean(lb, dim) if average else torch.sum(lb, dim)


def log_logistic_256(x, mean, logvar, average=False, reduce=True, dim=None):
    """(:54-66) 256-bin discretised logistic (continuous inputs, use_logit=False).  The [B x D] / dim=1 / sum case is one
    fused row kernel; a log-variance that is ONE value broadcast over the batch (fully_conv's decoder_logstd, passed as
    an expanded view) stays one value, so its gradient is reduced inside the kernel."""
    if (not average) and dim == 1 and mean.dim() == 2 and x.shape == mean.shape and mean.is_cuda and logvar.shape == mean.shape:
        if all(s == 0 for s in logvar.stride()):
            return ops.LogLogistic256.apply(x, mean, logvar.reshape(-1)[:1])
        return ops.LogLogistic256.apply(x, mean, logvar)
    bin_size = 1. / 256.
    scale = torch.exp(logvar)
    xs = (torch.floor(x / bin_size) * bin_size - mean) / scale
    ll = torch.log(torch.sigmoid(xs + bin_size / scale) - torch.sigmoid(xs) + 1e-7)
    return torch.mean(ll, dim) if average else torch.sum(ll, dim)
