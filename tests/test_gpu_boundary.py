"""GPU parity of the reference-named free functions at the boundary (utils/distributions.py, BaseModel.log_p_z_exemplar):
values against the goldens of the real reference, and -- the reference's versions being differentiable -- gradients
against the same formulas evaluated by torch autograd in fp64."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
import smoke_case

pytestmark = pytest.mark.gpu


def rel(a, b):
    return smoke_case.rel(np, a, b)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_distance_wrappers_match_reference_golden(golden):
    """G1 / G2: utils.distributions.pairwise_distance and log_normal_diag_vectorized (reference :12-25)."""
    from utils.distributions import pairwise_distance, log_normal_diag_vectorized
    g = golden("g1_g2_distance")
    for zdim in (40, 256):
        z, m = gi.latents(11 + zdim, 16, 257, zdim)
        pd = pairwise_distance(dev(z), dev(m)).cpu().numpy()
        ref = g["pd_z%d" % zdim]
        assert np.abs(pd - ref).max() <= np.spacing(np.abs(ref).max())
        for p in (-1.0, 0.3):
            lv = torch.full((1, zdim), p).cuda()
            ln, pair = log_normal_diag_vectorized(dev(z), dev(m), lv)
            key = "ln_z%d_p%s" % (zdim, str(p).replace("-", "m").replace(".", "_"))
            assert rel(ln.cpu().numpy(), g[key]) < 1e-6
            assert tuple(pair.shape) == (16, 257)


def _ref_log_normal(x, mean, log_var):
    sd = log_var.mul(0.5).exp()
    xs, ms = x / sd, mean / sd
    d = (xs ** 2).sum(1, keepdim=True) + (ms ** 2).sum(1).unsqueeze(0) - 2 * xs @ ms.t()
    return -0.5 * torch.sum(log_var + np.log(2 * np.pi), dim=1) - 0.5 * d, d


@pytest.mark.parametrize("B,C,zd", [(16, 257, 40), (5, 33, 7), (70, 130, 256)])
def test_distance_wrappers_are_differentiable(B, C, zd):
    """The reference's pairwise_distance / log_normal_diag_vectorized are plain torch: callers may differentiate through
    them.  Gradients of a random contraction against torch autograd on the same formula in fp64."""
    from utils.distributions import pairwise_distance, log_normal_diag_vectorized
    z, m = gi.latents(5 + B, B, C, zd)
    G = np.random.RandomState(1).standard_normal((B, C)).astype(np.float32)
    zt = dev(z).requires_grad_(True); mt = dev(m).requires_grad_(True)
    (pairwise_distance(zt, mt) * dev(G)).sum().backward()
    z64 = dev(z).double().requires_grad_(True); m64 = dev(m).double().requires_grad_(True)
    d64 = (z64 ** 2).sum(1, keepdim=True) + (m64 ** 2).sum(1).unsqueeze(0) - 2 * z64 @ m64.t()
    (d64 * dev(G).double()).sum().backward()
    assert rel(zt.grad.cpu().numpy(), z64.grad.cpu().numpy()) < 1e-4
    assert rel(mt.grad.cpu().numpy(), m64.grad.cpu().numpy()) < 1e-4
    # log_normal_diag_vectorized: gradients wrt x, mean and the [1 x z] log-variance
    lv = np.linspace(-1.0, 0.3, zd).astype(np.float32)[None, :]
    zt = dev(z).requires_grad_(True); mt = dev(m).requires_grad_(True); lt = dev(lv).requires_grad_(True)
    ln, pair = log_normal_diag_vectorized(zt, mt, lt)
    assert ln.requires_grad and pair.requires_grad
    (ln * dev(G)).sum().backward()
    z64 = dev(z).double().requires_grad_(True); m64 = dev(m).double().requires_grad_(True)
    l64 = dev(lv).double().requires_grad_(True)
    ln64, _ = _ref_log_normal(z64, m64, l64)
    (ln64 * dev(G).double()).sum().backward()
    assert rel(ln.detach().cpu().numpy(), ln64.detach().cpu().numpy()) < 1e-5
    assert rel(zt.grad.cpu().numpy(), z64.grad.cpu().numpy()) < 1e-4
    assert rel(mt.grad.cpu().numpy(), m64.grad.cpu().numpy()) < 1e-4
    assert rel(lt.grad.cpu().numpy(), l64.grad.cpu().numpy()) < 1e-4


def test_log_p_z_sum_false_matches_golden_and_is_differentiable(golden):
    """BaseModel.log_p_z(sum=False) -> log_p_z_exemplar (reference :98-109,126-127): the [B x C] matrix of G3 and its gradient."""
    from models.VAE import VAE
    g = golden("g3_prior")
    model = VAE(smoke_case.vae_args()).cuda()
    B, C, N, seed = 8, 300, 120, 21
    z_np, c_np = gi.clustered_latents(seed, B, C, 40)
    zi_np, ci_np = gi.mask_indices(seed + 1, B, C, N)
    for mode in ("train", "test"):
        model.train(mode == "train")
        logvar = torch.full((C, 40), -1.3).cuda()
        with torch.no_grad():
            prob = model.log_p_z((dev(z_np), dev(zi_np)), (dev(c_np), logvar, dev(ci_np)), sum=False)
        ref = g["small_%s_prob" % mode]
        got = prob.cpu().numpy()
        assert np.array_equal(np.isinf(got), np.isinf(ref))
        fin = np.isfinite(ref)
        assert rel(got[fin], ref[fin]) < 1e-5
        # with gradients: same values, and d/d(z, c, logvar) of a contraction over the finite entries vs torch fp64
        zt = dev(z_np).requires_grad_(True); ct = dev(c_np).requires_grad_(True)
        plv = torch.tensor([-1.3], device="cuda", requires_grad=True)
        probg = model.log_p_z((zt, dev(zi_np)), (ct, plv * torch.ones((C, 40), device="cuda"), dev(ci_np)), sum=False)
        assert probg.requires_grad
        assert rel(probg.detach().cpu().numpy()[fin], ref[fin]) < 1e-5
        W = torch.from_numpy(np.random.RandomState(2).standard_normal((B, C)).astype(np.float32)).cuda()
        finite = torch.from_numpy(fin).cuda()
        (torch.where(finite, probg, torch.zeros_like(probg)) * W).sum().backward()
        z64 = dev(z_np).double().requires_grad_(True); c64 = dev(c_np).double().requires_grad_(True)
        p64 = torch.tensor([-1.3], device="cuda", dtype=torch.float64, requires_grad=True)
        ln64, _ = _ref_log_normal(z64, c64, (p64 * torch.ones((1, 40), device="cuda", dtype=torch.float64)))
        (torch.where(finite, ln64, torch.zeros_like(ln64)) * W.double()).sum().backward()
        assert rel(zt.grad.cpu().numpy(), z64.grad.cpu().numpy()) < 1e-4
        assert rel(ct.grad.cpu().numpy(), c64.grad.cpu().numpy()) < 1e-4
        assert rel(plv.grad.cpu().numpy(), p64.grad.cpu().numpy()) < 1e-4


@pytest.mark.parametrize("B,D,scalar", [(37, 53, False), (4, 12288, True), (100, 784, False), (1, 7, True)])
def test_log_logistic_256_row_kernel(golden, B, D, scalar):
    """utils.distributions.log_logistic_256 (reference :54-66) on the fused row kernel: values against the oracle (and G6
    for its case), gradients wrt mean and log-variance -- a full [B x D] tensor or ONE broadcast value, the way
    models/fully_conv.py feeds its decoder_logstd -- against torch autograd on the reference's formula in fp64."""
    import evae_oracle as orc
    from utils.distributions import log_logistic_256
    if (B, D) == (37, 53):
        rs = np.random.RandomState(51)          # replay tools/gen_goldens.py::g6 up to its log_logistic_256 inputs
        R, I, O = 37, 53, 24
        rs.standard_normal((R, I)); [rs.standard_normal(sh) for sh in ((O, I), O, (O, I), O)]; rs.standard_normal((R, O))
        rs.standard_normal((R, I)); rs.random_sample((R, I)); rs.standard_normal((R, I)); rs.uniform(-6, 2, (R, I))
    else:
        rs = np.random.RandomState(B + D)
    xc = ((rs.randint(0, 256, (B, D)) + 0.5) / 256).astype(np.float32)
    mc = rs.uniform(1 / 512., 1 - 1 / 512., (B, D)).astype(np.float32)
    ls = rs.uniform(-4.5, 0, (B, D)).astype(np.float32)
    if scalar:
        ls = np.full((B, D), -1.7, np.float32)
    if (B, D) == (37, 53):
        g = golden("g6_layers")
        got = log_logistic_256(dev(xc), dev(mc), dev(ls), dim=1).cpu().numpy()
        assert rel(got, g["log_logistic_256"]) < 1e-5
    out = log_logistic_256(dev(xc), dev(mc), dev(ls), dim=1)
    assert rel(out.cpu().numpy(), orc.log_logistic_256(xc, mc, ls)) < 1e-5           # fp32 like the reference
    # gradients away from the saturated tails (there sigmoid(u) - sigmoid(v) is rounding noise around the 1e-7 floor in
    # fp32 -- in the reference as well -- and an fp64 evaluation is no yardstick): means near x, moderate scales
    mc = np.clip(xc + rs.uniform(-0.08, 0.08, (B, D)), 1 / 512., 1 - 1 / 512.).astype(np.float32)
    if not scalar:
        ls = rs.uniform(-3.0, 0, (B, D)).astype(np.float32)
    mt = dev(mc).requires_grad_(True)
    if scalar:
        lt = torch.tensor([-1.7], device="cuda", requires_grad=True)
        lv_in = lt.expand(B, D)
    else:
        lt = dev(ls).requires_grad_(True)
        lv_in = lt
    out = log_logistic_256(dev(xc), mt, lv_in, dim=1)
    w = dev(np.random.RandomState(3).standard_normal(B).astype(np.float32))
    (out * w).sum().backward()
    m64 = dev(mc).double().requires_grad_(True)
    l64 = (torch.tensor([-1.7], device="cuda", dtype=torch.float64) if scalar else dev(ls).double()).requires_grad_(True)
    x64 = dev(xc).double()
    scale = torch.exp(l64.expand(B, D) if scalar else l64)
    xs = (torch.floor(x64 * 256) / 256 - m64) / scale
    ref = torch.log(torch.sigmoid(xs + 1 / (256 * scale)) - torch.sigmoid(xs) + 1e-7).sum(1)
    (ref * w.double()).sum().backward()
    assert rel(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 1e-5
    assert rel(mt.grad.cpu().numpy(), m64.grad.cpu().numpy()) < 1e-4
    assert rel(lt.grad.cpu().numpy(), l64.grad.cpu().numpy()) < 1e-4
