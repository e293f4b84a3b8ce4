"""GPU parity tests of the raw C-ABI entry points (through evae.ops) against the oracle.
Run on a real MI355X:  python -m pytest tests -m gpu"""
import os

import numpy as np
import pytest
import torch

import evae_oracle as orc
import golden_inputs as gi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def ops():
    from evae import ops as o
    o._lib.load()
    return o


# ---------------------------------------------------------------- prior
@pytest.mark.parametrize("B,C,zd,masked", [(8, 300, 40, True), (100, 1000, 40, True), (100, 25000, 40, True),
                                           (100, 25000, 40, False), (5, 7, 3, True), (130, 70, 256, False),
                                           (300, 2000, 40, True), (1, 1, 40, False), (64, 1000, 100, True),
                                           (20, 150, 294, True), (9, 70, 512, False)])
def test_prior_fwd_matches_oracle(ops, B, C, zd, masked):
    z, c = gi.clustered_latents(100 + B + C, B, C, zd)
    zi, ci = gi.mask_indices(7 + B, B, C, max(C // 2, 4))
    lv = np.linspace(-1.5, -0.5, zd).astype(np.float32)
    m, s, n, prob = ops.prior_lse_fwd(dev(z), dev(c), dev(lv), dev(zi) if masked else None,
                                      dev(ci) if masked else None, want_prob=True)
    lp, lse = ops.prior_merge(m, s, n, C)
    ref_prob = orc.log_p_z_exemplar(z, zi, c, lv[None, :], ci, test=not masked)
    ref = orc.logsumexp_rows(ref_prob)
    assert rel(lp.cpu().numpy(), ref) < 1e-5          # north_star bar is 1e-4
    pm, ps, pn = orc.prior_partials(z, zi, c, lv, ci, masked)
    assert np.array_equal(n.cpu().numpy(), pn)
    raw = prob.cpu().numpy()
    assert np.array_equal(np.isinf(raw), np.isinf(ref_prob))
    fin = np.isfinite(ref_prob)
    denom = (C - pn)[:, None] * np.ones_like(ref_prob)
    assert rel((raw - np.log(denom))[fin], ref_prob[fin]) < 1e-5


@pytest.mark.parametrize("B,C,zd,offset", [(2048, 33000, 40, 0.0), (1500, 45001, 40, 3.0), (1100, 61100, 16, 0.0),
                                           (1024, 65536, 48, -2.0), (4100, 20000, 28, 0.0)])
def test_prior_evaluator_sized_calls_on_the_streaming_bf16_kernel(ops, B, C, zd, offset):
    """IWAE-sized, unmasked calls at z <= 48 (thousands of samples x all exemplars) run on prior_x6_lse_kernel
    (csrc/evae_prior_gemm.hip): queries' split fragments in registers, online log-sum-exp over the exemplar tiles, six bf16
    products per pair.  Against the fp64 oracle on a sample of the queries (the full [B x C] matrix is not built on the host),
    at the bar of the fp32 kernels, and against the fp32 matrix-core kernel (split-bf16 pipe switched off)."""
    z, c = gi.clustered_latents(900 + B, B, C, zd)
    z = (z + offset).astype(np.float32); c = (c + offset).astype(np.float32)
    lv = np.linspace(-1.2, -0.4, zd).astype(np.float32)
    m, s, n, _ = ops.prior_lse_fwd(dev(z), dev(c), dev(lv))
    lp, _ = ops.prior_merge(m, s, n, C)
    lp = lp.cpu().numpy()
    pick = np.random.RandomState(B).choice(B, size=96, replace=False)
    ref = orc.logsumexp_rows(orc.log_p_z_exemplar(z[pick], None, c, lv[None, :], None, test=True))
    assert rel(lp[pick], ref) < 1e-5
    assert float(n.abs().max()) == 0.0
    ops.gemm_x6_configure(0, -1)
    try:
        m2, s2, n2, _ = ops.prior_lse_fwd(dev(z), dev(c), dev(lv))
        lp2, _ = ops.prior_merge(m2, s2, n2, C)
    finally:
        ops.gemm_x6_configure(1, -1)
    lp2 = lp2.cpu().numpy()
    assert rel(lp, lp2) < 2e-6
    assert not np.array_equal(lp, lp2)          # two different kernels: identical bits would mean the switch did nothing


@pytest.mark.parametrize("B,C,zd,masked", [(8, 300, 40, True), (100, 1000, 40, True), (100, 25000, 40, True),
                                           (100, 25000, 40, False), (5, 7, 4, True), (130, 70, 64, False),
                                           (300, 2000, 40, True), (1, 1, 40, False), (129, 129, 24, True),
                                           (128, 127, 8, False), (700, 3000, 56, True)])
def test_prior_fwd_matrix_core_kernel_matches_oracle(ops, B, C, zd, masked):
    """The forward without the probability matrix and z <= 64 runs on the matrix cores (expanded-form distances):
    same bar as the direct-difference kernel above, on full, ragged and single-element tiles."""
    z, c = gi.clustered_latents(200 + B + C, B, C, zd)
    zi, ci = gi.mask_indices(9 + B, B, C, max(C // 2, 4))
    lv = np.linspace(-1.5, -0.5, zd).astype(np.float32)
    m, s, n, prob = ops.prior_lse_fwd(dev(z), dev(c), dev(lv), dev(zi) if masked else None, dev(ci) if masked else None)
    assert prob is None
    lp, lse = ops.prior_merge(m, s, n, C)
    ref = orc.logsumexp_rows(orc.log_p_z_exemplar(z, zi, c, lv[None, :], ci, test=not masked))
    assert rel(lp.cpu().numpy(), ref) < 1e-5          # north_star bar is 1e-4
    pm, ps, pn = orc.prior_partials(z, zi, c, lv, ci, masked)
    assert np.array_equal(n.cpu().numpy(), pn)


def test_prior_golden_c2(ops, golden):
    g = golden("g3_prior")
    z, c = gi.clustered_latents(22, 100, 25000, 40)
    zi, ci = gi.mask_indices(23, 100, 25000, 50000)
    gout = np.random.RandomState(24).standard_normal(100).astype(np.float32)
    lv = np.full(40, -1.3, np.float32)
    for mode in ("train", "test"):
        masked = mode == "train"
        zt = dev(z).requires_grad_(True); ct = dev(c).requires_grad_(True); lvt = dev(lv).requires_grad_(True)
        lp = ops.PriorLogP.apply(zt, ct, lvt, dev(zi) if masked else None, dev(ci) if masked else None)
        (lp * dev(gout)).sum().backward()
        assert rel(lp.detach().cpu().numpy(), g["c2_%s_logp" % mode]) < 1e-5
        assert rel(zt.grad.cpu().numpy(), g["c2_%s_dz" % mode]) < 1e-4
        assert rel(lvt.grad.sum().item(), g["c2_%s_dplv" % mode]) < 1e-4
        dc = ct.grad.cpu().numpy()
        assert rel(dc[:64], g["c2_%s_dc_head" % mode]) < 1e-4
        assert rel(dc.astype(np.float64).sum(0), g["c2_%s_dc_colsum" % mode]) < 1e-4
        assert rel(np.linalg.norm(dc.astype(np.float64), axis=1), g["c2_%s_dc_rownorm" % mode]) < 1e-4


@pytest.mark.parametrize("B,C,zd,masked", [(8, 300, 40, True), (100, 1000, 40, True), (5, 7, 3, True),
                                           (130, 70, 256, False), (200, 500, 40, True), (33, 129, 100, False),
                                           # matrix-core backward: every z_dim group up to 56, several query tiles
                                           # (atomic dcentres), ragged last exemplar tile, 60 / 64 fall back to the VALU kernel
                                           (100, 3125, 40, True), (300, 777, 8, True), (129, 128, 56, False),
                                           (64, 1000, 24, True), (100, 257, 60, True), (40, 300, 64, False),
                                           # fully_conv on 28 x 28 with the default bottleneck: z = 6 * 7 * 7 = 294; the limit 512
                                           (20, 150, 294, True), (9, 70, 512, False)])
def test_prior_bwd_matches_oracle(ops, B, C, zd, masked):
    z, c = gi.clustered_latents(200 + B + C, B, C, zd)
    zi, ci = gi.mask_indices(9 + B, B, C, max(C // 2, 4))
    lv = np.linspace(-1.0, 0.2, zd).astype(np.float32)
    gout = np.random.RandomState(B).standard_normal(B).astype(np.float32)
    dz, dc, dlv, lse = orc.prior_grads(z.astype(np.float64), zi, c.astype(np.float64), lv.astype(np.float64), ci,
                                       masked, gout.astype(np.float64))
    m, s, n, _ = ops.prior_lse_fwd(dev(z), dev(c), dev(lv), dev(zi) if masked else None, dev(ci) if masked else None)
    lp, lse_t = ops.prior_merge(m, s, n, C)
    gz, gc, glv = ops.prior_lse_bwd(dev(z), dev(c), dev(lv), dev(zi) if masked else None,
                                    dev(ci) if masked else None, lse_t, dev(gout))
    # fp32 kernel vs fp64 oracle; the bar for the ELBO is 1e-4 relative
    assert rel(gz.cpu().numpy(), dz) < 1e-4
    assert rel(gc.cpu().numpy(), dc) < 1e-4
    assert rel(glv.cpu().numpy(), dlv) < 1e-4


def _prior_fwd_bwd_vs_oracle(ops, z, c, lv, zi, ci, masked, gout, tol_lp=1e-5, tol_g=1e-4, strict=False):
    """forward log p and the three gradients of the fused prior against the fp64 oracle (reference arithmetic:
    utils/distributions.py:12-25 computes the distance in fp64)"""
    C = len(c)
    dz, dc, dlv, _ = orc.prior_grads(z.astype(np.float64), zi, c.astype(np.float64), lv.astype(np.float64), ci, masked,
                                     gout.astype(np.float64))
    ref = orc.log_p_z(z.astype(np.float64), zi, c.astype(np.float64), lv[None, :].astype(np.float64), ci, test=not masked)
    a = (dev(z), dev(c), dev(lv), dev(zi) if masked else None, dev(ci) if masked else None)
    m, s, n, _ = ops.prior_lse_fwd(*a)
    lp, lse_t = ops.prior_merge(m, s, n, C)
    gz, gc, glv = ops.prior_lse_bwd(*a, lse_t, dev(gout))
    # softmax weights exp(p_ij - lse_i) are formed from fp32 log-densities here and in the reference alike (its p_ij is the
    # fp64 distance cast to fp32, utils/distributions.py:18): half an ulp of |lse| is a relative error of every weight
    # (that bound is for near-tied exemplars, whose weight ratio no fp32 log-density resolves; `strict`: the caller's inputs
    # have clear nearest exemplars, and then the (max, log sum) token of the merge keeps the weights exact at any magnitude)
    if not strict:
        tol_g = max(tol_g, 1.5 * 2.0 ** -24 * float(np.abs(ref).max()))
    assert rel(lp.cpu().numpy(), ref) < tol_lp
    assert rel(gz.cpu().numpy(), dz) < tol_g
    assert rel(gc.cpu().numpy(), dc) < tol_g
    assert rel(glv.cpu().numpy(), dlv) < tol_g


@pytest.mark.parametrize("zd", [8, 40, 56, 100, 294])
@pytest.mark.parametrize("masked", [True, False])
def test_prior_gradients_stay_normalised_at_huge_log_densities(ops, zd, masked):
    """|log p| ~ 1e7 .. 1e9 -- an untrained fully_conv net at its He-initialised scale puts its latents there (golden G21; the
    reference's own loss is 6e7 on that input).  The backward recomputes exp(p_ij - lse_i): with the rounded fp32 lse that
    exponent is off by ulp(lse) = 8 .. 64 nats, i.e. every row of the gradient by a factor e^(+-8..64) (r02 behaviour, found by
    the G21 test in r03: encoder gradient norms 4e6 x the reference's).  The merge now hands the backward the row maximum and
    the log of the normalised sum apart, and the direct-difference backward sums its chunks in the forward's order, so
    p_ij - max_i is exact: gradients against float64 at 1e-4 with NO allowance for the magnitude.  All three kernel families:
    z <= 56 matrix-core kernels (their direct-difference blocks), the VALU kernels, z > 64 (GEMM path's guard fallback)."""
    B, C = 37, 300
    rs = np.random.RandomState(zd)
    z = (rs.standard_normal((B, zd)) * 6000.0).astype(np.float32)
    c = (rs.standard_normal((C, zd)) * 30.0).astype(np.float32)
    lv = np.full(zd, -1.2, np.float32)
    zi, ci = gi.mask_indices(5, B, C, 400)
    ci[:3] = zi[:3, 0]                                   # a few leave-one-out hits
    gout = rs.standard_normal(B).astype(np.float32)
    _prior_fwd_bwd_vs_oracle(ops, z, c, lv, zi, ci, masked, gout, tol_lp=1e-6, tol_g=1e-4, strict=True)


@pytest.mark.parametrize("offset,scale", [(10.0, 1.0), (30.0, 1.0), (100.0, 1.0), (-300.0, 1.0), (0.0, 12.0), (25.0, 40.0)])
@pytest.mark.parametrize("masked", [True, False])
def test_prior_offset_and_large_scale_latents(ops, offset, scale, masked):
    """The matrix-core kernels evaluate |z|^2 + |c|^2 - 2 z.c in fp32 where the reference works in fp64
    (utils/distributions.py:13-18).  A common offset of the latent cloud is removed by centring, widely spread latents
    fall through the norm guard to direct differences: the 1e-5 bar holds either way."""
    B, C, zd = 100, 5000, 40
    z, c = gi.clustered_latents(300 + int(abs(offset)) + int(scale), B, C, zd)
    z = (z * scale + offset).astype(np.float32); c = (c * scale + offset).astype(np.float32)
    zi, ci = gi.mask_indices(11, B, C, 9000)
    lv = np.full(zd, -1.3, np.float32)
    gout = np.random.RandomState(3).standard_normal(B).astype(np.float32)
    _prior_fwd_bwd_vs_oracle(ops, z, c, lv, zi, ci, masked, gout)


@pytest.mark.parametrize("B,C,zd,masked", [(100, 3000, 256, True), (300, 2000, 128, False), (64, 1000, 100, True),
                                           (20, 150, 294, True), (9, 70, 512, False), (130, 257, 68, True),
                                           (1, 1, 256, False), (5, 3, 72, True)])
def test_prior_large_latent_sizes_run_as_gemms_on_the_matrix_cores(ops, B, C, zd, masked, gemm_pipe):
    """z > 64 (fully_conv: 256 on 64 x 64 inputs, 294 on 28 x 28): forward = GEMM with a log-sum-exp epilogue, backward =
    three GEMMs (evae_prior_gemm.hip), against the fp64 oracle; odd sizes are padded by the staging pass."""
    z, c = gi.clustered_latents(500 + B + C, B, C, zd)
    zi, ci = gi.mask_indices(17 + B, B, C, max(C // 2, 4))
    lv = np.linspace(-1.0, 0.2, zd).astype(np.float32)
    gout = np.random.RandomState(B).standard_normal(B).astype(np.float32)
    _prior_fwd_bwd_vs_oracle(ops, z, c, lv, zi, ci, masked, gout)


def test_prior_gemm_path_with_an_offset_latent_cloud(ops, gemm_pipe):
    B, C, zd = 100, 2000, 256
    z, c = gi.clustered_latents(77, B, C, zd)
    z = (z * 0.5 + 40.0).astype(np.float32); c = (c * 0.5 + 40.0).astype(np.float32)
    zi, ci = gi.mask_indices(19, B, C, 5000)
    lv = np.full(zd, -0.4, np.float32)
    gout = np.random.RandomState(4).standard_normal(B).astype(np.float32)
    _prior_fwd_bwd_vs_oracle(ops, z, c, lv, zi, ci, True, gout)


@pytest.mark.parametrize("B,C,zd,masked", [(100, 3125, 40, True), (300, 777, 8, True), (129, 128, 56, False),
                                           (130, 70, 64, False), (1, 1, 40, False),
                                           # z > 64: the GEMM path's guard flag releases the direct-difference kernels instead
                                           (100, 1500, 256, True), (20, 150, 294, False)])
def test_prior_direct_difference_path_of_the_matrix_core_kernels(ops, B, C, zd, masked):
    """Norm limit 0: every block of the matrix-core kernels takes its guard's direct-difference path."""
    z, c = gi.clustered_latents(400 + B + C, B, C, zd)
    zi, ci = gi.mask_indices(13 + B, B, C, max(C // 2, 4))
    lv = np.linspace(-1.0, 0.2, zd).astype(np.float32)
    gout = np.random.RandomState(B).standard_normal(B).astype(np.float32)
    ops.prior_set_norm_limit(0.0)
    try:
        _prior_fwd_bwd_vs_oracle(ops, z, c, lv, zi, ci, masked, gout)
    finally:
        ops.prior_set_norm_limit(-1.0)



def test_prior_bwd_many_tiles_per_block(ops):
    """More exemplar tiles than blocks (70 000 exemplars: two 128-row tiles per split), against the fp64 oracle."""
    B, C, zd = 100, 70000, 40
    z, c = gi.clustered_latents(31, B, C, zd)
    zi, ci = gi.mask_indices(32, B, C, 50000)
    lv = np.full(zd, -0.7, np.float32)
    gout = np.random.RandomState(5).standard_normal(B).astype(np.float32)
    dz, dc, dlv, lse = orc.prior_grads(z.astype(np.float64), zi, c.astype(np.float64), lv.astype(np.float64), ci, True,
                                       gout.astype(np.float64))
    m, s, n, _ = ops.prior_lse_fwd(dev(z), dev(c), dev(lv), dev(zi), dev(ci))
    lp, lse_t = ops.prior_merge(m, s, n, C)
    gz, gc, glv = ops.prior_lse_bwd(dev(z), dev(c), dev(lv), dev(zi), dev(ci), lse_t, dev(gout))
    assert rel(gz.cpu().numpy(), dz) < 1e-4
    assert rel(gc.cpu().numpy(), dc) < 1e-4
    assert rel(glv.cpu().numpy(), dlv) < 1e-4


def test_prior_sharded_merge_matches_single(ops):
    """R logical shards on one GPU (uneven, one empty) merge to the single-shard answer."""
    B, C, zd = 100, 11500, 40
    z, c = gi.clustered_latents(77, B, C, zd)
    zi, ci = gi.mask_indices(78, B, C, 23000)
    lv = np.full(zd, -0.9, np.float32)
    cuts = [0, 1438, 1438, 2876, 4314, 5752, 7189, 8626, 10063, 11500]
    ms, ss, ns = [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        m, s, n, _ = ops.prior_lse_fwd(dev(z), dev(c[a:b]), dev(lv), dev(zi), dev(ci[a:b]))
        ms.append(m); ss.append(s); ns.append(n)
    lp, _ = ops.prior_merge(torch.stack(ms), torch.stack(ss), torch.stack(ns), C)
    ref = orc.log_p_z(z, zi, c, lv[None, :], ci, test=False)
    assert rel(lp.cpu().numpy(), ref) < 1e-5


# ---------------------------------------------------------------- top-K
def test_topk_golden_bit_exact(ops, golden):
    g = golden("g4_topk")
    for tag, (B, C, zd, seed) in {"c2": (100, 25000, 40, 31), "c5": (64, 100000, 256, 32)}.items():
        z, c = gi.clustered_latents(seed, B, C, zd)
        idx, val = ops.pairdist_topk(dev(z), dev(c), 10)
        assert np.array_equal(idx.cpu().numpy(), g[tag + "_idx"].astype(np.int64)), tag
        assert np.array_equal(val.cpu().numpy(), g[tag + "_val"]), tag


def test_knn_golden_bit_exact(ops, golden):
    g = golden("g5_knn")
    zv, zt = gi.clustered_latents(41, 100, 60000, 40)
    idx, _ = ops.pairdist_topk(dev(zv), dev(zt), 20, sqrt=True)
    assert np.array_equal(idx.cpu().numpy(), g["idx"].astype(np.int64))


@pytest.mark.parametrize("B,N,zd,k", [(3, 5, 2, 5), (130, 1000, 7, 1), (17, 64, 40, 64), (100, 777, 33, 20)])
def test_topk_small_and_ragged(ops, B, N, zd, k):
    z, c = gi.latents(5 + B, B, N, zd)
    idx, val = ops.pairdist_topk(dev(z), dev(c), k)
    ov, oi = orc.topk_smallest(orc.pairdist_direct_f64(z, c), k)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(val.cpu().numpy(), ov)


@pytest.mark.parametrize("B,N,zd,k,sqrt,offset", [(64, 100000, 256, 10, False, 0.0), (100, 25000, 40, 10, False, 0.0),
                                                 (37, 5000, 24, 64, True, 0.0), (130, 4099, 40, 7, False, 0.0),
                                                 (3, 2048, 8, 16, False, 0.0),
                                                 # B > 64 over >= 384 tiles: the two-term split-bf16 screen with fused cache norms
                                                 (100, 100000, 256, 10, False, 0.0), (128, 60000, 64, 10, True, 0.0),
                                                 (100, 100000, 256, 10, False, 3.0), (77, 50001 // 4 * 4, 48, 5, False, -8.0),
                                                 # several query tiles: every (row tile, query tile) block keeps its own row norms
                                                 (300, 70000, 32, 10, False, 1.5), (513, 50048, 16, 33, True, 0.0)])
def test_topk_screening_path_equals_exact_scan(ops, B, N, zd, k, sqrt, offset):
    """Large caches take the matrix-core screening + exact re-ranking path; it must return the very same indices and
    values as the exact fp64 scan kernel (forced with EVAE_TOPK_EXACT_SCAN=1 in a child process), including on
    duplicated exemplars (ties broken by index), clustered data, and latents with a common offset (norms far larger than
    the distances: the screen's error bound is what keeps the candidate set complete)."""
    import subprocess, sys, tempfile
    z, c = gi.clustered_latents(300 + B + N, B, N, zd)
    z = (z + np.float32(offset)).astype(np.float32); c = (c + np.float32(offset)).astype(np.float32)
    c[N // 2:N // 2 + 50] = c[:50]                    # exact duplicates -> ties
    c[-7:] = z[:7] if B >= 7 else c[-7:]              # zero-distance hits at the very end of the cache
    idx, val = ops.pairdist_topk(dev(z), dev(c), k, sqrt=sqrt)
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "z.npy"), z); np.save(os.path.join(td, "c.npy"), c)
        code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from evae import ops;"
                "z=torch.from_numpy(np.load(%r)).cuda(); c=torch.from_numpy(np.load(%r)).cuda();"
                "i,v=ops.pairdist_topk(z,c,%d,sqrt=%r); np.save(%r,i.cpu().numpy()); np.save(%r,v.cpu().numpy())"
                % (os.path.join(ROOT, "exemplar-vae_amd"), os.path.join(td, "z.npy"), os.path.join(td, "c.npy"), k, sqrt,
                   os.path.join(td, "i.npy"), os.path.join(td, "v.npy")))
        subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, EVAE_TOPK_EXACT_SCAN="1"))
        ri, rv = np.load(os.path.join(td, "i.npy")), np.load(os.path.join(td, "v.npy"))
    assert np.array_equal(idx.cpu().numpy(), ri)
    assert np.array_equal(val.cpu().numpy(), rv)


@pytest.mark.parametrize("B,N,zd,k,sqrt,offset", [(100, 100000, 256, 10, False, 0.0), (100, 25000, 40, 10, False, 0.0),
                                                 (64, 100000, 256, 10, False, 3.0), (128, 60000, 64, 20, True, 0.0),
                                                 (37, 5000, 24, 32, True, 0.0), (3, 2048, 8, 16, False, -8.0),
                                                 (77, 50001 // 4 * 4, 48, 5, False, 0.0)])
def test_topk_stream_kernel_equals_the_two_launch_form(ops, golden, monkeypatch, B, N, zd, k, sqrt, offset):
    """EVAE_TOPK_STREAM=1 (r06, opt-in): the cache scan + top-K as ONE launch (csrc/evae_topk_screen.hip::topk_stream_kernel: per-block
    candidate lists from lane-local bounds, flags instead of a second launch, the same exact re-ranking) returns the very indices and
    values of the default two-launch form -- duplicated exemplars (ties by index), zero-distance hits at the end of the cache, a
    common offset (norms far above the distances) included; and G4's reference indices at c2 / c5 sizes."""
    z, c = gi.clustered_latents(500 + B + N, B, N, zd)
    z = (z + np.float32(offset)).astype(np.float32); c = (c + np.float32(offset)).astype(np.float32)
    c[N // 2:N // 2 + 50] = c[:50]
    c[-7:] = z[:7] if B >= 7 else c[-7:]
    monkeypatch.delenv("EVAE_TOPK_STREAM", raising=False)
    ri, rv = ops.pairdist_topk(dev(z), dev(c), k, sqrt=sqrt)
    monkeypatch.setenv("EVAE_TOPK_STREAM", "1")
    idx, val = ops.pairdist_topk(dev(z), dev(c), k, sqrt=sqrt)
    assert torch.equal(idx, ri) and torch.equal(val, rv)
    if (B, N, zd, k, offset) == (100, 25000, 40, 10, 0.0):
        g = golden("g4_topk")
        for tag, (b_, n_, z_) in (("c2", (100, 25000, 40)), ("c5", (64, 100000, 256))):
            zz, cc = gi.clustered_latents(31 if tag == "c2" else 32, b_, n_, z_)
            got, gval = ops.pairdist_topk(dev(zz), dev(cc), 10)
            assert np.array_equal(got.cpu().numpy(), g[tag + "_idx"].astype(np.int64)), tag
            assert np.array_equal(gval.cpu().numpy(), g[tag + "_val"]), tag


def test_pairwise_distance_golden(ops, golden):
    g = golden("g1_g2_distance")
    for zdim in (40, 256):
        z, m = gi.latents(11 + zdim, 16, 257, zdim)
        pd = ops.pairwise_distance(dev(z), dev(m)).cpu().numpy()
        ref = g["pd_z%d" % zdim]
        assert np.abs(pd - ref).max() <= np.spacing(np.abs(ref).max())
        assert (pd == ref).mean() > 0.999


def test_topk_ties_index_order(ops):
    c = np.zeros((300, 8), np.float32); c[::3] = 1.0
    z = np.zeros((4, 8), np.float32)
    idx, _ = ops.pairdist_topk(dev(z), dev(c), 10)
    expect = np.asarray([i for i in range(300) if i % 3][:10])
    assert np.array_equal(idx.cpu().numpy(), np.tile(expect, (4, 1)))


def test_topk_sharded_merge(ops):
    B, N, zd, k = 100, 20000, 40, 10
    z, c = gi.clustered_latents(91, B, N, zd)
    full, _ = ops.pairdist_topk(dev(z), dev(c), k)
    cuts = [0, 2500, 2500, 9000, 20000]
    vals, idxs = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b - a < k:
            vals.append(torch.full((B, k), float("inf"), device="cuda")); idxs.append(torch.full((B, k), -1, dtype=torch.int64, device="cuda"))
            continue
        i, v = ops.pairdist_topk(dev(z), dev(c[a:b]), k, index_base=a)
        vals.append(v); idxs.append(i)
    mi, mv = ops.topk_merge(torch.stack(vals), torch.stack(idxs))
    assert torch.equal(mi, full)


@pytest.mark.parametrize("n,C", [(1000, 25000), (1, 5), (7, 3), (1030, 40), (16384, 100000)])
def test_select_exemplars_marks_repeats(ops, n, C):
    """evae_select_exemplars (static-shape `unique` of reference models/BaseModel.py:265-266): every slot keeps its dataset
    row, the first slot naming a position keeps it as c_idx, repeats get PRIOR_MASK_ALL; the count is #unique."""
    rs = np.random.RandomState(n + C)
    pos = rs.randint(0, C, size=n).astype(np.int64)
    cand = rs.permutation(4 * C)[:C].astype(np.int64)
    sel, cidx, cnt = ops.select_exemplars(dev(pos), dev(cand), want_count=True)
    first = np.zeros(n, bool)
    first[np.unique(pos, return_index=True)[1]] = True
    assert np.array_equal(sel.cpu().numpy(), cand[pos])
    assert np.array_equal(cidx.cpu().numpy(), np.where(first, cand[pos], ops.PRIOR_MASK_ALL))
    assert int(cnt.item()) == int(first.sum())


# ---------------------------------------------------------------- dense layers
@pytest.mark.parametrize("M,K,N", [(37, 53, 24), (100, 784, 300), (1000, 300, 300), (257, 40, 300), (5000, 784, 300),
                                   (130, 294, 40)])
def test_gated_dense_fwd_bwd(ops, M, K, N, gemm_pipe):
    rs = np.random.RandomState(M + K)
    x = rs.standard_normal((M, K)).astype(np.float32)
    wh = (rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32); bh = (rs.standard_normal(N) * 0.1).astype(np.float32)
    wg = (rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32); bg = (rs.standard_normal(N) * 0.1).astype(np.float32)
    gout = rs.standard_normal((M, N)).astype(np.float32)
    x64, wh64, wg64 = x.astype(np.float64), wh.astype(np.float64), wg.astype(np.float64)
    y, saved = orc.gated_dense(x64, wh64, bh.astype(np.float64), wg64, bg.astype(np.float64))
    dx, gr = orc.gated_dense_bwd(x64, wh64, wg64, saved, gout.astype(np.float64))
    t = [dev(a).requires_grad_(True) for a in (x, wh, bh, wg, bg)]
    out = ops.gated_dense(t[0], t[1], t[2], t[3], t[4])
    out.backward(dev(gout))
    assert rel(out.detach().cpu().numpy(), y) < 2e-6
    assert rel(t[0].grad.cpu().numpy(), dx) < 1e-5
    for ti, k in ((1, "wh"), (2, "bh"), (3, "wg"), (4, "bg")):
        assert rel(t[ti].grad.cpu().numpy(), gr[k]) < 1e-5, k


def test_gated_dense_row_gather(ops):
    rs = np.random.RandomState(3)
    data = rs.standard_normal((500, 784)).astype(np.float32)
    rows = rs.randint(0, 500, 1000).astype(np.int64)
    wh = (rs.standard_normal((300, 784)) / 28).astype(np.float32); wg = (rs.standard_normal((300, 784)) / 28).astype(np.float32)
    b = np.zeros(300, np.float32)
    w = [dev(a).requires_grad_(True) for a in (wh, b, wg, b)]
    out = ops.gated_dense(dev(data), w[0], w[1], w[2], w[3], rows=dev(rows))
    gout = rs.standard_normal((1000, 300)).astype(np.float32)
    out.backward(dev(gout))
    x64 = data[rows].astype(np.float64)
    y, saved = orc.gated_dense(x64, wh.astype(np.float64), b.astype(np.float64), wg.astype(np.float64), b.astype(np.float64))
    _, gr = orc.gated_dense_bwd(x64, wh.astype(np.float64), wg.astype(np.float64), saved, gout.astype(np.float64), need_dx=False)
    assert rel(out.detach().cpu().numpy(), y) < 2e-6
    assert rel(w[0].grad.cpu().numpy(), gr["wh"]) < 1e-5
    assert rel(w[2].grad.cpu().numpy(), gr["wg"]) < 1e-5
    assert rel(w[1].grad.cpu().numpy(), gr["bh"]) < 1e-5


def test_dense_operand_beyond_2gib(ops):
    """Maximum sizes: the fast tile loads address non-gathered operands with 31-bit byte offsets; a 2.3 GiB input
    must take the 64-bit path (plain), and a row gather out of it must work (gathered operands always use 64-bit
    pointers).  Checked on the last rows, which lie beyond the 2 GiB mark."""
    M, K, N = 800_000, 784, 40
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.randn((M, K), device="cuda", generator=g)
    w = torch.randn((N, K), device="cuda", generator=g) / 28
    b = torch.zeros(N, device="cuda")
    tail = slice(M - 300, M)
    ref = (x[tail].double() @ w.double().T).float()
    y = ops.linear(x, w, b, 0, 0.0, 0.0)
    assert rel(y[tail].cpu().numpy(), ref.cpu().numpy()) < 2e-6
    rows = torch.arange(M - 300, M, device="cuda")
    yg = ops.linear(x, w, b, 0, 0.0, 0.0, rows=rows)
    assert rel(yg.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    wh = torch.randn((64, K), device="cuda", generator=g) / 28; wg = torch.randn((64, K), device="cuda", generator=g) / 28
    bz = torch.zeros(64, device="cuda")
    og = ops.gated_dense(x, wh, bz, wg, bz, rows=rows)
    xr = x[tail].double()
    refg = ((xr @ wh.double().T) * torch.sigmoid(xr @ wg.double().T)).float()
    assert rel(og.cpu().numpy(), refg.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("K", [4, 28, 32, 36, 60, 64, 68, 96, 100])
def test_dense_k_tails(ops, K, gemm_pipe):
    """K tails of every length around the 32-wide slab (zero fill by out-of-range buffer offsets / masked gathers)."""
    rs = np.random.RandomState(K)
    M, N = 200, 72
    x = rs.standard_normal((M + 50, K)).astype(np.float32)
    w = rs.standard_normal((N, K)).astype(np.float32); b = rs.standard_normal(N).astype(np.float32)
    gout = rs.standard_normal((M, N)).astype(np.float32)
    rows = rs.randint(0, M + 50, M).astype(np.int64)
    for gather in (False, True):
        xin = x[rows] if gather else x[:M]
        t = [dev(x if gather else x[:M].copy()).requires_grad_(not gather), dev(w).requires_grad_(True), dev(b).requires_grad_(True)]
        out = ops.linear(t[0], t[1], t[2], 0, 0.0, 0.0, rows=dev(rows) if gather else None)
        out.backward(dev(gout))
        assert rel(out.detach().cpu().numpy(), xin.astype(np.float64) @ w.astype(np.float64).T + b) < 2e-6
        assert rel(t[1].grad.cpu().numpy(), gout.astype(np.float64).T @ xin.astype(np.float64)) < 1e-5
        assert rel(t[2].grad.cpu().numpy(), gout.astype(np.float64).sum(0)) < 1e-5


@pytest.fixture
def x6_all_rows(ops):
    """Drive every eligible launch through the split-bf16 kernel (csrc/evae_gemm_x6.h), whatever its row count (and keep the
    one-launch thin kernels, which would take hidden-width layers of a few thousand rows, to batch-sized row counts)."""
    ops.gemm_x6_configure(1, 0)
    thin = ops.thin_configure(-1)
    ops.thin_configure(128)
    yield
    ops.gemm_x6_configure(1, 2048)
    ops.thin_configure(thin)


@pytest.mark.parametrize("M,K,N", [(37, 52, 24), (130, 300, 300), (1000, 300, 300), (257, 40, 300), (5000, 784, 300),
                                   (129, 296, 40), (2500, 300, 300)])
@pytest.mark.parametrize("spread", [0, 6])
def test_x6_gated_dense_forward_holds_the_fp32_bar(ops, x6_all_rows, M, K, N, spread):
    """Gated layer forward on the bf16 pipe with three-term operand splits: against the fp64 oracle at the tolerance of the
    fp32-MFMA kernel, and no worse than that kernel by more than a small factor -- also when the operands span 10^+-spread
    (per-column scales: the split of each element is relative to ITS magnitude)."""
    rs = np.random.RandomState(M + K + spread)
    x = rs.standard_normal((M, K)).astype(np.float32)
    wh = (rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32); bh = (rs.standard_normal(N) * 0.1).astype(np.float32)
    wg = (rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32); bg = (rs.standard_normal(N) * 0.1).astype(np.float32)
    if spread:
        sc = (10.0 ** rs.uniform(-spread, spread, K)).astype(np.float32)
        x = x * sc; wh = wh / sc; wg = wg / sc
    y, _ = orc.gated_dense(x.astype(np.float64), wh.astype(np.float64), bh.astype(np.float64), wg.astype(np.float64),
                           bg.astype(np.float64))
    t = [dev(a) for a in (x, wh, bh, wg, bg)]
    out6 = ops.gated_dense(*t).cpu().numpy()
    ops.gemm_x6_configure(0, -1)
    out32 = ops.gated_dense(*t).cpu().numpy()
    ops.gemm_x6_configure(1, -1)
    e6, e32 = rel(out6, y), rel(out32, y)
    assert e6 < 2e-6, (e6, e32)
    assert e6 < 2.0 * e32 + 2e-7, (e6, e32)
    assert not np.array_equal(out6, out32) or M < 2000     # the two pipes round differently: equal bits = the switch did nothing


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("M,K,N", [(300, 300, 784), (1000, 300, 40), (2049, 64, 129 * 4)])
def test_x6_linear_forward_holds_the_fp32_bar(ops, x6_all_rows, act, M, K, N):
    rs = np.random.RandomState(M + N + act)
    x = rs.standard_normal((M, K)).astype(np.float32)
    w = (rs.standard_normal((N, K)) * 4 / np.sqrt(K)).astype(np.float32); b = (rs.standard_normal(N) * 0.1).astype(np.float32)
    pre = x.astype(np.float64) @ w.astype(np.float64).T + b
    y = 1 / (1 + np.exp(-pre)) if act == 1 else (np.clip(pre, -6, 2) if act == 2 else pre)
    out6 = ops.linear(dev(x), dev(w), dev(b), act, -6.0, 2.0).cpu().numpy()
    ops.gemm_x6_configure(0, -1)
    out32 = ops.linear(dev(x), dev(w), dev(b), act, -6.0, 2.0).cpu().numpy()
    ops.gemm_x6_configure(1, -1)
    e6, e32 = rel(out6, y), rel(out32, y)
    assert e6 < 2e-6 and e6 < 2.0 * e32 + 2e-7, (e6, e32)


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("M,K,N", [(37, 53, 24), (100, 300, 784), (1000, 300, 40)])
def test_linear_fwd_bwd(ops, act, M, K, N, gemm_pipe):
    rs = np.random.RandomState(M + N + act)
    x = rs.standard_normal((M, K)).astype(np.float32)
    w = (rs.standard_normal((N, K)) * 4 / np.sqrt(K)).astype(np.float32); b = (rs.standard_normal(N) * 0.1).astype(np.float32)
    gout = rs.standard_normal((M, N)).astype(np.float32)
    pre = x.astype(np.float64) @ w.astype(np.float64).T + b
    if act == 1:
        y = 1 / (1 + np.exp(-pre)); dpre = gout * y * (1 - y)
    elif act == 2:
        y = np.clip(pre, -6, 2); dpre = gout * ((pre > -6) & (pre < 2))
    else:
        y = pre; dpre = gout.astype(np.float64)
    t = [dev(a).requires_grad_(True) for a in (x, w, b)]
    out = ops.linear(t[0], t[1], t[2], act, -6.0, 2.0)
    out.backward(dev(gout))
    assert rel(out.detach().cpu().numpy(), y) < 2e-6
    assert rel(t[0].grad.cpu().numpy(), dpre @ w) < 1e-5
    assert rel(t[1].grad.cpu().numpy(), dpre.T @ x) < 1e-5
    assert rel(t[2].grad.cpu().numpy(), dpre.sum(0)) < 1e-5


def test_layers_golden(ops, golden):
    g = golden("g6_layers")
    rs = np.random.RandomState(51)
    R, I, O = 37, 53, 24
    x = rs.standard_normal((R, I)).astype(np.float32)
    wh = (rs.standard_normal((O, I)) * 0.2).astype(np.float32); bh = (rs.standard_normal(O) * 0.1).astype(np.float32)
    wg = (rs.standard_normal((O, I)) * 0.2).astype(np.float32); bg = (rs.standard_normal(O) * 0.1).astype(np.float32)
    gout = rs.standard_normal((R, O)).astype(np.float32)
    t = [dev(a).requires_grad_(True) for a in (x, wh, bh, wg, bg)]
    out = ops.gated_dense(*t)
    out.backward(dev(gout))
    assert rel(out.detach().cpu().numpy(), g["gd_y"]) < 1e-5
    assert rel(t[0].grad.cpu().numpy(), g["gd_dx"]) < 1e-5
    assert rel(t[1].grad.cpu().numpy(), g["gd_dwh"]) < 1e-5
    assert rel(t[4].grad.cpu().numpy(), g["gd_dbg"]) < 1e-5


# ---------------------------------------------------------------- latent / loss rows
def test_reparam_logq_and_densities(ops):
    rs = np.random.RandomState(8)
    B, zd, D = 100, 40, 784
    mu = rs.standard_normal((B, zd)).astype(np.float32); lv = rs.uniform(-6, 2, (B, zd)).astype(np.float32)
    eps = rs.standard_normal((B, zd)).astype(np.float32)
    gz = rs.standard_normal((B, zd)).astype(np.float32); gq = rs.standard_normal(B).astype(np.float32)
    t = [dev(a).requires_grad_(True) for a in (mu, lv)]
    z, logq = ops.ReparamLogQ.apply(t[0], t[1], dev(eps))
    ((z * dev(gz)).sum() + (logq * dev(gq)).sum()).backward()
    mu64, lv64, e64 = mu.astype(np.float64), lv.astype(np.float64), eps.astype(np.float64)
    z_ref = e64 * np.exp(0.5 * lv64) + mu64
    assert rel(z.detach().cpu().numpy(), z_ref) < 1e-6
    assert rel(logq.detach().cpu().numpy(), orc.log_normal_diag(z_ref, mu64, lv64)) < 1e-5
    # analytic: logq = sum -0.5(lv + log2pi + eps^2) -> d/dmu = 0 (+gz), d/dlv = -0.5 (+ gz*eps*std/2)
    assert rel(t[0].grad.cpu().numpy(), gz) < 1e-4
    assert rel(t[1].grad.cpu().numpy(), gz * e64 * np.exp(0.5 * lv64) * 0.5 - 0.5 * gq[:, None]) < 1e-4
    x = rs.standard_normal((B, zd)).astype(np.float32)
    tt = [dev(a).requires_grad_(True) for a in (x, mu, lv)]
    out = ops.LogNormalDiag.apply(*tt)
    (out * dev(gq)).sum().backward()
    assert rel(out.detach().cpu().numpy(), orc.log_normal_diag(x.astype(np.float64), mu64, lv64)) < 1e-5
    d = x.astype(np.float64) - mu64
    assert rel(tt[0].grad.cpu().numpy(), -gq[:, None] * d / np.exp(lv64)) < 1e-5
    assert rel(tt[2].grad.cpu().numpy(), gq[:, None] * -0.5 * (1 - d * d / np.exp(lv64))) < 1e-5
    xm = (1 / (1 + np.exp(-rs.standard_normal((B, D)) * 8))).astype(np.float32)
    xb = (rs.random_sample((B, D)) < 0.3).astype(np.float32)
    mt = dev(xm).requires_grad_(True)
    re = ops.BernoulliLL.apply(dev(xb), mt)
    (re * dev(gq)).sum().backward()
    # fp32 oracle: the clamp constants 1e-5 / 1-1e-5 are fp32 roundings in the reference too
    assert rel(re.detach().cpu().numpy(), orc.log_bernoulli(xb, xm)) < 1e-5
    p = np.clip(xm, np.float32(1e-5), np.float32(1 - 1e-5)).astype(np.float64)
    inside = (xm >= np.float32(1e-5)) & (xm <= np.float32(1 - 1e-5))
    assert rel(mt.grad.cpu().numpy(), gq[:, None] * (xb / p - (1 - xb) / (1 - p)) * inside) < 1e-5


def test_adam_normgrad_golden(ops, golden):
    g = golden("g8_adam")
    ps = [dev(g["p0_%d" % i]) for i in range(4)]
    ms = [torch.zeros_like(p) for p in ps]; vs = [torch.zeros_like(p) for p in ps]
    for step in range(3):
        grads = [dev(g["g%d_%d" % (step, i)]) for i in range(4)]
        ops.adam_normgrad_step(ps, grads, ms, vs, step + 1, 5e-4, 0.9, 0.999, 1e-8, 0.0)
        for i in range(4):
            assert rel(ps[i].cpu().numpy(), g["p%d_%d" % (step + 1, i)]) < 1e-6
    for i in range(4):
        assert rel(ms[i].cpu().numpy(), g["m_%d" % i]) < 1e-6
        assert rel(vs[i].cpu().numpy(), g["v_%d" % i]) < 1e-6


def test_adam_normgrad_ragged_sizes_vs_oracle(ops):
    """Tensor sizes around the kernel's block / vector boundaries (1 element, odd counts, one above ANB * ACHUNK elements,
    a weight-decay run) against the numpy restatement of reference utils/optimizer.py:32-80."""
    rs = np.random.RandomState(21)
    sizes = [1, 7, 1023, 2049, 300 * 784, 128 * 2048 + 5, 40]
    for wd in (0.0, 1e-3):
        p_np = [rs.standard_normal(n).astype(np.float32) for n in sizes]
        m_np = [np.zeros(n, np.float32) for n in sizes]; v_np = [np.zeros(n, np.float32) for n in sizes]
        ps = [dev(a.copy()) for a in p_np]
        ms = [torch.zeros_like(p) for p in ps]; vs = [torch.zeros_like(p) for p in ps]
        for step in range(1, 4):
            g_np = [(rs.standard_normal(n) * 10.0 ** rs.uniform(-3, 1)).astype(np.float32) for n in sizes]
            ops.adam_normgrad_step(ps, [dev(g) for g in g_np], ms, vs, step, 5e-4, 0.9, 0.999, 1e-8, wd)
            for i in range(len(sizes)):
                p_np[i], m_np[i], v_np[i] = orc.adam_normgrad_step(p_np[i], g_np[i], m_np[i], v_np[i], step, weight_decay=wd)
                assert rel(ps[i].cpu().numpy(), p_np[i]) < 1e-6, (sizes[i], step)
                assert rel(ms[i].cpu().numpy(), m_np[i]) < 1e-5, (sizes[i], step)
                assert rel(vs[i].cpu().numpy(), v_np[i]) < 1e-5, (sizes[i], step)


def test_fused_elementwise_backward_launches_match_their_parts():
    """evae_bernoulli_sigmoid_bwd == evae_bernoulli_ll_bwd then evae_act_bwd(sigmoid);
    evae_reparam_logq_bwd_hardtanh == (dz + dz2) -> evae_reparam_logq_bwd -> evae_act_bwd(hardtanh)."""
    import ctypes as C
    from evae import _lib
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    rs = np.random.RandomState(22)
    B, D, zd = 37, 784, 40
    xm = dev((1 / (1 + np.exp(-rs.standard_normal((B, D)) * 8))).astype(np.float32))     # saturates: clamp region hit
    xb = dev((rs.random_sample((B, D)) < 0.3).astype(np.float32))
    c = dev(rs.standard_normal(B).astype(np.float32))
    d1 = torch.empty_like(xm); d2 = torch.empty_like(xm); fused = torch.empty_like(xm)
    _lib.check(lib.evae_bernoulli_ll_bwd(vp(xb), vp(xm), vp(c), B, D, vp(d1), st), "a")
    _lib.check(lib.evae_act_bwd(vp(d1), vp(xm), B * D, 1, 0.0, 0.0, vp(d2), st), "b")
    _lib.check(lib.evae_bernoulli_sigmoid_bwd(vp(xb), vp(xm), vp(c), B, D, vp(fused), st), "c")
    assert rel(fused.cpu().numpy(), d2.cpu().numpy()) < 1e-6
    mu, eps, dz, dz2 = (dev(rs.standard_normal((B, zd)).astype(np.float32)) for _ in range(4))
    pre = dev(rs.uniform(-8, 4, (B, zd)).astype(np.float32))            # part of it outside Hardtanh(-6, 2)
    lv = pre.clamp(-6.0, 2.0)
    z = mu + eps * torch.exp(0.5 * lv)
    dmu_a = torch.empty_like(mu); dlv_a = torch.empty_like(mu); dpre_a = torch.empty_like(mu)
    _lib.check(lib.evae_reparam_logq_bwd(vp(mu), vp(lv), vp(eps), vp(z), vp(dz + dz2), vp(c), B, zd, vp(dmu_a), vp(dlv_a), st), "d")
    _lib.check(lib.evae_act_bwd(vp(dlv_a), vp(pre), B * zd, 2, -6.0, 2.0, vp(dpre_a), st), "e")
    dmu_b = torch.empty_like(mu); dpre_b = torch.empty_like(mu)
    _lib.check(lib.evae_reparam_logq_bwd_hardtanh(vp(mu), vp(lv), vp(eps), vp(z), vp(dz), vp(dz2), vp(c), vp(pre), -6.0, 2.0,
                                                  B, zd, vp(dmu_b), vp(dpre_b), st), "f")
    assert rel(dmu_b.cpu().numpy(), dmu_a.cpu().numpy()) < 1e-6
    assert rel(dpre_b.cpu().numpy(), dpre_a.cpu().numpy()) < 1e-6
    assert float((dpre_b == 0).float().mean()) > 0.1                    # the clipped entries really are in the sample


def test_batch_prologue_gather_binarise_eps(ops):
    """evae_batch_prologue: exact gather without binarisation; Bernoulli(p) frequencies and N(0,1) moments with it;
    same (seed, counter) -> same draws, another counter -> other draws."""
    rs = np.random.RandomState(23)
    N, B, D, zd = 500, 100, 784, 40
    data = dev(rs.random_sample((N + 8, D)).astype(np.float32))[:N]           # a row-strided view, like the resident set
    idx = dev(rs.randint(0, N, B).astype(np.int64))
    sc = lambda seed, ctr: torch.tensor([seed, ctr], dtype=torch.int64, device="cuda")
    x0 = torch.empty((B, D), device="cuda"); e0 = torch.empty((B, zd), device="cuda")
    ops.batch_prologue(data, idx, False, sc(5, 0), x0, e0)
    assert torch.equal(x0, data[idx])
    xs, es = [], []
    for ctr in range(60):
        x = torch.empty((B, D), device="cuda"); e = torch.empty((B, zd), device="cuda")
        ops.batch_prologue(data, idx, True, sc(5, ctr), x, e)
        xs.append(x); es.append(e)
    x_again = torch.empty((B, D), device="cuda"); e_again = torch.empty((B, zd), device="cuda")
    ops.batch_prologue(data, idx, True, sc(5, 7), x_again, e_again)
    assert torch.equal(x_again, xs[7]) and torch.equal(e_again, es[7])
    assert not torch.equal(xs[7], xs[8]) and not torch.equal(es[7], es[8])
    ops.batch_prologue(data, idx, True, sc(6, 7), x_again, e_again)
    assert not torch.equal(e_again, es[7])
    X = torch.stack(xs)                                                       # [60 x B x D] of {0, 1}
    assert set(np.unique(X.cpu().numpy())) <= {0.0, 1.0}
    freq = X.mean(dim=0)
    p = data[idx]
    # per pixel: |freq - p| is a binomial deviation with sigma <= 0.5 / sqrt(60) = 0.065; averaged over 78 400 pixels
    assert float((freq - p).abs().max()) < 0.33
    assert abs(float((freq - p).mean())) < 2e-3
    E = torch.stack(es).double()                                              # 240 000 draws
    assert abs(float(E.mean())) < 0.01 and abs(float(E.var()) - 1.0) < 0.01
    assert abs(float((E ** 3).mean())) < 0.03 and abs(float((E ** 4).mean()) - 3.0) < 0.08
    assert float(E.abs().max()) > 3.5 and bool(torch.isfinite(E).all())
    # no correlation between the two outputs of a Box-Muller pair or between neighbouring steps
    flat = E.reshape(60, -1)
    assert abs(float((flat[:, 0::2] * flat[:, 1::2]).mean())) < 0.01
    assert abs(float((flat[:-1] * flat[1:]).mean())) < 0.01


def test_empty_inputs(ops):
    """The empty cases of the domain: a rank whose exemplar shard is empty (C = 0), an empty batch (B = 0), an empty
    row-gather list, an optimizer without tensors.  Nothing may crash; results are the neutral elements."""
    zd = 40
    z = dev(np.random.RandomState(1).standard_normal((6, zd)).astype(np.float32))
    c = dev(np.random.RandomState(2).standard_normal((50, zd)).astype(np.float32))
    lv = dev(np.zeros(zd, np.float32))
    # empty shard: max = -inf, sumexp = 0, nothing masked; merging it with a real shard changes nothing
    m0, s0, n0, _ = ops.prior_lse_fwd(z, c[:0], lv)
    assert bool(torch.isneginf(m0).all()) and float(s0.abs().sum()) == 0.0 and float(n0.abs().sum()) == 0.0
    m1, s1, n1, _ = ops.prior_lse_fwd(z, c, lv)
    lp_a, lse_a = ops.prior_merge(torch.stack((m1, m0)), torch.stack((s1, s0)), torch.stack((n1, n0)), 50)
    lp_b, lse_b = ops.prior_merge(m1, s1, n1, 50)
    assert torch.equal(lp_a, lp_b) and torch.equal(lse_a, lse_b)
    gz, gc, glv = ops.prior_lse_bwd(z, c[:0], lv, None, None, lse_b, torch.ones(6, device="cuda"))
    assert gc.shape == (0, zd) and float(gz.abs().sum()) == 0.0 and float(glv.abs().sum()) == 0.0
    # empty batch
    mb, sb, nb, _ = ops.prior_lse_fwd(z[:0], c, lv)
    assert mb.numel() == 0
    gz, gc, glv = ops.prior_lse_bwd(z[:0], c, lv, None, None, lse_b[:0], torch.ones(0, device="cuda"))
    assert gz.shape == (0, zd) and float(gc.abs().sum()) == 0.0
    # dense layers over zero rows: outputs empty, weight gradients zero
    w = [dev(a).requires_grad_(True) for a in (np.ones((8, zd), np.float32), np.zeros(8, np.float32),
                                               np.ones((8, zd), np.float32), np.zeros(8, np.float32))]
    out = ops.gated_dense(z[:0], *w)
    assert out.shape == (0, 8)
    out.sum().backward()
    assert all(float(t.grad.abs().sum()) == 0.0 for t in w)
    rows = torch.zeros(0, dtype=torch.int64, device="cuda")
    assert ops.gated_dense(z, *[t.detach() for t in w], rows=rows).shape == (0, 8)
    # optimizer without tensors, prologue without rows
    ops.adam_normgrad_step([], [], [], [], 1, 5e-4, 0.9, 0.999, 1e-8, 0.0)
    ops.batch_prologue(c, rows, True, torch.tensor([1, 0], dtype=torch.int64, device="cuda"), torch.empty((0, zd), device="cuda"),
                       torch.empty((0, 8), device="cuda"))
    torch.cuda.synchronize()


def test_workspace_growth_keeps_the_old_buffer_alive(ops):
    """A captured hipGraph has the workspace address baked in: a later, larger request under the same name must not free
    the buffer the graph still writes to."""
    d = torch.device("cuda", torch.cuda.current_device())
    a = ops._workspace("test_ws_growth", 1000, d)
    pa = a.data_ptr()
    b = ops._workspace("test_ws_growth", 5000, d)
    assert b.data_ptr() != pa and b.numel() >= 5000
    assert any(t.data_ptr() == pa for t in ops._ws_retired)
    assert ops._workspace("test_ws_growth", 4000, d).data_ptr() == b.data_ptr()


@pytest.mark.parametrize("M,K,N,R", [(300, 784, 300, 1000), (25000, 784, 300, 50000), (130, 48, 7, 200), (1, 16, 64, 5),
                                     (257, 560, 300, 300)])
def test_gated_dense_forward_on_the_uint8_store(ops, M, K, N, R):
    """First encoder layer on the byte store (grey pixels k/255): bytes are exact in bf16, fp32 weights split exactly into
    three bf16 terms -- the bf16-MFMA kernel is held to the fp32 kernel's bar against the fp64 oracle."""
    rs = np.random.RandomState(M + K)
    q = (rs.randint(0, 256, (R, K)) * (rs.random_sample((R, K)) < 0.4)).astype(np.uint8)
    rows = rs.randint(0, R, size=M).astype(np.int64)
    wh = (rs.standard_normal((N, K)) * 0.1).astype(np.float32); wg = (rs.standard_normal((N, K)) * 0.1).astype(np.float32)
    bh = (rs.standard_normal(N) * 0.1).astype(np.float32); bg = (rs.standard_normal(N) * 0.1).astype(np.float32)
    store = torch.zeros(R * K + 64, dtype=torch.uint8, device="cuda")          # slack behind the last row
    xs = store[:R * K].view(R, K); xs.copy_(torch.from_numpy(q))
    prep = ops.u8_prepare(dev(wh), dev(wg))
    s = torch.empty((M, N), device="cuda")
    out = ops.gated_dense_fwd_u8(xs, dev(rows), 1.0 / 255.0, prep, dev(bh), dev(bg), N, save_s=s)
    x64 = q[rows].astype(np.float64) / 255.0
    h = x64 @ wh.astype(np.float64).T + bh; g = 1.0 / (1.0 + np.exp(-(x64 @ wg.astype(np.float64).T + bg)))
    assert rel(out.cpu().numpy(), h * g) < 2e-6
    assert rel(s.cpu().numpy(), g) < 2e-6
    # and against the fp32 kernel on the fp32 copy of the same rows (the two are interchangeable)
    ref32 = ops.gated_dense(dev((q.astype(np.float32) / 255.0)), dev(wh), dev(bh), dev(wg), dev(bg), rows=dev(rows))
    assert rel(out.cpu().numpy(), ref32.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("M,K,N,R", [(300, 784, 600, 1000), (25100, 784, 600, 50000), (130, 48, 7, 200), (1, 16, 64, 5),
                                     (257, 560, 600, 300)])
def test_weight_gradient_on_the_uint8_store(ops, M, K, N, R):
    """dW = dy^T x(rows) / 255 and db on the byte store (dy split exactly into three bf16 terms, bytes exact in bf16):
    against the fp64 product and against the fp32 kernel on the fp32 copy of the same rows."""
    rs = np.random.RandomState(M + K + 1)
    q = (rs.randint(0, 256, (R, K)) * (rs.random_sample((R, K)) < 0.4)).astype(np.uint8)
    rows = rs.randint(0, R, size=M).astype(np.int64)
    dy = (rs.standard_normal((M, N)) * np.exp(rs.uniform(-6, 2, (M, 1)))).astype(np.float32)     # rows of very different scale
    store = torch.zeros(R * K + 64, dtype=torch.uint8, device="cuda")
    xs = store[:R * K].view(R, K); xs.copy_(torch.from_numpy(q))
    dw, db = ops.dense_bwd_weight_u8(dev(dy), xs, dev(rows), 1.0 / 255.0)
    ref = dy.astype(np.float64).T @ (q[rows].astype(np.float64) / 255.0)
    assert rel(dw.cpu().numpy(), ref) < 2e-6
    assert rel(db.cpu().numpy(), dy.astype(np.float64).sum(0)) < 2e-6
    dw32, db32 = ops._bwd_weight(dev(dy), dev(q.astype(np.float32) / 255.0), dev(rows), K)
    assert rel(dw.cpu().numpy(), dw32.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("M,K,Z", [(100, 300, 40), (37, 300, 40), (256, 128, 64), (5, 48, 8)])
def test_heads_and_sample_in_one_call(ops, M, K, Z):
    """evae_heads_reparam_fwd (both encoder heads as one split-K GEMM + one finish launch that also samples) against the fp64
    formulas of models/VAE.py:24-26, models/BaseModel.py:79-82 and utils/distributions.py:28-33."""
    from evae import _lib
    lib = _lib.load()
    rs = np.random.RandomState(M + K)
    x = rs.standard_normal((M, K)).astype(np.float32)
    wm = (rs.standard_normal((Z, K)) * 0.1).astype(np.float32); bm = (rs.standard_normal(Z) * 0.1).astype(np.float32)
    wl = (rs.standard_normal((Z, K)) * 0.5).astype(np.float32); bl = (rs.standard_normal(Z) * 0.5).astype(np.float32)
    eps = rs.standard_normal((M, Z)).astype(np.float32)
    t = {k: dev(v) for k, v in dict(x=x, wm=wm, bm=bm, wl=wl, bl=bl, eps=eps).items()}
    out = {k: torch.full((M, Z), float("nan"), device="cuda") for k in ("mean", "pre", "lv", "z")}
    logq = torch.full((M,), float("nan"), device="cuda")
    nb = lib.evae_heads_reparam_fwd_workspace_bytes(M, K, Z)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    p = lambda a: a.data_ptr()
    _lib.check(lib.evae_heads_reparam_fwd(p(t["x"]), M, K, K, p(t["wm"]), p(t["bm"]), p(t["wl"]), p(t["bl"]), Z, -6.0, 2.0, p(t["eps"]),
                                          p(out["mean"]), p(out["pre"]), p(out["lv"]), p(out["z"]), p(logq), p(ws), nb,
                                          torch.cuda.current_stream().cuda_stream), "heads_reparam_fwd")
    x64 = x.astype(np.float64)
    mean = x64 @ wm.astype(np.float64).T + bm
    pre = x64 @ wl.astype(np.float64).T + bl
    lv = np.clip(pre, -6.0, 2.0)
    z = mean + eps * np.exp(0.5 * lv)
    lq = (-0.5 * (lv + np.log(2 * np.pi) + (z - mean) ** 2 / np.exp(lv))).sum(1)
    assert rel(out["mean"].cpu().numpy(), mean) < 2e-6 and rel(out["pre"].cpu().numpy(), pre) < 2e-6
    # (the clamped log-variance is held to the bar of the pre-activation it is clipped from: an fp32 dot product's error scales
    # with the magnitude of its terms, not with the clipped value)
    assert np.abs(out["lv"].cpu().numpy() - lv).max() < 2e-6 * np.abs(pre).max() and rel(out["z"].cpu().numpy(), z) < 5e-6
    assert rel(logq.cpu().numpy(), lq) < 5e-6


def test_step_head_launch_also_transposes_weights_for_the_data_gradients(ops):
    """evae_batch_prologue_u8_prepare with transposition jobs: the buffers are w^T with the row stride evae_dense_bwd_data_wt
    expects, the weight split and the batch are what the separate launches produce, and the data gradient fed with the
    prepared buffer equals evae_dense_bwd_data bit for bit."""
    from evae import _lib
    lib = _lib.load()
    rs = np.random.RandomState(9)
    B, D, R, H, Z, M = 16, 784, 300, 300, 40, 25000
    q = (rs.randint(0, 256, (R, D)) * (rs.random_sample((R, D)) < 0.4)).astype(np.uint8)
    store = torch.zeros((R + B) * D + 64, dtype=torch.uint8, device="cuda"); xs = store[:(R + B) * D].view(R + B, D)
    xs[:R].copy_(torch.from_numpy(q))
    idx = dev(rs.randint(0, R, size=B).astype(np.int64)); seed = dev(np.array([5, 7], dtype=np.int64))
    wh, wg = dev((rs.standard_normal((H, D)) * 0.1).astype(np.float32)), dev((rs.standard_normal((H, D)) * 0.1).astype(np.float32))
    wm = dev((rs.standard_normal((Z, H)) * 0.1).astype(np.float32))
    w2h, w2g = dev((rs.standard_normal((H, H)) * 0.1).astype(np.float32)), dev((rs.standard_normal((H, H)) * 0.1).astype(np.float32))
    t1 = torch.zeros(lib.evae_dense_bwd_data_wt_bytes(Z, H, 1), dtype=torch.uint8, device="cuda")
    t2 = torch.zeros(lib.evae_dense_bwd_data_wt_bytes(H, H, 2), dtype=torch.uint8, device="cuda")
    prep = torch.zeros(lib.evae_dense_u8_prepared_bytes(H, D), dtype=torch.uint8, device="cuda")
    x = torch.empty((B, D), device="cuda"); eps = torch.empty((B, Z), device="cuda")
    ops.batch_prologue_u8(xs[:R], idx, False, seed, 255.0, x, xs[R:], eps, prepare=(wh, wg, prep, [(wm, None, t1), (w2h, w2g, t2)]))
    assert torch.equal(prep, ops.u8_prepare(wh, wg))
    assert np.array_equal(x.cpu().numpy(), q[idx.cpu().numpy()].astype(np.float32) / np.float32(255.0))     # IEEE division, as the fp32 dataset
    ld1, ld2 = lib.evae_dense_bwd_data_wt_ld(Z), lib.evae_dense_bwd_data_wt_ld(H)
    assert torch.equal(t1.view(torch.float32)[:H * ld1].view(H, ld1)[:, :Z], wm.t())
    tt = t2.view(torch.float32)[:2 * H * ld2].view(2, H, ld2)
    assert torch.equal(tt[0][:, :H], w2h.t()) and torch.equal(tt[1][:, :H], w2g.t())
    # the data gradient with the prepared buffer == the one that transposes itself
    dq = dev((rs.standard_normal((M, 2 * H)) * 0.1).astype(np.float32))
    a1 = dev(rs.standard_normal((M, H)).astype(np.float32)); s1 = dev(rs.random_sample((M, H)).astype(np.float32))
    nb = lib.evae_dense_bwd_data_workspace_bytes(M, H, H, 2)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    outs = []
    st = torch.cuda.current_stream().cuda_stream
    for wT in (None, t2):
        o = torch.full((M, 2 * H), float("nan"), device="cuda")
        _lib.check(lib.evae_dense_bwd_data_wt(dq.data_ptr(), w2h.data_ptr(), dq.data_ptr() + 4 * H, w2g.data_ptr(), M, H, 2 * H, H,
                                              a1.data_ptr(), s1.data_ptr(), o.data_ptr(), o.data_ptr() + 4 * H, 2 * H,
                                              None if wT is None else wT.data_ptr(), ws.data_ptr(), nb, st), "bwd_data_wt")
        outs.append(o)
    assert torch.equal(outs[0], outs[1])


def test_gated_dense_autograd_on_the_uint8_store(ops):
    """ops.gated_dense on the byte store (the modular path's exemplar encoder, e.g. hvae_2level): output and the four parameter
    gradients against the fp32 Function on the fp32 copy of the same rows."""
    rs = np.random.RandomState(3)
    M, K, N, R = 1500, 784, 300, 4000
    q = (rs.randint(0, 256, (R, K)) * (rs.random_sample((R, K)) < 0.4)).astype(np.uint8)
    rows = dev(rs.randint(0, R, size=M).astype(np.int64))
    store = torch.zeros(R * K + 64, dtype=torch.uint8, device="cuda"); xs = store[:R * K].view(R, K); xs.copy_(torch.from_numpy(q))
    x32 = dev(q.astype(np.float32) / 255.0)
    par = [dev((rs.standard_normal(sh) * 0.1).astype(np.float32)) for sh in ((N, K), (N,), (N, K), (N,))]
    g = dev(rs.standard_normal((M, N)).astype(np.float32))
    res = []
    for x, kw in ((xs, dict(x_scale=1.0 / 255.0)), (x32, {})):
        ps = [p.clone().requires_grad_(True) for p in par]
        out = ops.gated_dense(x, ps[0], ps[1], ps[2], ps[3], rows=rows, **kw)
        out.backward(g)
        res.append([out.detach()] + [p.grad for p in ps])
    for a, b in zip(*res):
        assert rel(a.cpu().numpy(), b.cpu().numpy()) < 3e-6


def test_weight_gradient_on_the_uint8_store_in_phases(ops):
    """phase 3 (byte gather-transpose alone: what a training step issues during its forward pass, dy not yet known) followed
    by phase 4 (everything else) = the one-call weight gradient, bit for bit; likewise phases 1 + 2."""
    from evae import _lib
    lib = _lib.load()
    rs = np.random.RandomState(5)
    M, K, N, R = 1100, 784, 600, 3000
    q = (rs.randint(0, 256, (R, K)) * (rs.random_sample((R, K)) < 0.4)).astype(np.uint8)
    rows = dev(rs.randint(0, R, size=M).astype(np.int64))
    dy = dev((rs.standard_normal((M, N)) * 0.3).astype(np.float32))
    store = torch.zeros(R * K + 64, dtype=torch.uint8, device="cuda"); xs = store[:R * K].view(R, K); xs.copy_(torch.from_numpy(q))
    dw0, db0 = ops.dense_bwd_weight_u8(dy, xs, rows, 1.0 / 255.0, ws_name="t_ph0")
    nb = lib.evae_dense_bwd_weight_u8_workspace_bytes(M, N, K)
    st = torch.cuda.current_stream().cuda_stream
    for first, second in ((3, 4), (1, 2)):
        ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        dw = torch.full((N, K), float("nan"), device="cuda"); db = torch.full((N,), float("nan"), device="cuda")
        none = None
        _lib.check(lib.evae_dense_bwd_weight_u8_phased(none if first == 3 else dy.data_ptr(), M, N, N, xs.data_ptr(), rows.data_ptr(), K, K,
                                                       1.0 / 255.0, none if first == 3 else dw.data_ptr(),
                                                       none if first == 3 else db.data_ptr(), ws.data_ptr(), nb, first, st), "phase a")
        _lib.check(lib.evae_dense_bwd_weight_u8_phased(dy.data_ptr(), M, N, N, xs.data_ptr(), rows.data_ptr(), K, K, 1.0 / 255.0,
                                                       dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, second, st), "phase b")
        assert torch.equal(dw, dw0) and torch.equal(db, db0), (first, second)


@pytest.mark.parametrize("M1,M2", [(600, 100), (1000, 36), (2568, 100)])
def test_data_gradient_that_writes_bf16_tile_images(ops, gemm_pipe, M1, M2):
    """evae_dense_bwd_data_img + evae_dense_bwd_weight_u8(dy = NULL): the layer-above data gradient leaves (dh, dg) as the
    three-term bf16 tile images the byte-store weight gradient reads -- same dW / db as the fp32 buffer + pre-pass route,
    with the rows arriving in two launches (exemplar rows, batch rows)."""
    import ctypes as C
    from evae import _lib
    lib = _lib.load()
    rs = np.random.RandomState(12)
    H, D, R = 300, 784, 2000
    M = M1 + M2              # (the second launch may end off a multiple of 8: its last rows are zero-filled up to one)
    q = (rs.randint(0, 256, (R, D)) * (rs.random_sample((R, D)) < 0.3)).astype(np.uint8)
    rows = dev(rs.randint(0, R, size=M).astype(np.int64))
    store = torch.zeros(R * D + 64, dtype=torch.uint8, device="cuda"); xs = store[:R * D].view(R, D); xs.copy_(torch.from_numpy(q))
    dq2 = dev((rs.standard_normal((M, 2 * H)) * 0.1).astype(np.float32))
    w2h = dev((rs.standard_normal((H, H)) * 0.1).astype(np.float32)); w2g = dev((rs.standard_normal((H, H)) * 0.1).astype(np.float32))
    a1 = dev(rs.standard_normal((M, H)).astype(np.float32)); s1 = dev(rs.random_sample((M, H)).astype(np.float32))
    # route A: fp32 [dh | dg] buffer, then the weight gradient with its transposing pre-pass
    dq1 = torch.empty((M, 2 * H), device="cuda")
    ops._bwd_data(dq2.data_ptr(), w2h, dq2.data_ptr() + 4 * H, w2g, M, H, 2 * H, "cuda", out_prev=a1, s_prev=s1, out=dq1,
                  dg_ptr=dq1.data_ptr() + 4 * H, ldo=2 * H)
    dwA, dbA = ops.dense_bwd_weight_u8(dq1, xs, rows, 1.0 / 255.0, ws_name="t_wgrad_a")
    # route B: tile images written by the data gradient itself
    nb = lib.evae_dense_bwd_weight_u8_workspace_bytes(M, 2 * H, D)
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    off, nslab = C.c_size_t(0), C.c_int(0)
    _lib.check(lib.evae_dense_bwd_weight_u8_images(M, 2 * H, D, C.byref(off), C.byref(nslab)), "images")
    img = ws.data_ptr() + off.value
    st = ops._stream()
    wsd = torch.zeros(lib.evae_dense_bwd_data_workspace_bytes(M, H, H, 2), dtype=torch.uint8, device="cuda")
    for m0, mm in ((0, M1), (M1, M2)):
        _lib.check(lib.evae_dense_bwd_data_img(C.c_void_p(dq2.data_ptr() + 4 * m0 * 2 * H), ops._p(w2h),
                                               C.c_void_p(dq2.data_ptr() + 4 * m0 * 2 * H + 4 * H), ops._p(w2g), mm, H, 2 * H, H,
                                               C.c_void_p(a1.data_ptr() + 4 * m0 * H), C.c_void_p(s1.data_ptr() + 4 * m0 * H),
                                               C.c_void_p(img), nslab.value, m0, None, ops._p(wsd), wsd.numel(), st), "bwd_data_img")
    dwB = torch.empty((2 * H, D), device="cuda"); dbB = torch.empty(2 * H, device="cuda")
    _lib.check(lib.evae_dense_bwd_weight_u8(None, M, 2 * H, 2 * H, ops._p(xs), ops._p(rows), D, D, 1.0 / 255.0, ops._p(dwB),
                                            ops._p(dbB), ops._p(ws), ws.numel(), st), "bwd_weight_u8")
    assert rel(dwB.cpu().numpy(), dwA.cpu().numpy()) < 2e-6
    assert rel(dbB.cpu().numpy(), dbA.cpu().numpy()) < 2e-6


def test_grouped_thin_weight_gradients_equal_the_single_launches(ops):
    """evae_dense_bwd_weight_group: the batch rows' four leaf weight gradients of a training step (decoder layers, log-variance
    head; contraction = 100 batch rows) in ONE launch -- the same kernel body per job, so bit-identical to four
    evae_dense_bwd_weight launches, and both against float64."""
    import ctypes as C
    from evae import _lib
    lib = _lib.load()
    rs = np.random.RandomState(31)
    B = 100
    shapes = [(784, 300), (600, 300), (600, 40), (40, 300), (8, 12), (132, 68)]      # (N, K) of dw [N x K]
    dys = [dev((rs.standard_normal((B, n)) * 0.1).astype(np.float32)) for n, _ in shapes]
    xs = [dev(rs.standard_normal((B, k)).astype(np.float32)) for _, k in shapes]
    arr = (_lib.WgradJob * len(shapes))()
    dws = [torch.empty((n, k), device="cuda") for n, k in shapes]
    dbs = [torch.empty(n, device="cuda") for n, _ in shapes]
    for i, ((n, k), dy, x) in enumerate(zip(shapes, dys, xs)):
        arr[i].dy, arr[i].x, arr[i].dw, arr[i].db = dy.data_ptr(), x.data_ptr(), dws[i].data_ptr(), dbs[i].data_ptr()
        arr[i].M, arr[i].N, arr[i].K, arr[i].ldy, arr[i].ldx = B, n, k, n, k
    _lib.check(lib.evae_dense_bwd_weight_group(C.cast(arr, C.c_void_p), len(shapes), ops._stream()), "group")
    for i, ((n, k), dy, x) in enumerate(zip(shapes, dys, xs)):
        ref_w = dy.double().t() @ x.double(); ref_b = dy.double().sum(0)
        assert rel(dws[i].cpu().numpy(), ref_w.cpu().numpy()) < 2e-6, shapes[i]
        assert rel(dbs[i].cpu().numpy(), ref_b.cpu().numpy()) < 2e-6, shapes[i]
        # the single-launch entry point on the same operands
        dw1 = torch.empty((n, k), device="cuda"); db1 = torch.empty(n, device="cuda")
        nb = lib.evae_dense_bwd_weight_workspace_bytes(B, n, k)
        ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        _lib.check(lib.evae_dense_bwd_weight(ops._p(dy), B, n, n, ops._p(x), None, k, k, ops._p(dw1), ops._p(db1), 0, ops._p(ws), nb,
                                             ops._stream()), "single")
        assert torch.equal(dw1, dws[i]) and torch.equal(db1, dbs[i]), shapes[i]
    # a job the group cannot take (contraction over more than 128 rows) is refused, nothing launched
    arr[0].M = 200
    assert lib.evae_dense_bwd_weight_group(C.cast(arr, C.c_void_p), len(shapes), ops._stream()) != 0


def test_grouped_finish_of_split_k_weight_gradients_equals_the_separate_finishes(ops):
    """evae_dense_bwd_weight_finish_group: the byte layer's and two fp32 layers' split-K planes summed by ONE launch -- same
    bodies, same order, so bit-identical to the GEMM + own-finish entry points."""
    import ctypes as C
    from evae import _lib
    lib = _lib.load()
    rs = np.random.RandomState(41)
    M, H, D, Z, R = 2500, 300, 784, 40, 4000
    q = (rs.randint(0, 256, (R, D)) * (rs.random_sample((R, D)) < 0.3)).astype(np.uint8)
    store = torch.zeros(R * D + 64, dtype=torch.uint8, device="cuda"); xs = store[:R * D].view(R, D); xs.copy_(torch.from_numpy(q))
    rows = dev(rs.randint(0, R, size=M).astype(np.int64))
    dq1 = dev((rs.standard_normal((M, 2 * H)) * 0.1).astype(np.float32)); dq2 = dev((rs.standard_normal((M, 2 * H)) * 0.1).astype(np.float32))
    dm = dev((rs.standard_normal((M, Z)) * 0.1).astype(np.float32))
    a1 = dev(rs.standard_normal((M, H)).astype(np.float32)); a2 = dev(rs.standard_normal((M, H)).astype(np.float32))
    st = ops._stream()
    # separate entry points
    dw1, db1 = ops.dense_bwd_weight_u8(dq1, xs, rows, 1.0 / 255.0, ws_name="t_fg_a")
    ref = {}
    for tag, dy, x, n, k in (("w2", dq2, a1, 2 * H, H), ("wm", dm, a2, Z, H)):
        dw = torch.empty((n, k), device="cuda"); db = torch.empty(n, device="cuda")
        nb = lib.evae_dense_bwd_weight_workspace_bytes(M, n, k); ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        _lib.check(lib.evae_dense_bwd_weight(ops._p(dy), M, n, n, ops._p(x), None, k, k, ops._p(dw), ops._p(db), 0, ops._p(ws), nb, st), tag)
        ref[tag] = (dw, db)
    # GEMMs without their finish, then the grouped finish
    nb1 = lib.evae_dense_bwd_weight_u8_workspace_bytes(M, 2 * H, D); ws1 = torch.zeros(nb1, dtype=torch.uint8, device="cuda")
    g1 = torch.empty((2 * H, D), device="cuda"); gb1 = torch.empty(2 * H, device="cuda")
    _lib.check(lib.evae_dense_bwd_weight_u8_phased(ops._p(dq1), M, 2 * H, 2 * H, ops._p(xs), ops._p(rows), D, D, 1.0 / 255.0, ops._p(g1),
                                                   ops._p(gb1), ops._p(ws1), nb1, 16, st), "u8 gemm only")
    outs, wss = {}, {}
    for tag, dy, x, n, k in (("w2", dq2, a1, 2 * H, H), ("wm", dm, a2, Z, H)):
        dw = torch.empty((n, k), device="cuda"); db = torch.empty(n, device="cuda")
        nb = lib.evae_dense_bwd_weight_workspace_bytes(M, n, k); ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        _lib.check(lib.evae_dense_bwd_weight_phased(ops._p(dy), M, n, n, ops._p(x), None, k, k, ops._p(dw), ops._p(db), 0, ops._p(ws), nb, 1,
                                                    st), tag)
        outs[tag] = (dw, db); wss[tag] = ws
    arr = (_lib.WgradFinishJob * 3)()
    for i, (byte, n, k, ldx, xs_, dw, db, ws) in enumerate(((1, 2 * H, D, D, 1.0 / 255.0, g1, gb1, ws1),
                                                            (0, 2 * H, H, H, 1.0, outs["w2"][0], outs["w2"][1], wss["w2"]),
                                                            (0, Z, H, H, 1.0, outs["wm"][0], outs["wm"][1], wss["wm"]))):
        arr[i].byte_rows, arr[i].M, arr[i].N, arr[i].K, arr[i].ldy, arr[i].ldx, arr[i].x_scale = byte, M, n, k, n, ldx, xs_
        arr[i].dw, arr[i].db, arr[i].ws, arr[i].ws_bytes = dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel()
    _lib.check(lib.evae_dense_bwd_weight_finish_group(C.cast(arr, C.c_void_p), 3, st), "finish group")
    assert torch.equal(g1, dw1) and torch.equal(gb1, db1)
    assert torch.equal(outs["w2"][0], ref["w2"][0]) and torch.equal(outs["w2"][1], ref["w2"][1])
    # (the narrow head's own finish -- narrow_finish_kernel, r06 -- sums its planes 16 ways, the grouped one 8 ways: same planes, rounding apart)
    assert rel(outs["wm"][0].cpu().numpy(), ref["wm"][0].cpu().numpy()) < 1e-6 and rel(outs["wm"][1].cpu().numpy(), ref["wm"][1].cpu().numpy()) < 1e-6
    assert rel(g1.cpu().numpy(), (dq1.double().t() @ (xs[rows].double() / 255.0)).cpu().numpy()) < 2e-6


@pytest.mark.parametrize("M,K,Z", [(100, 300, 40), (37, 300, 40), (200, 128, 64), (5, 48, 6)])
def test_heads_sample_density_function_and_its_gradients(ops, M, K, Z):
    """evae.ops.HeadsReparamFn (mean head, Hardtanh log-variance head, sample, log q in two launches; backward in four) against
    the float64 tensor expressions of reference models/VAE.py:24-26, BaseModel.py:79-82, utils/distributions.py:28-33, with
    upstream gradients on all four outputs.  (200, 128, 64) and (5, 48, 6) take the un-grouped weight-gradient calls."""
    g = torch.Generator(device="cuda").manual_seed(3)
    r = lambda *s, k=1.0: torch.randn(*s, device="cuda", generator=g) * k
    h, wm, bm, wl, bl, eps = r(M, K), r(Z, K, k=0.1), r(Z, k=0.1), r(Z, K, k=0.3), r(Z), r(M, Z)
    gz, gm, glv, gq = r(M, Z), r(M, Z), r(M, Z), r(M)
    leaves = [t.clone().requires_grad_() for t in (h, wm, bm, wl, bl)]
    z, mu, lv, lq = ops.heads_reparam(*leaves, eps, -6.0, 2.0)
    ((z * gz).sum() + (mu * gm).sum() + (lv * glv).sum() + (lq * gq).sum()).backward()
    ref = [t.double().clone().requires_grad_() for t in (h, wm, bm, wl, bl)]
    h_, wm_, bm_, wl_, bl_ = ref
    mu_ = h_ @ wm_.t() + bm_
    lv_ = torch.nn.functional.hardtanh(h_ @ wl_.t() + bl_, -6.0, 2.0)
    z_ = mu_ + eps.double() * torch.exp(0.5 * lv_)
    lq_ = (-0.5 * (lv_ + np.log(2 * np.pi) + (z_ - mu_) ** 2 / torch.exp(lv_))).sum(1)
    ((z_ * gz.double()).sum() + (mu_ * gm.double()).sum() + (lv_ * glv.double()).sum() + (lq_ * gq.double()).sum()).backward()
    assert (lv_.detach().abs() == 6.0).any() or (lv_.detach() == 2.0).any() or M < 10      # the clamp is exercised
    for a, b in ((z, z_), (mu, mu_), (lv, lv_), (lq, lq_)):
        assert rel(a.detach().cpu().numpy(), b.detach().cpu().numpy()) < 2e-6
    for name, a, b in zip("h wm bm wl bl".split(), leaves, ref):
        assert rel(a.grad.cpu().numpy(), b.grad.cpu().numpy()) < 5e-6, name
    # only z and log q used (the training step): None upstream gradients for the moments
    leaves2 = [t.clone().requires_grad_() for t in (h, wm, bm, wl, bl)]
    z2, _, _, lq2 = ops.heads_reparam(*leaves2, eps, -6.0, 2.0)
    ((z2 * gz).sum() + (lq2 * gq).sum()).backward()
    for t in ref:
        t.grad = None
    mu_ = h_ @ wm_.t() + bm_
    lv_ = torch.nn.functional.hardtanh(h_ @ wl_.t() + bl_, -6.0, 2.0)
    z_ = mu_ + eps.double() * torch.exp(0.5 * lv_)
    lq_ = (-0.5 * (lv_ + np.log(2 * np.pi) + (z_ - mu_) ** 2 / torch.exp(lv_))).sum(1)
    ((z_ * gz.double()).sum() + (lq_ * gq.double()).sum()).backward()
    for name, a, b in zip("h wm bm wl bl".split(), leaves2, ref):
        assert rel(a.grad.cpu().numpy(), b.grad.cpu().numpy()) < 5e-6, name


@pytest.mark.parametrize("two", [False, True])
@pytest.mark.parametrize("average", [False, True])
def test_elbo_function_and_its_gradients(ops, two, average):
    """evae.ops.ElboFn (one launch each way) against the tensor expressions of reference models/BaseModel.py:71-77 and the
    two-level grouping of AbsHModel.py:88-106; beta as a number and as a device scalar."""
    B = 100
    g = torch.Generator(device="cuda").manual_seed(11)
    ins = [torch.randn(B, device="cuda", generator=g) * s for s in (50.0, 30.0, 30.0, 20.0, 20.0)]
    up = [torch.randn((), device="cuda", generator=g) for _ in range(3)] if average else \
         [torch.randn(B, device="cuda", generator=g) for _ in range(3)]
    for beta in (0.37, torch.tensor([0.37], device="cuda")):
        leaves = [t.clone().requires_grad_() for t in ins]
        RE, q1, p1, q2, p2 = leaves
        loss, RE_o, KL = ops.elbo(RE, q1, p1, beta, average, q2 if two else None, p2 if two else None)
        ((loss * up[0]).sum() + (RE_o * up[1]).sum() + (KL * up[2]).sum()).backward()
        ref = [t.double().clone().requires_grad_() for t in ins]
        RE_, q1_, p1_, q2_, p2_ = ref
        KL_ = (q1_ - p1_) + (q2_ - p2_) if two else q1_ - p1_
        loss_ = -RE_ + 0.37 * KL_
        outs_ = (loss_.mean(), RE_.mean(), KL_.mean()) if average else (loss_, RE_, KL_)
        sum((o * u.double()).sum() for o, u in zip(outs_, up)).backward()
        for a, b in zip((loss, RE_o, KL), outs_):
            assert a.shape == b.shape and rel(a.detach().cpu().numpy(), b.detach().cpu().numpy()) < 1e-6
        for i, (a, b) in enumerate(zip(leaves, ref)):
            if i >= 3 and not two:
                assert a.grad is None
            else:
                assert rel(a.grad.cpu().numpy(), b.grad.cpu().numpy()) < 1e-6, i


@pytest.mark.parametrize("M,N,K,ldy,ldx", [(25100, 40, 300, 40, 300), (11600, 40, 300, 40, 300), (4099, 8, 67, 12, 68),
                                           (2048, 64, 64, 64, 64), (9000, 36, 784, 600, 784)])
def test_narrow_weight_gradient_streams_the_rows(ops, M, N, K, ldy, ldx):
    """evae_dense_bwd_weight for a narrow output (N <= 64: the encoder heads' [40 x 300] over all C + B rows) takes the
    streaming kernel (narrow_wgrad_kernel: row slices x 64-column tiles, partial planes, the common finish) instead of a GEMM
    whose one row tile is mostly empty: dw and db against float64, strided dy / x, accumulate, and equal to the GEMM path
    (EVAE_WGRAD_NARROW=0 is read once per process, so the comparison is against float64 with the GEMM kernel's bar)."""
    g = torch.Generator(device="cuda").manual_seed(M + N)
    dyb = torch.randn(M, ldy, device="cuda", generator=g) * 0.05
    xb = torch.randn(M, ldx, device="cuda", generator=g)
    dy, x = dyb[:, :N], xb[:, :K]
    lib = ops._lib.load()
    nb = lib.evae_dense_bwd_weight_workspace_bytes(M, N, K)
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    dw = torch.full((N, K), 7.0, device="cuda"); db = torch.full((N,), 7.0, device="cuda")
    call = lambda acc: ops._lib.check(lib.evae_dense_bwd_weight(ops._p(dy), M, N, ldy, ops._p(x), None, K, ldx, ops._p(dw), ops._p(db), acc,
                                                                ops._p(ws), nb, ops._stream()), "bwd_weight")
    call(0)
    ref = dy.double().t() @ x.double(); refb = dy.double().sum(0)
    assert rel(dw.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    assert rel(db.cpu().numpy(), refb.cpu().numpy()) < 2e-6
    call(1)                                                       # accumulate on top
    assert rel(dw.cpu().numpy(), 2 * ref.cpu().numpy()) < 2e-6
    assert rel(db.cpu().numpy(), 2 * refb.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("M,K,Z", [(100, 300, 40), (37, 300, 40), (200, 128, 64), (5, 48, 8)])
def test_heads_density_function_and_its_gradients(ops, M, K, Z):
    """evae.ops.HeadsDensityFn (p(z1 | z2): mean head, Hardtanh log-variance head and log N(zq | mean, exp(logvar)) in two launches;
    backward in four) against the float64 tensor expressions of reference models/AbsHModel.py:17-20,99-100 and
    utils/distributions.py:28-33, upstream gradients on all three outputs, and on the density alone (the training step)."""
    g = torch.Generator(device="cuda").manual_seed(9)
    r = lambda *s, k=1.0: torch.randn(*s, device="cuda", generator=g) * k
    h, wm, bm, wl, bl, zq = r(M, K), r(Z, K, k=0.1), r(Z, k=0.1), r(Z, K, k=0.3), r(Z), r(M, Z)
    gm, glv, gp = r(M, Z), r(M, Z), r(M)
    for only_density in (False, True):
        leaves = [t.clone().requires_grad_() for t in (h, wm, bm, wl, bl, zq)]
        mu, lv, lp = ops.heads_density(*leaves, -6.0, 2.0)
        obj = (lp * gp).sum() if only_density else (mu * gm).sum() + (lv * glv).sum() + (lp * gp).sum()
        obj.backward()
        ref = [t.double().clone().requires_grad_() for t in (h, wm, bm, wl, bl, zq)]
        h_, wm_, bm_, wl_, bl_, zq_ = ref
        mu_ = h_ @ wm_.t() + bm_
        lv_ = torch.nn.functional.hardtanh(h_ @ wl_.t() + bl_, -6.0, 2.0)
        lp_ = (-0.5 * (lv_ + np.log(2 * np.pi) + (zq_ - mu_) ** 2 / torch.exp(lv_))).sum(1)
        obj_ = (lp_ * gp.double()).sum() if only_density else (mu_ * gm.double()).sum() + (lv_ * glv.double()).sum() + (lp_ * gp.double()).sum()
        obj_.backward()
        for a, b in ((mu, mu_), (lv, lv_), (lp, lp_)):
            assert rel(a.detach().cpu().numpy(), b.detach().cpu().numpy()) < 2e-6
        for name, a, b in zip("h wm bm wl bl zq".split(), leaves, ref):
            assert rel(a.grad.cpu().numpy(), b.grad.cpu().numpy()) < 5e-6, (name, only_density)


@pytest.mark.parametrize("R,B", [(196, 100), (1, 1), (3, 17), (300, 130), (64, 16)])
def test_elbo_tail_as_two_launches_equals_the_one_launch_form(ops, R, B):
    """evae_prior_merge_coef + evae_elbo_assemble (r04: merge on the prior's stream, assembly beside the backward) against
    evae_prior_elbo_fwd_coef on the same partial rows (some rows and whole queries empty), and against a float64 merge"""
    import ctypes as C
    lib = ops._lib.load()
    rng = np.random.default_rng(R * 1000 + B)
    pm = rng.normal(-300.0, 40.0, (R, B)).astype(np.float32)
    ps = rng.uniform(1.0, 50.0, (R, B)).astype(np.float32)
    pn = rng.integers(0, 3, (R, B)).astype(np.float32)
    empty = rng.random((R, B)) < 0.1
    pm[empty] = -np.inf; ps[empty] = 0.0
    if B > 4:
        pm[:, 3] = -np.inf; ps[:, 3] = 0.0          # a query with no unmasked exemplar at all
    RE = rng.normal(-90.0, 5.0, B).astype(np.float32); lq = rng.normal(-40.0, 3.0, B).astype(np.float32)
    beta, ct = 0.37, 5000.0
    d = {k: dev(v) for k, v in dict(pm=pm, ps=ps, pn=pn, RE=RE, lq=lq).items()}
    vp = lambda t_: C.c_void_p(t_.data_ptr())
    st = ops._stream()
    outs = []
    for two in (False, True):
        o = {k: torch.full((n_,), 7.0, device="cuda") for k, n_ in dict(logp=B, lse=2 * B, loss=B, KL=B, means=3, cRE=B, cKL=B, nck=B).items()}
        if two:
            ops._lib.check(lib.evae_prior_merge_coef(vp(d["pm"]), vp(d["ps"]), vp(d["pn"]), R, B, B, ct, None, beta, vp(o["logp"]),
                                                     vp(o["lse"]), vp(o["cRE"]), vp(o["cKL"]), vp(o["nck"]), st), "merge_coef")
            ops._lib.check(lib.evae_elbo_assemble(vp(o["logp"]), vp(d["RE"]), vp(d["lq"]), None, beta, B, vp(o["loss"]), vp(o["KL"]),
                                                  vp(o["means"]), st), "assemble")
        else:
            ops._lib.check(lib.evae_prior_elbo_fwd_coef(vp(d["pm"]), vp(d["ps"]), vp(d["pn"]), R, B, B, ct, vp(d["RE"]), vp(d["lq"]), None,
                                                        beta, vp(o["logp"]), vp(o["lse"]), vp(o["loss"]), vp(o["KL"]), vp(o["means"]),
                                                        vp(o["cRE"]), vp(o["cKL"]), vp(o["nck"]), st), "elbo_fwd_coef")
        torch.cuda.synchronize()
        outs.append({k: v.cpu().numpy() for k, v in o.items()})
    a, b = outs
    for k in ("cRE", "cKL", "nck"):
        assert np.array_equal(a[k], b[k]), k
    fin = np.isfinite(a["logp"])
    assert np.array_equal(fin, np.isfinite(b["logp"]))
    # float64 merge of the same partials
    M = pm.max(0).astype(np.float64)
    with np.errstate(invalid="ignore"):
        S = np.where(np.isinf(pm), 0.0, ps.astype(np.float64) * np.exp(pm.astype(np.float64) - M)).sum(0)
    ref_lp = M + np.log(np.where(S > 0, S, 1.0)) - np.log(ct - pn.astype(np.float64).sum(0))
    assert rel(b["logp"][fin], ref_lp[fin]) < 2e-7
    assert rel(a["logp"][fin], b["logp"][fin]) < 2e-7
    assert np.array_equal(a["lse"][:B], b["lse"][:B])                      # the token's row maximum is exact
    assert np.abs(a["lse"][B:][fin] - b["lse"][B:][fin]).max() < 1e-5
    for k in ("loss", "KL"):
        assert rel(a[k][fin], b[k][fin]) < 2e-7, k
    if fin.all():
        assert rel(a["means"], b["means"]) < 1e-6


@pytest.mark.parametrize("B,C,zd,masked,limit", [(100, 25000, 40, True, None), (100, 25000, 40, False, None), (100, 200, 40, True, None),
                                                 (128, 30720, 40, True, None), (1, 1, 40, False, None), (7, 129, 8, True, None),
                                                 (100, 3125, 56, True, None), (100, 1000, 40, True, 0.0), (33, 11500, 40, True, None),
                                                 (64, 1000, 4, False, 0.0)])
def test_prior_of_a_training_step_in_one_launch(ops, B, C, zd, masked, limit):
    """evae_prior_train_step (forward partials, two-level last-arriver merge, backward on the products each block holds) against
    the three-launch path it replaces (evae_prior_lse_fwd + evae_prior_merge + evae_prior_lse_bwd) and against float64; launched
    repeatedly with CHANGING inputs on the same buffers, so that a stale token / partial row from the previous launch
    (another XCD's L2, this CU's L1) cannot pass; limit = 0 sends every tile through the direct-difference path"""
    import ctypes as C_
    lib = ops._lib.load()
    assert ops.prior_train_applies(B, C, zd)
    if limit is not None:
        lib.evae_prior_set_norm_limit(C_.c_float(limit))
    try:
        lv = np.linspace(-1.5, -0.5, zd).astype(np.float32)
        beta = 0.73
        out = None
        for it in range(6):
            z, c = gi.clustered_latents(500 + 17 * it + B + C, B, C, zd)
            z = (z * (1.0 + 0.3 * it)).astype(np.float32)
            zi, ci = gi.mask_indices(7 + B + it, B, C, max(C // 2, 4))
            if masked and C > 8 and it > 0:
                ci[6] = -3                                            # a slot masked for every query (EVAE_PRIOR_MASK_ALL)
            dz_, dc_, dlv_ = dev(z), dev(c), dev(lv)
            zi_ = dev(zi) if masked else None; ci_ = dev(ci) if masked else None
            out = ops.prior_train_step(dz_, dc_, dlv_, zi_, ci_, C, beta, out=out)
            logp, token, coef, gz, gc, glv = out
            m, s, n, _ = ops.prior_lse_fwd(dz_, dc_, dlv_, zi_, ci_)
            lp_ref, tok_ref = ops.prior_merge(m, s, n, C)
            g = torch.full((B,), -beta / B, device="cuda")
            rz, rc, rlv = ops.prior_lse_bwd(dz_, dc_, dlv_, zi_, ci_, tok_ref, g)
            torch.cuda.synchronize()
            fin = torch.isfinite(lp_ref).cpu().numpy()
            assert np.array_equal(fin, torch.isfinite(logp).cpu().numpy())
            assert rel(logp.cpu().numpy()[fin], lp_ref.cpu().numpy()[fin]) < 3e-7, it
            assert np.array_equal(token[0].cpu().numpy(), tok_ref[0].cpu().numpy()), it             # row maxima: exact
            assert np.abs(token[1].cpu().numpy()[fin] - tok_ref[1].cpu().numpy()[fin]).max() < 2e-5, it
            for a, b, name in ((gz, rz, "dz"), (gc, rc, "dcentres"), (glv, rlv, "dlogvar")):
                assert rel(a.cpu().numpy(), b.cpu().numpy()) < 3e-5, (name, it)
            assert np.array_equal(coef[0].cpu().numpy(), np.full(B, -1.0 / B, np.float32))
            assert np.allclose(coef[1].cpu().numpy(), beta / B) and np.allclose(coef[2].cpu().numpy(), -beta / B)
            assert int(ops.prior_train_state(torch.device("cuda", 0))[10]) == 0                     # no block's wait ran out
            if it == 0 and B * C <= 4_000_000 and limit is None:
                # float64 restatement (oracle primitives): log p and its gradient through the mean-of-batch coefficient
                lp64 = orc.logsumexp_rows(orc.log_p_z_exemplar(z, zi, c, lv[None, :], ci, test=not masked))
                ok = np.isfinite(lp64)
                assert rel(logp.cpu().numpy()[ok], lp64[ok]) < 1e-5
    finally:
        if limit is not None:
            lib.evae_prior_set_norm_limit(C_.c_float(-1.0))


@pytest.mark.parametrize("B,C,N,zd", [(100, 25000, 50000, 40), (100, 4000, 4000, 40), (7, 300, 150, 8), (128, 11500, 23000, 40), (33, 129, 129, 56)])
def test_prior_of_a_training_step_over_the_draws_of_distinct_rows(ops, B, C, N, zd):
    """evae_prior_train_step_rows (r06): the one-launch prior of a step that encoded each DISTINCT image of its draw once (C draws with
    replacement from N images; rows / inv / rep / mult from evae_host_dedup, distinct rows padded with multiplicity 0) -- log p, token,
    dz, dlogvar BIT-equal to evae_prior_train_step over the gathered per-draw centres (same kernel, the centres read through the row
    map), and its folded centre gradients equal to multiplicity x the per-draw gradient of the representative draw, padding rows zero."""
    import ctypes as C_
    lib = ops._lib.load()
    rs = np.random.RandomState(C + B)
    draws = torch.from_numpy(rs.randint(0, N, size=C).astype(np.int64))
    cap = (C + 7) // 8 * 8
    rows = torch.zeros(cap, dtype=torch.int64); inv = torch.zeros(C, dtype=torch.int64); rep = torch.zeros(cap, dtype=torch.int64)
    mult = torch.zeros(cap, dtype=torch.float32)
    nu = lib.evae_host_dedup(C_.c_void_p(draws.data_ptr()), C, N, cap, C_.c_void_p(rows.data_ptr()), C_.c_void_p(inv.data_ptr()),
                             C_.c_void_p(rep.data_ptr()), C_.c_void_p(mult.data_ptr()))
    assert 0 < nu <= C
    Cd = (nu + 7) // 8 * 8
    z, cu = gi.clustered_latents(900 + B + C, B, Cd, zd)                 # encodings of the distinct rows (padding rows: anything)
    lv = np.linspace(-1.2, -0.4, zd).astype(np.float32)
    zi = rs.randint(0, N, size=B).astype(np.int64)
    zi[:3] = draws.numpy()[:3]                                           # leave-one-out hits on the draws
    dz_, dcu, dlv_ = dev(z), dev(cu), dev(lv)
    inv_d, rep_d, mult_d, ci_d, zi_d = inv.cuda(), rep[:Cd].cuda(), mult[:Cd].cuda(), draws.cuda(), dev(zi)
    f = dict(device="cuda", dtype=torch.float32)
    beta = 0.61
    for it in range(3):                                                  # repeated launches on the same buffers (stale tokens cannot pass)
        out = (torch.empty(B, **f), torch.empty((2, B), **f), None, torch.empty((B, zd), **f), torch.full((Cd, zd), 7.0, **f), torch.empty(zd, **f))
        dc_draws = torch.empty((C, zd), **f)
        ops.prior_train_step_rows(dz_, dcu, (inv_d, rep_d, mult_d), dlv_, zi_d, ci_d, C, beta, out, dc_draws)
        cx = dcu[inv_d].contiguous()                                     # every draw's centre
        ref = ops.prior_train_step(dz_, cx, dlv_, zi_d, ci_d, C, beta, want_coef=False)
        torch.cuda.synchronize()
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), it
        assert torch.equal(out[3], ref[3]) and torch.equal(out[5], ref[5]), it
        assert torch.equal(dc_draws, ref[4]), it
        fold = mult_d[:, None] * ref[4][rep_d]
        assert torch.equal(out[4], fold), it
        assert float(out[4][nu:].abs().max()) == 0.0 if Cd > nu else True
        dz_ = dz_ * 1.01


@pytest.mark.parametrize("M,N,K,ldd", [(100, 300, 300, 300), (100, 300, 40, 600), (100, 300, 784, 300), (100, 784, 600, 784),
                                       (7, 20, 36, 20), (128, 296, 588, 296), (1438, 300, 300, 300), (3000, 300, 784, 300)])
def test_gated_backward_with_the_gate_derivative_in_the_operand_load(ops, M, N, K, ldd):
    """evae_gated_dense_bwd (r04): a gated layer's backward wrt its input -- [dh | dg] = (dout s, dout (h s)(1 - s)) and
    dx = dh Wh + dg Wg, reference utils/nn.py:62-68 under autograd -- as ONE launch for batch-sized row counts (the gate
    derivative formed in the data gradient's operand load) against the two-launch form (evae_gated_dense_bwd_input_ld +
    evae_dense_bwd_data) bit for bit, and both against float64; dout may be a column block of a wider gradient (ldd > N: one
    half of a torch.cat's).  The last shape is past the thin kernels' row bound: the entry point takes the two launches itself."""
    from evae import _lib
    lib = _lib.load()
    rs = np.random.RandomState(M + N + K)
    wide = dev((rs.standard_normal((M, ldd)) * 0.1).astype(np.float32))
    dout = wide[:, ldd - N:]                                         # (16-byte aligned: N and ldd are multiples of 4)
    h = dev(rs.standard_normal((M, N)).astype(np.float32))
    s = torch.sigmoid(dev(rs.standard_normal((M, N)).astype(np.float32)))
    gout = h * s
    wh = dev((rs.standard_normal((N, K)) * 0.05).astype(np.float32)); wg = dev((rs.standard_normal((N, K)) * 0.05).astype(np.float32))
    st = ops._stream()
    nb = lib.evae_dense_bwd_data_workspace_bytes(M, N, K, 2)
    ws = torch.zeros(max(nb, 256), dtype=torch.uint8, device="cuda")
    dpre = torch.full((M, 2 * N), float("nan"), device="cuda"); dx = torch.full((M, K), float("nan"), device="cuda")
    _lib.check(lib.evae_gated_dense_bwd(ops._p(dout), ldd, ops._p(gout), ops._p(s), M, N, ops._p(wh), ops._p(wg), K, ops._p(dpre), 2 * N,
                                        ops._p(dx), K, ops._p(ws), ws.numel(), st), "gated_dense_bwd")
    dpre2 = torch.empty((M, 2 * N), device="cuda"); dx2 = torch.empty((M, K), device="cuda")
    base = dpre2.data_ptr()
    _lib.check(lib.evae_gated_dense_bwd_input_ld(ops._p(dout), ldd, ops._p(gout), ops._p(s), M, N, ops._vp(base), ops._vp(base + 4 * N), 2 * N,
                                                 st), "bwd_input_ld")
    _lib.check(lib.evae_dense_bwd_data(ops._vp(base), ops._p(wh), ops._vp(base + 4 * N), ops._p(wg), M, N, 2 * N, K, None, None, ops._p(dx2),
                                       None, K, ops._p(ws), ws.numel(), st), "bwd_data")
    assert torch.equal(dpre, dpre2)
    assert torch.equal(dx, dx2)
    d64, s64, g64 = dout.double(), s.double(), gout.double()
    dh, dg = d64 * s64, d64 * g64 * (1 - s64)
    ref = dh @ wh.double() + dg @ wg.double()
    assert rel(dx.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    assert rel(dpre.cpu().numpy(), torch.cat((dh, dg), 1).cpu().numpy()) < 1e-6
    # through autograd: the modular layer's gradients against float64 autograd
    x = dev(rs.standard_normal((M, K)).astype(np.float32)).requires_grad_(True)
    bh = dev(np.zeros(N, np.float32)); bg = dev(np.zeros(N, np.float32))
    whp, wgp = wh.clone().requires_grad_(True), wg.clone().requires_grad_(True)
    y = ops.GatedDenseFn.apply(x, None, whp, bh, wgp, bg)
    y.backward(dout)
    x6 = x.detach().double().requires_grad_(True); wh6 = wh.double().requires_grad_(True); wg6 = wg.double().requires_grad_(True)
    y6 = (x6 @ wh6.t()) * torch.sigmoid(x6 @ wg6.t())
    y6.backward(dout.double())
    for got, ref_ in ((x.grad, x6.grad), (whp.grad, wh6.grad), (wgp.grad, wg6.grad)):
        assert rel(got.cpu().numpy(), ref_.cpu().numpy()) < 3e-6
