"""Pins the oracle (oracle/evae_oracle.py) to golden vectors produced by the real reference
(tools/gen_goldens.py).  CPU only.  Tolerances: fp32 summation-order noise only."""
import numpy as np

import evae_oracle as orc
import golden_inputs as gi


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_g1_pairwise_distance(golden):
    g = golden("g1_g2_distance")
    for zdim in (40, 256):
        z, m = gi.latents(11 + zdim, 16, 257, zdim)
        pd = orc.pairwise_distance(z, m)
        ref = g["pd_z%d" % zdim]
        # fp64 accumulate + one rounding: at most 1 ulp apart (BLAS summation order in fp64)
        assert np.abs(pd - ref).max() <= np.spacing(np.abs(ref).max())
        assert (pd == ref).mean() > 0.999
        assert np.array_equal(orc.pairdist_direct_f64(z, m) == ref, np.ones_like(ref, bool)) or \
            (orc.pairdist_direct_f64(z, m) == ref).mean() > 0.999


def test_g2_log_normal_diag_vectorized(golden):
    g = golden("g1_g2_distance")
    for zdim in (40, 256):
        z, m = gi.latents(11 + zdim, 16, 257, zdim)
        for p, key in ((-1.0, "m1_0"), (0.3, "0_3")):
            ln, _ = orc.log_normal_diag_vectorized(z, m, np.full((1, zdim), p, np.float32))
            assert rel(ln, g["ln_z%d_p%s" % (zdim, key)]) < 1e-6


def _prior_case(tag):
    B, C, N, seed = {"small": (8, 300, 120, 21), "c2": (100, 25000, 50000, 22)}[tag]
    z, c = gi.clustered_latents(seed, B, C, 40)
    zi, ci = gi.mask_indices(seed + 1, B, C, N)
    gout = np.random.RandomState(seed + 2).standard_normal(B).astype(np.float32)
    return z, c, zi, ci, gout, np.float32(-1.3)


def test_g3_prior_forward_and_grads_small(golden):
    g = golden("g3_prior")
    z, c, zi, ci, gout, plv = _prior_case("small")
    lv = np.full((len(c), 40), plv, np.float32)
    for mode in ("train", "test"):
        test = mode == "test"
        prob = orc.log_p_z_exemplar(z, zi, c, lv, ci, test)
        ref = g["small_%s_prob" % mode]
        assert np.array_equal(np.isinf(prob), np.isinf(ref))
        fin = np.isfinite(ref)
        assert rel(prob[fin], ref[fin]) < 1e-6
        lp = orc.logsumexp_rows(prob)
        assert rel(lp, g["small_%s_logp" % mode]) < 1e-6
        dz, dc, dlv, _ = orc.prior_grads(z, zi, c, lv[0], ci, not test, gout)
        assert rel(dz, g["small_%s_dz" % mode]) < 2e-5
        assert rel(dc, g["small_%s_dc" % mode]) < 2e-5
        assert rel(dlv.sum(), g["small_%s_dplv" % mode]) < 2e-5
        # shard form: 3 uneven shards (one empty) merge to the same log-prior
        cuts = [0, 0, 101, 300]
        parts = [orc.prior_partials(z, zi, c[a:b], lv[0], ci[a:b], not test) for a, b in zip(cuts[:-1], cuts[1:])]
        merged = orc.prior_merge([p[0] for p in parts], [p[1] for p in parts], [p[2] for p in parts], len(c))
        assert rel(merged, g["small_%s_logp" % mode]) < 1e-6


def test_g3_prior_c2_size(golden):
    g = golden("g3_prior")
    z, c, zi, ci, gout, plv = _prior_case("c2")
    lv = np.full((1, 40), plv, np.float32)
    for mode in ("train", "test"):
        test = mode == "test"
        lp = orc.log_p_z(z, zi, c, lv, ci, test)
        assert rel(lp, g["c2_%s_logp" % mode]) < 1e-6
        z64, c64 = z.astype(np.float64), c.astype(np.float64)
        dz, dc, dlv, _ = orc.prior_grads(z64, zi, c64, lv[0].astype(np.float64), ci, not test, gout.astype(np.float64))
        assert rel(dz, g["c2_%s_dz" % mode]) < 1e-4
        assert rel(dlv.sum(), g["c2_%s_dplv" % mode]) < 1e-4
        assert rel(dc[:64], g["c2_%s_dc_head" % mode]) < 1e-4
        assert rel(dc.sum(0), g["c2_%s_dc_colsum" % mode]) < 1e-4
        assert rel(np.linalg.norm(dc, axis=1), g["c2_%s_dc_rownorm" % mode]) < 1e-4


def test_g4_topk_indices_bit_exact(golden):
    g = golden("g4_topk")
    for tag, (B, C, zdim, seed) in {"c2": (100, 25000, 40, 31), "c5": (64, 100000, 256, 32)}.items():
        z, c = gi.clustered_latents(seed, B, C, zdim)
        vals, idx = orc.nearest_exemplars_topk(z, c, 10)
        assert np.array_equal(idx, g[tag + "_idx"].astype(np.int64))
        assert g[tag + "_gap"][0] > 0
        # the exact-arithmetic direct form used by the HIP kernel selects the same indices
        v2, i2 = orc.topk_smallest(orc.pairdist_direct_f64(z, c), 10)
        assert np.array_equal(i2, idx)


def test_g5_find_nearest_neighbors_bit_exact(golden):
    g = golden("g5_knn")
    zv, zt = gi.clustered_latents(41, 100, 60000, 40)
    idx = orc.find_nearest_neighbors(zv, zt)
    assert np.array_equal(idx, g["idx"].astype(np.int64))


def test_g6_layers(golden):
    g = golden("g6_layers")
    rs = np.random.RandomState(51)
    R, I, O = 37, 53, 24
    x = rs.standard_normal((R, I)).astype(np.float32)
    wh = (rs.standard_normal((O, I)) * 0.2).astype(np.float32); bh = (rs.standard_normal(O) * 0.1).astype(np.float32)
    wg = (rs.standard_normal((O, I)) * 0.2).astype(np.float32); bg = (rs.standard_normal(O) * 0.1).astype(np.float32)
    gout = rs.standard_normal((R, O)).astype(np.float32)
    y, saved = orc.gated_dense(x, wh, bh, wg, bg)
    assert rel(y, g["gd_y"]) < 2e-6
    dx, gr = orc.gated_dense_bwd(x, wh, wg, saved, gout)
    assert rel(dx, g["gd_dx"]) < 1e-5
    for k, n in (("wh", "gd_dwh"), ("bh", "gd_dbh"), ("wg", "gd_dwg"), ("bg", "gd_dbg")):
        assert rel(gr[k], g[n]) < 1e-5
    w8 = wh * 8
    pre = orc.linear(x, w8, bh)
    assert rel(orc.sigmoid(pre), g["nl_sigmoid_y"]) < 2e-6
    assert rel(orc.hardtanh(pre, -6, 2), g["nl_hardtanh_y"]) < 2e-6
    assert rel(pre, g["nl_none_y"]) < 2e-6
    dpre = gout * ((pre > -6) & (pre < 2))
    assert rel(dpre @ w8, g["nl_hardtanh_dx"]) < 1e-5
    assert rel(dpre.T @ x, g["nl_hardtanh_dw"]) < 1e-5
    xm = 1 / (1 + np.exp(-rs.standard_normal((R, I)) * 6)).astype(np.float32)
    xb = (rs.random_sample((R, I)) < 0.3).astype(np.float32)
    assert rel(orc.log_bernoulli(xb, xm.astype(np.float32)), g["log_bernoulli"]) < 1e-6
    mu = rs.standard_normal((R, I)).astype(np.float32); lv = rs.uniform(-6, 2, (R, I)).astype(np.float32)
    assert rel(orc.log_normal_diag(x, mu, lv), g["log_normal_diag"]) < 1e-6
    xc = ((rs.randint(0, 256, (R, I)) + 0.5) / 256).astype(np.float32)
    mc = rs.uniform(1 / 512., 1 - 1 / 512., (R, I)).astype(np.float32)
    ls = rs.uniform(-4.5, 0, (R, I)).astype(np.float32)
    assert rel(orc.log_logistic_256(xc, mc, ls), g["log_logistic_256"]) < 1e-5


def vae_case(tag):
    B, C, N, seed = {"small": (16, 200, 500, 61), "c1": (100, 1000, 4000, 62)}[tag]
    p = orc.vae_init_params(np.random.RandomState(123))
    data = gi.gray_images(seed, N)
    rs = np.random.RandomState(seed + 1)
    bidx = rs.randint(0, N, size=(B, 1)).astype(np.int64)
    x = (rs.random_sample((B, 784)) < np.clip(data[bidx[:, 0]] + 0.1, 0, 1)).astype(np.float32)
    eps = rs.standard_normal((B, 40)).astype(np.float32)
    ex_idx = rs.randint(0, N, size=(C,)).astype(np.int64)
    ex_idx[:3] = bidx[:3, 0]
    return p, data, bidx, x, eps, ex_idx


def test_g7_vae_calculate_loss_and_grads(golden):
    g = golden("g7_vae_loss")
    for tag in ("small", "c1"):
        p, data, bidx, x, eps, ex_idx = vae_case(tag)
        fwd = orc.vae_calculate_loss(p, x, bidx, eps, ("images", data[ex_idx], ex_idx), beta=0.37)
        for k in ("loss", "RE", "KL"):
            assert rel(fwd[k], g["%s_train_%s" % (tag, k)]) < 1e-5, (tag, k)
        grads = orc.vae_loss_backward(p, x, bidx, eps, fwd, beta=0.37)
        for name in orc.VAE_PARAM_NAMES:
            ref_norm = g["%s_gnorm_%s" % (tag, name)][0]
            got = np.linalg.norm(grads[name].astype(np.float64))
            assert abs(got - ref_norm) <= 2e-4 * max(ref_norm, 1e-6), (tag, name, got, ref_norm)
            assert rel(grads[name].reshape(-1)[:16], g["%s_ghead_%s" % (tag, name)]) < 5e-4, (tag, name)
        # evaluation: cache of the whole dataset as the embedding, no mask
        cz, clv, _ = orc.vae_q_z(p, data, prior=True)
        assert rel(cz[:32], g[tag + "_cache_head"]) < 1e-5
        ev = orc.vae_calculate_loss(p, x, None, eps, ("embedding", cz, clv, np.arange(len(cz))), training=False)
        for k in ("loss", "RE", "KL"):
            assert rel(ev[k], g["%s_eval_%s" % (tag, k)]) < 1e-5, (tag, k)


def test_g8_adam_normgrad(golden):
    g = golden("g8_adam")
    for i in range(4):
        p = g["p0_%d" % i]
        m = np.zeros_like(p); v = np.zeros_like(p)
        for step in range(3):
            p, m, v = orc.adam_normgrad_step(p, g["g%d_%d" % (step, i)], m, v, step + 1)
            assert rel(p, g["p%d_%d" % (step + 1, i)]) < 1e-6
        assert rel(m, g["m_%d" % i]) < 1e-6 and rel(v, g["v_%d" % i]) < 1e-6
