#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (read-only at /root/reference) on CPU.

Run in the build container only:   python tools/gen_goldens.py
The reference never travels to the GPU box; only the small .npz outputs under tests/golden/ do.
Inputs come from tests/golden_inputs.py (numpy seeds), so tests regenerate them instead of
storing them.  Nothing from the reference is copied: this script calls its functions.
"""
import os
import sys
import types
from argparse import Namespace

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, os.path.join(ROOT, "tests"))
sys.path.insert(2, os.path.join(ROOT, "oracle"))
for name in ("torchvision", "torchvision.datasets", "wget"):      # only the dataset loader needs them
    sys.modules.setdefault(name, types.ModuleType(name))

import numpy as np
import torch

import golden_inputs as gi
import evae_oracle as orc

from utils.distributions import pairwise_distance, log_normal_diag_vectorized           # noqa: E402
from utils.distributions import log_normal_diag, log_bernoulli, log_logistic_256         # noqa: E402
from utils.knn_on_latent import find_nearest_neighbors                                   # noqa: E402
from utils.nn import GatedDense, NonLinear                                               # noqa: E402
from utils.optimizer import AdamNormGrad                                                 # noqa: E402
from models.VAE import VAE                                                               # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(os.cpu_count())
T = torch.from_numpy


def vae_args(**kw):
    a = dict(prior="exemplar_prior", input_type="binary", input_size=[1, 28, 28], hidden_size=300,
             z1_size=40, z2_size=40, model_name="vae", device="cpu", number_components=1000,
             training_set_size=50000, approximate_prior=False, approximate_k=10, no_mask=False,
             no_attention=False, same_variational_var=False, use_logit=False, lambd=1e-4,
             bottleneck=6, dataset_name="dynamic_mnist", continuous=False)
    a.update(kw)
    return Namespace(**a)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def tie_gap(dist, k):
    """min over rows of the gap between consecutive order statistics 1..k+1 (relative)."""
    part = np.sort(dist, axis=1)[:, :k + 1]
    gaps = np.diff(part, axis=1)
    return float(gaps.min()), float((gaps / np.maximum(part[:, 1:], 1e-30)).min())


# ---- G1 / G2 ------------------------------------------------------------------------------------
def g1_g2():
    out = {}
    for zdim in (40, 256):
        z, m = gi.latents(11 + zdim, 16, 257, zdim)
        out["pd_z%d" % zdim] = pairwise_distance(T(z), T(m)).numpy()
        for p in (-1.0, 0.3):
            lv = torch.full((1, zdim), p)
            ln, _ = log_normal_diag_vectorized(T(z), T(m), lv)
            out["ln_z%d_p%s" % (zdim, str(p).replace("-", "m").replace(".", "_"))] = ln.numpy()
    save("g1_g2_distance", **out)


# ---- G3: prior with mask / test / sum=False, and gradients -----------------------------------------
def g3():
    model = VAE(vae_args())
    out = {}
    for tag, (B, C, N, seed) in {"small": (8, 300, 120, 21), "c2": (100, 25000, 50000, 22)}.items():
        z_np, c_np = gi.clustered_latents(seed, B, C, 40)
        zi_np, ci_np = gi.mask_indices(seed + 1, B, C, N)
        gout = np.random.RandomState(seed + 2).standard_normal(B).astype(np.float32)
        plv0 = -1.3
        for mode in ("train", "test"):
            z = T(z_np).clone().requires_grad_(True)
            c = T(c_np).clone().requires_grad_(True)
            plv = torch.tensor([plv0], requires_grad=True)
            logvar = plv * torch.ones((C, 40))
            model.train(mode == "train")
            lp = model.log_p_z((z, T(zi_np)), (c, logvar, T(ci_np)))
            (lp * T(gout)).sum().backward()
            out["%s_%s_logp" % (tag, mode)] = lp.detach().numpy()
            out["%s_%s_dz" % (tag, mode)] = z.grad.numpy()
            out["%s_%s_dplv" % (tag, mode)] = plv.grad.numpy()
            if tag == "small":
                out["%s_%s_dc" % (tag, mode)] = c.grad.numpy()
                with torch.no_grad():
                    out["%s_%s_prob" % (tag, mode)] = model.log_p_z(
                        (T(z_np), T(zi_np)), (T(c_np), logvar.detach(), T(ci_np)), sum=False).numpy()
            else:
                out["%s_%s_dc_head" % (tag, mode)] = c.grad.numpy()[:64]
                out["%s_%s_dc_colsum" % (tag, mode)] = c.grad.double().sum(0).numpy()
                out["%s_%s_dc_rownorm" % (tag, mode)] = c.grad.double().norm(dim=1).numpy().astype(np.float32)
    save("g3_prior", **out)


# ---- G4: distance + top-k (BaseModel.py:263-264) ---------------------------------------------------
def g4():
    out = {}
    for tag, (B, C, zdim, seed) in {"c2": (100, 25000, 40, 31), "c5": (64, 100000, 256, 32)}.items():
        z, c = gi.clustered_latents(seed, B, C, zdim)
        d = pairwise_distance(T(z), T(c))
        vals, idx = d.topk(k=10, largest=False, dim=1)
        gap_abs, gap_rel = tie_gap(d.numpy(), 10)
        assert gap_abs > 0, "fp32 tie inside top-(k+1): regenerate with another seed"
        out[tag + "_idx"] = idx.numpy().astype(np.int32)
        out[tag + "_val"] = vals.numpy()
        out[tag + "_gap"] = np.asarray([gap_abs, gap_rel])
        # cross-check: the build's tie rule equals torch.topk on tie-free rows
        ov, oi = orc.topk_smallest(d.numpy(), 10)
        assert np.array_equal(oi, idx.numpy())
    save("g4_topk", **out)


# ---- G5: find_nearest_neighbors (knn_on_latent.py:4-9) ----------------------------------------------
def g5():
    zv, zt = gi.clustered_latents(41, 100, 60000, 40)
    idx = find_nearest_neighbors(T(zv), T(zt), None).numpy()
    dist = np.sqrt(orc.pairdist_direct_f64(zv, zt))
    gap_abs, gap_rel = tie_gap(dist, 20)
    assert gap_abs > 0
    save("g5_knn", idx=idx.astype(np.int32), gap=np.asarray([gap_abs, gap_rel]))


# ---- G6: dense layers (utils/nn.py:29-69) -----------------------------------------------------------
def g6():
    rs = np.random.RandomState(51)
    R, I, O = 37, 53, 24
    x = rs.standard_normal((R, I)).astype(np.float32)
    wh = (rs.standard_normal((O, I)) * 0.2).astype(np.float32); bh = (rs.standard_normal(O) * 0.1).astype(np.float32)
    wg = (rs.standard_normal((O, I)) * 0.2).astype(np.float32); bg = (rs.standard_normal(O) * 0.1).astype(np.float32)
    gout = rs.standard_normal((R, O)).astype(np.float32)
    gd = GatedDense(I, O)
    gd.load_state_dict({"h.weight": T(wh), "h.bias": T(bh), "g.weight": T(wg), "g.bias": T(bg)})
    xt = T(x).clone().requires_grad_(True)
    y = gd(xt)
    y.backward(T(gout))
    out = dict(gd_y=y.detach().numpy(), gd_dx=xt.grad.numpy(), gd_dwh=gd.h.weight.grad.numpy(),
               gd_dbh=gd.h.bias.grad.numpy(), gd_dwg=gd.g.weight.grad.numpy(), gd_dbg=gd.g.bias.grad.numpy())
    for name, act in (("sigmoid", torch.nn.Sigmoid()), ("hardtanh", torch.nn.Hardtanh(-6., 2.)), ("none", None)):
        nl = NonLinear(I, O, activation=act)
        nl.load_state_dict({"linear.weight": T(wh * 8), "linear.bias": T(bh)})
        xt = T(x).clone().requires_grad_(True)
        y = nl(xt)
        y.backward(T(gout))
        out["nl_%s_y" % name] = y.detach().numpy()
        out["nl_%s_dx" % name] = xt.grad.numpy()
        out["nl_%s_dw" % name] = nl.linear.weight.grad.numpy()
        out["nl_%s_db" % name] = nl.linear.bias.grad.numpy()
    # reconstruction / density terms
    xm = 1 / (1 + np.exp(-rs.standard_normal((R, I)) * 6)).astype(np.float32)
    xb = (rs.random_sample((R, I)) < 0.3).astype(np.float32)
    out["log_bernoulli"] = log_bernoulli(T(xb), T(xm.astype(np.float32)), dim=1).numpy()
    mu = rs.standard_normal((R, I)).astype(np.float32); lv = rs.uniform(-6, 2, (R, I)).astype(np.float32)
    out["log_normal_diag"] = log_normal_diag(T(x), T(mu), T(lv), dim=1).numpy()
    xc = ((rs.randint(0, 256, (R, I)) + 0.5) / 256).astype(np.float32)
    mc = rs.uniform(1 / 512., 1 - 1 / 512., (R, I)).astype(np.float32)
    ls = rs.uniform(-4.5, 0, (R, I)).astype(np.float32)
    out["log_logistic_256"] = log_logistic_256(T(xc), T(mc), T(ls), dim=1).numpy()
    save("g6_layers", **out)


# ---- G6b: the convolution modules, utils/nn.py:72-114 (GatedConv2d, Conv2d), forward + all gradients -------------
G6_CONV_CASES = [   # (kind, Cin, Cout, k, stride, pad, H, W, N, activation)
    ("gated", 1, 32, 7, 1, 3, 28, 28, 3, None), ("gated", 32, 32, 3, 2, 1, 28, 28, 2, None),
    ("gated", 32, 64, 5, 1, 2, 14, 14, 2, None), ("gated", 64, 6, 3, 1, 1, 7, 7, 3, None),
    ("gated", 3, 32, 3, 2, 1, 16, 12, 2, "elu"),
    ("plain", 64, 1, 1, 1, 0, 28, 28, 2, "sigmoid"), ("plain", 64, 3, 1, 1, 0, 16, 16, 2, "hardtanh"),
    ("plain", 32, 48, 3, 1, 1, 9, 9, 2, None),
]


def g6_conv():
    from utils.nn import GatedConv2d, Conv2d
    acts = {None: None, "elu": torch.nn.ELU(), "sigmoid": torch.nn.Sigmoid(), "hardtanh": torch.nn.Hardtanh(-4.5, 0.)}
    out = {}
    for i, (kind, ci, co, k, st, pd, H, W, N, act) in enumerate(G6_CONV_CASES):
        rs = np.random.RandomState(600 + i)
        x = rs.standard_normal((N, ci, H, W)).astype(np.float32)
        sc = 1.0 / np.sqrt(ci * k * k)
        wh = (rs.standard_normal((co, ci, k, k)) * sc).astype(np.float32); bh = (rs.standard_normal(co) * 0.1).astype(np.float32)
        wg = (rs.standard_normal((co, ci, k, k)) * sc).astype(np.float32); bg = (rs.standard_normal(co) * 0.1).astype(np.float32)
        if kind == "gated":
            m = GatedConv2d(ci, co, k, st, pd, activation=acts[act])
            m.load_state_dict({"h.weight": T(wh), "h.bias": T(bh), "g.weight": T(wg), "g.bias": T(bg)})
        else:
            m = Conv2d(ci, co, k, st, pd, activation=acts[act])
            m.load_state_dict({"conv.weight": T(wh), "conv.bias": T(bh)})
        xt = T(x).clone().requires_grad_(True)
        y = m(xt)
        gout = rs.standard_normal(tuple(y.shape)).astype(np.float32)
        y.backward(T(gout))
        # big tensors are kept as every 5th element + their fp64 L2 norm (fixtures stay small)
        for key, arr in (("y", y.detach().numpy()), ("dx", xt.grad.numpy())):
            out["c%d_%s" % (i, key)] = arr.reshape(-1)[::5].copy()
            out["c%d_%s_norm" % (i, key)] = np.asarray(np.linalg.norm(arr.astype(np.float64)))
        for name, prm in m.named_parameters():
            out["c%d_d_%s" % (i, name)] = prm.grad.numpy()
    save("g6_conv_layers", **out)


# ---- G7: VAE.calculate_loss, train (exact prior) and eval, with injected eps / exemplar indices -----
def load_params(model, p):
    model.load_state_dict({k: T(v.copy()) for k, v in p.items()})


def g7():
    out = {}
    for tag, (B, C, N, seed) in {"small": (16, 200, 500, 61), "c1": (100, 1000, 4000, 62)}.items():
        args = vae_args(number_components=C, training_set_size=N)
        model = VAE(args)
        p = orc.vae_init_params(np.random.RandomState(123))
        load_params(model, p)
        data = gi.gray_images(seed, N).astype(np.float32)            # dataset tensor (un-binarised)
        rs = np.random.RandomState(seed + 1)
        bidx = rs.randint(0, N, size=(B, 1)).astype(np.int64)
        x = (rs.random_sample((B, 784)) < np.clip(data[bidx[:, 0]] + 0.1, 0, 1)).astype(np.float32)
        eps = rs.standard_normal((B, 40)).astype(np.float32)
        ex_idx = rs.randint(0, N, size=(C,)).astype(np.int64)
        ex_idx[:3] = bidx[:3, 0]                                       # force some leave-one-out hits
        dataset = torch.utils.data.TensorDataset(T(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
        model.reparameterize = lambda mu, logvar: T(eps) * logvar.mul(0.5).exp() + mu
        orig_randint = torch.randint
        torch.randint = lambda low=0, high=None, size=None, **kw: T(ex_idx)
        try:
            model.train()
            model.zero_grad()
            loss, RE, KL = model.calculate_loss((T(x), T(bidx)), beta=0.37, average=False, dataset=dataset)
            loss.mean().backward()
        finally:
            torch.randint = orig_randint
        out[tag + "_train_loss"] = loss.detach().numpy()
        out[tag + "_train_RE"] = RE.detach().numpy()
        out[tag + "_train_KL"] = KL.detach().numpy()
        for k, v in model.named_parameters():
            g = v.grad
            out[tag + "_gnorm_" + k] = np.asarray([0.0 if g is None else g.double().norm().item()])
            if g is not None:
                out[tag + "_ghead_" + k] = g.reshape(-1)[:16].numpy().copy()
        # evaluation mode: embedding = cache of the whole dataset, no mask (evaluation.py:15,26,56-59)
        model.eval()
        with torch.no_grad():
            cz, clv = model.cache_z(dataset)
            emb = (cz, clv, torch.arange(len(cz)))
            loss, RE, KL = model.calculate_loss((T(x), None), average=False, exemplars_embedding=emb)
        out[tag + "_eval_loss"] = loss.numpy(); out[tag + "_eval_RE"] = RE.numpy(); out[tag + "_eval_KL"] = KL.numpy()
        out[tag + "_cache_head"] = cz.numpy()[:32]
    save("g7_vae_loss", **out)


# ---- G8: AdamNormGrad, 3 steps (optimizer.py:32-80) -------------------------------------------------
def g8():
    import warnings
    rs = np.random.RandomState(71)
    shapes = [(5, 7), (7,), (1,), (3, 2, 2)]
    ps = [torch.nn.Parameter(T(rs.standard_normal(s).astype(np.float32))) for s in shapes]
    grads = [[rs.standard_normal(s).astype(np.float32) * (10.0 ** (i - 1)) for s in shapes] for i in range(3)]
    opt = AdamNormGrad(ps, lr=5e-4)
    out = {}
    for i, s in enumerate(shapes):
        out["p0_%d" % i] = ps[i].detach().numpy().copy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for step in range(3):
            for p_, g in zip(ps, grads[step]):
                p_.grad = T(g.copy())
            opt.step()
            for i, p_ in enumerate(ps):
                out["g%d_%d" % (step, i)] = grads[step][i]
                out["p%d_%d" % (step + 1, i)] = p_.detach().numpy().copy()
    for i, p_ in enumerate(ps):
        out["m_%d" % i] = opt.state[p_]["exp_avg"].numpy()
        out["v_%d" % i] = opt.state[p_]["exp_avg_sq"].numpy()
    save("g8_adam", **out)


# ---- G9: calculate_loss of the other architectures (hvae_2level, convhvae_2level, single_conv) -------------
def seeded_state_dict(model, seed, gain=1.0):
    """Deterministic weights for any architecture: walk the state_dict in order and draw from one RandomState
    (scaled like He-init for matrices / filters; weight-norm g kept positive; BatchNorm buffers untouched)."""
    rs = np.random.RandomState(seed)
    sd = {}
    for k, v in model.state_dict().items():
        shp = tuple(v.shape)
        if "normalization" in k or k.endswith("num_batches_tracked"):
            sd[k] = v.clone()
        elif k.endswith("weight_g"):
            sd[k] = T(((0.5 + rs.random_sample(shp)) * gain).astype(np.float32))
        elif len(shp) >= 2:
            fan_in = int(np.prod(shp[1:]))
            sd[k] = T((rs.standard_normal(shp) * np.sqrt(2.0 / fan_in) * gain).astype(np.float32))
        elif k in ("prior_log_variance",):
            sd[k] = T(np.asarray([-1.2], np.float32))
        else:
            sd[k] = T((rs.standard_normal(shp) * 0.05).astype(np.float32))
    return sd


G9_CASES = {
    "hvae_2level": dict(model_name="hvae_2level", input_size=[1, 28, 28], input_type="binary", B=8, C=64, N=200),
    "convhvae_2level": dict(model_name="convhvae_2level", input_size=[1, 28, 28], input_type="binary", B=6, C=40, N=120),
    "single_conv": dict(model_name="single_conv", input_size=[3, 16, 16], input_type="continuous", B=4, C=24, N=60,
                        bottleneck=1, z1_size=16, use_logit=False),
}


G19_CASES = {      # the other input geometries of the reference's datasets: RGB 32x32, non-square grey 28x20, binary single_conv
    "convhvae_cifar": dict(model_name="convhvae_2level", dataset_name="cifar10", input_size=[3, 32, 32], input_type="continuous",
                           continuous=True, B=4, C=16, N=40),
    "convhvae_frey": dict(model_name="convhvae_2level", dataset_name="freyfaces", input_size=[1, 28, 20], input_type="gray",
                          continuous=True, B=4, C=16, N=40),
    # gain: twelve residual blocks of He-scaled random filters blow the activations up by 2^12; |log p| ~ 5e7 is beyond what
    # fp32 resolves in ANY implementation (the reference's own gradients are ~1e8 there), so this case runs at a sane scale
    "single_conv_mnist": dict(model_name="single_conv", input_size=[1, 28, 28], input_type="binary", bottleneck=6, z1_size=294,
                              B=4, C=16, N=40, gain=0.35),
}


G21_CASES = {      # the same single_conv geometry at gain 1 (|log p| ~ 5e7, gradients ~ 1e8 in the reference too):
    # what an untrained He-initialised net does in fp32 -- finite, and equal to the reference to fp32 noise at that magnitude
    "single_conv_mnist_gain1": dict(model_name="single_conv", input_size=[1, 28, 28], input_type="binary", bottleneck=6,
                                    z1_size=294, B=4, C=16, N=40, gain=1.0),
}


def g21():
    _model_cases(G21_CASES, "g21_single_conv_gain1")


G22_CASES = {      # convhvae_2level with MORE THAN 1 024 DISTINCT exemplar rows (1 600 draws with replacement from 6 000
    # images name ~1 400), the size at which the build's exemplar encoder switches to its pixel-image convolution stack
    # (evae.ops.GatedConvStackFn) -- and few enough images that the draw has duplicates (the build encodes each distinct image once);
    # reference models/convHVAE_2level.py:13-97
    "convhvae_stack": dict(model_name="convhvae_2level", input_size=[1, 28, 28], input_type="binary", B=8, C=1600, N=6000),
}


def g22():
    _model_cases(G22_CASES, "g22_convhvae_stack")


def g9():
    _model_cases(G9_CASES, "g9_models")


def g19():
    _model_cases(G19_CASES, "g19_models_geometries")


def _model_cases(cases, fixture):
    from utils.utils import importing_model
    out = {}
    for tag, cfg in cases.items():
        cfg = dict(cfg)
        B, C, N = cfg.pop("B"), cfg.pop("C"), cfg.pop("N")
        gain = cfg.pop("gain", 1.0)
        args = vae_args(number_components=C, training_set_size=N, **cfg)
        torch.manual_seed(0)
        model = importing_model(args)(args)
        model.load_state_dict(seeded_state_dict(model, 77, gain))
        D = int(np.prod(args.input_size))
        rs = np.random.RandomState(91)
        if args.input_type == "binary":
            data = gi.gray_images(92, N, D)
            x = (rs.random_sample((B, D)) < 0.3).astype(np.float32)
        else:
            data = ((rs.randint(0, 256, (N, D)) + 0.5) / 256).astype(np.float32)
            x = ((rs.randint(0, 256, (B, D)) + 0.5) / 256).astype(np.float32)
        bidx = rs.randint(0, N, size=(B, 1)).astype(np.int64)
        ex_idx = rs.randint(0, N, size=(C,)).astype(np.int64)
        ex_idx[:2] = bidx[:2, 0]
        zsz = args.z1_size
        eps_list = [rs.standard_normal((B, zsz)).astype(np.float32) for _ in range(2)]
        it = {"i": 0}

        def reparam(mu, logvar, it=it, eps_list=eps_list):
            e = T(eps_list[it["i"] % 2]).reshape(mu.shape); it["i"] += 1
            return e * logvar.mul(0.5).exp() + mu
        model.reparameterize = reparam
        dataset = torch.utils.data.TensorDataset(T(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
        orig = torch.randint
        torch.randint = lambda low=0, high=None, size=None, **kw: T(ex_idx)
        try:
            model.train(); model.zero_grad(); it["i"] = 0
            loss, RE, KL = model.calculate_loss((T(x), T(bidx)), beta=0.6, average=False, dataset=dataset)
            loss.mean().backward()
        finally:
            torch.randint = orig
        out[tag + "_train_loss"] = loss.detach().numpy(); out[tag + "_train_RE"] = RE.detach().numpy()
        out[tag + "_train_KL"] = KL.detach().numpy()
        names, norms = [], []
        for k, v in model.named_parameters():
            names.append(k); norms.append(0.0 if v.grad is None else v.grad.double().norm().item())
        out[tag + "_gnorms"] = np.asarray(norms)
        model.eval()
        with torch.no_grad():
            it["i"] = 0
            cz, clv = model.cache_z(dataset)
            loss, RE, KL = model.calculate_loss((T(x), None), average=False,
                                                exemplars_embedding=(cz, clv, torch.arange(len(cz))))
        out[tag + "_eval_loss"] = loss.numpy(); out[tag + "_eval_RE"] = RE.numpy(); out[tag + "_eval_KL"] = KL.numpy()
        out[tag + "_cache_head"] = cz.numpy()[:16]
        print(tag, "params", len(names), "train loss mean", float(loss.mean()))
    save(fixture, **out)


# ---- G10: approximate prior (cache + top-k), models/BaseModel.py:256-271 via calculate_loss ------------------
def g10():
    B, C, N, k = 16, 300, 1000, 10
    args = vae_args(number_components=C, training_set_size=N, approximate_prior=True, approximate_k=k)
    model = VAE(args)
    p = orc.vae_init_params(np.random.RandomState(123))
    load_params(model, p)
    data = gi.gray_images(63, N).astype(np.float32)
    rs = np.random.RandomState(64)
    bidx = rs.choice(N, size=(B, 1), replace=False).astype(np.int64)
    x = (rs.random_sample((B, 784)) < np.clip(data[bidx[:, 0]] + 0.1, 0, 1)).astype(np.float32)
    eps = rs.standard_normal((B, 40)).astype(np.float32)
    cand = rs.choice(N, size=C, replace=False).astype(np.int64)    # distinct candidates: no exact distance ties
    dataset = torch.utils.data.TensorDataset(T(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    model.reparameterize = lambda mu, logvar: T(eps) * logvar.mul(0.5).exp() + mu
    model.train()
    with torch.no_grad():
        cache = model.cache_z(dataset)
    cache0 = cache[0].numpy().copy()
    orig = torch.randint
    torch.randint = lambda low=0, high=None, size=None, **kw: T(cand.copy())
    try:
        model.zero_grad()
        loss, RE, KL = model.calculate_loss((T(x), T(bidx)), beta=0.8, average=False, cache=cache, dataset=dataset)
        loss.mean().backward()
    finally:
        torch.randint = orig
    # tie check on the top-k boundary of the candidate distances
    zq = model.q_z(T(x))[0].detach()
    d = pairwise_distance(zq, T(cache0)[T(cand)]).numpy()
    assert tie_gap(d, k)[0] > 0
    out = dict(loss=loss.detach().numpy(), RE=RE.detach().numpy(), KL=KL.detach().numpy(),
               cache_after=cache[0].detach().numpy(), cache_before=cache0,
               gnorm=np.asarray([v.grad.double().norm().item() for _, v in model.named_parameters()]))
    save("g10_approx", **out)


# ---- G20: config-5 geometry -- single_conv on 3x64x64, z1 = 256 (bottleneck 1), 256-bin logistic likelihood, ----------
#      approximate (cache + top-k) prior, through calculate_loss; then the evaluation path against the whole cache
G20 = dict(B=4, C=24, N=48, k=3, gain=0.35)


def g20():
    from utils.utils import importing_model
    B, C, N, k, gain = (G20[x] for x in ("B", "C", "N", "k", "gain"))
    args = vae_args(model_name="single_conv", dataset_name="celeba", input_size=[3, 64, 64], input_type="continuous",
                    continuous=True, use_logit=False, bottleneck=1, z1_size=256, number_components=C,
                    training_set_size=N, approximate_prior=True, approximate_k=k)
    torch.manual_seed(0)
    model = importing_model(args)(args)
    model.load_state_dict(seeded_state_dict(model, 78, gain))
    D = int(np.prod(args.input_size))
    rs = np.random.RandomState(93)
    data = ((rs.randint(0, 256, (N, D)) + 0.5) / 256).astype(np.float32)
    bidx = rs.choice(N, size=(B, 1), replace=False).astype(np.int64)
    x = np.clip(data[bidx[:, 0]] + rs.randint(-6, 7, (B, D)).astype(np.float32) / 256, 0.5 / 256, 255.5 / 256).astype(np.float32)
    cand = rs.choice(N, size=C, replace=False).astype(np.int64)    # distinct candidates: no exact distance ties
    eps = rs.standard_normal((B, args.z1_size)).astype(np.float32)
    model.reparameterize = lambda mu, logvar: T(eps).reshape(mu.shape) * logvar.mul(0.5).exp() + mu
    dataset = torch.utils.data.TensorDataset(T(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    model.train()
    with torch.no_grad():
        cache = model.cache_z(dataset)
    cache0 = cache[0].numpy().copy()
    orig = torch.randint
    torch.randint = lambda low=0, high=None, size=None, **kw: T(cand.copy())
    try:
        model.zero_grad()
        loss, RE, KL = model.calculate_loss((T(x), T(bidx)), beta=0.7, average=False, cache=cache, dataset=dataset)
        loss.mean().backward()
    finally:
        torch.randint = orig
    zq = model.q_z(T(x))[0].detach()
    d = pairwise_distance(zq, T(cache0)[T(cand)]).numpy()
    assert tie_gap(d, k)[0] > 0
    out = dict(loss=loss.detach().numpy(), RE=RE.detach().numpy(), KL=KL.detach().numpy(),
               cache_before=cache0, cache_after=cache[0].detach().numpy(),
               gnorms=np.asarray([0.0 if v.grad is None else v.grad.double().norm().item() for _, v in model.named_parameters()]))
    model.eval()
    with torch.no_grad():
        cz, clv = model.cache_z(dataset)
        loss, RE, KL = model.calculate_loss((T(x), None), average=False, exemplars_embedding=(cz, clv, torch.arange(len(cz))))
    out.update(eval_loss=loss.numpy(), eval_RE=RE.numpy(), eval_KL=KL.numpy())
    print("g20 train loss mean", float(out["loss"].mean()), "KL", out["KL"], "eval", float(loss.mean()))
    save("g20_c5_geometry", **out)


# ---- G23: G20's model (single_conv, 3 x 64 x 64, z1 = 256, cache + top-k prior) on a batch of 64 images: 16 384 pixels even in
#      the 96-channel 16 x 16 runs, the size at which the build hands all four residual runs, the convolutions around them and the
#      weight norm to its pixel-image operators (evae.ops.ResStackFn / PlainConvFn / WeightNormSetFn); reference models/fully_conv.py:12-81
G23 = dict(B=64, C=160, N=320, k=3, gain=0.35)


def g23():
    from utils.utils import importing_model
    B, C, N, k, gain = (G23[x] for x in ("B", "C", "N", "k", "gain"))
    args = vae_args(model_name="single_conv", dataset_name="celeba", input_size=[3, 64, 64], input_type="continuous",
                    continuous=True, use_logit=False, bottleneck=1, z1_size=256, number_components=C,
                    training_set_size=N, approximate_prior=True, approximate_k=k)
    torch.manual_seed(0)
    model = importing_model(args)(args)
    model.load_state_dict(seeded_state_dict(model, 79, gain))
    D = int(np.prod(args.input_size))
    data, x, bidx, cand, eps = gi.g23_inputs(B, C, N, D, args.z1_size)
    model.reparameterize = lambda mu, logvar: T(eps).reshape(mu.shape) * logvar.mul(0.5).exp() + mu
    dataset = torch.utils.data.TensorDataset(T(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    model.train()
    with torch.no_grad():
        cache = model.cache_z(dataset)
        zq0 = model.q_z(T(x))[0].numpy()
    cache0 = cache[0].numpy().copy()
    # the candidate draw: of 400 seeded draws the one whose k / k+1 boundary (what decides WHICH rows are re-encoded) is widest, so that
    # two fp32 implementations of the encoder pick the same neighbours; stored in the fixture (an input, like the seeds of the others)
    c1 = cache0.copy(); c1[bidx[:, 0]] = zq0
    best = (-1.0, None)
    for sd in range(400):
        cd = np.random.RandomState(1000 + sd).choice(N, size=C, replace=False).astype(np.int64)
        dd = np.sort(((zq0[:, None, :].astype(np.float64) - c1[cd][None].astype(np.float64)) ** 2).sum(-1), axis=1)
        g = float(((dd[:, k] - dd[:, k - 1]) / dd[:, k]).min())
        if g > best[0]:
            best = (g, cd)
    cand = best[1]
    print("g23 candidate draw: boundary gap %.2e relative" % best[0])
    orig = torch.randint
    torch.randint = lambda low=0, high=None, size=None, **kw: T(cand.copy())
    try:
        model.zero_grad()
        loss, RE, KL = model.calculate_loss((T(x), T(bidx)), beta=0.7, average=False, cache=cache, dataset=dataset)
        loss.mean().backward()
    finally:
        torch.randint = orig
    zq = model.q_z(T(x))[0].detach()
    cache1 = cache0.copy(); cache1[bidx[:, 0]] = zq.numpy()
    d = pairwise_distance(zq, T(cache1)[T(cand)]).numpy()
    gap = tie_gap(d, k)
    assert gap[0] > 0, gap
    ds = np.sort(d.astype(np.float64), axis=1)
    bgap = float(((ds[:, k] - ds[:, k - 1]) / ds[:, k]).min())
    near = np.unique(np.argsort(d, axis=1, kind="stable")[:, :k].reshape(-1))
    names = [n for n, _ in model.named_parameters()]
    out = dict(loss=loss.detach().numpy(), RE=RE.detach().numpy(), KL=KL.detach().numpy(),
               cache_before_head=cache0[:32], cache_after=cache[0].detach().numpy(), n_neighbours=np.asarray(len(near)),
               topk_boundary_rel_gap=np.asarray(bgap), cand=cand,
               gnorms=np.asarray([0.0 if v.grad is None else v.grad.double().norm().item() for _, v in model.named_parameters()]))
    model.eval()
    with torch.no_grad():
        cz, clv = model.cache_z(dataset)
        loss, RE, KL = model.calculate_loss((T(x), None), average=False, exemplars_embedding=(cz, clv, torch.arange(len(cz))))
    out.update(eval_loss=loss.numpy(), eval_RE=RE.numpy(), eval_KL=KL.numpy())
    print("g23 train loss mean", float(out["loss"].mean()), "KL mean", float(out["KL"].mean()), "neighbours", len(near),
          "gap", gap, "eval", float(loss.mean()), "params", len(names))
    save("g23_c5_window_size", **out)


# ---- G11: evaluation loops (utils/evaluation.py:11-33, 72-103): ELBO over a loader and IWAE log-likelihood ----
def g11():
    from utils.evaluation import evaluate_loss, calculate_likelihood
    N, NT, S = 400, 12, 50
    args = vae_args(number_components=N, training_set_size=N)
    args.batch_size = 5
    model = VAE(args)
    load_params(model, orc.vae_init_params(np.random.RandomState(123)))
    data = gi.binary_images(71, N)
    test = gi.binary_images(72, NT)
    train_ds = torch.utils.data.TensorDataset(T(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    test_ds = torch.utils.data.TensorDataset(T(test), torch.zeros(NT))
    loader = torch.utils.data.DataLoader(test_ds, batch_size=5, shuffle=False)
    eps_rs = np.random.RandomState(73)

    def reparam(mu, logvar):
        e = T(eps_rs.standard_normal(tuple(mu.shape)).astype(np.float32))
        return e * logvar.mul(0.5).exp() + mu
    model.reparameterize = reparam
    with torch.no_grad():
        elbo, re, kl = evaluate_loss(args, model, loader, dataset=train_ds)
        model.eval()
        emb = (lambda z, lv: (z, lv, torch.arange(len(z))))(*model.cache_z(train_ds))
        ll = calculate_likelihood(args, model, loader, S=S, exemplars_embedding=emb)
    save("g11_eval", elbo=np.asarray([elbo, re, kl]), ll=np.asarray([ll]))


# ---- G12: the checkpoint wire format (density_estimation.py:148-156 + utils/utils.py:22-32) ------------------
def g12():
    """A checkpoint written by the REFERENCE's save_model after two optimizer steps of a tiny vae, plus what the
    reference holds after loading it back and taking one more step -- the build must load the file (model and
    optimizer) and continue on the same trajectory."""
    import warnings
    from utils.utils import save_model, load_model
    args = vae_args(input_size=[1, 8, 8], hidden_size=16, z1_size=8, z2_size=8, number_components=10, training_set_size=50)
    torch.manual_seed(12)
    model = VAE(args)
    opt = AdamNormGrad(model.parameters(), lr=5e-4)
    rs = np.random.RandomState(120)
    names = [n for n, _ in model.named_parameters()]
    out = {}

    def step(tag):
        for n, p_ in model.named_parameters():
            g = rs.standard_normal(tuple(p_.shape)).astype(np.float32)
            out["%s_grad_%s" % (tag, n)] = g
            p_.grad = T(g.copy())
        opt.step()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        step("s1"); step("s2")
        path = os.path.join(OUT, "g12_checkpoint.pth")
        content = {'epoch': 7, 'state_dict': model.state_dict(), 'optimizer': opt.state_dict(), 'best_loss': 123.5, 'e': 3}
        save_model(path + ".tmp", path, content)
        # what a fresh reference model + optimizer hold after load_model, then one more step
        torch.manual_seed(99)
        model2 = VAE(args)
        opt2 = AdamNormGrad(model2.parameters(), lr=5e-4)
        ck = load_model(path, model2, opt2)
        assert ck['epoch'] == 7 and ck['e'] == 3
        for n, p_ in model2.named_parameters():
            out["loaded_" + n] = p_.detach().numpy().copy()
            out["loaded_m_" + n] = opt2.state[p_]["exp_avg"].numpy().copy()
            out["loaded_v_" + n] = opt2.state[p_]["exp_avg_sq"].numpy().copy()
            out["loaded_step_" + n] = np.asarray(opt2.state[p_]["step"])
        model, opt = model2, opt2
        step("s3")
        for n, p_ in model2.named_parameters():
            out["after_" + n] = p_.detach().numpy().copy()
    out["names"] = np.asarray(names)
    save("g12_checkpoint", **out)
    print("g12_checkpoint.pth %8.1f KB" % (os.path.getsize(path) / 1024))


# ---- G13: the loops either side of the hot path (utils/training.py:15-51, knn_on_latent.py:32-74, evaluation.py:106-140)
def g13():
    """Two epochs of the reference's train_one_epoch on a tiny dataset (last batch partial), then its kNN report and its
    final_evaluation.  z = mean (reparameterize patched) so that the run is deterministic and a captured step can
    reproduce it; exemplar indices come from the seeded CPU generator in both trees."""
    import tempfile, warnings
    import utils.evaluation as ev
    from utils.training import train_one_epoch
    from utils.knn_on_latent import report_knn_on_latent
    from utils.utils import save_model
    N, NV, B, C = 200, 64, 32, 50
    args = vae_args(number_components=C, training_set_size=N)
    args.batch_size, args.dynamic_binarization, args.warmup, args.S = B, False, 100, 20
    model = VAE(args)
    load_params(model, orc.vae_init_params(np.random.RandomState(123)))
    model.reparameterize = lambda mu, logvar: mu
    mk = lambda seed, n: T(gi.binary_images(seed, n))
    train_ds = torch.utils.data.TensorDataset(mk(81, N), torch.arange(N).reshape(-1, 1), torch.arange(N) % 10)
    val_ds = torch.utils.data.TensorDataset(mk(82, NV), (torch.arange(NV) * 3) % 10)
    test_ds = torch.utils.data.TensorDataset(mk(83, NV), (torch.arange(NV) * 7) % 10)
    L = lambda ds: torch.utils.data.DataLoader(ds, batch_size=B, shuffle=False)
    train_loader, val_loader, test_loader = L(train_ds), L(val_ds), L(test_ds)
    opt = AdamNormGrad(model.parameters(), lr=5e-4)
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(130)
        out["epoch1"] = np.asarray(train_one_epoch(1, args, train_loader, model, opt))
        out["epoch2"] = np.asarray(train_one_epoch(2, args, train_loader, model, opt))
        for n, p_ in model.named_parameters():
            out["sum_" + n] = np.asarray(p_.detach().double().sum().item())
            out["norm_" + n] = np.asarray(p_.detach().double().norm().item())
        model.eval()
        for flag in (True, False):
            d = {"3": [], "5": [], "7": [], "15": []}
            report_knn_on_latent(train_loader, val_loader, test_loader, model, "", d, args, val=flag)
            out["knn_val" if flag else "knn_test"] = np.asarray([d[k][0] for k in ("3", "5", "7", "15")])
        ev.visualize_reconstruction = lambda *a, **k: None       # plotting, outside the path
        ev.visualize_generation = lambda *a, **k: None
        with tempfile.TemporaryDirectory() as tmp:
            tmp = tmp + "/"
            save_model(tmp + "c.tmp", tmp + "best.model", {'epoch': 2, 'state_dict': model.state_dict(),
                                                           'optimizer': opt.state_dict(), 'best_loss': 0.0, 'e': 0})
            with torch.no_grad():                                  # as density_estimation.py calls it
                ev.final_evaluation(train_loader, test_loader, val_loader, tmp + "best.model", model, opt, args, tmp)
            out["final"] = np.asarray([float(torch.load(tmp + "vae." + k, weights_only=False))
                                       for k in ("test_log_likelihood", "test_loss", "test_re", "test_kl")])
            out["log_txt"] = np.asarray(open(tmp + "vae_experiment_log.txt").read())
    save("g13_loops", **out)
    print(out["epoch1"], out["epoch2"], out["knn_val"], out["knn_test"], out["final"])


# ---- G14: raw arrays -> loaders (utils/load_data/base_load_data.py:8-120) ---------------------------------------
def g14():
    """The reference's load_dataset pipeline on synthetic 8-bit images behind a stub obtain_data: split, value range,
    fixed-seed binarisation of the evaluation splits, dataset tuples.  Three variants: dynamically binarised, grey with
    (x + 0.5) / 256, grey with the logit transform (noise from the global numpy generator)."""
    from types import SimpleNamespace
    from utils.load_data.base_load_data import base_load_data
    rs = np.random.RandomState(140)
    raw_train = rs.randint(0, 256, (300, 8, 8)).astype(np.uint8); lab_train = rs.randint(0, 10, 300)
    raw_test = rs.randint(0, 256, (60, 8, 8)).astype(np.uint8); lab_test = rs.randint(0, 10, 60)

    class stub(base_load_data):
        def obtain_data(self):
            return (SimpleNamespace(data=T(raw_train), train_labels=T(lab_train)),
                    SimpleNamespace(data=T(raw_test), test_labels=T(lab_test)))
    out = {"raw_train": raw_train, "lab_train": lab_train, "raw_test": raw_test, "lab_test": lab_test}
    variants = {"dyn": dict(input_type="binary", dynamic_binarization=True, continuous=False, use_logit=False),
                "grey": dict(input_type="gray", dynamic_binarization=False, continuous=True, use_logit=False),
                "logit": dict(input_type="gray", dynamic_binarization=False, continuous=False, use_logit=True)}
    for tag, kw in variants.items():
        a = Namespace(dataset_name="dynamic_mnist", input_size=[1, 8, 8], training_set_size=250, batch_size=32, test_batch_size=20,
                      use_training_data_init=0, number_components=10, lambd=1e-4, **kw)
        np.random.seed(141)
        tr, va, te, a2 = stub(a, no_binarization=(tag != "dyn")).load_dataset()
        out[tag + "_x_train"], out[tag + "_idx"], out[tag + "_y_train"] = (t.numpy() for t in tr.dataset.tensors)
        out[tag + "_x_val"], out[tag + "_y_val"] = (t.numpy() for t in va.dataset.tensors)
        out[tag + "_x_test"], out[tag + "_y_test"] = (t.numpy() for t in te.dataset.tensors)
        out[tag + "_input_type"] = np.asarray(a2.input_type)
        out[tag + "_batch"] = np.asarray([tr.batch_size, va.batch_size, te.batch_size])
    save("g14_load_data", **out)


# ---- G15: the other priors behind the same API (models/BaseModel.py:111-128: standard, vampprior) -------------
def g15():
    from utils.evaluation import evaluate_loss
    out = {}
    B, D = 16, 64
    x = gi.binary_images(151, B, D)
    test = gi.binary_images(152, 24, D)
    eps = np.random.RandomState(153).standard_normal((B, 8)).astype(np.float32)
    out["eps"] = eps
    for prior in ("standard", "vampprior"):
        args = vae_args(prior=prior, input_size=[1, 8, 8], hidden_size=32, z1_size=8, z2_size=8, number_components=20,
                        training_set_size=100)
        args.pseudoinputs_mean, args.pseudoinputs_std, args.use_training_data_init = 0.05, 0.01, False
        args.batch_size = B
        torch.manual_seed(150)
        model = VAE(args)
        model.train()
        for k, v in model.state_dict().items():
            out[prior + "_sd_" + k] = v.numpy().copy()
        model.reparameterize = lambda mu, logvar: T(eps[:mu.shape[0]]) * logvar.mul(0.5).exp() + mu
        loss, RE, KL = model.calculate_loss((T(x), torch.arange(B).reshape(-1, 1)), 0.7, average=False)
        loss.mean().backward()
        out[prior + "_loss"], out[prior + "_RE"], out[prior + "_KL"] = (t.detach().numpy() for t in (loss, RE, KL))
        for n, p_ in model.named_parameters():
            out[prior + "_gnorm_" + n] = np.asarray(0.0 if p_.grad is None else p_.grad.double().norm().item())
        model.eval()
        model.reparameterize = lambda mu, logvar: mu
        loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(T(test), torch.zeros(24)), batch_size=8)
        with torch.no_grad():
            out[prior + "_eval"] = np.asarray(evaluate_loss(args, model, loader, dataset=None))
    save("g15_priors", **out)


# ---- G16: option flags consumed on the path (utils/nn.py:38-47 no_attention, models/BaseModel.py:103-107 no_mask) ----
def g16():
    out = {}
    B, D, N, C = 16, 64, 120, 40
    data = gi.gray_images(161, N, D)
    x = (gi.binary_images(162, B, D))
    bidx = np.random.RandomState(163).randint(0, N, (B, 1)).astype(np.int64)
    eps = np.random.RandomState(164).standard_normal((B, 8)).astype(np.float32)
    out["eps"], out["bidx"] = eps, bidx
    dataset = torch.utils.data.TensorDataset(T(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    for tag, kw in (("no_attention", dict(no_attention=True)), ("no_mask", dict(no_mask=True)), ("plain", dict())):
        args = vae_args(input_size=[1, 8, 8], hidden_size=32, z1_size=8, z2_size=8, number_components=C, training_set_size=N, **kw)
        torch.manual_seed(160)
        model = VAE(args)
        model.train()
        for k, v in model.state_dict().items():
            out[tag + "_sd_" + k] = v.numpy().copy()
        model.reparameterize = lambda mu, logvar: T(eps[:mu.shape[0]]) * logvar.mul(0.5).exp() + mu
        torch.manual_seed(165)                                  # the exemplar draw
        loss, RE, KL = model.calculate_loss((T(x), T(bidx)), 0.6, average=False, dataset=dataset)
        loss.mean().backward()
        out[tag + "_loss"], out[tag + "_RE"], out[tag + "_KL"] = (t.detach().numpy() for t in (loss, RE, KL))
        for n, p_ in model.named_parameters():
            out[tag + "_gnorm_" + n] = np.asarray(0.0 if p_.grad is None else p_.grad.double().norm().item())
    save("g16_options", **out)


# ---- G17: grey-level inputs through the MLP models (AbsModel.py:26-37 / AbsHModel.py:70-86, log_logistic_256) ----
def g17():
    from models.HVAE_2level import VAE as HVAE
    out = {}
    B, D, N, C = 12, 64, 90, 30
    data = gi.gray_images(171, N, D)
    x = np.clip(gi.gray_images(172, B, D) + 0.002, 0.0, 1.0).astype(np.float32)
    bidx = np.random.RandomState(173).randint(0, N, (B, 1)).astype(np.int64)
    out["bidx"] = bidx
    dataset = torch.utils.data.TensorDataset(T(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    for tag, cls, name in (("vae_gray", VAE, "vae"), ("hvae_gray", HVAE, "hvae_2level")):
        args = vae_args(model_name=name, input_type="gray", continuous=True, input_size=[1, 8, 8], hidden_size=32, z1_size=8,
                        z2_size=8, number_components=C, training_set_size=N)
        torch.manual_seed(170)
        model = cls(args)
        model.train()
        for k, v in model.state_dict().items():
            out[tag + "_sd_" + k] = v.numpy().copy()
        rs = np.random.RandomState(174)
        model.reparameterize = lambda mu, logvar: T(rs.standard_normal(tuple(mu.shape)).astype(np.float32)) * logvar.mul(0.5).exp() + mu
        torch.manual_seed(175)
        loss, RE, KL = model.calculate_loss((T(x), T(bidx)), 0.8, average=False, dataset=dataset)
        loss.mean().backward()
        out[tag + "_loss"], out[tag + "_RE"], out[tag + "_KL"] = (t.detach().numpy() for t in (loss, RE, KL))
        for n, p_ in model.named_parameters():
            out[tag + "_gnorm_" + n] = np.asarray(0.0 if p_.grad is None else p_.grad.double().norm().item())
    save("g17_grey_mlp", **out)


# ---- G18: state_dict names and shapes of every architecture x input geometry the reference's datasets produce --------
def g18():
    import json
    from utils.utils import importing_model
    combos = [("vae", "dynamic_mnist", [1, 28, 28], "binary"), ("vae", "freyfaces", [1, 28, 20], "gray"),
              ("hvae_2level", "omniglot", [1, 28, 28], "binary"), ("hvae_2level", "freyfaces", [1, 28, 20], "gray"),
              ("convhvae_2level", "fashion_mnist", [1, 28, 28], "binary"), ("convhvae_2level", "cifar10", [3, 32, 32], "continuous"),
              ("convhvae_2level", "freyfaces", [1, 28, 20], "gray"),
              ("single_conv", "dynamic_mnist", [1, 28, 28], "binary"), ("single_conv", "celeba", [3, 64, 64], "continuous")]
    out = {}
    for name, ds, isz, it in combos:
        bott = 1 if isz[1] == 64 else 6
        z1 = bott * (isz[1] // 4) * (isz[2] // 4) if name == "single_conv" else 40
        args = vae_args(model_name=name, dataset_name=ds, input_size=isz, input_type=it, bottleneck=bott, z1_size=z1,
                        continuous=(it != "binary"))
        args.rs_blocks = 4
        torch.manual_seed(0)
        m = importing_model(args)(args)
        out["%s|%s" % (name, ds)] = {"input_size": isz, "input_type": it, "bottleneck": bott, "z1_size": z1,
                                      "entries": [[k, list(v.shape)] for k, v in m.state_dict().items()]}
    path = os.path.join(OUT, "g18_state_dict_shapes.json")
    json.dump(out, open(path, "w"))
    print("g18_state_dict_shapes.json %.1f KB, %d architectures" % (os.path.getsize(path) / 1024, len(out)))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1_g2", "g3", "g4", "g5", "g6", "g6_conv", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19", "g20", "g21", "g22", "g23"]
    for w in which:
        globals()[w]()
