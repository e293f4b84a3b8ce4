"""GPU parity of the drop-in model API against the goldens generated from the real reference
(tests/golden/g7_vae_loss.npz) and against the oracle."""
import os

import numpy as np
import pytest
import torch

import evae_oracle as orc
import golden_inputs as gi
import smoke_case

pytestmark = pytest.mark.gpu


def rel(a, b):
    return smoke_case.rel(np, a, b)


@pytest.mark.parametrize("fused", [True, False])
def test_vae_train_step_matches_oracle(fused):
    smoke_case.run(torch, np, orc, verbose=True, fused=fused)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("tag,B,C,N,seed", [("small", 16, 200, 500, 61), ("c1", 100, 1000, 4000, 62)])
def test_vae_calculate_loss_matches_reference_golden(golden, tag, B, C, N, seed, fused):
    """ELBO / RE / KL per sample and gradient norms vs the REAL reference (1e-4 relative bar)."""
    g = golden("g7_vae_loss")
    args = smoke_case.vae_args(number_components=C, training_set_size=N)
    model, p = smoke_case.build_model(torch, np, orc, args)
    data, bidx, x, eps, ex_idx = smoke_case.make_case(np, B, C, N, seed, gi)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    model._draw_eps = lambda like: torch.from_numpy(eps).to(like.device)
    model._use_fused = fused          # one-node fused path (evae/fused_vae.py) vs the modular autograd path
    orig = torch.randint
    torch.randint = lambda low=0, high=None, size=None, **kw: torch.from_numpy(ex_idx)
    try:
        model.train()
        loss, RE, KL = model.calculate_loss((torch.from_numpy(x).cuda(), torch.from_numpy(bidx).cuda()), beta=0.37,
                                            average=False, dataset=dataset)
        loss.mean().backward()
    finally:
        torch.randint = orig
    for k, v in (("loss", loss), ("RE", RE), ("KL", KL)):
        assert rel(v.detach().cpu().numpy(), g["%s_train_%s" % (tag, k)]) < 1e-4, k
    for name, prm in model.named_parameters():
        ref_norm = g["%s_gnorm_%s" % (tag, name)][0]
        got = float(prm.grad.double().norm().item())
        assert abs(got - ref_norm) <= 5e-4 * max(ref_norm, 1e-6), (name, got, ref_norm)
        assert rel(prm.grad.reshape(-1)[:16].cpu().numpy(), g["%s_ghead_%s" % (tag, name)]) < 1e-3, name
    # evaluation: whole-dataset cache as embedding, no mask (utils/evaluation.py path)
    model.eval()
    with torch.no_grad():
        cz, clv = model.cache_z(dataset)
        emb = (cz, clv, torch.arange(len(cz)))
        loss, RE, KL = model.calculate_loss((torch.from_numpy(x).cuda(), None), average=False, exemplars_embedding=emb)
    assert rel(cz[:32].cpu().numpy(), g[tag + "_cache_head"]) < 1e-5
    for k, v in (("loss", loss), ("RE", RE), ("KL", KL)):
        assert rel(v.cpu().numpy(), g["%s_eval_%s" % (tag, k)]) < 1e-4, k


def test_state_dict_keys_match_reference_names():
    args = smoke_case.vae_args()
    model, p = smoke_case.build_model(torch, np, orc, args)
    assert list(model.state_dict().keys()) == orc.VAE_PARAM_NAMES


def test_no_cpu_fallback():
    from evae import ops, _lib
    with pytest.raises(_lib.EvaeError):
        ops.prior_lse_fwd(torch.zeros(2, 4), torch.zeros(3, 4), torch.zeros(4))


@pytest.mark.parametrize("feed", ["device_indices", "host_loader", "foreign_images", "host_loader_staged_upload"])
def test_graphed_step_matches_eager(feed, monkeypatch):
    """hipGraph replay of the whole step (evae/graph.py) == the same steps launched eagerly.  `host_loader`: CPU batches
    as a DataLoader yields them (the step gathers the images from the resident dataset by index); `foreign_images`:
    batches that are NOT rows of the dataset (the step must notice and upload the images instead).  A step this thin uploads
    its control block directly on its own stream; `..._staged_upload` forces the double-buffered path of the large steps
    (evae_ctl_upload)."""
    if feed == "host_loader_staged_upload":
        monkeypatch.setenv("EVAE_CTL_DIRECT", "0")
        feed = "host_loader"
    from evae.graph import GraphedTrainStep
    from utils.optimizer import AdamNormGrad
    B, C, N = 32, 500, 2000
    data = gi.binary_images(5, N)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    results = []
    for use_graph in (False, True):
        args = smoke_case.vae_args(number_components=C, training_set_size=N, batch_size=B)
        model, _ = smoke_case.build_model(torch, np, orc, args)
        model.train()
        opt = AdamNormGrad(model.parameters(), lr=5e-4)
        torch.manual_seed(3); torch.cuda.manual_seed(3)
        runner = GraphedTrainStep(model, opt, dataset, B, False) if use_graph else None
        losses = []
        for it in range(7):
            xb = torch.from_numpy(data[it * B:(it + 1) * B])
            if feed == "foreign_images":
                xb = 1.0 - xb
            ib = torch.arange(it * B, (it + 1) * B).reshape(-1, 1)
            if runner is not None:
                if feed != "host_loader":
                    xb = xb.cuda()
                if feed == "device_indices":
                    ib = ib.cuda()
                losses.append(runner(xb, ib, 0.5)[0].item())
            else:
                xb, ib = xb.cuda(), ib.cuda()
                if feed != "foreign_images":
                    # the captured step draws eps with the counter-based generator of its prologue launch
                    # (seed = torch's seed when the runner was built, counter = step number): same draw here
                    from evae import ops as _ops
                    eps = torch.empty((B, args.z1_size), device="cuda")
                    _ops.batch_prologue(torch.from_numpy(data).cuda(), ib.reshape(-1).contiguous(), False,
                                        torch.tensor([3, it], dtype=torch.int64, device="cuda"), torch.empty_like(xb), eps)
                    model._eps_override = eps
                opt.zero_grad()
                loss, RE, KL = model.calculate_loss((xb, ib), 0.5, average=True, dataset=dataset)
                loss.backward()
                opt.step()
                losses.append(loss.item())
        results.append((losses, {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}))
        if runner is not None:
            assert runner.graph is not None                       # steps 4.. were replays
            assert runner.by_index == (feed != "foreign_images")
            assert runner._direct == (os.environ.get("EVAE_CTL_DIRECT") != "0")
    (l0, p0), (l1, p1) = results
    assert rel(np.asarray(l1), np.asarray(l0)) < 1e-5
    for k in p0:
        assert rel(p1[k], p0[k]) < 1e-5, k


@pytest.mark.parametrize("model_name", ["hvae_2level", "convhvae_2level"])
def test_graphed_step_other_architectures(model_name):
    """The captured step is not specific to the fused `vae` node: the modular autograd path of the hierarchical and
    convolutional models replays from a hipGraph too, with the same trajectory as eager launching."""
    from evae.graph import GraphedTrainStep
    from utils.optimizer import AdamNormGrad
    from utils.utils import importing_model
    B, C, N = 16, 120, 600
    data = gi.binary_images(6, N)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    results = []
    for use_graph in (False, True):
        args = smoke_case.vae_args(model_name=model_name, number_components=C, training_set_size=N, batch_size=B)
        torch.manual_seed(5); torch.cuda.manual_seed(5)
        model = importing_model(args)(args).cuda()
        model.train()
        opt = AdamNormGrad(model.parameters(), lr=5e-4)
        torch.manual_seed(3); torch.cuda.manual_seed(3)
        runner = GraphedTrainStep(model, opt, dataset, B, False) if use_graph else None
        losses = []
        for it in range(6):
            xb = torch.from_numpy(data[it * B:(it + 1) * B]).cuda()
            ib = torch.arange(it * B, (it + 1) * B).reshape(-1, 1).cuda()
            if runner is not None:
                losses.append(runner(xb, ib, 0.5)[0].item())
            else:
                opt.zero_grad()
                loss, RE, KL = model.calculate_loss((xb, ib), 0.5, average=True, dataset=dataset)
                loss.backward()
                opt.step()
                losses.append(loss.item())
        results.append(losses)
        if runner is not None:
            assert runner.graph is not None and not runner.failed     # steps 4.. were replays
    assert rel(np.asarray(results[1]), np.asarray(results[0])) < 2e-5


def test_graphed_step_with_the_approximate_prior():
    """The kNN-approximate prior (reference models/BaseModel.py:256-271) inside the captured step: its exemplar union is a
    fixed list of B * k slots with masked repeats instead of `unique`, so the step replays from a hipGraph -- same
    trajectory (losses, parameters, latent cache) as eager launching; and the fixed-slot form equals the reference's
    `unique` form (the sharded / no_mask code path) on the same step."""
    from evae.graph import GraphedTrainStep
    from utils.optimizer import AdamNormGrad
    B, C, N, k = 16, 300, 800, 5
    data = gi.binary_images(8, N)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    eps_all = torch.from_numpy(np.random.RandomState(4).standard_normal((8, B, 40)).astype(np.float32)).cuda()
    results = []
    for use_graph in (False, True):
        args = smoke_case.vae_args(number_components=C, training_set_size=N, batch_size=B, approximate_prior=True, approximate_k=k)
        model, _ = smoke_case.build_model(torch, np, orc, args)
        model.train()
        it_box = {"i": 0}
        eps_static = torch.zeros((B, 40), device="cuda")
        model._draw_eps = lambda like: eps_static          # a static buffer: refreshed before every step, replay-safe
        opt = AdamNormGrad(model.parameters(), lr=5e-4)
        with torch.no_grad():
            cache = tuple(model.cache_z(dataset))
        torch.manual_seed(3); torch.cuda.manual_seed(3)
        runner = GraphedTrainStep(model, opt, dataset, B, False) if use_graph else None
        if runner is not None:
            cache = runner.set_cache(cache)
        losses = []
        for it in range(7):
            eps_static.copy_(eps_all[it])
            xb = torch.from_numpy(data[it * B:(it + 1) * B]).cuda()
            ib = torch.arange(it * B, (it + 1) * B).reshape(-1, 1).cuda()
            if runner is not None:
                losses.append(runner(xb, ib, 0.5)[0].item())
            else:
                opt.zero_grad()
                loss, RE, KL = model.calculate_loss((xb, ib), 0.5, average=True, dataset=dataset, cache=cache)
                loss.backward()
                opt.step()
                losses.append(loss.item())
        results.append((losses, cache[0].detach().cpu().numpy().copy(),
                        {kk: v.detach().cpu().numpy().copy() for kk, v in model.named_parameters()}))
        if runner is not None:
            assert runner.graph is not None and not runner.failed
    (l0, c0, p0), (l1, c1, p1) = results
    assert rel(np.asarray(l1), np.asarray(l0)) < 2e-5
    assert rel(c1, c0) < 2e-5
    for kk in p0:
        assert rel(p1[kk], p0[kk]) < 2e-5, kk
    # fixed slots with masked repeats (the fused node, then the modular autograd path) == unique (the reference's formulation,
    # modular path) on one step with the same weights / draws: loss, gradients, refreshed cache
    outs = []
    for fused, no_static in ((True, False), (False, False), (False, True)):
        args = smoke_case.vae_args(number_components=C, training_set_size=N, batch_size=B, approximate_prior=True, approximate_k=k)
        model, _ = smoke_case.build_model(torch, np, orc, args)
        model.train()
        model._use_fused = fused
        assert model._fused_config() == fused
        model._draw_eps = lambda like: eps_all[0]
        with torch.no_grad():
            cache = tuple(model.cache_z(dataset))
        torch.manual_seed(9)
        if no_static:
            import evae.ops as _ops
            orig = _ops.select_exemplars

            def unique_form(pos, cand):           # the reference's formulation through the same interface
                u = torch.unique(pos.view(-1))
                rows = cand[u]
                return rows, rows
            _ops.select_exemplars = unique_form
        try:
            xb = torch.from_numpy(data[:B]).cuda(); ib = torch.arange(B).reshape(-1, 1).cuda()
            loss, RE, KL = model.calculate_loss((xb, ib), 0.5, average=False, dataset=dataset, cache=cache)
            loss.mean().backward()
        finally:
            if no_static:
                _ops.select_exemplars = orig
        outs.append((loss.detach().cpu().numpy(), {kk: v.grad.cpu().numpy().copy() for kk, v in model.named_parameters()},
                     cache[0].detach().cpu().numpy().copy()))
    for other in (1, 2):
        assert rel(outs[0][0], outs[other][0]) < 1e-5
        assert rel(outs[0][2], outs[other][2]) < 1e-5
        for kk in outs[0][1]:
            assert rel(outs[0][1][kk], outs[other][1][kk]) < 1e-4, kk


def test_two_captured_steps_on_one_optimizer_keep_their_own_pointer_tables():
    """train_one_epoch keys its captured runners by (dataset, batch size, binarisation): two of them can share one optimizer.
    Each graph owns its AdamNormGrad pointer table (its own gradient buffers) -- alternating replays follow the eager
    trajectory."""
    from evae.graph import GraphedTrainStep
    from utils.optimizer import AdamNormGrad
    C, N = 300, 1200
    data = gi.binary_images(9, N)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    eps_static = {16: torch.zeros((16, 40), device="cuda"), 24: torch.zeros((24, 40), device="cuda")}
    eps_all = torch.from_numpy(np.random.RandomState(5).standard_normal((12, 24, 40)).astype(np.float32)).cuda()
    results = []
    for use_graph in (False, True):
        args = smoke_case.vae_args(number_components=C, training_set_size=N, batch_size=16)
        model, _ = smoke_case.build_model(torch, np, orc, args)
        model.train()
        model._use_fused = False                # modular path: eps through the _draw_eps hook
        cur = {"B": 16}
        model._draw_eps = lambda like: eps_static[cur["B"]]
        opt = AdamNormGrad(model.parameters(), lr=5e-4)
        torch.manual_seed(3); torch.cuda.manual_seed(3)
        runners = {b: GraphedTrainStep(model, opt, dataset, b, False) for b in (16, 24)} if use_graph else None
        losses = []
        for it in range(12):
            Bc = 16 if it % 2 == 0 else 24
            cur["B"] = Bc
            eps_static[Bc].copy_(eps_all[it, :Bc])
            xb = torch.from_numpy(data[it * 24:it * 24 + Bc]).cuda()
            ib = torch.arange(it * 24, it * 24 + Bc).reshape(-1, 1).cuda()
            if runners is not None:
                losses.append(runners[Bc](xb, ib, 0.5)[0].item())
            else:
                opt.zero_grad()
                loss, RE, KL = model.calculate_loss((xb, ib), 0.5, average=True, dataset=dataset)
                loss.backward()
                opt.step()
                losses.append(loss.item())
        results.append((losses, {kk: v.detach().cpu().numpy().copy() for kk, v in model.named_parameters()}))
        if runners is not None:
            assert all(r.graph is not None for r in runners.values())
    (l0, p0), (l1, p1) = results
    assert rel(np.asarray(l1), np.asarray(l0)) < 2e-5
    for kk in p0:
        assert rel(p1[kk], p0[kk]) < 2e-5, kk


def test_captured_step_after_resuming_a_checkpoint_with_unused_parameters():
    """density_estimation.py:102's resume flow (load_model -> optimizer.load_state_dict) for single_conv, whose
    BatchNorm2d parameters never receive a gradient: the trained parameters resume at step N, the unused ones have no state.
    The runner's first call steps eagerly and learns the participants; the replays continue the eager trajectory, and the
    unused parameters end with no optimizer state, as in the reference (utils/optimizer.py:50-57)."""
    from evae.graph import GraphedTrainStep
    from utils.optimizer import AdamNormGrad
    from utils.utils import importing_model
    B, C, N = 4, 24, 60
    rs = np.random.RandomState(17)
    data = ((rs.randint(0, 256, (N, 3 * 16 * 16)) + 0.5) / 256).astype(np.float32)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    eps_all = torch.from_numpy(rs.standard_normal((9, B, 16)).astype(np.float32)).cuda()
    eps_static = torch.zeros((B, 16), device="cuda")

    def make():
        args = smoke_case.vae_args(model_name="single_conv", input_size=[3, 16, 16], input_type="continuous", bottleneck=1,
                                   z1_size=16, use_logit=False, number_components=C, training_set_size=N, batch_size=B)
        model = importing_model(args)(args)
        model.load_state_dict(seeded_state_dict(model, 77, 0.35))
        model = model.cuda().train()
        model._draw_eps = lambda like: eps_static.reshape(like.shape)
        return model, AdamNormGrad(model.parameters(), lr=5e-4)

    def eager_step(model, opt, it):
        eps_static.copy_(eps_all[it])
        xb = torch.from_numpy(data[it * B:(it + 1) * B]).cuda()
        ib = torch.arange(it * B, (it + 1) * B).reshape(-1, 1).cuda()
        opt.zero_grad()
        loss, _, _ = model.calculate_loss((xb, ib), 0.5, average=True, dataset=dataset)
        loss.backward()
        opt.step()
        return loss.item()

    torch.manual_seed(3)
    model, opt = make()
    for it in range(3):
        eager_step(model, opt, it)
    ck_model = {k: v.clone() for k, v in model.state_dict().items()}
    ck_opt = opt.state_dict()
    import copy
    ck_opt = copy.deepcopy(ck_opt)
    n_state = len(ck_opt["state"])
    assert 0 < n_state < len(list(model.parameters()))            # the unused parameters carry no state
    results = []
    for use_graph in (False, True):
        model, opt = make()
        model.load_state_dict(ck_model)
        opt.load_state_dict(copy.deepcopy(ck_opt))
        torch.manual_seed(5)
        runner = GraphedTrainStep(model, opt, dataset, B, False) if use_graph else None
        losses = []
        for it in range(3, 9):
            if runner is None:
                losses.append(eager_step(model, opt, it))
            else:
                eps_static.copy_(eps_all[it])
                xb = torch.from_numpy(data[it * B:(it + 1) * B]).cuda()
                ib = torch.arange(it * B, (it + 1) * B).reshape(-1, 1).cuda()
                losses.append(runner(xb, ib, 0.5)[0].item())
        if runner is not None:
            assert runner.graph is not None and not runner.failed
        assert len(opt.state_dict()["state"]) == n_state          # no state for parameters the reference would skip
        steps = {int(st["step"]) for st in opt.state.values() if len(st)}
        assert steps == {9}
        results.append((losses, {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}))
    (l0, p0), (l1, p1) = results
    assert rel(np.asarray(l1), np.asarray(l0)) < 2e-5
    for k in p0:
        assert rel(p1[k], p0[k]) < 5e-5, k


# ---- the other architectures against goldens of the real reference ----------
def seeded_state_dict(model, seed, gain=1.0):
    """Same deterministic fill as tools/gen_goldens.py::seeded_state_dict (walks the state_dict in order)."""
    rs = np.random.RandomState(seed)
    sd = {}
    for k, v in model.state_dict().items():
        shp = tuple(v.shape)
        if "normalization" in k or k.endswith("num_batches_tracked"):
            sd[k] = v.clone()
        elif k.endswith("weight_g"):
            sd[k] = torch.from_numpy(((0.5 + rs.random_sample(shp)) * gain).astype(np.float32))
        elif len(shp) >= 2:
            fan_in = int(np.prod(shp[1:]))
            sd[k] = torch.from_numpy((rs.standard_normal(shp) * np.sqrt(2.0 / fan_in) * gain).astype(np.float32))
        elif k in ("prior_log_variance",):
            sd[k] = torch.from_numpy(np.asarray([-1.2], np.float32))
        else:
            sd[k] = torch.from_numpy((rs.standard_normal(shp) * 0.05).astype(np.float32))
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


G21_CASES = {      # single_conv at gain 1: |log p| ~ 1e8 in the untrained net (the reference's own gradients reach 3e9 there)
    "single_conv_mnist_gain1": dict(model_name="single_conv", input_size=[1, 28, 28], input_type="binary", bottleneck=6,
                                    z1_size=294, B=4, C=16, N=40, gain=1.0),
}
# (values, gradient norms, cache rows): what fp32 resolves at that magnitude -- the KL there is a difference of two ~1e8 terms of
# twelve un-normalised residual blocks, so agreement between two correct fp32 implementations is ~1e-4, not 1e-6
MODEL_CASE_TOL = {"single_conv_mnist_gain1": (1e-3, 3e-3, 1e-3)}


G22_CASES = {      # more than 1 024 exemplar rows: the exemplar encoder runs as evae.ops.GatedConvStackFn (pixel-image window kernels)
    "convhvae_stack": dict(model_name="convhvae_2level", input_size=[1, 28, 28], input_type="binary", B=8, C=1600, N=6000),
}


@pytest.mark.parametrize("tag", list(G9_CASES) + list(G19_CASES) + list(G21_CASES))
def test_other_architectures_match_reference_golden(golden, tag):
    g = golden("g9_models" if tag in G9_CASES else "g21_single_conv_gain1" if tag in G21_CASES else "g19_models_geometries")
    cfg = dict(G9_CASES[tag] if tag in G9_CASES else G21_CASES[tag] if tag in G21_CASES else G19_CASES[tag])
    _model_case_against_golden(g, tag, cfg)


def test_conv_stack_operator_matches_reference_golden(golden, gemm_pipe):
    """G22: convhvae_2level (reference models/convHVAE_2level.py:13-97) with 1 600 exemplar draws (~1 400 distinct
    images, encoded once each: models/BaseModel.py::_dedup_draws) and a 6 000-row cache_z -- sizes at which the exemplar encoder is evae.ops.GatedConvStackFn on the window kernels (csrc/evae_conv_win.h), asserted
    by counting the evae_cw_* calls -- against the real reference's loss / RE / KL (1e-4), gradient norms (3e-4) and cache rows, on both
    matrix pipes of the dense layers around it."""
    from evae import _lib
    with _lib.count_calls("evae_cw_") as n:
        _model_case_against_golden(golden("g22_convhvae_stack"), "convhvae_stack", dict(G22_CASES["convhvae_stack"]))
    # training step: first layer + 3 gated window layers forward, the same backward; cache_z: one more forward of the stack
    assert n.get("evae_cw_first_fwd", 0) >= 2 and n.get("evae_cw_fwd_gated", 0) >= 6, n
    assert n.get("evae_cw_bwd_data_gate", 0) >= 3 and n.get("evae_cw_bwd_weight", 0) >= 3 and n.get("evae_cw_first_bwd_weight", 0) >= 1, n


def _model_case_against_golden(g, tag, cfg):
    from utils.utils import importing_model
    tol_v, tol_g, tol_c = MODEL_CASE_TOL.get(tag, (1e-4, 3e-4, 1e-4))
    B, C, N = cfg.pop("B"), cfg.pop("C"), cfg.pop("N")
    gain = cfg.pop("gain", 1.0)
    args = smoke_case.vae_args(number_components=C, training_set_size=N, **cfg)
    model = importing_model(args)(args)
    model.load_state_dict(seeded_state_dict(model, 77, gain))
    model = model.to("cuda")
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
    eps_list = [rs.standard_normal((B, args.z1_size)).astype(np.float32) for _ in range(2)]
    it = {"i": 0}

    def draw(like):
        e = torch.from_numpy(eps_list[it["i"] % 2]).to(like.device).reshape(like.shape); it["i"] += 1
        return e
    model._draw_eps = draw
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    orig = torch.randint
    torch.randint = lambda low=0, high=None, size=None, **kw: torch.from_numpy(ex_idx)
    try:
        model.train(); model.zero_grad(); it["i"] = 0
        loss, RE, KL = model.calculate_loss((torch.from_numpy(x).cuda(), torch.from_numpy(bidx).cuda()), beta=0.6,
                                            average=False, dataset=dataset)
        loss.mean().backward()
    finally:
        torch.randint = orig
    for k, v in (("loss", loss), ("RE", RE), ("KL", KL)):
        assert np.isfinite(v.detach().cpu().numpy()).all(), (tag, k)
        assert rel(v.detach().cpu().numpy(), g["%s_train_%s" % (tag, k)]) < tol_v, (tag, k)
    norms = np.asarray([0.0 if p.grad is None else p.grad.double().norm().item() for _, p in model.named_parameters()])
    ref = g[tag + "_gnorms"]
    assert norms.shape == ref.shape and np.isfinite(norms).all()
    # (G21, |lse| ~ 1e8: the prior's backward keeps its softmax weights normalised through the (max, log sum) token of the merge --
    # r03; with the rounded lse the encoder's and prior_log_variance's gradients came out e^(ulp(lse)) off per row)
    assert np.all(np.abs(norms - ref) <= tol_g * np.maximum(ref, 1e-5)), (tag, (np.abs(norms - ref) / np.maximum(ref, 1e-5)).max())
    model.eval()
    with torch.no_grad():
        it["i"] = 0
        cz, clv = model.cache_z(dataset)
        loss, RE, KL = model.calculate_loss((torch.from_numpy(x).cuda(), None), average=False,
                                            exemplars_embedding=(cz, clv, torch.arange(len(cz))))
    assert rel(cz[:16].cpu().numpy(), g[tag + "_cache_head"]) < tol_c
    for k, v in (("loss", loss), ("RE", RE), ("KL", KL)):
        assert rel(v.cpu().numpy(), g["%s_eval_%s" % (tag, k)]) < tol_v, (tag, k)


@pytest.mark.parametrize("tag,env", [("single_conv", "EVAE_DECODER_STREAM"), ("convhvae_2level", "EVAE_HVAE_TWO_STREAM_CONV")])
def test_two_stream_steps_equal_the_one_stream_steps(tag, env, monkeypatch):
    """the steps of the convolutional models run on two streams (single_conv: the decoder beside the prior's exemplar set,
    models/BaseModel.py::_decoder_beside_prior; convhvae_2level: the batch rows beside the exemplar encoder, models/AbsHModel.py) -- same
    kernels, same per-stream workspaces, so loss / RE / KL and every gradient must be BIT-equal to the one-stream step (a cross-stream
    race would show here).  G9's model cases, both settings of the switch, twice each way round."""
    from utils.utils import importing_model
    cfg = dict(G9_CASES[tag]); B, C, N = cfg.pop("B"), cfg.pop("C"), cfg.pop("N")
    args = smoke_case.vae_args(number_components=C, training_set_size=N, **cfg)
    D = int(np.prod(args.input_size))
    rs = np.random.RandomState(191)
    if args.input_type == "binary":
        data = gi.gray_images(192, N, D); x = (rs.random_sample((B, D)) < 0.3).astype(np.float32)
    else:
        data = ((rs.randint(0, 256, (N, D)) + 0.5) / 256).astype(np.float32); x = ((rs.randint(0, 256, (B, D)) + 0.5) / 256).astype(np.float32)
    bidx = rs.randint(0, N, size=(B, 1)).astype(np.int64)
    ex_idx = rs.randint(0, N, size=(C,)).astype(np.int64)
    eps_list = [rs.standard_normal((B, args.z1_size)).astype(np.float32) for _ in range(2)]
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    outs = []
    for setting in ("1", "0", "1", "0"):
        monkeypatch.setenv(env, setting)
        model = importing_model(args)(args)
        model.load_state_dict(seeded_state_dict(model, 77, 0.35 if tag == "single_conv" else 1.0))
        model = model.to("cuda"); model.train()
        it = {"i": 0}

        def draw(like, it=it):
            e = torch.from_numpy(eps_list[it["i"] % 2]).to(like.device).reshape(like.shape); it["i"] += 1
            return e
        model._draw_eps = draw
        orig = torch.randint
        torch.randint = lambda low=0, high=None, size=None, **kw: torch.from_numpy(ex_idx)
        try:
            loss, RE, KL = model.calculate_loss((torch.from_numpy(x).cuda(), torch.from_numpy(bidx).cuda()), beta=0.6, average=False,
                                                dataset=dataset)
            loss.mean().backward()
        finally:
            torch.randint = orig
        torch.cuda.synchronize()
        outs.append(([t.detach().clone() for t in (loss, RE, KL)], {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    for (v, g_) in outs[1:]:
        for a_, b_ in zip(v, outs[0][0]):
            assert torch.equal(a_, b_)
        assert set(g_) == set(outs[0][1])
        for k in g_:
            if tag == "single_conv":
                assert torch.equal(g_[k], outs[0][1][k]), k
            else:
                # (the 2-level model's q(z2 | x) encoder serves the batch rows AND the exemplar rows: autograd adds the two
                #  contributions to one .grad in the order the streams deliver them -- the same two addends, either order)
                assert rel(g_[k].cpu().numpy(), outs[0][1][k].cpu().numpy()) < 1e-6, k


def test_single_conv_stack_keeps_no_head_weight_after_a_pass():
    """q_z(prior=True) never fetches q_z_logvar's weight from the stack's one-launch weight-norm set; the leftover (a non-leaf
    tensor holding the set's autograd graph) used to stay on the module until the next forward -- kept alive by it, pickled with it, and a
    head called without its stack's forward would have used it.  (copy.deepcopy of the whole model fails in the reference too:
    torch.nn.utils.weight_norm leaves a non-leaf `weight` on every weight-normed layer from construction on.)"""
    import copy, pickle
    from utils.utils import importing_model
    args = smoke_case.vae_args(model_name="single_conv", dataset_name="celeba", input_size=[3, 32, 32], input_type="continuous",
                               continuous=True, use_logit=False, bottleneck=1, z1_size=64, number_components=8, training_set_size=16)
    model = importing_model(args)(args).cuda(); model.train()
    mu, _ = model.q_z(torch.rand(16, 3 * 32 * 32, device="cuda"), prior=True)      # 16 x 32 x 32 pixels: the one-launch set is on
    assert "_head_w" not in model.q_z_layers.__dict__
    mu.sum().backward()
    xm, _ = model.p_x(torch.randn(16, 64, device="cuda"))
    assert "_head_w" not in model.p_x_layers.__dict__
    model.q_z_layers._head_w = {1: mu}                                              # even if one were there: not part of the state
    assert "_head_w" not in model.q_z_layers.__getstate__()
    model.q_z_layers.clear_heads()
    pickle.dumps(model.state_dict())


def test_c5_geometry_matches_reference_golden(golden):
    """BASELINE config 5's geometry through the model API: single_conv (models/fully_conv.py) on 3 x 64 x 64, z1 = 256
    (bottleneck 1), 256-bin logistic likelihood, approximate cache + top-k prior (models/BaseModel.py:256-271) -- loss,
    gradients and the cache refresh in training, then the evaluation path against the whole cache (G20)."""
    from utils.utils import importing_model
    g = golden("g20_c5_geometry")
    B, C, N, k, gain = 4, 24, 48, 3, 0.35
    args = smoke_case.vae_args(model_name="single_conv", dataset_name="celeba", input_size=[3, 64, 64], input_type="continuous",
                               continuous=True, use_logit=False, bottleneck=1, z1_size=256, number_components=C,
                               training_set_size=N, approximate_prior=True, approximate_k=k)
    model = importing_model(args)(args)
    model.load_state_dict(seeded_state_dict(model, 78, gain))
    model = model.to("cuda")
    D = int(np.prod(args.input_size))
    rs = np.random.RandomState(93)
    data = ((rs.randint(0, 256, (N, D)) + 0.5) / 256).astype(np.float32)
    bidx = rs.choice(N, size=(B, 1), replace=False).astype(np.int64)
    x = np.clip(data[bidx[:, 0]] + rs.randint(-6, 7, (B, D)).astype(np.float32) / 256, 0.5 / 256, 255.5 / 256).astype(np.float32)
    cand = rs.choice(N, size=C, replace=False).astype(np.int64)
    eps = rs.standard_normal((B, args.z1_size)).astype(np.float32)
    model._draw_eps = lambda like: torch.from_numpy(eps).to(like.device).reshape(like.shape)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    model.train()
    with torch.no_grad():
        cache = tuple(model.cache_z(dataset))
    assert rel(cache[0].cpu().numpy(), g["cache_before"]) < 1e-4
    orig = torch.randint
    torch.randint = lambda low=0, high=None, size=None, **kw: torch.from_numpy(cand.copy())
    try:
        model.zero_grad()
        loss, RE, KL = model.calculate_loss((torch.from_numpy(x).cuda(), torch.from_numpy(bidx).cuda()), beta=0.7,
                                            average=False, cache=cache, dataset=dataset)
        loss.mean().backward()
    finally:
        torch.randint = orig
    for kk, v in (("loss", loss), ("RE", RE), ("KL", KL)):
        assert rel(v.detach().cpu().numpy(), g[kk]) < 1e-4, kk
    assert rel(cache[0].detach().cpu().numpy(), g["cache_after"]) < 1e-4
    norms = np.asarray([0.0 if p.grad is None else p.grad.double().norm().item() for _, p in model.named_parameters()])
    ref = g["gnorms"]
    assert norms.shape == ref.shape
    assert np.all(np.abs(norms - ref) <= 1e-4 * np.maximum(ref, 1e-5)), (np.abs(norms - ref) / np.maximum(ref, 1e-5)).max()
    model.eval()
    with torch.no_grad():
        cz, clv = model.cache_z(dataset)
        loss, RE, KL = model.calculate_loss((torch.from_numpy(x).cuda(), None), average=False,
                                            exemplars_embedding=(cz, clv, torch.arange(len(cz))))
    for kk, v in (("loss", loss), ("RE", RE), ("KL", KL)):
        assert rel(v.cpu().numpy(), g["eval_" + kk]) < 1e-4, kk


def test_c5_window_operators_match_reference_golden(golden, gemm_pipe):
    """G23: G20's model -- single_conv (reference models/fully_conv.py:12-81) on 3 x 64 x 64, z1 = 256, cache + top-k
    prior (models/BaseModel.py:256-271) -- on a batch of 64 images and 94 re-encoded neighbours: 16 384 pixels even in the 96-channel
    16 x 16 runs, so every residual run is evae.ops.ResStackFn, the convolutions around them PlainConvFn and the weight norm the
    one-launch WeightNormSetFn (asserted by counting the library calls).  Against the real reference: loss / RE / KL 1e-4, gradient
    norms 3e-4, the cache refresh, then the evaluation path; on both matrix pipes."""
    from utils.utils import importing_model
    from evae import _lib
    g = golden("g23_c5_window_size")
    B, C, N, k, gain = 64, 160, 320, 3, 0.35
    args = smoke_case.vae_args(model_name="single_conv", dataset_name="celeba", input_size=[3, 64, 64], input_type="continuous",
                               continuous=True, use_logit=False, bottleneck=1, z1_size=256, number_components=C,
                               training_set_size=N, approximate_prior=True, approximate_k=k)
    model = importing_model(args)(args)
    model.load_state_dict(seeded_state_dict(model, 79, gain))
    model = model.to("cuda")
    D = int(np.prod(args.input_size))
    data, x, bidx, _, eps = gi.g23_inputs(B, C, N, D, args.z1_size)
    cand = g["cand"].astype(np.int64)                     # the generator's candidate draw (an input; chosen for a wide top-k boundary)
    model._draw_eps = lambda like: torch.from_numpy(eps).to(like.device).reshape(like.shape)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    model.train()
    with _lib.count_calls("evae_") as n:
        with torch.no_grad():
            cache = tuple(model.cache_z(dataset))
        assert rel(cache[0][:32].cpu().numpy(), g["cache_before_head"]) < 1e-4
        orig = torch.randint
        torch.randint = lambda low=0, high=None, size=None, **kw: torch.from_numpy(cand.copy())
        try:
            model.zero_grad()
            loss, RE, KL = model.calculate_loss((torch.from_numpy(x).cuda(), torch.from_numpy(bidx).cuda()), beta=0.7,
                                                average=False, cache=cache, dataset=dataset)
            loss.mean().backward()
        finally:
            torch.randint = orig
        torch.cuda.synchronize()
    # the window path ran: 8 residual runs forward in the step (encoder x 2 passes x 2 runs + decoder 2 runs ...), their backward,
    # the plain convolutions, the one-launch weight norm both ways -- and no layer-by-layer residual block
    assert n.get("evae_cw_res_run_fwd", 0) >= 8 and n.get("evae_cw_res_run_bwd", 0) >= 6, n
    assert n.get("evae_cw_plain_fwd", 0) >= 8 and n.get("evae_cw_plain_bwd_data", 0) >= 4, n
    assert n.get("evae_weight_norm_set_fwd", 0) >= 3 and n.get("evae_weight_norm_set_bwd", 0) >= 2, n
    assert "evae_conv2d_cl_fwd_res" not in n and "evae_conv2d_cl_fwd" not in n, n       # no layer-by-layer convolution served it
    for kk, v in (("loss", loss), ("RE", RE), ("KL", KL)):
        assert rel(v.detach().cpu().numpy(), g[kk]) < 1e-4, kk
    assert rel(cache[0].detach().cpu().numpy(), g["cache_after"]) < 1e-4
    norms = np.asarray([0.0 if p.grad is None else p.grad.double().norm().item() for _, p in model.named_parameters()])
    ref = g["gnorms"]
    assert norms.shape == ref.shape
    assert np.all(np.abs(norms - ref) <= 3e-4 * np.maximum(ref, 1e-5)), (np.abs(norms - ref) / np.maximum(ref, 1e-5)).max()
    model.eval()
    with torch.no_grad():
        cz, clv = model.cache_z(dataset)
        loss, RE, KL = model.calculate_loss((torch.from_numpy(x).cuda(), None), average=False,
                                            exemplars_embedding=(cz, clv, torch.arange(len(cz))))
    for kk, v in (("loss", loss), ("RE", RE), ("KL", KL)):
        assert rel(v.cpu().numpy(), g["eval_" + kk]) < 1e-4, kk


@pytest.mark.parametrize("fused", [True, False])
def test_vae_train_step_at_config2_size_matches_oracle(fused):
    """The benchmarked shapes themselves: B = 100, C = 25 000 gathered exemplar rows out of N = 50 000 (985 blocks on the
    XCD remap of the dominant launch, the split-K plans of both big weight gradients) -- loss / RE / KL, every gradient
    and the AdamNormGrad update against the oracle's train step."""
    smoke_case.run(torch, np, orc, B=100, C=25000, N=50000, seed=71, verbose=True, fused=fused)


def test_convhvae_train_step_at_config3_size_on_both_pipes():
    """c3 at its benchmarked size -- one `convhvae_2level` training step over C = 25 000 exemplar images on
    the split-bf16 pipe and on the fp32-MFMA pipe: loss / RE / KL agree to 1e-5, every gradient norm to 1e-3, and the latents
    of the first 8 exemplars (a convolutional encoder is independent per image) match float64 torch on the CPU to 1e-5."""
    import torch.nn.functional as F
    from evae import ops
    from utils.utils import importing_model
    B, Cn, N = 100, 25000, 50000
    data = gi.binary_images(0, N)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    args = smoke_case.vae_args(model_name="convhvae_2level", number_components=Cn, training_set_size=N, batch_size=B,
                               dataset_name="fashion_mnist")
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    model = importing_model(args)(args).cuda()
    model.train()
    rs = np.random.RandomState(33)
    ex_idx = rs.randint(0, N, size=(Cn,)).astype(np.int64)
    eps = [torch.from_numpy(rs.standard_normal((B, 40)).astype(np.float32)).cuda() for _ in range(2)]
    xb = torch.from_numpy(data[:B]).cuda(); ib = torch.arange(B).reshape(-1, 1).cuda()
    ex8 = torch.from_numpy(data[ex_idx[:8]]).cuda()
    res = []
    for pipe in (1, 0):
        ops.gemm_x6_configure(pipe, 2048)
        it = {"i": 0}

        def draw(like):
            e = eps[it["i"] % 2].reshape(like.shape); it["i"] += 1
            return e
        model._draw_eps = draw
        orig = torch.randint
        torch.randint = lambda low=0, high=None, size=None, **kw: torch.from_numpy(ex_idx)
        try:
            model.zero_grad()
            loss, RE, KL = model.calculate_loss((xb, ib), 0.5, average=True, dataset=dataset)
            loss.backward()
            with torch.no_grad():
                z8 = model.q_z(ex8, prior=True)[0]
        finally:
            torch.randint = orig
            ops.gemm_x6_configure(1, 2048)
        res.append((np.asarray([loss.item(), RE.item(), KL.item()]),
                    np.asarray([p.grad.double().norm().item() for p in model.parameters() if p.grad is not None]), z8.cpu().numpy()))
    assert np.isfinite(res[0][0]).all() and rel(res[0][0], res[1][0]) < 1e-5
    assert np.all(np.abs(res[0][1] - res[1][1]) <= 1e-3 * np.maximum(res[1][1], 1e-6))
    # float64 reference of q(z2|x) (reference models/convHVAE_2level.py:21-46 through utils/nn.py:72-97) on the first 8 exemplars
    sd = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
    h = ex8.double().cpu().reshape(-1, 1, 28, 28)
    for li, (st, pd) in enumerate(((1, 3), (2, 1), (1, 2), (2, 1), (1, 1))):
        pre = "q_z_layers.%d." % li
        h = F.conv2d(h, sd[pre + "h.weight"], sd[pre + "h.bias"], st, pd) * \
            torch.sigmoid(F.conv2d(h, sd[pre + "g.weight"], sd[pre + "g.bias"], st, pd))
    z64 = F.linear(h.reshape(8, -1), sd["q_z_mean.linear.weight"], sd["q_z_mean.linear.bias"]).numpy()
    for r in res:
        assert rel(r[2], z64) < 1e-5


def test_approximate_prior_matches_reference_golden(golden):
    """kNN-pruned exemplar prior (reference models/BaseModel.py:256-271): loss, gradients and the in-place cache
    refresh against the real reference."""
    g = golden("g10_approx")
    B, C, N, k = 16, 300, 1000, 10
    args = smoke_case.vae_args(number_components=C, training_set_size=N, approximate_prior=True, approximate_k=k)
    model, p = smoke_case.build_model(torch, np, orc, args)
    data = gi.gray_images(63, N).astype(np.float32)
    rs = np.random.RandomState(64)
    bidx = rs.choice(N, size=(B, 1), replace=False).astype(np.int64)
    x = (rs.random_sample((B, 784)) < np.clip(data[bidx[:, 0]] + 0.1, 0, 1)).astype(np.float32)
    eps = rs.standard_normal((B, 40)).astype(np.float32)
    cand = rs.choice(N, size=C, replace=False).astype(np.int64)    # distinct candidates: no exact distance ties
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    model._draw_eps = lambda like: torch.from_numpy(eps).to(like.device)
    model.train()
    with torch.no_grad():
        cache = tuple(model.cache_z(dataset))
    assert rel(cache[0].cpu().numpy(), g["cache_before"]) < 1e-5
    orig = torch.randint
    torch.randint = lambda low=0, high=None, size=None, **kw: torch.from_numpy(cand.copy())
    try:
        model.zero_grad()
        loss, RE, KL = model.calculate_loss((torch.from_numpy(x).cuda(), torch.from_numpy(bidx).cuda()), beta=0.8,
                                            average=False, cache=cache, dataset=dataset)
        loss.mean().backward()
    finally:
        torch.randint = orig
    for kk, v in (("loss", loss), ("RE", RE), ("KL", KL)):
        assert rel(v.detach().cpu().numpy(), g[kk]) < 1e-4, kk
    assert rel(cache[0].detach().cpu().numpy(), g["cache_after"]) < 1e-5      # refreshed rows identical
    norms = np.asarray([prm.grad.double().norm().item() for _, prm in model.named_parameters()])
    assert np.all(np.abs(norms - g["gnorm"]) <= 1e-3 * np.maximum(g["gnorm"], 1e-5))


def test_evaluation_loops_match_reference_golden(golden):
    """utils.evaluation.evaluate_loss / calculate_likelihood (ELBO over a loader, IWAE test log p(x)) vs the
    reference's loops on identical data, weights and eps stream."""
    from models.VAE import VAE
    from utils.evaluation import evaluate_loss, calculate_likelihood
    g = golden("g11_eval")
    N, NT, S = 400, 12, 50
    args = smoke_case.vae_args(number_components=N, training_set_size=N, batch_size=5)
    model, _ = smoke_case.build_model(torch, np, orc, args)
    data = gi.binary_images(71, N)
    test = gi.binary_images(72, NT)
    train_ds = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    test_ds = torch.utils.data.TensorDataset(torch.from_numpy(test), torch.zeros(NT))
    loader = torch.utils.data.DataLoader(test_ds, batch_size=5, shuffle=False)
    eps_rs = np.random.RandomState(73)
    model._draw_eps = lambda like: torch.from_numpy(eps_rs.standard_normal(tuple(like.shape)).astype(np.float32)).to(like.device)
    with torch.no_grad():
        elbo, re, kl = evaluate_loss(args, model, loader, dataset=train_ds)
        model.eval()
        cz, clv = model.cache_z(train_ds)
        ll = calculate_likelihood(args, model, loader, S=S, exemplars_embedding=(cz, clv, torch.arange(len(cz))))
    assert rel(np.asarray([elbo, re, kl]), g["elbo"]) < 1e-4
    assert abs(ll - g["ll"][0]) <= 1e-4 * abs(g["ll"][0])


def test_training_continues_from_a_reference_checkpoint(golden):
    """Load the checkpoint the reference wrote (two steps in), take the third step with the HIP optimizer: parameters
    equal the reference's own third step (the step count / bias correction travelled with the checkpoint)."""
    import os
    from models.VAE import VAE
    from utils.optimizer import AdamNormGrad
    from utils.utils import load_model
    g = golden("g12_checkpoint")
    args = smoke_case.vae_args(input_size=[1, 8, 8], hidden_size=16, z1_size=8, z2_size=8, number_components=10,
                               training_set_size=50)
    model = VAE(args).cuda()
    opt = AdamNormGrad(model.parameters(), lr=5e-4)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_checkpoint.pth")
    load_model(path, model, opt)
    for n, p in model.named_parameters():
        assert p.is_cuda and opt.state[p]["exp_avg"].is_cuda
        p.grad = torch.from_numpy(g["s3_grad_" + n]).cuda()
    opt.step()
    for n, p in model.named_parameters():
        assert rel(p.detach().cpu().numpy(), g["after_" + n]) < 1e-6, n


def test_epoch_loops_match_reference_golden(golden, tmp_path):
    """utils.training.train_one_epoch (two epochs, hipGraph step for the full batches, eager for the partial last one),
    utils.knn_on_latent.report_knn_on_latent and utils.evaluation.final_evaluation against the reference's own loops on
    the same data, weights and exemplar draws (tools/gen_goldens.py::g13; z = mean in both trees)."""
    from utils.optimizer import AdamNormGrad
    from utils.training import train_one_epoch
    from utils.knn_on_latent import report_knn_on_latent
    from utils.evaluation import final_evaluation
    from utils.utils import save_model
    g = golden("g13_loops")
    N, NV, B, C = 200, 64, 32, 50
    args = smoke_case.vae_args(number_components=C, training_set_size=N, batch_size=B, dynamic_binarization=False,
                               warmup=100, S=20)
    model, _ = smoke_case.build_model(torch, np, orc, args)
    model._draw_eps = lambda like: torch.zeros_like(like)
    mk = lambda seed, n: torch.from_numpy(gi.binary_images(seed, n))
    train_ds = torch.utils.data.TensorDataset(mk(81, N), torch.arange(N).reshape(-1, 1), torch.arange(N) % 10)
    val_ds = torch.utils.data.TensorDataset(mk(82, NV), (torch.arange(NV) * 3) % 10)
    test_ds = torch.utils.data.TensorDataset(mk(83, NV), (torch.arange(NV) * 7) % 10)
    L = lambda ds: torch.utils.data.DataLoader(ds, batch_size=B, shuffle=False)
    train_loader, val_loader, test_loader = L(train_ds), L(val_ds), L(test_ds)
    opt = AdamNormGrad(model.parameters(), lr=5e-4)
    torch.manual_seed(130)
    r1 = train_one_epoch(1, args, train_loader, model, opt)
    r2 = train_one_epoch(2, args, train_loader, model, opt)
    runners = list(model._graphed_steps.values())
    assert len(runners) == 1 and runners[0].graph is not None and runners[0].by_index     # the full batches were replays
    assert rel(np.asarray(r1), g["epoch1"]) < 1e-4
    assert rel(np.asarray(r2), g["epoch2"]) < 1e-4
    for n, p in model.named_parameters():
        assert abs(p.detach().double().norm().item() - float(g["norm_" + n])) <= 1e-4 * max(float(g["norm_" + n]), 1e-3), n
        assert abs(p.detach().double().sum().item() - float(g["sum_" + n])) <= 2e-4 * max(float(g["norm_" + n]), 1e-3), n
    model.eval()
    for flag, key in ((True, "knn_val"), (False, "knn_test")):
        d = {"3": [], "5": [], "7": [], "15": []}
        report_knn_on_latent(train_loader, val_loader, test_loader, model, "", d, args, val=flag)
        assert np.array_equal(np.asarray([d[k][0] for k in ("3", "5", "7", "15")]), g[key]), key
    out = str(tmp_path) + "/"
    save_model(out + "c.tmp", out + "best.model", {'epoch': 2, 'state_dict': model.state_dict(), 'optimizer': opt.state_dict(),
                                                  'best_loss': 0.0, 'e': 0})
    final_evaluation(train_loader, test_loader, val_loader, out + "best.model", model, opt, args, out)
    got = [float(torch.load(out + "vae." + k, weights_only=False)) for k in ("test_log_likelihood", "test_loss", "test_re", "test_kl")]
    assert rel(np.asarray(got), g["final"]) < 1e-4
    assert open(out + "vae_experiment_log.txt").read().split("\n")[0] == "FINAL EVALUATION ON TEST SET"


def test_end_to_end_run_on_grey_data_with_dynamic_binarisation(tmp_path, monkeypatch):
    """What density_estimation.py does, with this build's modules only: load_dataset from IDX files on disk (grey-level
    training rows, dynamically binarised per step, evaluation splits binarised once), importing_model, AdamNormGrad,
    epochs of train_one_epoch (shuffled CPU batches -> the captured step gathers them by index and binarises them in its
    prologue launch; the exemplars stay grey, as in the reference) + evaluate_loss + save_model, then final_evaluation."""
    import struct
    from argparse import Namespace
    from utils.load_data.data_loader_instances import load_dataset
    from utils.utils import importing_model, save_model
    from utils.optimizer import AdamNormGrad
    from utils.training import train_one_epoch
    from utils.evaluation import evaluate_loss, final_evaluation
    raw = tmp_path / "datasets" / "dynamic_mnist" / "MNIST" / "raw"
    raw.mkdir(parents=True)
    rs = np.random.RandomState(11)
    protos = (gi.gray_images(5, 10) * 255.0).reshape(10, 28, 28)               # ten grey prototypes

    def split(n):
        lab = rs.randint(0, 10, n)
        img = np.clip(protos[lab] + rs.normal(0, 12, (n, 28, 28)), 0, 255).astype(np.uint8)
        return img, lab.astype(np.uint8)

    def idx(name, arr):
        with open(raw / name, "wb") as f:
            f.write(struct.pack(">HBB", 0, 8, arr.ndim) + struct.pack(">" + "I" * arr.ndim, *arr.shape) + arr.tobytes())
    xtr, ytr = split(700); xte, yte = split(100)
    idx("train-images-idx3-ubyte", xtr); idx("train-labels-idx1-ubyte", ytr)
    idx("t10k-images-idx3-ubyte", xte); idx("t10k-labels-idx1-ubyte", yte)
    monkeypatch.chdir(tmp_path)
    args = smoke_case.vae_args(dataset_name="dynamic_mnist", number_components=200, batch_size=50, test_batch_size=50, S=20,
                               warmup=2, use_training_data_init=0, training_set_size=None)
    torch.manual_seed(4); np.random.seed(4)
    train_loader, val_loader, test_loader, args = load_dataset(args, training_num=600)
    assert args.dynamic_binarization is True and args.input_type == "binary" and args.training_set_size == 600
    grey = train_loader.dataset.tensors[0]
    assert float(((grey > 0.05) & (grey < 0.95)).float().mean()) > 0.2          # the training rows really are grey
    model = importing_model(args)(args).cuda()
    opt = AdamNormGrad(model.parameters(), lr=2e-3)
    out = str(tmp_path) + "/"
    val_hist, train_hist = [], []
    for epoch in range(1, 5):
        tr = train_one_epoch(epoch, args, train_loader, model, opt)
        with torch.no_grad():
            va = evaluate_loss(args, model, val_loader, dataset=train_loader.dataset)
        assert all(np.isfinite(tr)) and all(np.isfinite(va))
        train_hist.append(tr[1]); val_hist.append(va[0])           # reconstruction error / validation ELBO
        save_model(out + "ck.tmp", out + "best.model", {'epoch': epoch, 'state_dict': model.state_dict(),
                                                        'optimizer': opt.state_dict(), 'best_loss': va[0], 'e': 0})
    runner = list(model._graphed_steps.values())[0]
    assert runner.graph is not None and runner.by_index and runner.binarize       # 12 captured steps per epoch
    assert train_hist[-1] < train_hist[0] and val_hist[-1] < val_hist[0]          # it learns
    from utils.knn_on_latent import report_knn_on_latent
    knn = {"3": [], "7": []}
    model.eval()
    report_knn_on_latent(train_loader, val_loader, test_loader, model, out, knn, args, val=True)
    assert knn["3"][0] > 60.0 and knn["7"][0] > 60.0            # ten prototype classes: the latent space separates them (chance: 10 %)
    with torch.no_grad():
        final_evaluation(train_loader, test_loader, val_loader, out + "best.model", model, opt, args, out)
    ll = float(torch.load(out + "vae.test_log_likelihood", weights_only=False))
    elbo = float(torch.load(out + "vae.test_loss", weights_only=False))
    assert np.isfinite(ll) and np.isfinite(elbo) and ll <= elbo + 1e-3             # the IWAE bound is at least as tight


@pytest.mark.parametrize("prior", ["standard", "vampprior"])
def test_other_priors_match_reference_golden(golden, prior):
    """models.BaseModel.log_p_z for prior = 'standard' and 'vampprior' (the reference's default) behind the same API:
    calculate_loss, its gradients and evaluate_loss against the reference on identical weights, batch and eps."""
    from models.VAE import VAE
    from utils.evaluation import evaluate_loss
    g = golden("g15_priors")
    B, D = 16, 64
    args = smoke_case.vae_args(prior=prior, input_size=[1, 8, 8], hidden_size=32, z1_size=8, z2_size=8, number_components=20,
                               training_set_size=100, batch_size=B, pseudoinputs_mean=0.05, pseudoinputs_std=0.01,
                               use_training_data_init=False)
    model = VAE(args).cuda()
    sd = {k[len(prior) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prior + "_sd_")}
    assert set(sd) == set(model.state_dict().keys())
    model.load_state_dict(sd)
    model.train()
    eps = torch.from_numpy(g["eps"]).cuda()
    model._draw_eps = lambda like: eps[:like.shape[0]]
    x = torch.from_numpy(gi.binary_images(151, B, D)).cuda()
    loss, RE, KL = model.calculate_loss((x, torch.arange(B).reshape(-1, 1).cuda()), 0.7, average=False)
    loss.mean().backward()
    for name, t in (("loss", loss), ("RE", RE), ("KL", KL)):
        assert rel(t.detach().cpu().numpy(), g[prior + "_" + name]) < 1e-4, name
    for n, p in model.named_parameters():
        ref = float(g[prior + "_gnorm_" + n])
        got = 0.0 if p.grad is None else p.grad.double().norm().item()
        assert abs(got - ref) <= 1e-3 * max(ref, 1e-6), n
    model.eval()
    model._draw_eps = lambda like: torch.zeros_like(like)
    test = torch.from_numpy(gi.binary_images(152, 24, D))
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(test, torch.zeros(24)), batch_size=8)
    with torch.no_grad():
        ev = evaluate_loss(args, model, loader, dataset=None)
    assert rel(np.asarray(ev), g[prior + "_eval"]) < 1e-4


@pytest.mark.parametrize("tag,kw", [("no_attention", dict(no_attention=True)), ("no_mask", dict(no_mask=True)), ("plain", dict())])
def test_option_flags_match_reference_golden(golden, tag, kw):
    """args.no_attention (GatedDense degenerates to ReLU(h), modular path) and args.no_mask (no leave-one-out mask, fused
    path) against the reference: loss, RE, KL and gradient norms on identical weights, batch, eps and exemplar draw."""
    from models.VAE import VAE
    g = golden("g16_options")
    B, D, N, C = 16, 64, 120, 40
    args = smoke_case.vae_args(input_size=[1, 8, 8], hidden_size=32, z1_size=8, z2_size=8, number_components=C,
                               training_set_size=N, **kw)
    model = VAE(args).cuda()
    model.load_state_dict({k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "_sd_")})
    model.train()
    eps = torch.from_numpy(g["eps"]).cuda()
    model._draw_eps = lambda like: eps[:like.shape[0]]
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(gi.gray_images(161, N, D)), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    x = torch.from_numpy(gi.binary_images(162, B, D)).cuda()
    torch.manual_seed(165)
    loss, RE, KL = model.calculate_loss((x, torch.from_numpy(g["bidx"]).cuda()), 0.6, average=False, dataset=dataset)
    loss.mean().backward()
    for name, t in (("loss", loss), ("RE", RE), ("KL", KL)):
        assert rel(t.detach().cpu().numpy(), g[tag + "_" + name]) < 1e-4, name
    for n, p in model.named_parameters():
        ref = float(g[tag + "_gnorm_" + n])
        got = 0.0 if p.grad is None else p.grad.double().norm().item()
        assert abs(got - ref) <= 1e-3 * max(ref, 1e-6), n


@pytest.mark.parametrize("tag,model_name", [("vae_gray", "vae"), ("hvae_gray", "hvae_2level")])
def test_grey_inputs_through_mlp_models_match_reference_golden(golden, tag, model_name):
    """input_type = 'gray' (continuous=True): clamped means, decoder_logstd / p_x_logvar head and the 256-bin discretised
    logistic likelihood through vae and hvae_2level, against the reference on identical weights, batch, eps, exemplars."""
    from utils.utils import importing_model
    g = golden("g17_grey_mlp")
    B, D, N, C = 12, 64, 90, 30
    args = smoke_case.vae_args(model_name=model_name, input_type="gray", continuous=True, input_size=[1, 8, 8], hidden_size=32,
                               z1_size=8, z2_size=8, number_components=C, training_set_size=N)
    model = importing_model(args)(args).cuda()
    sd = {k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "_sd_")}
    assert list(sd) and set(sd) == set(model.state_dict().keys())
    model.load_state_dict(sd)
    model.train()
    rs = np.random.RandomState(174)
    model._draw_eps = lambda like: torch.from_numpy(rs.standard_normal(tuple(like.shape)).astype(np.float32)).to(like.device)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(gi.gray_images(171, N, D)), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    x = torch.from_numpy(np.clip(gi.gray_images(172, B, D) + 0.002, 0.0, 1.0).astype(np.float32)).cuda()
    torch.manual_seed(175)
    loss, RE, KL = model.calculate_loss((x, torch.from_numpy(g["bidx"]).cuda()), 0.8, average=False, dataset=dataset)
    loss.mean().backward()
    for name, t in (("loss", loss), ("RE", RE), ("KL", KL)):
        assert rel(t.detach().cpu().numpy(), g[tag + "_" + name]) < 1e-4, name
    for n, p in model.named_parameters():
        ref = float(g[tag + "_gnorm_" + n])
        got = 0.0 if p.grad is None else p.grad.double().norm().item()
        assert abs(got - ref) <= 1e-4 * max(ref, 1e-6), n


@pytest.mark.parametrize("model_name", ["vae", "hvae_2level", "convhvae_2level"])
def test_generation_helpers_run(model_name):
    """generate_x / reference_based_generation_x / reconstruct_x / generate_z (reference models/BaseModel.py:130-200): the
    helpers the reference's evaluation and analysis scripts call.  Random by construction: shapes, ranges, finiteness."""
    from utils.utils import importing_model
    N = 300
    args = smoke_case.vae_args(model_name=model_name, number_components=50, training_set_size=N)
    torch.manual_seed(2)
    model = importing_model(args)(args).cuda().eval()
    data = torch.from_numpy(gi.binary_images(9, N))
    dataset = torch.utils.data.TensorDataset(data, torch.arange(N).reshape(-1, 1), torch.zeros(N))
    with torch.no_grad():
        gx = model.generate_x(N=9, dataset=dataset)
        rx = model.reference_based_generation_x(N=6, reference_image=data[:1].cuda())
        rec = model.reconstruct_x(data[:7].cuda())
        gz = model.generate_z(N=9, dataset=dataset)
    zdim = args.z2_size if "hvae" in model_name else args.z1_size
    assert gx.shape == (9, 784) and rx.shape == (6, 784) and rec.shape == (7, 784) and gz.shape == (9, zdim)
    for t in (gx, rx, rec):
        assert bool(torch.isfinite(t).all()) and float(t.min()) >= 0.0 and float(t.max()) <= 1.0


# combinations the reference itself cannot run fail here too, loudly (same place, same reason): index -> why
_MATRIX_FAILS_LIKE_REFERENCE = {
    3: "top-k of approximate_k = 10 over 5 candidates (torch.topk raises in the reference, evae_pairdist_topk here)",
    6: "AbsModel.p_x has no x_logvar for continuous inputs with use_logit=True (unbound local in the reference)",
    7: "q_z reshapes to z1_size although the hvae encoder emits z2_size: z1 != z2 is not runnable",
    9: "GatedConv2d(no_attention=True) builds no `g` but its forward uses it",
}
_MATRIX = [
    # (model, input_size, input_type, extra args, batch, exemplars)
    ("vae", [1, 28, 28], "binary", dict(), 1, 7),
    ("vae", [1, 28, 28], "binary", dict(), 257, 33),
    ("vae", [1, 28, 28], "binary", dict(z1_size=3), 5, 40),
    ("vae", [1, 28, 28], "binary", dict(approximate_prior=True, approximate_k=10), 16, 5),      # fewer candidates than k
    ("vae", [1, 28, 28], "binary", dict(approximate_prior=True, approximate_k=3), 16, 64),
    ("vae", [1, 28, 20], "gray", dict(continuous=True, dataset_name="freyfaces"), 9, 30),
    ("vae", [3, 32, 32], "continuous", dict(continuous=True, use_logit=True, dataset_name="cifar10"), 6, 20),
    ("hvae_2level", [1, 28, 28], "binary", dict(z1_size=24, z2_size=56), 11, 50),
    ("hvae_2level", [1, 28, 28], "binary", dict(approximate_prior=True, z2_size=40), 8, 40),
    ("convhvae_2level", [1, 28, 28], "binary", dict(no_attention=True), 3, 12),
    ("convhvae_2level", [3, 32, 32], "continuous", dict(continuous=True, dataset_name="svhn"), 2, 10),
    ("single_conv", [1, 28, 28], "binary", dict(bottleneck=6, z1_size=294), 2, 9),
    ("single_conv", [3, 32, 32], "continuous", dict(continuous=True, bottleneck=2, z1_size=128, dataset_name="cifar10"), 2, 9),
]


@pytest.mark.parametrize("case", range(len(_MATRIX)))
def test_configuration_matrix_trains_one_step(case):
    """One training step (loss, backward, AdamNormGrad) for a spread of model x geometry x option combinations the
    reference's argument parser admits: nothing may be refused by a kernel-side limit, everything stays finite."""
    from utils.utils import importing_model
    from utils.optimizer import AdamNormGrad
    name, isz, itype, extra, B, C = _MATRIX[case]
    N = 90
    D = int(np.prod(isz))
    kw = dict(model_name=name, input_size=isz, input_type=itype, number_components=C, training_set_size=N, batch_size=B)
    kw.update(extra)
    args = smoke_case.vae_args(**kw)
    torch.manual_seed(100 + case)
    model = importing_model(args)(args).cuda()
    model.train()
    opt = AdamNormGrad(model.parameters(), lr=1e-4)
    rs = np.random.RandomState(case)
    if itype == "binary":
        data = (rs.random_sample((N, D)) < 0.3).astype(np.float32)
    else:
        data = ((rs.randint(0, 256, (N, D)) + 0.5) / 256).astype(np.float32)
        if extra.get("use_logit"):
            data = np.log(data / (1 - data)).astype(np.float32)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    idx = torch.from_numpy(rs.randint(0, N, (B, 1)).astype(np.int64))
    x = torch.from_numpy(data[idx[:, 0].numpy()]).cuda()
    cache = None
    if extra.get("approximate_prior"):
        with torch.no_grad():
            cache = tuple(model.cache_z(dataset))
    before = [p.detach().clone() for p in model.parameters()]
    opt.zero_grad()
    if case in _MATRIX_FAILS_LIKE_REFERENCE:
        with pytest.raises(Exception):
            loss, RE, KL = model.calculate_loss((x, idx.cuda()), 0.5, average=True, dataset=dataset, cache=cache)
            loss.backward()
        return
    loss, RE, KL = model.calculate_loss((x, idx.cuda()), 0.5, average=True, dataset=dataset, cache=cache)
    loss.backward()
    with_grad = [p.grad is not None for p in model.parameters()]      # fully_conv carries BatchNorm modules it never calls
    opt.step()
    assert all(bool(torch.isfinite(t).all()) for t in (loss, RE, KL))
    moved = 0
    for p, b, has in zip(model.parameters(), before, with_grad):
        assert bool(torch.isfinite(p).all())
        moved += int(has and not torch.equal(p.detach(), b))
    assert moved >= 0.9 * sum(with_grad) and sum(with_grad) > 0.6 * len(before)


@pytest.mark.parametrize("name,isz,itype,extra", [
    ("hvae_2level", [1, 28, 28], "binary", dict()),
    ("convhvae_2level", [1, 28, 28], "binary", dict()),
    ("convhvae_2level", [3, 32, 32], "continuous", dict(continuous=True, dataset_name="cifar10")),
    ("single_conv", [3, 32, 32], "continuous", dict(continuous=True, bottleneck=2, z1_size=128, dataset_name="cifar10")),
])
def test_evaluation_loops_run_for_every_architecture(name, isz, itype, extra):
    """evaluate_loss, calculate_likelihood (several images per pass) and report_knn_on_latent through the hierarchical and
    convolutional models: shapes and reshapes of the evaluation side, finite results, IWAE bound <= ELBO bound."""
    from utils.utils import importing_model
    from utils.evaluation import evaluate_loss, calculate_likelihood
    from utils.knn_on_latent import report_knn_on_latent
    N, NE, B = 96, 24, 12
    D = int(np.prod(isz))
    args = smoke_case.vae_args(model_name=name, input_size=isz, input_type=itype, number_components=N, training_set_size=N,
                               batch_size=B, **extra)
    torch.manual_seed(21)
    model = importing_model(args)(args).cuda().eval()
    rs = np.random.RandomState(3)
    mk = (lambda n: (rs.random_sample((n, D)) < 0.3).astype(np.float32)) if itype == "binary" else \
         (lambda n: ((rs.randint(0, 256, (n, D)) + 0.5) / 256).astype(np.float32))
    train = torch.utils.data.TensorDataset(torch.from_numpy(mk(N)), torch.arange(N).reshape(-1, 1), torch.arange(N) % 10)
    val = torch.utils.data.TensorDataset(torch.from_numpy(mk(NE)), torch.arange(NE) % 10)
    test = torch.utils.data.TensorDataset(torch.from_numpy(mk(NE)), (torch.arange(NE) * 3) % 10)
    L = lambda ds: torch.utils.data.DataLoader(ds, batch_size=B, shuffle=False)
    with torch.no_grad():
        elbo, re, kl = evaluate_loss(args, model, L(test), dataset=train)
        cz, clv = model.cache_z(train)
        emb = (cz, clv, torch.arange(len(cz)))
        ll = calculate_likelihood(args, model, L(test), S=40, exemplars_embedding=emb)
    assert all(np.isfinite(v) for v in (elbo, re, kl, ll))
    assert ll <= elbo + 0.02 * abs(elbo)                     # 40 importance samples already tighten the bound (or tie it)
    d = {"3": [], "7": []}
    report_knn_on_latent(L(train), L(val), L(test), model, "", d, args, val=True)
    report_knn_on_latent(L(train), L(val), L(test), model, "", d, args, val=False)
    assert all(len(v) == 2 and all(0.0 <= a <= 100.0 for a in v) for v in d.values())


def test_fused_backward_redoes_the_byte_gather_when_another_step_used_the_workspace():
    """The fused step transposes the gathered byte rows during its FORWARD pass into a named workspace (evae/fused_vae.py);
    a second model's forward on another dataset in between overwrites it, and the first backward must notice (generation
    counter) and redo the gather: same gradients as without the interleaved step."""
    B, C, N = 32, 300, 800
    def build(seed):
        args = smoke_case.vae_args(number_components=C, training_set_size=N)
        torch.manual_seed(3)
        model, _ = smoke_case.build_model(torch, np, orc, args)
        _, bidx, _, eps, ex_idx = smoke_case.make_case(np, B, C, N, seed, gi)
        data = np.ascontiguousarray(gi.binary_images(seed, N).reshape(N, -1).astype(np.float32))     # k/255 data: the byte store applies
        x = data[bidx.reshape(-1)]
        dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
        model._draw_eps = lambda like: torch.from_numpy(eps).to(like.device)
        model._use_fused = True
        model.train()
        return model, dataset, x, bidx, ex_idx

    def loss_of(m, ds, x, bidx, ex_idx):
        orig = torch.randint
        torch.randint = lambda low=0, high=None, size=None, **kw: torch.from_numpy(ex_idx)
        try:
            return m.calculate_loss((torch.from_numpy(x).cuda(), torch.from_numpy(bidx).cuda()), beta=0.5, average=True, dataset=ds)[0]
        finally:
            torch.randint = orig

    m1, ds1, x1, b1, e1 = build(71)
    assert m1.resident_u8(ds1, B) is not None            # the byte store (and with it the forward-time gather) is in play
    l = loss_of(m1, ds1, x1, b1, e1); l.backward()
    ref = {n: p.grad.clone() for n, p in m1.named_parameters()}
    m1.zero_grad(set_to_none=True)
    m2, ds2, x2, b2, e2 = build(72)
    l1 = loss_of(m1, ds1, x1, b1, e1)
    l2 = loss_of(m2, ds2, x2, b2, e2)                    # same workspace names, other rows
    l1.backward()
    l2.backward()
    for n, p in m1.named_parameters():
        assert torch.equal(p.grad, ref[n]), n


def test_two_level_step_fused_head_functions_at_c4_size():
    """hvae_2level at the benchmarked size (BASELINE configs[3]: 11 500 exemplars, batch 100): the two-stream step with each pair
    of heads + sample + density as one Function and the loss assembly as one (models/AbsHModel.py, evae.ops.HeadsReparamFn /
    ElboFn) against the same step through the separate modules, same noise: loss / RE / KL to 1e-5, every gradient norm to 1e-4."""
    from models import AbsHModel
    from utils.utils import importing_model
    from argparse import Namespace
    N, C, B = 23000, 11500, 100
    data = torch.from_numpy(gi.binary_images(9, N))
    dataset = torch.utils.data.TensorDataset(data, torch.arange(N).reshape(-1, 1), torch.zeros(N))
    args = Namespace(prior="exemplar_prior", input_type="binary", input_size=[1, 28, 28], hidden_size=300, z1_size=40, z2_size=40,
                     model_name="hvae_2level", device="cuda", number_components=C, training_set_size=N, approximate_prior=False,
                     approximate_k=10, no_mask=False, no_attention=False, same_variational_var=False, use_logit=False, lambd=1e-4,
                     bottleneck=1, dataset_name="dynamic_mnist", continuous=False, batch_size=B, dynamic_binarization=False,
                     warmup=100, S=5000)
    torch.manual_seed(21)
    model = importing_model(args)(args).cuda()
    model.train()
    x = data[:B].cuda(); idx = torch.arange(B, device="cuda").reshape(-1, 1)
    g = torch.Generator(device="cuda")
    out = []
    try:
        for fused in (True, False):
            AbsHModel._FUSED_HEADS = fused
            g.manual_seed(5); model._eps_generator = g          # the same z2 / z1 noise both times
            torch.manual_seed(77)                                # the same exemplar draw
            model.zero_grad(set_to_none=True)
            loss, RE, KL = model.calculate_loss((x, idx), 0.5, average=True, dataset=dataset)
            loss.backward()
            torch.cuda.synchronize()
            out.append(([float(loss.detach()), float(RE.detach()), float(KL.detach())], {k: float(p.grad.double().norm()) for k, p in model.named_parameters()
                                                              if p.grad is not None}))
    finally:
        AbsHModel._FUSED_HEADS = True
        model._eps_generator = None
    (v1, g1), (v0, g0) = out
    assert np.isfinite(v1).all() and set(g1) == set(g0) and len(g1) >= 40
    for a, b in zip(v1, v0):
        assert abs(a - b) <= 1e-5 * max(abs(b), 1.0), (v1, v0)
    for k in g0:
        assert abs(g1[k] - g0[k]) <= 1e-4 * max(g0[k], 1e-12), (k, g1[k], g0[k])


def test_two_level_step_at_c4_size_matches_the_oracle():
    """hvae_2level at the benchmarked size (BASELINE configs[3]: 11 500 exemplars, batch 100): per-sample loss / RE / KL of the
    two-stream training step against an fp64 restatement of reference models/AbsHModel.py:13-106 + models/HVAE_2level.py:15-66
    composed from the oracle's primitives (gated_dense, linear, hardtanh, log_normal_diag, log_bernoulli, log_p_z) on the same
    weights, noise and exemplar draw -- 1e-4 relative, north_star's bar (c4 was only compared with itself)."""
    from utils.utils import importing_model
    from argparse import Namespace
    N, C, B = 23000, 11500, 100
    data_np = gi.binary_images(9, N)
    data = torch.from_numpy(data_np)
    dataset = torch.utils.data.TensorDataset(data, torch.arange(N).reshape(-1, 1), torch.zeros(N))
    args = Namespace(prior="exemplar_prior", input_type="binary", input_size=[1, 28, 28], hidden_size=300, z1_size=40, z2_size=40,
                     model_name="hvae_2level", device="cuda", number_components=C, training_set_size=N, approximate_prior=False,
                     approximate_k=10, no_mask=False, no_attention=False, same_variational_var=False, use_logit=False, lambd=1e-4,
                     bottleneck=1, dataset_name="dynamic_mnist", continuous=False, batch_size=B, dynamic_binarization=False,
                     warmup=100, S=5000)
    torch.manual_seed(21)
    model = importing_model(args)(args).cuda()
    model.train()
    rs = np.random.RandomState(4)
    xi = rs.choice(N, size=B, replace=False).astype(np.int64)
    x_np = data_np[xi]
    eps2 = rs.standard_normal((B, 40)).astype(np.float32); eps1 = rs.standard_normal((B, 40)).astype(np.float32)
    ex_idx = rs.randint(0, N, size=C).astype(np.int64)
    ex_idx[:5] = xi[:5]                                        # leave-one-out hits
    draws = [eps2, eps1]
    model._draw_eps = lambda like: torch.from_numpy(draws.pop(0)).to(like.device)
    orig = torch.randint
    torch.randint = lambda low=0, high=None, size=None, **kw: torch.from_numpy(ex_idx.copy())
    beta = 0.6
    try:
        loss, RE, KL = model.calculate_loss((torch.from_numpy(x_np).cuda(), torch.from_numpy(xi).reshape(-1, 1).cuda()), beta,
                                            average=False, dataset=dataset)
    finally:
        torch.randint = orig
    # ---- fp64 restatement from the oracle's primitives ----
    P = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in model.state_dict().items()}

    def gated(x, name):
        y, _ = orc.gated_dense(x, P[name + ".h.weight"], P[name + ".h.bias"], P[name + ".g.weight"], P[name + ".g.bias"])
        return y

    def heads(t, mean, logvar):
        mu = orc.linear(t, P[mean + ".weight"], P[mean + ".bias"])
        lv = orc.hardtanh(orc.linear(t, P[logvar + ".linear.weight"], P[logvar + ".linear.bias"]), -6.0, 2.0)
        return mu, lv
    x64 = x_np.astype(np.float64)
    enc2 = lambda v: gated(gated(v, "q_z_layers.0"), "q_z_layers.1")
    q2_mu, q2_lv = heads(enc2(x64), "q_z_mean", "q_z_logvar")
    z2 = q2_mu + eps2 * np.exp(0.5 * q2_lv)
    joint = gated(np.concatenate((gated(x64, "q_z1_layers_x.0"), gated(z2, "q_z1_layers_z2.0")), 1), "q_z1_layers_joint.0")
    q1_mu, q1_lv = heads(joint, "q_z1_mean", "q_z1_logvar")
    z1 = q1_mu + eps1 * np.exp(0.5 * q1_lv)
    p1_mu, p1_lv = heads(gated(gated(z2, "p_z1_layers_z2.0"), "p_z1_layers_z2.1"), "p_z1_mean", "p_z1_logvar")
    top = gated(np.concatenate((gated(z1, "p_x_layers_z1.0"), gated(z2, "p_x_layers_z2.0")), 1), "p_x_layers_joint.0")
    x_mean = orc.sigmoid(orc.linear(top, P["p_x_mean.linear.weight"], P["p_x_mean.linear.bias"]))
    RE_ref = orc.log_bernoulli(x64, x_mean)
    centres = orc.linear(enc2(data_np[ex_idx].astype(np.float64)), P["q_z_mean.weight"], P["q_z_mean.bias"])
    clv = np.full((C, 40), float(P["prior_log_variance"][0]))
    log_pz2 = orc.log_p_z(z2, xi.reshape(-1, 1), centres, clv, ex_idx, test=False)
    KL_ref = (orc.log_normal_diag(z1, q1_mu, q1_lv) - orc.log_normal_diag(z1, p1_mu, p1_lv)
              + orc.log_normal_diag(z2, q2_mu, q2_lv) - log_pz2)
    loss_ref = -RE_ref + beta * KL_ref
    for name, got, ref in (("RE", RE, RE_ref), ("KL", KL, KL_ref), ("loss", loss, loss_ref)):
        assert rel(got.detach().cpu().numpy(), ref) < 1e-4, (name, rel(got.detach().cpu().numpy(), ref))


@pytest.mark.parametrize("model_name,C", [("hvae_2level", 11500), ("hvae_2level", 120), ("vae", 200)])
def test_deferred_grouped_weight_gradients_equal_the_inline_ones(model_name, C):
    """ops.deferred_wgrads (r04): the thin layers' weight gradients allocated where autograd asks for them and filled by grouped
    launches behind the backward pass -- every parameter's gradient bit-identical to the launch-per-layer form (the same
    kernel body per job), none left unfilled (the buffers are poisoned with NaN first through the allocator)"""
    from evae import ops
    from utils.utils import importing_model
    B, N = 100, 2 * C + 300
    data = torch.from_numpy(gi.binary_images(9, N))
    dataset = torch.utils.data.TensorDataset(data, torch.arange(N).reshape(-1, 1), torch.zeros(N))
    args = smoke_case.vae_args(model_name=model_name, number_components=C, training_set_size=N, batch_size=B)
    torch.manual_seed(21)
    model = importing_model(args)(args).cuda()
    model.train()
    model._use_fused = False                                      # the modular autograd path (the fused `vae` node has its own grouping)
    x = data[:B].cuda(); idx = torch.arange(B, device="cuda").reshape(-1, 1)
    g = torch.Generator(device="cuda")
    grads = []
    for deferred in (False, True):
        g.manual_seed(5); model._eps_generator = g
        torch.manual_seed(77)
        model.zero_grad(set_to_none=True)
        loss, RE, KL = model.calculate_loss((x, idx), 0.5, average=True, dataset=dataset)
        torch.cuda.synchronize()
        poison = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]      # what the backward's buffers will be cut from
        del poison
        if deferred:
            with ops.deferred_wgrads(loss):
                loss.backward()
                njobs = len(ops._DEFER[0]["jobs"])
            assert njobs >= (8 if model_name == "hvae_2level" else 3), njobs
        else:
            loss.backward()
        torch.cuda.synchronize()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    model._eps_generator = None
    a, b = grads
    assert set(a) == set(b) and len(a) >= 10
    for k in a:
        assert torch.isfinite(b[k]).all(), k
        assert torch.equal(a[k], b[k]), (k, float((a[k] - b[k]).abs().max()))


@pytest.mark.parametrize("model_name,C", [("hvae_2level", 120), ("vae", 200)])
def test_deferred_weight_gradients_with_gradients_already_there(model_name, C):
    """a deferred (or in-place accumulated) weight gradient is a buffer filled AFTER autograd has been handed it, which is
    only right while AccumulateGrad installs it (.grad is None).  With gradients already there (zero_grad(set_to_none=False),
    gradient accumulation over two backward passes) the scope must compute those leaves on the spot: two accumulated passes inside
    the scope == the same two passes without it, bit for bit, and nothing is deferred on the second pass."""
    from evae import ops
    from utils.utils import importing_model
    B, N = 100, 2 * C + 300
    data = torch.from_numpy(gi.binary_images(9, N))
    dataset = torch.utils.data.TensorDataset(data, torch.arange(N).reshape(-1, 1), torch.zeros(N))
    args = smoke_case.vae_args(model_name=model_name, number_components=C, training_set_size=N, batch_size=B)
    torch.manual_seed(21)
    model = importing_model(args)(args).cuda()
    model.train()
    model._use_fused = False
    x = data[:B].cuda(); idx = torch.arange(B, device="cuda").reshape(-1, 1)
    g = torch.Generator(device="cuda")
    grads = []
    for deferred in (False, True):
        model.zero_grad(set_to_none=True)
        for rep in range(2):                                      # the second pass meets the first one's gradients
            g.manual_seed(5 + rep); model._eps_generator = g
            torch.manual_seed(77 + rep)
            loss, RE, KL = model.calculate_loss((x, idx), 0.5, average=True, dataset=dataset)
            if deferred:
                with ops.deferred_wgrads(loss):
                    loss.backward()
                    njobs = len(ops._DEFER[0]["jobs"])
                assert (njobs > 0) if rep == 0 else (njobs == 0), (rep, njobs)
            else:
                loss.backward()
        torch.cuda.synchronize()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    model._eps_generator = None
    a, b = grads
    assert set(a) == set(b) and len(a) >= 10
    for k in a:
        assert torch.isfinite(b[k]).all(), k
        assert torch.equal(a[k], b[k]), (k, float((a[k] - b[k]).abs().max()))


def test_conv_two_level_step_at_c3_size_matches_the_oracle():
    """convhvae_2level at the benchmarked size (BASELINE configs[2]: 25 000 exemplars, batch 100): per-sample loss / RE / KL of
    the training step against an fp64 restatement of reference models/AbsHModel.py:13-106 + models/convHVAE_2level.py:13-103
    (convolutions: float64 torch on the CPU, reference utils/nn.py:72-97; densities and the exemplar prior: the oracle's
    log_normal_diag / log_bernoulli / log_p_z) on the same weights, noise and exemplar draw -- 1e-4 relative.  Of the 25 000
    centres of the oracle's prior, a 2 048-row shard (the first 1 024, the last 1 024: both ends of the launch grid, the
    leave-one-out rows included) comes from the float64 encoder -- and the GPU path's rows are held to it at 1e-5 --, the rest are
    the GPU path's own q(z2 | x) means (a convolutional encoder is independent per image; all 25 000 in float64 on the host
    would take minutes).  VERDICT r03 weak #1: c3 was only compared with itself."""
    import torch.nn.functional as F
    from utils.utils import importing_model
    B, Cn, N = 100, 25000, 50000
    data_np = gi.binary_images(2, N)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data_np), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    args = smoke_case.vae_args(model_name="convhvae_2level", number_components=Cn, training_set_size=N, batch_size=B,
                               dataset_name="fashion_mnist")
    torch.manual_seed(15); torch.cuda.manual_seed(15)
    model = importing_model(args)(args).cuda()
    model.train()
    rs = np.random.RandomState(61)
    xi = rs.choice(N, size=B, replace=False).astype(np.int64)
    x_np = data_np[xi]
    eps2 = rs.standard_normal((B, 40)).astype(np.float32); eps1 = rs.standard_normal((B, 40)).astype(np.float32)
    ex_idx = rs.randint(0, N, size=Cn).astype(np.int64)
    ex_idx[:5] = xi[:5]                                        # leave-one-out hits
    draws = [eps2, eps1]
    model._draw_eps = lambda like: torch.from_numpy(draws.pop(0)).to(like.device).reshape(like.shape)
    orig = torch.randint
    torch.randint = lambda low=0, high=None, size=None, **kw: torch.from_numpy(ex_idx.copy())
    beta = 0.6
    try:
        loss, RE, KL = model.calculate_loss((torch.from_numpy(x_np).cuda(), torch.from_numpy(xi).reshape(-1, 1).cuda()), beta,
                                            average=False, dataset=dataset)
        with torch.no_grad():
            centres_gpu = model.q_z(torch.from_numpy(data_np[ex_idx]).cuda(), prior=True)[0].double().cpu().numpy()
    finally:
        torch.randint = orig
    assert not draws
    # ---- fp64 restatement ----
    T = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
    P = {k: v.numpy() for k, v in T.items()}

    def gconv(h, name, st, pd):
        return F.conv2d(h, T[name + ".h.weight"], T[name + ".h.bias"], st, pd) * \
            torch.sigmoid(F.conv2d(h, T[name + ".g.weight"], T[name + ".g.bias"], st, pd))

    def stack(v, name, spec):
        h = torch.from_numpy(v).double().reshape(-1, 1, 28, 28)
        for li, (st, pd) in enumerate(spec):
            h = gconv(h, "%s.%d" % (name, li), st, pd)
        return h.reshape(h.shape[0], -1).numpy()

    def gated(v, name):
        y, _ = orc.gated_dense(v, P[name + ".h.weight"], P[name + ".h.bias"], P[name + ".g.weight"], P[name + ".g.bias"])
        return y

    def heads(t, mean, logvar):
        mu = orc.linear(t, P[mean + ".linear.weight"], P[mean + ".linear.bias"])
        lv = orc.hardtanh(orc.linear(t, P[logvar + ".linear.weight"], P[logvar + ".linear.bias"]), -6.0, 2.0)
        return mu, lv
    ENC2 = ((1, 3), (2, 1), (1, 2), (2, 1), (1, 1)); ENC1 = ((1, 1), (2, 1), (1, 1), (2, 1), (1, 1))
    x64 = x_np.astype(np.float64)
    q2_mu, q2_lv = heads(stack(x64, "q_z_layers", ENC2), "q_z_mean", "q_z_logvar")
    z2 = q2_mu + eps2 * np.exp(0.5 * q2_lv)
    joint = gated(np.concatenate((stack(x64, "q_z1_layers_x", ENC1), gated(z2, "q_z1_layers_z2.0")), 1), "q_z1_layers_joint.0")
    q1_mu, q1_lv = heads(joint, "q_z1_mean", "q_z1_logvar")
    z1 = q1_mu + eps1 * np.exp(0.5 * q1_lv)
    p1_mu, p1_lv = heads(gated(gated(z2, "p_z1_layers_z2.0"), "p_z1_layers_z2.1"), "p_z1_mean", "p_z1_logvar")
    pre = gated(np.concatenate((gated(z1, "p_x_layers_z1.0"), gated(z2, "p_x_layers_z2.0")), 1), "p_x_layers_joint_pre.0")
    h = torch.from_numpy(pre).reshape(-1, 1, 28, 28)
    for li in range(4):
        h = gconv(h, "p_x_layers_joint.%d" % li, 1, 1)
    x_mean = torch.sigmoid(F.conv2d(h, T["p_x_mean.conv.weight"], T["p_x_mean.conv.bias"])).reshape(B, -1).numpy()
    RE_ref = orc.log_bernoulli(x64, x_mean)
    # a 2 048-row shard of the centres (both ends of the launch grid, the leave-one-out rows among them) from the float64 encoder:
    # held against the GPU path's rows at 1e-5 AND handed to the oracle's prior in place of them
    sample = np.r_[0:1024, Cn - 1024:Cn]
    c64 = orc.linear(stack(data_np[ex_idx[sample]].astype(np.float64), "q_z_layers", ENC2), P["q_z_mean.linear.weight"], P["q_z_mean.linear.bias"])
    assert rel(centres_gpu[sample], c64) < 1e-5, rel(centres_gpu[sample], c64)
    centres_ref = centres_gpu.copy()
    centres_ref[sample] = c64
    clv = np.full((Cn, 40), float(P["prior_log_variance"][0]))
    log_pz2 = orc.log_p_z(z2, xi.reshape(-1, 1), centres_ref, clv, ex_idx, test=False)
    KL_ref = (orc.log_normal_diag(z1, q1_mu, q1_lv) - orc.log_normal_diag(z1, p1_mu, p1_lv)
              + orc.log_normal_diag(z2, q2_mu, q2_lv) - log_pz2)
    loss_ref = -RE_ref + beta * KL_ref
    for name, got, ref in (("RE", RE, RE_ref), ("KL", KL, KL_ref), ("loss", loss, loss_ref)):
        assert rel(got.detach().cpu().numpy(), ref) < 1e-4, (name, rel(got.detach().cpu().numpy(), ref))
    # ---- gradients at this size: the step's gradient wrt the centres from the oracle's prior (d mean-loss / d
    # log p(z2_i) = -beta / B), pulled back through the exemplar encoder over the 2 048-row shard -- in float64 torch on the host and by
    # the GPU path (GatedConvStackFn backward + the layers outside it) -- every encoder parameter's gradient, norm and entries
    lv_row = np.full((40,), float(P["prior_log_variance"][0]))
    _, dc_ref, _, _ = orc.prior_grads(z2, xi.reshape(-1, 1), centres_ref, lv_row, ex_idx, True, np.full((B,), -beta / B))
    up = np.ascontiguousarray(dc_ref[sample])
    enc = {k: T[k].clone().requires_grad_(True) for k in T if k.startswith("q_z_layers.") or k.startswith("q_z_mean.")}
    h = torch.from_numpy(data_np[ex_idx[sample]].astype(np.float64)).reshape(-1, 1, 28, 28)
    for li, (st, pd) in enumerate(ENC2):
        nm = "q_z_layers.%d" % li
        h = F.conv2d(h, enc[nm + ".h.weight"], enc[nm + ".h.bias"], st, pd) * \
            torch.sigmoid(F.conv2d(h, enc[nm + ".g.weight"], enc[nm + ".g.bias"], st, pd))
    mu64 = h.reshape(h.shape[0], -1) @ enc["q_z_mean.linear.weight"].t() + enc["q_z_mean.linear.bias"]
    mu64.backward(torch.from_numpy(up))
    model.zero_grad()
    from evae import _lib
    with _lib.count_calls("evae_cw_") as n:
        mu = model.q_z(torch.from_numpy(data_np[ex_idx[sample]]).cuda(), prior=True)[0]
        mu.backward(torch.from_numpy(up).float().cuda())
    assert n.get("evae_cw_bwd_weight", 0) >= 3 and n.get("evae_cw_bwd_data_gate", 0) >= 3, n        # the window operators did it
    named = dict(model.named_parameters())
    for k, r in enc.items():
        got, ref = named[k].grad.double().cpu().numpy(), r.grad.numpy()
        assert abs(np.linalg.norm(got) - np.linalg.norm(ref)) <= 3e-4 * np.linalg.norm(ref), (k, np.linalg.norm(got), np.linalg.norm(ref))
        assert np.abs(got - ref).max() <= 3e-4 * np.abs(ref).max(), (k, np.abs(got - ref).max() / np.abs(ref).max())


def test_single_conv_training_step_at_c5_size():
    """BASELINE configs[4] at its own size (only c5's top-K and geometry were under test): `single_conv`
    on 3 x 64 x 64, z = 256, approximate prior over 100 000 candidates drawn from a 100 000-image training set, k = 10, batch
    100 -- one eager training step (cache of all latents, refresh of the batch's rows, top-K among the candidates' cached
    latents, <= 1000 exemplars re-encoded, loss, backward) on the split-bf16 pipe and on the fp32-MFMA pipe: loss / RE / KL
    finite and equal to 1e-5, every gradient norm to 1e-3; the step's top-K (batch means against the 100 000 cached latents,
    reference models/BaseModel.py:263-264) bit-equal to the oracle's float64 scan with its (value, index) order; every batch row's KL
    against the oracle's prior and posterior densities over the exemplar set the reference would re-encode (1e-4)."""
    from evae import ops
    from utils.utils import importing_model
    from argparse import Namespace
    B, N, k = 100, 100000, 10
    isz = [3, 64, 64]
    args = Namespace(prior="exemplar_prior", input_type="continuous", input_size=isz, hidden_size=300, z1_size=256, z2_size=40,
                     model_name="single_conv", device="cuda", number_components=N, training_set_size=N, approximate_prior=True,
                     approximate_k=k, no_mask=False, no_attention=False, same_variational_var=False, use_logit=False, lambd=1e-4,
                     bottleneck=1, dataset_name="celeba", continuous=True, batch_size=B, dynamic_binarization=False, warmup=100,
                     S=5000, shard_exemplars=False, shard_batch=False)
    torch.manual_seed(44); torch.cuda.manual_seed(44)
    model = importing_model(args)(args).cuda()
    model.train()
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    # images with structure (a few prototypes + noise) so that neighbours in latent space mean something
    protos = torch.randint(0, 256, (32, 12288), device="cuda", generator=g).float()
    which = torch.randint(0, 32, (N,), device="cuda", generator=g)
    data_dev = ((protos[which] * 0.7 + torch.randint(0, 77, (N, 12288), device="cuda", generator=g, dtype=torch.int16).float()).floor() + 0.5) / 256
    del protos
    dataset = torch.utils.data.TensorDataset(data_dev, torch.arange(N).reshape(-1, 1), torch.zeros(N))
    idx_all = torch.arange(N, device="cuda").reshape(-1, 1)
    cand = torch.randperm(N, device="cuda", generator=g).cpu()    # the step's candidate draw (the reference: torch.randint on the host)
    rs = np.random.RandomState(12)
    eps = torch.from_numpy(rs.standard_normal((B, 256)).astype(np.float32)).cuda()
    res = []
    # (pipe, window): the r05 path (pixel-image window kernels, one-launch weight norm) on both GEMM pipes, then the r04 kernels
    # (channels-last implicit GEMMs + torch's weight norm: the path the reference-generated goldens G9 / G19 / G20 / G21 pin at small
    # sizes) -- every batch row's loss / RE / KL must agree between them at THIS size
    for pipe, window in ((1, True), (0, True), (1, False)):
        ops.gemm_x6_configure(pipe, 2048)
        orig = torch.randint
        stack_on = ops.CONV_STACK_ON
        try:
            ops.CONV_STACK_ON = window
            os.environ["EVAE_WN_SET"] = "1" if window else "0"
            with torch.no_grad():
                cache = tuple(t.clone() for t in model.cache_z(dataset))
            torch.randint = lambda low=0, high=None, size=None, **kw: cand.clone()
            model._draw_eps = lambda like: eps.reshape(like.shape)
            model.zero_grad()
            loss, RE, KL = model.calculate_loss((data_dev[500:500 + B], idx_all[500:500 + B]), 0.5, average=False, cache=cache,
                                                dataset=dataset)
            loss.mean().backward()
        finally:
            torch.randint = orig
            ops.CONV_STACK_ON = stack_on
            os.environ.pop("EVAE_WN_SET", None)
            ops.gemm_x6_configure(1, 2048)
        res.append((np.stack([t.detach().double().cpu().numpy().reshape(-1) for t in (loss, RE, KL)]),
                    np.asarray([p.grad.double().norm().item() for p in model.parameters() if p.grad is not None]), cache))
    assert np.isfinite(res[0][0]).all() and np.isfinite(res[0][1]).all() and res[0][0].shape == (3, B)
    for other in (1, 2):
        for j, what in enumerate(("loss", "RE", "KL")):          # every batch row
            assert rel(res[0][0][j], res[other][0][j]) < 2e-5, (other, what, rel(res[0][0][j], res[other][0][j]))
        assert np.all(np.abs(res[0][1] - res[other][1]) <= 1e-3 * np.maximum(res[other][1], 1e-6)), other
    # the step's top-K at this size against the oracle (the cache AFTER the step holds the batch's refreshed rows)
    cz = res[0][2][0]
    q = cz[500:500 + B].contiguous()
    idx, val = ops.pairdist_topk(q, cz, k)
    _, ref_idx = orc.nearest_exemplars_topk(q.double().cpu().numpy(), cz.double().cpu().numpy(), k)
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    # r06: the step's KL at THIS size against the oracle -- log q(z | x) - log p(z) with the exemplar set the
    # reference's get_approximate_nearest_exemplars arrives at (models/BaseModel.py:256-271: batch rows of the cache refreshed with their
    # means, top-k of every batch row among the candidates' cached latents, `unique`, those images re-encoded): the encoder's outputs
    # (pinned by G20 / G23) go to evae_oracle.log_p_z / log_normal_diag in float64
    with torch.no_grad():
        xb = data_dev[500:500 + B]
        mu, lv = model.q_z(xb)
        z = mu + eps.reshape(mu.shape) * torch.exp(0.5 * lv)
        cache0 = model.cache_z(dataset)[0].clone()
        cache0[500:500 + B] = mu
        near, _ = ops.pairdist_topk(mu, cache0[cand.cuda()].contiguous(), k)
        pos = torch.unique(near.reshape(-1))
        sel = cand.cuda()[pos]
        centres = model.q_z(data_dev[sel], prior=True)[0]
    assert 10 <= sel.numel() <= B * k
    plv = float(model.prior_log_variance.item())
    log_p = orc.log_p_z(z.double().cpu().numpy(), np.arange(500, 500 + B).reshape(-1, 1), centres.double().cpu().numpy(),
                        np.full((sel.numel(), 256), plv), sel.cpu().numpy(), test=False)
    log_q = orc.log_normal_diag(z.double().cpu().numpy(), mu.double().cpu().numpy(), lv.double().cpu().numpy())
    assert rel(res[0][0][2], log_q - log_p) < 1e-4, rel(res[0][0][2], log_q - log_p)


@pytest.mark.parametrize("upload", ["direct", "staged"])
def test_graphed_step_over_distinct_exemplar_rows_matches_eager(upload, monkeypatch):
    """EVAE_DEDUP=1 (r04): the captured step encodes the DISTINCT rows of the exemplar draw only (4 000 draws with replacement from
    4 000 images name ~2 530 of them), the prior sees every draw through a gather of the distinct rows' encodings, a distinct
    row's gradient is its multiplicity x one draw's -- same losses and same parameters after seven steps as the eager step that
    encodes every draw (reference models/BaseModel.py:243-254), to 1e-5."""
    monkeypatch.setenv("EVAE_DEDUP", "1")
    monkeypatch.setenv("EVAE_CTL_DIRECT", "1" if upload == "direct" else "0")
    from evae.graph import GraphedTrainStep
    from utils.optimizer import AdamNormGrad
    B, C, N = 100, 4000, 4000
    data = gi.binary_images(15, N)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    results = []
    for use_graph in (False, True):
        args = smoke_case.vae_args(number_components=C, training_set_size=N, batch_size=B)
        model, _ = smoke_case.build_model(torch, np, orc, args)
        model.train()
        opt = AdamNormGrad(model.parameters(), lr=5e-4)
        torch.manual_seed(13); torch.cuda.manual_seed(13)
        runner = GraphedTrainStep(model, opt, dataset, B, False) if use_graph else None
        losses = []
        for it in range(7):
            xb = torch.from_numpy(data[it * B:(it + 1) * B])
            ib = torch.arange(it * B, (it + 1) * B).reshape(-1, 1)
            if runner is not None:
                losses.append(runner(xb, ib, 0.5)[0].item())
                assert 0 < runner.dedup["distinct"] <= runner.dedup["cap"] < 0.92 * C
            else:
                xb, ib = xb.cuda(), ib.cuda()
                from evae import ops as _ops
                eps = torch.empty((B, args.z1_size), device="cuda")
                _ops.batch_prologue(torch.from_numpy(data).cuda(), ib.reshape(-1).contiguous(), False,
                                    torch.tensor([13, it], dtype=torch.int64, device="cuda"), torch.empty_like(xb), eps)
                model._eps_override = eps
                opt.zero_grad()
                loss, RE, KL = model.calculate_loss((xb, ib), 0.5, average=True, dataset=dataset)
                loss.backward()
                opt.step()
                losses.append(loss.item())
        results.append((losses, {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}))
        if runner is not None:
            assert runner.graph is not None and runner.by_index and runner.dedup is not None
    (l0, p0), (l1, p1) = results
    assert rel(np.asarray(l1), np.asarray(l0)) < 1e-5
    for k in p0:
        assert rel(p1[k], p0[k]) < 1e-5, k


@pytest.mark.parametrize("model_name", ["vae", "hvae_2level"])
def test_a_draw_with_too_many_distinct_rows_steps_eagerly_once(model_name, monkeypatch):
    """ADVICE r04 / VERDICT r05 weak #4: a draw whose distinct rows do not fit the captured step's fixed row count (eight standard
    deviations above the mean: forced here by making evae_host_dedup report it) no longer raises mid-training -- that ONE step is issued
    eagerly with every draw encoded (reference models/BaseModel.py:243-254, which encodes every draw anyway), the next replays again:
    same losses and parameters as the run without the event."""
    monkeypatch.setenv("EVAE_DEDUP", "1")
    from evae import _lib
    from evae.graph import GraphedTrainStep
    from utils.optimizer import AdamNormGrad
    from utils.utils import importing_model
    B, C, N = 100, 4000, 4000
    data = gi.binary_images(15, N)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    lib = _lib.load()
    real = lib.evae_host_dedup
    results = []
    for overflow_at in (None, 5):
        args = smoke_case.vae_args(model_name=model_name, number_components=C, training_set_size=N, batch_size=B)
        torch.manual_seed(5); torch.cuda.manual_seed(5)
        model = importing_model(args)(args).cuda()
        model.train()
        opt = AdamNormGrad(model.parameters(), lr=5e-4)
        torch.manual_seed(13); torch.cuda.manual_seed(13)
        runner = GraphedTrainStep(model, opt, dataset, B, False)
        calls = {"n": 0}

        def dedup(*a, calls=calls, overflow_at=overflow_at):
            calls["n"] += 1
            return -1 if calls["n"] - 1 == overflow_at else real(*a)
        monkeypatch.setattr(lib, "evae_host_dedup", dedup)
        losses = []
        for it in range(8):
            xb = torch.from_numpy(data[it * B:(it + 1) * B])
            ib = torch.arange(it * B, (it + 1) * B).reshape(-1, 1)
            losses.append(runner(xb, ib, 0.5)[0].item())
        monkeypatch.setattr(lib, "evae_host_dedup", real)
        assert runner.graph is not None and not runner.failed and runner.dedup is not None
        assert runner.overflow_steps == (0 if overflow_at is None else 1)
        results.append((losses, {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}))
    (l0, p0), (l1, p1) = results
    assert rel(np.asarray(l1), np.asarray(l0)) < 1e-5
    for k_ in p0:
        assert rel(p1[k_], p0[k_]) < 2e-5, k_


@pytest.mark.parametrize("C", [200, 4000])
def test_merged_element_wise_launches_leave_the_trajectory_unchanged(C, monkeypatch):
    """r06: the log-variance row's broadcast in the heads' launch, RE + unit coefficients + sigmoid gradient as one launch
    (evae_bernoulli_unit_step) and the ELBO's assembly + log-variance gradient's sum in the reparameterisation's backward
    (evae_reparam_logq_bwd_hardtanh_tail) compute what the five launches they replace computed, in the same order: a captured `vae`
    run with them is the run without them (EVAE_NODE_MERGE=0), and each merged entry point was actually called."""
    from evae import _lib
    from evae.graph import GraphedTrainStep
    from utils.optimizer import AdamNormGrad
    from utils.utils import importing_model
    B, N = 100, 4000
    data = gi.binary_images(17, N)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    results = []
    for mask in ("0", "15"):
        monkeypatch.setenv("EVAE_NODE_MERGE", mask)
        args = smoke_case.vae_args(model_name="vae", number_components=C, training_set_size=N, batch_size=B)
        torch.manual_seed(5); torch.cuda.manual_seed(5)
        model = importing_model(args)(args).cuda()
        model.train()
        opt = AdamNormGrad(model.parameters(), lr=5e-4)
        torch.manual_seed(13); torch.cuda.manual_seed(13)
        runner = GraphedTrainStep(model, opt, dataset, B, False)
        out = []
        with _lib.count_calls("evae_") as n:
            for it in range(7):
                xb = torch.from_numpy(data[it * B:(it + 1) * B])
                ib = torch.arange(it * B, (it + 1) * B).reshape(-1, 1)
                out.append(runner(xb, ib, 0.5).cpu().numpy().copy())
        assert runner.graph is not None and not runner.failed
        merged = ("evae_heads_reparam_fwd_bcast", "evae_bernoulli_unit_step", "evae_reparam_logq_bwd_hardtanh_tail")
        replaced = ("evae_broadcast_scalar", "evae_sum_small", "evae_elbo_assemble", "evae_bernoulli_sigmoid_bwd")
        # (the runner's first call steps the reference's way -- no unit-upstream promise --, so the un-merged Bernoulli launches
        #  appear once in either run)
        if mask == "0":
            assert not any(n.get(f, 0) for f in merged), n
        else:
            assert all(n.get(f, 0) >= 2 for f in merged), n
            assert n.get("evae_broadcast_scalar", 0) == 0 and n.get("evae_sum_small", 0) == 0, n
        results.append((np.asarray(out), {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}))
    (l0, p0), (l1, p1) = results
    assert np.array_equal(l0, l1), np.abs(l0 - l1).max()
    for k_ in p0:
        assert np.array_equal(p0[k_], p1[k_]), k_


@pytest.mark.parametrize("model_name", ["vae", "hvae_2level"])
def test_control_block_handed_over_by_the_first_launch_equals_the_copied_one(model_name, monkeypatch):
    """r06: on the byte store the step's first launch takes the control block from the staging block a device-side parity word
    names and copies it on (evae_batch_prologue_u8_step); the optimizer's last launch flips the parity.  Same trajectory as the
    device-to-device copy in front of the graph (EVAE_CTL_HANDOVER=0) across warm-up, capture, replays, one eager every-draw step
    (a draw that does not fit) and one step_eagerly in between; parity and step counter in step with the host's count."""
    monkeypatch.setenv("EVAE_DEDUP", "1")
    monkeypatch.setenv("EVAE_CTL_DIRECT", "0")         # (the staged upload, as at c2: thin steps upload straight into the block)
    from evae import _lib
    from evae.graph import GraphedTrainStep
    from utils.optimizer import AdamNormGrad
    from utils.utils import importing_model
    B, C, N = 100, 4000, 4000
    data = gi.binary_images(15, N)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    lib = _lib.load()
    real = lib.evae_host_dedup
    results = []
    for handover in ("0", "1"):
        monkeypatch.setenv("EVAE_CTL_HANDOVER", handover)
        args = smoke_case.vae_args(model_name=model_name, number_components=C, training_set_size=N, batch_size=B)
        torch.manual_seed(5); torch.cuda.manual_seed(5)
        model = importing_model(args)(args).cuda()
        model.train()
        opt = AdamNormGrad(model.parameters(), lr=5e-4)
        torch.manual_seed(13); torch.cuda.manual_seed(13)
        runner = GraphedTrainStep(model, opt, dataset, B, False)
        on = handover == "1" and model_name == "vae"       # (the byte store -- and with it the hand-over -- is the fused `vae` step's)
        assert runner._handover == on and not runner._direct
        calls = {"n": 0}

        def dedup(*a, calls=calls):
            calls["n"] += 1
            return -1 if calls["n"] - 1 == 4 else real(*a)
        monkeypatch.setattr(lib, "evae_host_dedup", dedup)
        losses = []
        with _lib.count_calls("evae_batch_prologue_u8_step") as n_step:
            for it in range(10):
                xb = torch.from_numpy(data[it * B:(it + 1) * B])
                ib = torch.arange(it * B, (it + 1) * B).reshape(-1, 1)
                step = runner.step_eagerly if it == 6 else runner
                if it in (3, 7):                 # (index lists that are already on the device go into the staging block on the step's stream)
                    ib = ib.cuda()
                losses.append(step(xb, ib, 0.5)[0].item())
                if on:
                    assert int(runner._ho_state[0]) == runner._calls & 1, it
                    assert int(runner.ctl[runner._o_seed + 1]) == runner._calls - 1, it
        monkeypatch.setattr(lib, "evae_host_dedup", real)
        assert runner.graph is not None and not runner.failed and runner.overflow_steps == 1
        assert (n_step.get("evae_batch_prologue_u8_step", 0) > 0) == on
        results.append((losses, {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}))
    (l0, p0), (l1, p1) = results
    assert rel(np.asarray(l1), np.asarray(l0)) < 1e-5
    for k_ in p0:
        assert rel(p1[k_], p0[k_]) < 2e-5, k_


@pytest.mark.parametrize("model_name", ["hvae_2level", "convhvae_2level", "vae_modular"])
def test_graphed_modular_step_over_distinct_exemplar_rows_matches_eager(model_name, monkeypatch):
    """EVAE_DEDUP on the modular autograd path (r04): get_exemplar_set encodes the DISTINCT rows of the draw and hands the prior
    every draw's encoding through ops.ExpandRowsFn (forward: gather; backward: multiplicity x one draw's gradient) -- the
    hierarchical, convolutional and un-fused `vae` models replay the same trajectory as the eager step that encodes every draw
    (reference models/BaseModel.py:243-254), 2 400 draws with replacement from 2 400 images."""
    monkeypatch.setenv("EVAE_DEDUP", "1")
    from evae.graph import GraphedTrainStep
    from utils.optimizer import AdamNormGrad
    from utils.utils import importing_model
    B, C, N = 16, 2400, 2400
    data = gi.binary_images(16, N)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    results = []
    for use_graph in (False, True):
        args = smoke_case.vae_args(model_name="vae" if model_name == "vae_modular" else model_name, number_components=C,
                                   training_set_size=N, batch_size=B)
        torch.manual_seed(5); torch.cuda.manual_seed(5)
        model = importing_model(args)(args).cuda()
        if model_name == "vae_modular":
            model._use_fused = False
        model.train()
        opt = AdamNormGrad(model.parameters(), lr=5e-4)
        torch.manual_seed(3); torch.cuda.manual_seed(3)
        runner = GraphedTrainStep(model, opt, dataset, B, False) if use_graph else None
        losses = []
        for it in range(6):
            xb = torch.from_numpy(data[it * B:(it + 1) * B]).cuda()
            ib = torch.arange(it * B, (it + 1) * B).reshape(-1, 1).cuda()
            if runner is not None:
                losses.append(runner(xb, ib, 0.5)[0].item())
                assert 0 < runner.dedup["distinct"] <= runner.dedup["cap"] < 0.92 * C
            else:
                opt.zero_grad()
                loss, RE, KL = model.calculate_loss((xb, ib), 0.5, average=True, dataset=dataset)
                loss.backward()
                opt.step()
                losses.append(loss.item())
        results.append((losses, {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}))
        if runner is not None:
            assert runner.graph is not None and not runner.failed and runner.dedup is not None
    (l0, p0), (l1, p1) = results
    assert rel(np.asarray(l1), np.asarray(l0)) < 2e-5
    for k_ in p0:
        assert rel(p1[k_], p0[k_]) < 5e-5, k_


def test_distinct_rows_step_at_config2_size_equals_the_every_draw_step(monkeypatch):
    """BASELINE configs[1] itself (25 000 draws from 50 000 images, batch 100): five captured steps that encode the ~19 700
    distinct images of each draw (19 968 rows) against five captured steps that encode every draw (EVAE_DEDUP=0) from the same
    seeds -- the same losses and the same parameters to 1e-5 (the two differ in the order of a few sums only)."""
    from evae.graph import GraphedTrainStep
    from utils.optimizer import AdamNormGrad
    B, C, N = 100, 25000, 50000
    data = gi.binary_images(0, N)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    results = []
    for dedup in ("1", "0"):
        monkeypatch.setenv("EVAE_DEDUP", dedup)
        args = smoke_case.vae_args(number_components=C, training_set_size=N, batch_size=B)
        model, _ = smoke_case.build_model(torch, np, orc, args)
        model.train()
        opt = AdamNormGrad(model.parameters(), lr=5e-4)
        torch.manual_seed(23); torch.cuda.manual_seed(23)
        runner = GraphedTrainStep(model, opt, dataset, B, True)
        losses = []
        for it in range(5):
            xb = torch.from_numpy(data[it * B:(it + 1) * B])
            ib = torch.arange(it * B, (it + 1) * B).reshape(-1, 1)
            losses.append(runner(xb, ib, 0.5)[0].item())
        assert runner.graph is not None
        assert (runner.dedup is not None) == (dedup == "1")
        if dedup == "1":
            assert runner.dedup["cap"] == 19968 and 19300 < runner.dedup["distinct"] <= 19968
        results.append((losses, {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}))
    (l1, p1), (l0, p0) = results
    assert rel(np.asarray(l1), np.asarray(l0)) < 1e-5
    for k_ in p0:
        assert rel(p1[k_], p0[k_]) < 1e-5, k_


def test_captured_distinct_rows_steps_at_config2_size_match_the_oracle(monkeypatch):
    """the BENCHMARKED object itself -- the captured step that encodes the distinct rows of the draw, at
    BASELINE configs[1]'s size (B = 100, C = 25 000 draws with replacement from N = 50 000) -- against the oracle's train step
    (reference utils/training.py:27-40 + models/BaseModel.py:54-77,243-254 + utils/optimizer.py:32-80), three steps in a row with
    the same injected draws and noise: per-step loss / RE / KL to 1e-4, every parameter after the third step to 1e-5."""
    from evae.graph import GraphedTrainStep
    from utils.optimizer import AdamNormGrad
    B, C, N, beta = 100, 25000, 50000, 0.5
    data = gi.binary_images(0, N)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    monkeypatch.setenv("EVAE_DEDUP", "1")
    args = smoke_case.vae_args(number_components=C, training_set_size=N, batch_size=B)
    model, p = smoke_case.build_model(torch, np, orc, args)
    model.train()
    opt = AdamNormGrad(model.parameters(), lr=5e-4)
    rs = np.random.RandomState(91)
    nsteps = 4                                                    # (the runner's first calls step eagerly, then capture, then replay)
    draws = [rs.randint(0, N, size=(C,)).astype(np.int64) for _ in range(nsteps)]
    epss = [rs.standard_normal((B, 40)).astype(np.float32) for _ in range(nsteps)]
    eps_dev = torch.zeros((B, 40), device="cuda")
    model._draw_eps = lambda like: eps_dev                        # static buffer: the captured launches read it at every replay
    cur = {"i": 0}
    orig = torch.randint

    def fake_randint(low=0, high=None, size=None, out=None, **kw):
        d = torch.from_numpy(draws[cur["i"]])
        if out is not None:
            out.copy_(d[:out.numel()])
            return out
        return d
    monkeypatch.setattr(torch, "randint", fake_randint)
    runner = GraphedTrainStep(model, opt, dataset, B, False)
    got = []
    from evae import _lib as _evl
    with _evl.count_calls("evae_") as ncalls:
        for it in range(nsteps):
            cur["i"] = it
            eps_dev.copy_(torch.from_numpy(epss[it])); torch.cuda.synchronize()
            xb = torch.from_numpy(data[it * B:(it + 1) * B]); ib = torch.arange(it * B, (it + 1) * B).reshape(-1, 1)
            out = runner(xb, ib, beta)
            torch.cuda.synchronize()
            got.append([float(out[0].item()), float(out[1].item()), float(out[2].item())])
    monkeypatch.setattr(torch, "randint", orig)
    assert runner.graph is not None and runner.dedup is not None and runner.dedup["cap"] == 19968
    # (r06) the head launch of the step handed the control block over and built layer 2's weight images from the second call on:
    # the stand-alone image launches ran in the first call only
    assert ncalls.get("evae_batch_prologue_u8_step", 0) >= 3 and ncalls.get("evae_p6_pack_rows", 0) == 1 \
        and ncalls.get("evae_p6_pack_cols", 0) == 1, {k_: v_ for k_, v_ in ncalls.items() if "p6_pack" in k_ or "prologue" in k_}
    # the oracle, same inputs
    opt_state = {}
    po = {k: v.copy() for k, v in p.items()}
    for it in range(nsteps):
        x = data[it * B:(it + 1) * B]; bidx = np.arange(it * B, (it + 1) * B).reshape(-1, 1)
        fwd, _ = orc.vae_train_step(po, opt_state, x, bidx, epss[it], data[draws[it]], draws[it], beta, lr=5e-4)
        ref = [float(np.mean(fwd["loss"])), float(np.mean(fwd["RE"])), float(np.mean(fwd["KL"]))]
        for name, a_, b_ in zip(("loss", "RE", "KL"), got[it], ref):
            sign = -1.0 if name == "RE" and a_ * b_ < 0 else 1.0      # (the loop's running sums carry -RE)
            assert abs(sign * a_ - b_) <= 1e-4 * max(abs(b_), 1e-30), (it, name, a_, b_)
    # Adam's first steps are sign-like (m / sqrt(v) ~ +-1 whatever the gradient's size), so an element whose gradient is at the
    # rounding level of the two implementations may move by lr in either direction: every parameter to 1e-5 of its tensor's largest
    # entry, except at most one element in 10 000, which stays within the 4 x lr such an element can travel in four steps
    worst = {}
    for name, prm in model.named_parameters():
        a_ = prm.detach().cpu().numpy().astype(np.float64); b_ = po[name].astype(np.float64)
        err = np.abs(a_ - b_) / max(np.abs(b_).max(), 1e-30)
        worst[name] = (float(err.max()), float((err > 1e-5).mean()))
        assert np.abs(a_ - b_).max() <= 4 * 5e-4 * 1.01, (name, worst[name])
        assert (err > 1e-5).mean() <= 1e-4, (name, worst[name])
    print("captured c2 steps vs oracle: worst (max rel err, fraction above 1e-5):", max(worst.values()))


@pytest.mark.gpu
def test_single_conv_on_28x28_inputs_window_path_matches_the_r04_kernels():
    """`single_conv` on 1 x 28 x 28 binary inputs (14 x 14 and 7 x 7 grids, bottleneck 6) with a batch large enough for the pixel-image
    window kernels (128 images + 150 exemplars): one training step's per-row loss / RE / KL and every gradient against the same step on
    the r04 kernels (evae.ops.CONV_STACK_ON = False, torch's weight norm) -- the path the reference-generated goldens pin at 4 images."""
    from evae import ops
    from utils.utils import importing_model
    B, C, N = 128, 150, 400
    args = smoke_case.vae_args(model_name="single_conv", input_size=[1, 28, 28], input_type="binary", bottleneck=6, z1_size=294,
                               number_components=C, training_set_size=N, batch_size=B)
    torch.manual_seed(21)
    model = importing_model(args)(args).cuda()
    model.train()
    rs = np.random.RandomState(8)
    data = (rs.random_sample((N, 784)) < 0.3).astype(np.float32)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    idx = torch.from_numpy(rs.randint(0, N, (B, 1)).astype(np.int64))
    x = torch.from_numpy(data[idx[:, 0].numpy()]).cuda()
    ex = torch.from_numpy(rs.randint(0, N, (C,)).astype(np.int64))
    eps = torch.from_numpy(rs.standard_normal((B, 294)).astype(np.float32)).cuda()
    res = []
    for window in (True, False):
        orig, stack_on = torch.randint, ops.CONV_STACK_ON
        try:
            ops.CONV_STACK_ON = window
            os.environ["EVAE_WN_SET"] = "1" if window else "0"
            torch.randint = lambda low=0, high=None, size=None, **kw: ex.clone()
            model._draw_eps = lambda like: eps.reshape(like.shape)
            model.zero_grad()
            loss, RE, KL = model.calculate_loss((x, idx.cuda()), 0.5, average=False, dataset=dataset)
            loss.mean().backward()
        finally:
            torch.randint = orig
            ops.CONV_STACK_ON = stack_on
            os.environ.pop("EVAE_WN_SET", None)
        res.append((np.stack([t.detach().double().cpu().numpy().reshape(-1) for t in (loss, RE, KL)]),
                    [None if p.grad is None else p.grad.double().cpu().numpy() for p in model.parameters()]))
    assert np.isfinite(res[0][0]).all()
    for j, what in enumerate(("loss", "RE", "KL")):
        assert rel(res[0][0][j], res[1][0][j]) < 2e-5, (what, rel(res[0][0][j], res[1][0][j]))
    for a, b in zip(res[0][1], res[1][1]):
        assert (a is None) == (b is None)
        if a is not None:
            assert np.abs(a - b).max() <= 2e-4 * max(np.abs(b).max(), 1e-8)
