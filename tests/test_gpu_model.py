"""GPU parity of the drop-in model API against the goldens generated from the real reference
(tests/golden/g7_vae_loss.npz) and against the oracle."""
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


def test_graphed_step_matches_eager():
    """hipGraph replay of the whole step (evae/graph.py) == the same steps launched eagerly."""
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
        results.append((losses, {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}))
        if runner is not None:
            assert runner.graph is not None                       # steps 4.. were replays
    (l0, p0), (l1, p1) = results
    assert rel(np.asarray(l1), np.asarray(l0)) < 1e-5
    for k in p0:
        assert rel(p1[k], p0[k]) < 1e-5, k
