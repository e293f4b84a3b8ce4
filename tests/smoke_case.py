"""One small `vae` training step on cuda:0 through the drop-in API (models.VAE.VAE.calculate_loss ->
backward -> utils.optimizer.AdamNormGrad.step), checked against the numpy oracle on identical inputs.
Shared by __graft_entry__.smoke() and tests/test_gpu_model.py."""
from argparse import Namespace


def vae_args(**kw):
    a = dict(prior="exemplar_prior", input_type="binary", input_size=[1, 28, 28], hidden_size=300,
             z1_size=40, z2_size=40, model_name="vae", device="cuda", number_components=1000,
             training_set_size=50000, approximate_prior=False, approximate_k=10, no_mask=False,
             no_attention=False, same_variational_var=False, use_logit=False, lambd=1e-4,
             bottleneck=6, dataset_name="dynamic_mnist", continuous=False, batch_size=100,
             dynamic_binarization=False, warmup=100, S=50)
    a.update(kw)
    return Namespace(**a)


def make_case(np, B, C, N, seed, gi):
    data = gi.gray_images(seed, N)
    rs = np.random.RandomState(seed + 1)
    bidx = rs.randint(0, N, size=(B, 1)).astype(np.int64)
    x = (rs.random_sample((B, 784)) < np.clip(data[bidx[:, 0]] + 0.1, 0, 1)).astype(np.float32)
    eps = rs.standard_normal((B, 40)).astype(np.float32)
    ex_idx = rs.randint(0, N, size=(C,)).astype(np.int64)
    ex_idx[:3] = bidx[:3, 0]
    return data, bidx, x, eps, ex_idx


def build_model(torch, np, orc, args, seed=123):
    from models.VAE import VAE
    model = VAE(args).to(args.device)
    p = orc.vae_init_params(np.random.RandomState(seed))
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
    return model, p


def rel(np, a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def run(torch, np, orc, B=16, C=200, N=500, seed=61, beta=0.37, verbose=False, tol=1e-4, fused=True):
    import golden_inputs as gi
    from utils.optimizer import AdamNormGrad
    args = vae_args(number_components=C, training_set_size=N)
    model, p = build_model(torch, np, orc, args)
    data, bidx, x, eps, ex_idx = make_case(np, B, C, N, seed, gi)
    dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
    model._draw_eps = lambda like: torch.from_numpy(eps).to(like.device)
    model._use_fused = fused
    orig_randint = torch.randint
    torch.randint = lambda low=0, high=None, size=None, **kw: torch.from_numpy(ex_idx)
    opt = AdamNormGrad(model.parameters(), lr=5e-4)
    try:
        model.train()
        opt.zero_grad()
        loss, RE, KL = model.calculate_loss((torch.from_numpy(x).cuda(), torch.from_numpy(bidx).cuda()),
                                            beta=beta, average=False, dataset=dataset)
        loss.mean().backward()
    finally:
        torch.randint = orig_randint
    fwd = orc.vae_calculate_loss(p, x, bidx, eps, ("images", data[ex_idx], ex_idx), beta=beta)
    errs = {k: rel(np, v.detach().cpu().numpy(), fwd[k]) for k, v in (("loss", loss), ("RE", RE), ("KL", KL))}
    grads = orc.vae_loss_backward(p, x, bidx, eps, fwd, beta=beta)
    gerr = {}
    for name, prm in model.named_parameters():
        gerr[name] = rel(np, prm.grad.cpu().numpy(), grads[name])
    # optimizer parity on IDENTICAL gradients (Adam's first step is sign-like, so feeding it the
    # oracle's gradients instead would amplify 1e-6 gradient noise on near-zero entries)
    dev_grads = {name: prm.grad.cpu().numpy().copy() for name, prm in model.named_parameters()}
    opt.step()
    perr = {}
    for name, prm in model.named_parameters():
        ref, _, _ = orc.adam_normgrad_step(p[name], dev_grads[name], np.zeros_like(p[name]),
                                           np.zeros_like(p[name]), 1, lr=5e-4)
        perr[name] = rel(np, prm.detach().cpu().numpy(), ref)
    if verbose:
        print("loss/RE/KL rel err:", errs)
        print("max grad rel err: %.3g   max param rel err after AdamNormGrad: %.3g"
              % (max(gerr.values()), max(perr.values())))
    assert max(errs.values()) < tol, errs
    assert max(gerr.values()) < 2e-5, gerr
    assert max(perr.values()) < 1e-6, perr
    return errs, gerr, perr
