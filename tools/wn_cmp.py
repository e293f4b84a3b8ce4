import os, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "exemplar-vae_amd"); sys.path.insert(0, "oracle")   # (a debugging tool: test helpers only)
import smoke_case
from test_gpu_model import G9_CASES, seeded_state_dict
from utils.utils import importing_model
def run(flag, perturb=False):
    os.environ["EVAE_WN_SET"] = flag
    cfg = dict(G9_CASES["single_conv"]); B, C, N = cfg.pop("B"), cfg.pop("C"), cfg.pop("N"); gain = cfg.pop("gain", 1.0)
    args = smoke_case.vae_args(number_components=C, training_set_size=N, **cfg)
    model = importing_model(args)(args); model.load_state_dict(seeded_state_dict(model, 77, gain)); model = model.to("cuda")
    if perturb:
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith("weight_g"): p.mul_(1 + 1.2e-7)
    D = int(np.prod(args.input_size)); rs = np.random.RandomState(91)
    x = torch.from_numpy(((rs.randint(0, 256, (B, D)) + 0.5) / 256).astype(np.float32)).cuda()
    torch.manual_seed(3)
    model.train(); model.zero_grad()
    mu, lv = model.q_z(x)
    xm, xl = model.p_x(mu)
    (xm.square().sum() + (mu * lv).sum()).backward()
    return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, mu.detach().clone(), xm.detach().clone()
a, mua, xa = run("0"); b, mub, xb = run(sys.argv[1] if len(sys.argv) > 1 else "1", len(sys.argv) > 2)
print("mu diff", (mua - mub).abs().max().item(), "xm diff", (xa - xb).abs().max().item() / xa.abs().max().item())
for n in a:
    d = (a[n] - b[n]).abs().max().item() / max(a[n].abs().max().item(), 1e-12)
    if d > 1e-5 and ("8" in n or "mean" in n): print("%-40s rel diff %.3e  shape %s" % (n, d, tuple(a[n].shape)))
