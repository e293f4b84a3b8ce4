import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
import torch
from evae import ops
torch.manual_seed(0)
B, C, Z = 128, 127, 8
z = torch.randn(B, Z, device="cuda") * 1.5
c = torch.randn(C, Z, device="cuda")
lv = torch.randn(Z, device="cuda") * 0.3 - 1.0
m, s, n, _ = ops.prior_lse_fwd(z, c, lv, None, None)
lp, lse = ops.prior_merge(m, s, n, C)
def ref(cc, count):
    zz, c2, l2 = z.double(), cc.double(), lv.double()
    d = (((zz[:, None, :] - c2[None]) ** 2) * torch.exp(-l2)).sum(-1)
    p = -0.5 * d - 0.5 * (l2 + math.log(2 * math.pi)).sum()
    return torch.logsumexp(p, 1) - math.log(count)
r0 = ref(c, C)
r1 = ref(torch.cat([c, torch.zeros(1, Z, device="cuda")]), C)
print("vs exact      ", (lp.double() - r0).abs().max().item())
print("vs +zero row  ", (lp.double() - r1).abs().max().item())
bad = (lp.double() - r0).abs() > 1e-3
print("bad queries:", bad.nonzero().flatten().tolist()[:40], "of", int(bad.sum()))
