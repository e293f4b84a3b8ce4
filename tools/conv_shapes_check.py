#!/usr/bin/env python3
"""Every convolution shape of models/fully_conv.py (c5) through evae.ops against torch's own conv2d, one by one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
import torch
import torch.nn.functional as F
from evae import ops
torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shapes = [(3, 64, 64, 64, 2), (64, 64, 32, 32, 1), (64, 128, 32, 32, 2), (128, 128, 16, 16, 1), (128, 1, 16, 16, 1),
          (1, 128, 32, 32, 1), (128, 64, 64, 64, 1), (64, 3, 64, 64, 1)]
for (C, Co, H, W, s) in shapes:
    x = torch.randn(N, C, H, W, device="cuda", requires_grad=True)
    w = (torch.randn(Co, C, 3, 3, device="cuda") / (3 * C ** 0.5)).requires_grad_(True)
    b = torch.randn(Co, device="cuda").requires_grad_(True)
    y = ops.conv2d(x, w, b, s, 1)
    torch.cuda.synchronize(); print("fwd ok", (C, Co, H, W, s), flush=True)
    g = torch.randn_like(y)
    y.backward(g)
    torch.cuda.synchronize(); print("bwd ok", flush=True)
    if N > 64:      # big batches: only that nothing faults and the head of the batch matches a small run
        x2 = x.detach()[:4].clone().requires_grad_(True)
        y2 = ops.conv2d(x2, w.detach(), b.detach(), s, 1)
        print("   big batch: head rel %.1e" % float((y[:4] - y2).abs().max() / y2.abs().max()), flush=True)
        continue
    xr, wr, br = [t.detach().double().requires_grad_(True) for t in (x, w, b)]
    yr = F.conv2d(xr, wr, br, s, 1); yr.backward(g.double())
    rel = lambda a, c: float((a.double() - c).abs().max() / c.abs().max())
    print("   rel: y %.1e dx %.1e dw %.1e db %.1e" % (rel(y, yr), rel(x.grad, xr.grad), rel(w.grad, wr.grad), rel(b.grad, br.grad)), flush=True)
