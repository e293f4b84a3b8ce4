#!/usr/bin/env python3
"""One gated conv layer of convhvae_2level's encoder, forward + backward, for a rocprofv3 kernel trace:
   tools/conv_layer_trace.py <layer 1..5> <images> [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
import torch
from evae import ops
L = int(sys.argv[1]); N = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
layers = [(1, 28, 32, 7, 1, 3), (32, 28, 32, 3, 2, 1), (32, 14, 64, 5, 1, 2), (64, 14, 64, 3, 2, 1), (64, 7, 6, 3, 1, 1)]
C, H, Co, k, s, p = layers[L - 1]
dev = torch.device("cuda")
x = torch.randn(N, C, H, H, device=dev, requires_grad=(C > 1))
wh = (torch.randn(Co, C, k, k, device=dev) * 0.05).requires_grad_(True)
wg = (torch.randn(Co, C, k, k, device=dev) * 0.05).requires_grad_(True)
bh = torch.zeros(Co, device=dev, requires_grad=True); bg = torch.zeros(Co, device=dev, requires_grad=True)
y = ops.gated_conv2d(x, wh, bh, wg, bg, s, p)
g = torch.randn_like(y)
for _ in range(reps):
    yy = ops.gated_conv2d(x, wh, bh, wg, bg, s, p)
    yy.backward(g)
torch.cuda.synchronize()
