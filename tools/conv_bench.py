#!/usr/bin/env python3
"""Timing of the implicit-GEMM convolution kernels on the layer shapes of convhvae_2level (N images)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
import torch
from evae import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda")


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


layers = [(1, 28, 32, 7, 1, 3), (32, 28, 32, 3, 2, 1), (32, 14, 64, 5, 1, 2), (64, 14, 64, 3, 2, 1), (64, 7, 6, 3, 1, 1)]
tot = {"fwd": 0.0, "bwd": 0.0}
for (C, H, Co, k, s, p) in layers:
    x = torch.randn(N, C, H, H, device=dev, requires_grad=(C > 1))      # the first layer reads data: no dx
    wh = (torch.randn(Co, C, k, k, device=dev) * 0.05).requires_grad_(True)
    wg = (torch.randn(Co, C, k, k, device=dev) * 0.05).requires_grad_(True)
    bh = torch.zeros(Co, device=dev, requires_grad=True); bg = torch.zeros(Co, device=dev, requires_grad=True)
    OH = (H + 2 * p - k) // s + 1
    flops = 2.0 * N * OH * OH * Co * 2 * C * k * k
    y = ops.gated_conv2d(x, wh, bh, wg, bg, s, p)
    g = torch.randn_like(y)
    t_f = timeit(lambda: ops.gated_conv2d(x, wh, bh, wg, bg, s, p))

    def fb():
        yy = ops.gated_conv2d(x, wh, bh, wg, bg, s, p)
        yy.backward(g)
    t_fb = timeit(fb)
    tot["fwd"] += t_f; tot["bwd"] += t_fb - t_f
    print("gated conv C=%2d H=%2d Co=%2d k=%d s=%d: fwd %8.1f us (%5.1f TFLOP/s)  fwd+bwd %8.1f us (bwd %5.1f TFLOP/s)"
          % (C, H, Co, k, s, t_f, flops / t_f / 1e6, t_fb, 2 * flops / max(t_fb - t_f, 1e-9) / 1e6))
print("encoder q_z_layers, N=%d images: fwd %.2f ms, bwd %.2f ms" % (N, tot["fwd"] / 1e3, tot["bwd"] / 1e3))
