#!/usr/bin/env python3
"""Forward time of single convolutions at fully_conv's shapes (HIP events).  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
import torch
from evae import ops
torch.manual_seed(0)


def timeit(fn, n=10, warm=3):
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


for (N, C, H, Co, k, s) in ((100, 48, 64, 48, 3, 1), (100, 48, 32, 48, 3, 1), (100, 96, 32, 96, 3, 1), (100, 96, 16, 96, 3, 1),
                            (100, 64, 64, 64, 3, 1), (100, 32, 64, 32, 3, 1), (100, 48, 64, 3, 3, 1), (100, 96, 64, 48, 3, 1)):
    x = torch.randn(N, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, C, k, k, device="cuda") * 0.03; b = torch.zeros(Co, device="cuda")
    with torch.no_grad():
        us = timeit(lambda: ops.conv2d(x, w, b, s, 1))
    fl = 2.0 * N * (H // s) * (H // s) * C * k * k * Co
    print("conv %3d -> %3d %dx%d %3dx%3d N=%d: %8.1f us %6.1f TFLOP/s" % (C, Co, k, k, H, H, N, us, fl / us / 1e6), flush=True)
