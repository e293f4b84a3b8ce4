#!/usr/bin/env python3
"""What the vendor fp32 GEMM reaches on this box at our shapes (a yardstick, not part of the product)."""
import torch
dev = torch.device("cuda")
torch.backends.cuda.matmul.allow_tf32 = False
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / n
for (M, K, N, tb) in [(25000, 784, 600, True), (25000, 300, 600, True), (25000, 600, 300, False), (600, 25000, 784, False),
                      (8192, 8192, 8192, True), (104448, 784, 600, True), (25088, 800, 640, True)]:
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev)
    fn = (lambda: A @ B.t()) if tb else (lambda: A @ B)
    us = t(fn)
    print("M=%6d K=%6d N=%5d %s: %8.1f us  %6.1f TFLOP/s" % (M, K, N, "NT" if tb else "NN", us, 2.0 * M * K * N / us / 1e6))
