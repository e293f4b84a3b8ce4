#!/usr/bin/env python3
"""Launch one kernel family a few times (for rocprofv3 --pmc passes).  usage: gemm_probe.py <which> [reps]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
import torch
from evae import ops, _lib
lib = _lib.load(); dev = torch.device("cuda"); p, st = ops._p, ops._stream
which = sys.argv[1] if len(sys.argv) > 1 else "fwd1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
N, D, H, Z, M = 50000, 784, 300, 40, 25100
data = (torch.rand(N, D, device=dev) < 0.13).float()
rows = torch.randint(0, N, (M,), device=dev)
wh = torch.randn(H, D, device=dev) * 0.05; wg = torch.randn(H, D, device=dev) * 0.05; b = torch.zeros(H, device=dev)
out = torch.empty(M, H, device=dev); h = torch.empty_like(out); s = torch.empty_like(out)
ws = torch.zeros(256 << 20, dtype=torch.uint8, device=dev); nws = ws.numel()
dpre = torch.randn(M, 2 * H, device=dev); dw = torch.empty(2 * H, D, device=dev); db = torch.empty(2 * H, device=dev)
w2h = torch.randn(H, H, device=dev) * 0.05; w2g = torch.randn(H, H, device=dev) * 0.05
dx = torch.empty(M, 2 * H, device=dev)
for _ in range(reps):
    if which == "fwd1":
        lib.evae_gated_dense_fwd(p(data), p(rows), M, D, D, p(wh), p(b), p(wg), p(b), H, p(out), None, p(s), p(ws), nws, st())
    elif which == "wgrad1":
        lib.evae_dense_bwd_weight(p(dpre), M, 2 * H, 2 * H, p(data), p(rows), D, D, p(dw), p(db), 0, p(ws), nws, st())
    elif which == "dgrad2":
        lib.evae_dense_bwd_data(p(dpre), p(w2h), C.c_void_p(dpre.data_ptr() + 4 * H), p(w2g), M, H, 2 * H, H, p(h), p(s), p(dx), C.c_void_p(dx.data_ptr() + 4 * H), 2 * H, p(ws), nws, st())
torch.cuda.synchronize()
