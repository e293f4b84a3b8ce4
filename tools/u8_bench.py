#!/usr/bin/env python3
"""Encoder layer 1 forward at the c2 shape: fp32 store + fp32 MFMA vs uint8 store + three-term bf16 MFMA.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
import torch
from evae import ops, _lib
if os.environ.get("EVAE_LIB_PATH"):
    _lib.LIB_PATH = os.environ["EVAE_LIB_PATH"]
lib = _lib.load(); p, st = ops._p, ops._stream
torch.manual_seed(0)
N, D, H, M = 50000, 784, 300, 25000


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); fn(); fn(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / 4)
    ts.sort()
    return ts[len(ts) // 2]


q = (torch.randint(0, 256, (N, D), device="cuda") * (torch.rand(N, D, device="cuda") < 0.2)).to(torch.uint8)
store = torch.zeros(N * D + 64, dtype=torch.uint8, device="cuda"); xs = store[:N * D].view(N, D); xs.copy_(q)
xf = q.float() / 255.0
rows = torch.randint(0, N, (M,), device="cuda")
wh = torch.randn(H, D, device="cuda") * 0.05; wg = torch.randn(H, D, device="cuda") * 0.05; b = torch.zeros(H, device="cuda")
out = torch.empty(M, H, device="cuda"); s = torch.empty_like(out)
ws = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
fl = 2.0 * M * D * 2 * H
us = timeit(lambda: lib.evae_gated_dense_fwd(p(xf), p(rows), M, D, D, p(wh), p(b), p(wg), p(b), H, p(out), None, p(s), p(ws), ws.numel(), st()))
print("fp32 store, fp32 MFMA : %7.1f us  %6.1f TFLOP/s" % (us, fl / us / 1e6))
prep = ops.u8_prepare(wh, wg)
us_p = timeit(lambda: ops.u8_prepare(wh, wg, out=prep))
us = timeit(lambda: ops.gated_dense_fwd_u8(xs, rows, 1.0 / 255.0, prep, b, b, H, out=out, save_s=s))
print("uint8 store, 3 x bf16 : %7.1f us  %6.1f TFLOP/s fp32-equivalent (%.1f executed bf16 TFLOP/s = %.3f of 2500); weight split %.1f us"
      % (us, fl / us / 1e6, 3 * fl / us / 1e6, 3 * fl / us / 1e6 / 2500, us_p))

dy = torch.randn(M + 100, 2 * H, device="cuda") * 0.01
rows2 = torch.randint(0, N, (M + 100,), device="cuda")
dw = torch.empty(2 * H, D, device="cuda"); db = torch.empty(2 * H, device="cuda")
nb = lib.evae_dense_bwd_weight_workspace_bytes(M + 100, 2 * H, D); ws2 = torch.zeros(nb, dtype=torch.uint8, device="cuda")
fl = 2.0 * (M + 100) * D * 2 * H
us = timeit(lambda: lib.evae_dense_bwd_weight(p(dy), M + 100, 2 * H, 2 * H, p(xf), p(rows2), D, D, p(dw), p(db), 0, p(ws2), nb, st()))
print("weight gradient fp32  : %7.1f us  %6.1f TFLOP/s" % (us, fl / us / 1e6))
us = timeit(lambda: ops.dense_bwd_weight_u8(dy, xs, rows2, 1.0 / 255.0, dw=dw, db=db))
print("weight gradient uint8 : %7.1f us  %6.1f TFLOP/s fp32-equivalent (gather-transpose + dy split + GEMM + finish)" % (us, fl / us / 1e6))
