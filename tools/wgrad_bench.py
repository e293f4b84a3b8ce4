#!/usr/bin/env python3
"""The two weight gradients of the headline step (c2: 25 100 rows), alone on the machine: entry-point time, the phases of the
byte layer's (gather-transpose / dy split / GEMM + finish), error against float64 on sampled rows of dw.  GPU box only.
  EVAE_SK_LOCAL=0 python tools/wgrad_bench.py     the r02 block order of the fp32 split-K weight gradient"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
import torch
from evae import ops, _lib
lib = _lib.load(); p, st = ops._p, ops._stream
torch.manual_seed(0)
N, D, H, M = 50000, 784, 300, 25100


def timeit(fn, n=20, warm=6):
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


def err(dw, dy, x, rows_pick):
    ref = dy[:, rows_pick].double().t() @ x.double()
    return float((dw[rows_pick].double() - ref).abs().max() / ref.abs().max())


q = (torch.randint(0, 256, (N, D), device="cuda") * (torch.rand(N, D, device="cuda") < 0.2)).to(torch.uint8)
store = torch.zeros(N * D + 64, dtype=torch.uint8, device="cuda"); xs = store[:N * D].view(N, D); xs.copy_(q)
rows = torch.randint(0, N, (M,), device="cuda")
dy = torch.randn(M, 2 * H, device="cuda") * 0.01
dw = torch.empty(2 * H, D, device="cuda"); db = torch.empty(2 * H, device="cuda")
pick = torch.tensor([0, 1, 77, 299, 300, 511, 599], device="cuda")
xg = xs[rows].float() / 255.0

# ---- byte layer (L1): phases
nb = lib.evae_dense_bwd_weight_u8_workspace_bytes(M, 2 * H, D)
w = torch.zeros(nb, dtype=torch.uint8, device="cuda")
call = lambda ph: _lib.check(lib.evae_dense_bwd_weight_u8_phased(p(dy), M, 2 * H, 2 * H, p(xs), p(rows), D, xs.stride(0), 1.0 / 255.0,
                                                                 p(dw), p(db), p(w), nb, ph, st()), "u8 phase")
full = lambda: _lib.check(lib.evae_dense_bwd_weight_u8(p(dy), M, 2 * H, 2 * H, p(xs), p(rows), D, xs.stride(0), 1.0 / 255.0,
                                                       p(dw), p(db), p(w), nb, st()), "u8")
full(); torch.cuda.synchronize()
e1 = err(dw, dy, xg, pick); edb = float((db.double() - dy.double().sum(0)).abs().max() / dy.double().sum(0).abs().max())
fl = 2.0 * M * D * 2 * H
t_full = timeit(full); t3 = timeit(lambda: call(3)); t1 = timeit(lambda: call(1)); t2 = timeit(lambda: call(2)); t4 = timeit(lambda: call(4))
print("L1 byte weight gradient [600 x 784] over %d rows: entry %.1f us (%.1f TFLOP/s alg); gather-transpose %.1f, pre-passes %.1f "
      "(dy split = %.1f), GEMM + finish %.1f, without the gather (as in the step) %.1f; max rel err dw %.2e db %.2e"
      % (M, t_full, fl / t_full / 1e6, t3, t1, t1 - t3, t2, t4, e1, edb))

# ---- fp32 layer 2
a1 = torch.randn(M, H, device="cuda")
dw2 = torch.empty(2 * H, H, device="cuda"); db2 = torch.empty(2 * H, device="cuda")
nb2 = lib.evae_dense_bwd_weight_workspace_bytes(M, 2 * H, H); w2 = torch.zeros(nb2, dtype=torch.uint8, device="cuda")
f2 = lambda: _lib.check(lib.evae_dense_bwd_weight(p(dy), M, 2 * H, 2 * H, p(a1), None, H, H, p(dw2), p(db2), 0, p(w2), nb2, st()), "wgrad2")
f2p = lambda ph: _lib.check(lib.evae_dense_bwd_weight_phased(p(dy), M, 2 * H, 2 * H, p(a1), None, H, H, p(dw2), p(db2), 0, p(w2), nb2, ph, st()), "wgrad2")
f2(); torch.cuda.synchronize()
e2 = err(dw2, dy, a1, pick)
fl2 = 2.0 * M * H * 2 * H
t = timeit(f2); tg = timeit(lambda: f2p(1)); tf = timeit(lambda: f2p(2))
print("L2 fp32 weight gradient [600 x 300] over %d rows: entry %.1f us (%.1f TFLOP/s), GEMM %.1f, finish %.1f; max rel err %.2e"
      % (M, t, fl2 / t / 1e6, tg, tf, e2))
# mean head (the 77-us side-stream launch of the r02 timeline): [40 x 300]
dm = torch.randn(M, 40, device="cuda") * 0.01
dwm = torch.empty(40, H, device="cuda"); dbm = torch.empty(40, device="cuda")
nb3 = lib.evae_dense_bwd_weight_workspace_bytes(M, 40, H); w3 = torch.zeros(nb3, dtype=torch.uint8, device="cuda")
f3 = lambda: _lib.check(lib.evae_dense_bwd_weight(p(dm), M, 40, 40, p(a1), None, H, H, p(dwm), p(dbm), 0, p(w3), nb3, st()), "wgrad head")
f3(); torch.cuda.synchronize()
e3 = float((dwm.double() - dm.double().t() @ a1.double()).abs().max() / (dm.double().t() @ a1.double()).abs().max())
print("mean-head weight gradient [40 x 300]: %.1f us; max rel err %.2e" % (timeit(f3), e3))
