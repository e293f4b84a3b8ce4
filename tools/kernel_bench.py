#!/usr/bin/env python3
"""Per-kernel timing at the c2 shapes (HIP events, median of N launches).  GPU box only."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
import torch
from evae import ops, _lib

lib = _lib.load()
dev = torch.device("cuda")
p, st = ops._p, ops._stream


def timeit(fn, n=20, warm=3):
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


def report(name, us, flops):
    print("%-52s %9.1f us  %7.1f TFLOP/s  (%.1f%% of 157.3)" % (name, us, flops / us / 1e6, 100 * flops / us / 1e6 / 157.3))


def main():
    torch.manual_seed(0)
    N, D, H, Z = 50000, 784, 300, 40
    data = (torch.rand(N, D, device=dev) < 0.13).float()
    for M in (25000, 3125, 100):
        rows = torch.randint(0, N, (M,), device=dev)
        wh = torch.randn(H, D, device=dev) * 0.05; wg = torch.randn(H, D, device=dev) * 0.05
        b = torch.zeros(H, device=dev)
        out = torch.empty(M, H, device=dev); h = torch.empty_like(out); s = torch.empty_like(out)
        wsf = torch.zeros(64 << 20, dtype=torch.uint8, device=dev); nwf = wsf.numel()
        report("gated_fwd L1 M=%d K=784 N=300 (gather)" % M,
               timeit(lambda: lib.evae_gated_dense_fwd(p(data), p(rows), M, D, D, p(wh), p(b), p(wg), p(b), H, p(out), None, p(s), p(wsf), nwf, st())),
               2.0 * M * D * 2 * H)
        w2h = torch.randn(H, H, device=dev) * 0.05; w2g = torch.randn(H, H, device=dev) * 0.05
        out2 = torch.empty(M, H, device=dev)
        report("gated_fwd L2 M=%d K=300 N=300" % M,
               timeit(lambda: lib.evae_gated_dense_fwd(p(out), None, M, H, H, p(w2h), p(b), p(w2g), p(b), H, p(out2), None, p(s), p(wsf), nwf, st())),
               2.0 * M * H * 2 * H)
        wm = torch.randn(Z, H, device=dev) * 0.05; bm = torch.zeros(Z, device=dev); y = torch.empty(M, Z, device=dev)
        report("linear_fwd mean M=%d K=300 N=40" % M,
               timeit(lambda: lib.evae_linear_fwd(p(out), None, M, H, H, p(wm), p(bm), Z, 0, 0.0, 0.0, p(y), None, p(wsf), nwf, st())),
               2.0 * M * H * Z)
        dpre = torch.randn(M, 2 * H, device=dev); dh = dpre; dx = torch.empty(M, H, device=dev); dg2 = torch.empty(M, H, device=dev)
        vp = lambda a: C.c_void_p(a)
        report("bwd_data dual M=%d N=300+300 K=300 (+gate epi)" % M,
               timeit(lambda: lib.evae_dense_bwd_data(p(dpre), p(w2h), vp(dpre.data_ptr() + 4 * H), p(w2g), M, H, 2 * H, H, p(h), p(s), p(dx), p(dg2), H, p(wsf), nwf, st())),
               2.0 * M * 2 * H * H)
        dy = torch.randn(M, Z, device=dev)
        report("bwd_data mean M=%d N=40 K=300" % M,
               timeit(lambda: lib.evae_dense_bwd_data(p(dy), p(wm), None, None, M, Z, Z, H, None, None, p(dx), None, H, p(wsf), nwf, st())),
               2.0 * M * Z * H)
        nb = max(lib.evae_dense_bwd_weight_workspace_bytes(M, 2 * H, D), lib.evae_dense_bwd_weight_workspace_bytes(M, 2 * H, H))
        ws = torch.zeros(nb, dtype=torch.uint8, device=dev); dw = torch.empty(2 * H, D, device=dev); db = torch.empty(2 * H, device=dev)
        report("bwd_weight L1 M=%d N=600 K=784 (gather, +db)" % M,
               timeit(lambda: lib.evae_dense_bwd_weight(p(dpre), M, 2 * H, 2 * H, p(data), p(rows), D, D, p(dw), p(db), 0, p(ws), nb, st())),
               2.0 * M * 2 * H * D)
        dw2 = torch.empty(2 * H, H, device=dev)
        report("bwd_weight L2 M=%d N=600 K=300 (+db)" % M,
               timeit(lambda: lib.evae_dense_bwd_weight(p(dpre), M, 2 * H, 2 * H, p(out), None, H, H, p(dw2), p(db), 0, p(ws), nb, st())),
               2.0 * M * 2 * H * H)
    # prior
    B, Cn = 100, 25000
    z = torch.randn(B, Z, device=dev); c = torch.randn(Cn, Z, device=dev); lv = torch.full((Z,), -1.0, device=dev)
    zi = torch.randint(0, N, (B,), device=dev); ci = torch.randint(0, N, (Cn,), device=dev)
    us = timeit(lambda: ops.prior_lse_fwd(z, c, lv, zi, ci))
    print("prior fwd B=100 C=25000: %.1f us" % us)
    m, s_, n, _ = ops.prior_lse_fwd(z, c, lv, zi, ci); lp, lse = ops.prior_merge(m, s_, n, Cn); g = torch.randn(B, device=dev)
    print("prior bwd B=100 C=25000: %.1f us" % timeit(lambda: ops.prior_lse_bwd(z, c, lv, zi, ci, lse, g)))
    S, Cn2 = 5000, 50000
    z2 = torch.randn(S, Z, device=dev); c2 = torch.randn(Cn2, Z, device=dev)
    us = timeit(lambda: ops.prior_lse_fwd(z2, c2, lv), n=10)
    print("prior fwd S=5000 C=50000 (one IWAE image): %.1f us -> %.2f G pair/s, %.1f TFLOP/s (3 flop/pair-dim)" % (us, S * Cn2 / us / 1e3, 3.0 * S * Cn2 * Z / us / 1e6))
    q = torch.randn(100, Z, device=dev)
    print("topk k=10 B=100 N=25000 z=40: %.1f us" % timeit(lambda: ops.pairdist_topk(q, c, 10)))
    q5 = torch.randn(64, 256, device=dev); c5 = torch.randn(100000, 256, device=dev)
    us = timeit(lambda: ops.pairdist_topk(q5, c5, 10), n=10)
    print("topk k=10 B=64 N=100000 z=256: %.1f us -> %.1f GB/s cache scan" % (us, 100000 * 256 * 4 / us / 1e3))


if __name__ == "__main__":
    main()
