#!/usr/bin/env python3
"""GatedDense forward L1 at M values that give an exact number of block rounds, under the EVAE_GEMM_DBG switches the
shipped kernel still carries (4: skip the epilogue, 512: shader-clock probe).  The in-loop ablations quoted in DESIGN.md
(no loads / no stores / no barrier / L1-resident loads) were compile-time experiments of round 1 and are gone.
Each setting runs in a fresh process (the flag is read once); EVAE_LIB_PATH selects an alternative build for A/B runs."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
    import torch
    from evae import ops, _lib
    if os.environ.get("EVAE_LIB_PATH"): _lib.LIB_PATH = os.environ["EVAE_LIB_PATH"]   # A/B builds on the same box
    lib = _lib.load(); dev = torch.device("cuda"); p, st = ops._p, ops._stream
    torch.manual_seed(0)
    N, D, H = 50000, 784, 300
    data = (torch.rand(N, D, device=dev) < 0.13).float()
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
    wh = torch.randn(H, D, device=dev) * 0.05; wg = torch.randn(H, D, device=dev) * 0.05; b = torch.zeros(H, device=dev)
    for M in [int(x) for x in sys.argv[2].split(",")]:
        rows = torch.randint(0, N, (M,), device=dev)
        if os.environ.get("ROWS") == "seq": rows = torch.arange(M, device=dev) % N
        if os.environ.get("ROWS") == "none": rows = None; data = torch.cat([data] * (1 + M // N))[:max(M, N)]
        out = torch.empty(M, H, device=dev); h = torch.empty_like(out); s = torch.empty_like(out)
        fn = lambda: lib.evae_gated_dense_fwd(p(data), p(rows) if rows is not None else None, M, D, D, p(wh), p(b), p(wg), p(b), H, p(out), None, p(s), p(ws), ws.numel(), st())
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            a.record(); fn(); e.record(); e.synchronize(); ts.append(a.elapsed_time(e) * 1e3)
        ts.sort(); us = ts[len(ts) // 2]
        if int(os.environ.get("EVAE_GEMM_DBG", "0")) & 512:
            s.zero_(); fn(); torch.cuda.synchronize()
            c, w, n = s[M - 1, :3].tolist()
            print("    clock probe: %d blocks, %.0f shader ticks / %.0f ticks@100MHz per block -> %.0f MHz" % (n, c / n, w / n, c / w * 100.0))
        blocks = ((M + 127) // 128) * 5
        print("  dbg=%s M=%6d blocks=%5d: %7.1f us  %6.1f TFLOP/s (useful)  %6.1f (padded)" % (
            os.environ.get("EVAE_GEMM_DBG", "0") + " " + os.path.basename(os.environ.get("EVAE_LIB_PATH", "")) + " rows=" + os.environ.get("ROWS", "rand"), M, blocks, us, 2.0 * M * D * 2 * H / us / 1e6,
            2.0 * blocks * 128 * 128 * 800 / us / 1e6))
else:
    Ms = sys.argv[2] if len(sys.argv) > 2 else "13056,25000,26112,52224,104448"
    for dbg in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("0", "4", "512")):
        env = dict(os.environ, EVAE_GEMM_DBG=dbg)
        subprocess.run([sys.executable, __file__, "child", Ms], env=env, check=False)
