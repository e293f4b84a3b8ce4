"""Split-bf16 fp32 GEMM (csrc/evae_gemm_x6.h) against the fp32-MFMA kernel on the layer shapes of the headline step:
time per launch (HIP events) and error against float64.  python tools/x6_bench.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "exemplar-vae_amd"))
from evae import ops, _lib
if os.environ.get('EVAE_LIB_PATH'):
    _lib.LIB_PATH = os.environ['EVAE_LIB_PATH']


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M, K, N in ((25100, 300, 300), (25100, 784, 300), (100000, 300, 300), (5000, 300, 300)):
    torch.manual_seed(0)
    x = torch.randn(M, K, device="cuda"); wh = torch.randn(N, K, device="cuda") / K ** 0.5; wg = torch.randn(N, K, device="cuda") / K ** 0.5
    bh = torch.randn(N, device="cuda") * 0.1; bg = torch.randn(N, device="cuda") * 0.1
    ref = ((x.double() @ wh.double().T + bh.double()) * torch.sigmoid(x.double() @ wg.double().T + bg.double()))
    fl = 2.0 * M * K * 2 * N
    for on in (1, 0):
        ops.gemm_x6_configure(on, 0)
        with torch.no_grad():
            us = timeit(lambda: ops.gated_dense(x, wh, bh, wg, bg))
            out = ops.gated_dense(x, wh, bh, wg, bg)
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        print("gated fwd M=%d K=%d N=%d  %s: %.1f us  %.1f TFLOP/s (fp32-equivalent)  max rel err %.2e" %
              (M, K, N, "x6  " if on else "fp32", us, fl / us / 1e6, err), flush=True)

# data gradient of a gated layer with the gate derivative of the layer below in the epilogue (the L2 shape of the headline step)
for M, N, K in ((25000, 300, 300), (25000, 40, 300), (100000, 300, 300)):
    torch.manual_seed(1)
    dy = torch.randn(M, 2 * N, device="cuda"); w1 = torch.randn(N, K, device="cuda") / N ** 0.5; w2 = torch.randn(N, K, device="cuda") / N ** 0.5
    outp = torch.randn(M, K, device="cuda"); sp = torch.rand(M, K, device="cuda")
    buf = torch.empty(M, 2 * K, device="cuda")
    dxr = dy[:, :N].double() @ w1.double() + dy[:, N:].double() @ w2.double()
    ref = dxr * sp.double()
    fl = 2.0 * M * K * 2 * N
    for on in (1, 0):
        ops.gemm_x6_configure(on, 0)
        fn = lambda: ops._bwd_data(dy.data_ptr(), w1, dy.data_ptr() + 4 * N, w2, M, N, 2 * N, dy.device, out_prev=outp, s_prev=sp,
                                   out=buf.data_ptr(), dg_ptr=buf.data_ptr() + 4 * K, ldo=2 * K)
        us = timeit(fn)
        err = float((buf[:, :K].double() - ref).abs().max() / ref.abs().max())
        print("gate dgrad M=%d N=%d+%d K=%d  %s: %.1f us  %.1f TFLOP/s (fp32-equivalent)  max rel err dh %.2e" %
              (M, N, N, K, "x6  " if on else "fp32", us, fl / us / 1e6, err), flush=True)

# weight gradient dW = dy^T x (+ db): both operands k-major -> gemm_x6t_kernel
for M, N, K in ((25100, 600, 300), (25100, 600, 784), (100000, 600, 300)):
    torch.manual_seed(2)
    dy = torch.randn(M, N, device="cuda"); x = torch.randn(M, K, device="cuda")
    ref = dy.double().T @ x.double()
    fl = 2.0 * M * N * K
    lib = _lib.load()
    for on in (1, 0):
        ops.gemm_x6_configure(on, 0)
        dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
        nb = lib.evae_dense_bwd_weight_workspace_bytes(M, N, K)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        fn = lambda: _lib.check(lib.evae_dense_bwd_weight(ops._p(dy), M, N, N, ops._p(x), None, K, K, ops._p(dw), ops._p(db), 0,
                                                          ops._p(ws), ws.numel(), ops._stream()), "wgrad")
        us = timeit(fn)
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
        errb = float((db.double() - dy.double().sum(0)).abs().max() / dy.double().sum(0).abs().max())
        print("weight grad M=%d N=%d K=%d  %s: %.1f us  %.1f TFLOP/s (fp32-equivalent)  max rel err dw %.2e db %.2e" %
              (M, N, K, "x6  " if on else "fp32", us, fl / us / 1e6, err, errb), flush=True)
