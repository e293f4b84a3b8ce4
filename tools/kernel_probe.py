#!/usr/bin/env python3
"""Launch ONE kernel family a few times, at the shape it has in a benchmarked configuration, for `rocprofv3 --pmc` /
`--kernel-trace --stats` passes (tools/profile_round2.sh).  usage: kernel_probe.py <which> [reps]
  fwd1        GatedDense forward, encoder layer 1 at c2: 25 000 gathered rows x 784 -> 2 x 300
  u8fwd1 / u8wgrad1   the same layer on the uint8 store (three-term bf16 MFMA): forward / weight gradient
  fwd2        GatedDense forward, encoder layer 2: 25 000 x 300 -> 2 x 300
  dgrad2      data gradient of encoder layer 2 (dual pair, gate-backward epilogue)
  wgrad1      weight gradient of encoder layer 1 ([600 x 784] + db, gathered rows)
  wgrad2      weight gradient of encoder layer 2
  prior_iwae  prior forward, 4 x 5000 importance samples x 50 000 exemplars, z = 40 (prior_fwd_mfma_kernel)
  prior_c5    prior forward, 5000 samples x 100 000 exemplars, z = 256 (GEMM + log-sum-exp epilogue)
  prior_train prior forward + backward at the training shape B = 100, C = 25 000, z = 40
  topk_c2 / topk_c5   evae_pairdist_topk, k = 10 (100 x 25 000 x 40 / 100 x 100 000 x 256)
  conv5_fwd / conv5_bwd   gated conv 32 -> 64, 5 x 5, 14 x 14, 25 000 images (c3): forward / data + weight gradient
  conv96_fwd  conv 96 -> 96, 3 x 3, 16 x 16, 1100 images (c5)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd")); sys.path.insert(1, os.path.join(ROOT, "tests"))
import torch
from evae import ops, _lib
lib = _lib.load(); dev = torch.device("cuda"); p, st = ops._p, ops._stream
which = sys.argv[1] if len(sys.argv) > 1 else "fwd1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
N, D, H, Z, M = 50000, 784, 300, 40, 25000
vp = lambda a: C.c_void_p(a)


def dense_setup():
    g = {}
    g["data"] = (torch.rand(N, D, device=dev) < 0.13).float()
    g["rows"] = torch.randint(0, N, (M,), device=dev)
    g["wh"] = torch.randn(H, D, device=dev) * 0.05; g["wg"] = torch.randn(H, D, device=dev) * 0.05; g["b"] = torch.zeros(H, device=dev)
    g["out"] = torch.empty(M, H, device=dev); g["h"] = torch.empty(M, H, device=dev); g["s"] = torch.rand(M, H, device=dev)
    g["ws"] = torch.zeros(256 << 20, dtype=torch.uint8, device=dev)
    g["dpre"] = torch.randn(M, 2 * H, device=dev); g["dw"] = torch.empty(2 * H, D, device=dev); g["db"] = torch.empty(2 * H, device=dev)
    g["w2h"] = torch.randn(H, H, device=dev) * 0.05; g["w2g"] = torch.randn(H, H, device=dev) * 0.05
    g["dx"] = torch.empty(M, 2 * H, device=dev); g["a1"] = torch.randn(M, H, device=dev); g["dw2"] = torch.empty(2 * H, H, device=dev)
    return g


if which in ("fwd1", "fwd2", "dgrad2", "wgrad1", "wgrad2"):
    g = dense_setup(); ws = g["ws"]; nws = ws.numel()
    for _ in range(reps):
        if which == "fwd1":
            lib.evae_gated_dense_fwd(p(g["data"]), p(g["rows"]), M, D, D, p(g["wh"]), p(g["b"]), p(g["wg"]), p(g["b"]), H, p(g["out"]), None, p(g["s"]), p(ws), nws, st())
        elif which == "fwd2":
            lib.evae_gated_dense_fwd(p(g["a1"]), None, M, H, H, p(g["w2h"]), p(g["b"]), p(g["w2g"]), p(g["b"]), H, p(g["out"]), None, p(g["s"]), p(ws), nws, st())
        elif which == "dgrad2":
            lib.evae_dense_bwd_data(p(g["dpre"]), p(g["w2h"]), vp(g["dpre"].data_ptr() + 4 * H), p(g["w2g"]), M, H, 2 * H, H, p(g["h"]), p(g["s"]), p(g["dx"]), vp(g["dx"].data_ptr() + 4 * H), 2 * H, p(ws), nws, st())
        elif which == "wgrad1":
            lib.evae_dense_bwd_weight(p(g["dpre"]), M, 2 * H, 2 * H, p(g["data"]), p(g["rows"]), D, D, p(g["dw"]), p(g["db"]), 0, p(ws), nws, st())
        else:
            lib.evae_dense_bwd_weight(p(g["dpre"]), M, 2 * H, 2 * H, p(g["a1"]), None, H, H, p(g["dw2"]), p(g["db"]), 0, p(ws), nws, st())
elif which in ("u8fwd1", "u8wgrad1"):
    R = N
    q = (torch.randint(0, 256, (R, D), device=dev) * (torch.rand(R, D, device=dev) < 0.2)).to(torch.uint8)
    store = torch.zeros(R * D + 64, dtype=torch.uint8, device=dev); xs = store[:R * D].view(R, D); xs.copy_(q)
    rows = torch.randint(0, R, (M + 100,), device=dev)
    wh = torch.randn(H, D, device=dev) * 0.05; wg = torch.randn(H, D, device=dev) * 0.05; b = torch.zeros(H, device=dev)
    out = torch.empty(M, H, device=dev); s_ = torch.empty_like(out)
    prep = ops.u8_prepare(wh, wg)
    dy = torch.randn(M + 100, 2 * H, device=dev) * 0.01
    dw = torch.empty(2 * H, D, device=dev); db = torch.empty(2 * H, device=dev)
    for _ in range(reps):
        if which == "u8fwd1":
            ops.gated_dense_fwd_u8(xs, rows[:M], 1.0 / 255.0, prep, b, b, H, out=out, save_s=s_)
        else:
            ops.dense_bwd_weight_u8(dy, xs, rows, 1.0 / 255.0, dw=dw, db=db)
elif which.startswith("prior"):
    S, Cn, zd = {"prior_iwae": (20000, 50000, 40), "prior_c5": (5000, 100000, 256), "prior_train": (100, 25000, 40)}[which]
    z = torch.randn(1, zd, device=dev) + 0.3 * torch.randn(S, zd, device=dev)
    if which == "prior_train":
        z = torch.randn(S, zd, device=dev)
    c = torch.randn(Cn, zd, device=dev); lv = torch.full((zd,), -0.5, device=dev)
    zi = torch.randint(0, N, (S,), device=dev); ci = torch.randint(0, N, (Cn,), device=dev)
    for _ in range(reps):
        if which == "prior_train":
            m, s_, n, _ = ops.prior_lse_fwd(z, c, lv, zi, ci)
            lp, lse = ops.prior_merge(m, s_, n, Cn)
            ops.prior_lse_bwd(z, c, lv, zi, ci, lse, torch.ones(S, device=dev))
        else:
            ops.prior_lse_fwd(z, c, lv)
elif which.startswith("topk"):
    import golden_inputs as gi
    Bq, Nn, zd = (100, 25000, 40) if which == "topk_c2" else (100, 100000, 256)
    zz, cc = gi.clustered_latents(3, Bq, Nn, zd)
    q = torch.from_numpy(zz).to(dev); cache = torch.from_numpy(cc).to(dev)
    for _ in range(reps):
        ops.pairdist_topk(q, cache, 10, want_val=False)
elif which in ("conv5_fwd", "conv5_bwd"):
    x = torch.randn(25000, 32, 14, 14, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(which == "conv5_bwd")
    wh = (torch.randn(64, 32, 5, 5, device=dev) * 0.03).requires_grad_(which == "conv5_bwd")
    wg = (torch.randn(64, 32, 5, 5, device=dev) * 0.03).requires_grad_(which == "conv5_bwd")
    b = torch.zeros(64, device=dev)
    for _ in range(reps):
        if which == "conv5_fwd":
            with torch.no_grad():
                ops.gated_conv2d(x, wh, b, wg, b, 1, 2)
        else:
            y = ops.gated_conv2d(x, wh, b, wg, b, 1, 2)
            y.backward(torch.ones_like(y))
            x.grad = None; wh.grad = None; wg.grad = None
elif which == "conv96_fwd":
    x = torch.randn(1100, 96, 16, 16, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(96, 96, 3, 3, device=dev) * 0.03; b = torch.zeros(96, device=dev)
    with torch.no_grad():
        for _ in range(reps):
            ops.conv2d(x, w, b, 1, 1)
else:
    raise SystemExit("unknown probe " + which)
torch.cuda.synchronize()
