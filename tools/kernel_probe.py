#!/usr/bin/env python3
"""Launch ONE kernel family a few times, at the shape it has in a benchmarked configuration, for `rocprofv3 --pmc` /
`--kernel-trace --stats` passes (tools/profile_round.sh).  usage: kernel_probe.py <which> [reps]
  fwd1        GatedDense forward, encoder layer 1 at c2: 25 000 gathered rows x 784 -> 2 x 300
  u8fwd1 / u8wgrad1   the same layer on the uint8 store (three-term bf16 MFMA): forward / weight gradient
  fwd2        GatedDense forward, encoder layer 2: 25 000 x 300 -> 2 x 300
  dgrad2      data gradient of encoder layer 2 (dual pair, gate-backward epilogue)
  wgrad1      weight gradient of encoder layer 1 ([600 x 784] + db, gathered rows)
  wgrad2      weight gradient of encoder layer 2
  the c2 step's pre-split bf16 image path (csrc/evae_gemm_p6.h), the launches of evae/fused_vae.py's p6 mode:
  u8fwd1_img  encoder layer 1 on the uint8 store, output + the image of its transpose (u8_gemm_kernel<true>)
  fwd2_p6     encoder layer 2 forward over h1^T's image (gemm_p6_kernel<1, 128, true>)
  hdgrad2_img the mean head's data gradient, gate-backward epilogue -> [dh2 | dg2]^T's image (gemm_x6_kernel<2, 0, 64, 3>)
  dgrad2_p6   encoder layer 2's data gradient over that image, (dh1, dg1) as the byte layer's tile images (gemm_p6_kernel<9, 64, true>)
  wgrad2_p6   encoder layer 2's weight gradient from the two images (gemm_p6_kernel<3, 64, false> + finish)
  hwgrad      the heads' weight gradient [40 x 300] over 25 100 rows (narrow_wgrad_mfma_kernel + narrow_finish_kernel)
  prior_iwae  prior forward, 4 x 5000 importance samples x 50 000 exemplars, z = 40 (prior_fwd_mfma_kernel)
  prior_c5    prior forward, 5000 samples x 100 000 exemplars, z = 256 (GEMM + log-sum-exp epilogue)
  prior_train prior forward + backward at the training shape B = 100, C = 25 000, z = 40 (three launches, the modular form)
  prior_train1  the same as ONE launch (evae_prior_train_step: what a captured step runs)
  topk_c2 / topk_c5   evae_pairdist_topk, k = 10 (100 x 25 000 x 40 / 100 x 100 000 x 256)
  conv5_fwd / conv5_bwd   gated conv 32 -> 64, 5 x 5, 14 x 14, 25 000 images (c3) on the channels-last kernels: forward / data + weight gradient
  cw5_fwd / cw5_bwd / cw5_wgrad   the same layer over the 19 968 encoded rows of a c3 step on the window kernels (csrc/evae_conv_win.h):
              forward (conv_win_kernel<0, 2, 2, 320>), data gradient + gate derivative (conv_win_kernel<1, 4, 1, 576>), weight gradient
              (conv_wgrad_win_kernel<13 | 12, 192, 8, 1, false>)
  res96_fwd / res96_bwd / res96_wgrad   a residual block 96 -> 96, 3 x 3, 32 x 32, 100 images (c5's decoder) on the window kernels
  cw1_fwd / cw1_wgrad   first layer 1 -> 32, 7 x 7, 28 x 28 (conv_first_kernel / conv_first_wgrad_kernel), 20 224 images
  cw2_bwd     data gradient of the stride-2 layer 32 -> 32, 3 x 3 into the first layer's 28 x 28 grid (four parity-class launches)
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
N, D, H, Z = 50000, 784, 300, 40
# exemplar rows of the large launches: what a captured c2 / c3 step encodes (the distinct rows of its 25 000 draws, evae/graph.py)
M = int(os.environ.get("EVAE_PROBE_ROWS", "19968"))
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
elif which in ("u8fwd1_img", "fwd2_p6", "hdgrad2_img", "dgrad2_p6", "wgrad2_p6", "hwgrad"):
    Mp = M + 100
    nks = lib.evae_p6_nks_rows(Mp)
    img = lambda rows: torch.zeros(lib.evae_p6_image_bytes(rows, nks), dtype=torch.uint8, device=dev)
    t_h1, t_dq2 = img(H + 1), img(2 * H)
    a1 = torch.randn(Mp, H, device=dev) * (torch.rand(Mp, H, device=dev) < 0.5); s1 = torch.rand(Mp, H, device=dev)
    a2 = torch.randn(Mp, H, device=dev); s2 = torch.rand(Mp, H, device=dev)
    dq2 = torch.randn(Mp, 2 * H, device=dev) * 0.01
    w2h = torch.randn(H, H, device=dev) * 0.05; w2g = torch.randn(H, H, device=dev) * 0.05; b = torch.zeros(H, device=dev)
    _lib.check(lib.evae_p6_pack_cols(p(a1), None, Mp, H, H, -1, nks, p(t_h1), t_h1.numel(), st()), "pack h1")
    _lib.check(lib.evae_p6_fill_row(p(t_h1), nks, H, 1.0, 0, Mp, st()), "ones row")
    _lib.check(lib.evae_p6_pack_cols(p(dq2), None, Mp, 2 * H, 2 * H, -1, nks, p(t_dq2), t_dq2.numel(), st()), "pack dq2")
    w2_img = torch.zeros(lib.evae_p6_image_bytes((H + 63) // 64 * 128, lib.evae_p6_nks(H)), dtype=torch.uint8, device=dev)
    w2t_img = torch.zeros(lib.evae_p6_image_bytes(H, lib.evae_p6_nks(2 * H)), dtype=torch.uint8, device=dev)
    _lib.check(lib.evae_p6_pack_rows(p(w2h), p(w2g), H, H, H, 1, p(w2_img), w2_img.numel(), st()), "pack w2")
    _lib.check(lib.evae_p6_pack_cols(p(w2h), p(w2g), H, H, H, -1, lib.evae_p6_nks(2 * H), p(w2t_img), w2t_img.numel(), st()), "pack w2t")
    out = torch.empty(M, H, device=dev); so = torch.empty(M, H, device=dev)
    if which == "u8fwd1_img":
        R = N
        q = (torch.randint(0, 256, (R, D), device=dev) * (torch.rand(R, D, device=dev) < 0.2)).to(torch.uint8)
        store = torch.zeros(R * D + 64, dtype=torch.uint8, device=dev); xs = store[:R * D].view(R, D); xs.copy_(q)
        rows = torch.randint(0, R, (M,), device=dev)
        wh = torch.randn(H, D, device=dev) * 0.05; wg = torch.randn(H, D, device=dev) * 0.05
        prep = ops.u8_prepare(wh, wg)
    elif which == "dgrad2_p6":
        nbw = lib.evae_dense_bwd_weight_u8_workspace_bytes(Mp, 2 * H, D)
        wsw = torch.zeros(nbw, dtype=torch.uint8, device=dev)
        off, nslab = C.c_size_t(0), C.c_int(0)
        _lib.check(lib.evae_dense_bwd_weight_u8_images(Mp, 2 * H, D, C.byref(off), C.byref(nslab)), "images")
    elif which == "hdgrad2_img":
        dmean = torch.randn(M, Z, device=dev) * 0.01; wm = torch.randn(Z, H, device=dev) * 0.05
        nbh = lib.evae_dense_bwd_data_workspace_bytes(M, Z, H, 1); wsh = torch.zeros(nbh, dtype=torch.uint8, device=dev)
    elif which == "wgrad2_p6":
        nbw = lib.evae_dense_bwd_weight_p6_workspace_bytes(nks, 2 * H, H); wsw = torch.zeros(nbw, dtype=torch.uint8, device=dev)
        dw2 = torch.empty(2 * H, H, device=dev); db2 = torch.empty(2 * H, device=dev)
    elif which == "hwgrad":
        dmean = torch.randn(Mp, Z, device=dev) * 0.01
        nbw = lib.evae_dense_bwd_weight_workspace_bytes(Mp, Z, H); wsw = torch.zeros(nbw, dtype=torch.uint8, device=dev)
        dwm = torch.empty(Z, H, device=dev); dbm = torch.empty(Z, device=dev)
    for _ in range(reps):
        if which == "u8fwd1_img":
            _lib.check(lib.evae_gated_dense_fwd_u8_timg(p(xs), p(rows), M, D, D, 1.0 / 255.0, p(prep), p(b), p(b), H, p(out), p(so), p(t_h1),
                                                        nks, 0, 0, st()), which)
        elif which == "fwd2_p6":
            _lib.check(lib.evae_gated_dense_fwd_p6t(p(t_h1), nks, M, H, p(w2_img), p(b), p(b), H, p(out), p(so), st()), which)
        elif which == "hdgrad2_img":
            _lib.check(lib.evae_dense_bwd_data_timg(p(dmean), p(wm), None, None, M, Z, Z, H, p(a2), p(s2), None, None, 2 * H, None,
                                                    p(t_dq2), nks, 0, 0, p(wsh), nbh, st()), which)
        elif which == "dgrad2_p6":
            _lib.check(lib.evae_dense_bwd_data_p6t(p(t_dq2), nks, M, 2 * H, p(w2t_img), H, p(a1), p(s1), None, None, 0,
                                                   vp(wsw.data_ptr() + off.value), nslab.value, 0, st()), which)
        elif which == "wgrad2_p6":
            _lib.check(lib.evae_dense_bwd_weight_p6(p(t_dq2), p(t_h1), nks, 2 * H, H, p(dw2), p(db2), p(wsw), nbw, st()), which)
        else:
            _lib.check(lib.evae_dense_bwd_weight(p(dmean), Mp, Z, Z, p(a2), None, H, H, p(dwm), p(dbm), 0, p(wsw), nbw, st()), which)
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
    S, Cn, zd = {"prior_iwae": (20000, 50000, 40), "prior_c5": (5000, 100000, 256), "prior_train": (100, 25000, 40),
                 "prior_train1": (100, 25000, 40)}[which]
    z = torch.randn(1, zd, device=dev) + 0.3 * torch.randn(S, zd, device=dev)
    if which.startswith("prior_train"):
        z = torch.randn(S, zd, device=dev)
    c = torch.randn(Cn, zd, device=dev); lv = torch.full((zd,), -0.5, device=dev)
    zi = torch.randint(0, N, (S,), device=dev); ci = torch.randint(0, N, (Cn,), device=dev)
    for _ in range(reps):
        if which == "prior_train1":      # the captured step's form: forward, merge and backward as ONE launch (csrc/evae_prior_train.h)
            ops.prior_train_step(z, c, lv, zi, ci, float(Cn), 0.7)
        elif which == "prior_train":
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
elif which in ("cw5_fwd", "cw5_bwd", "cw5_wgrad"):
    pr = ops.conv_window_probe(int(os.environ.get("EVAE_PROBE_ROWS", "19968")), 32, 14, 64, 5, 1, out_planar=True)
    fn = pr[{"cw5_fwd": "fwd", "cw5_bwd": "dgrad", "cw5_wgrad": "wgrad"}[which]]
    for _ in range(reps):
        fn()
elif which in ("res96_fwd", "res96_bwd", "res96_wgrad"):
    pr = ops.res_window_probe(100, 96, 32)
    fn = pr[{"res96_fwd": "fwd", "res96_bwd": "dgrad", "res96_wgrad": "wgrad"}[which]]
    for _ in range(reps * 4):
        fn()
elif which == "cw2_bwd":
    pr = ops.conv_window_probe(int(os.environ.get("EVAE_PROBE_ROWS", "19968")), 32, 28, 32, 3, 2)
    for _ in range(reps):
        pr["dgrad"]()
elif which in ("cw1_fwd", "cw1_wgrad"):
    n = int(os.environ.get("EVAE_PROBE_ROWS", "19968"))
    d = _lib.ConvDesc(n, 1, 28, 28, 32, 7, 7, 1, 3)
    x = (torch.rand(n, 28, 28, device=dev) < 0.3).float()
    wh = torch.randn(32, 1, 7, 7, device=dev) * 0.1; wg = torch.randn(32, 1, 7, 7, device=dev) * 0.1; b = torch.zeros(32, device=dev)
    oimg = torch.empty(int(lib.evae_cw_image_bytes(n * 784, 32)), dtype=torch.uint8, device=dev); s_ = torch.empty(n, 28, 28, 32, device=dev)
    dy = torch.randn(n, 28, 28, 64, device=dev) * 0.1
    dw = torch.empty(64, 49, device=dev); db = torch.empty(64, device=dev)
    ws = torch.empty(int(lib.evae_cw_first_workspace_bytes()), dtype=torch.uint8, device=dev)
    for _ in range(reps):
        if which == "cw1_fwd":
            _lib.check(lib.evae_cw_first_fwd(p(x), C.byref(d), p(wh), p(b), p(wg), p(b), p(oimg), 1, p(s_), None, st()), "evae_cw_first_fwd")
        else:
            _lib.check(lib.evae_cw_first_bwd_weight(p(dy), p(x), C.byref(d), p(dw), p(db), p(ws), ws.numel(), st()), "evae_cw_first_bwd_weight")
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
