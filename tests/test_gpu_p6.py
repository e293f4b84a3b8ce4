"""GPU parity tests of the pre-split bf16 operand path (csrc/evae_gemm_p6.h, csrc/evae_p6_image.h; include/evae_hip.h "p6"):
the hidden GatedDense layer of the exemplar rows' chain (reference utils/nn.py:44-69 at models/BaseModel.py:243-248's row count)
-- forward, data gradient, weight gradient over operand images, and the producers that write those images from their
epilogues -- against the fp64 oracle at the fp32 kernels' bar, and image against image bit for bit.
Run on a real MI355X:  python -m pytest tests -m gpu"""
import ctypes as C

import numpy as np
import pytest
import torch

import evae_oracle as orc

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def env():
    from evae import ops as o, _lib
    lib = _lib.load()

    class E:
        pass
    e = E()
    e.ops, e.lib, e.chk, e.st = o, lib, _lib.check, o._stream

    def image(rows, nks):
        return torch.zeros(lib.evae_p6_image_bytes(rows, nks), dtype=torch.uint8, device="cuda")

    def pack_cols(x, x2=None, ones_row=-1, nks=None, rows=None):
        """image of X^T for X [Kd x R] (+ x2 stacked along the contraction)"""
        Kd, R = x.shape
        nks = lib.evae_p6_nks((2 if x2 is not None else 1) * Kd) if nks is None else nks
        img = image(max(R, ones_row + 1) if rows is None else rows, nks)
        e.chk(lib.evae_p6_pack_cols(o._p(x), o._p(x2) if x2 is not None else None, Kd, R, x.stride(0), ones_row, nks, o._p(img),
                                    img.numel(), e.st()), "pack_cols")
        return img, nks

    def pack_gated(wh, wg):
        N, K = wh.shape
        img = image((N + 63) // 64 * 128, lib.evae_p6_nks(K))
        e.chk(lib.evae_p6_pack_rows(o._p(wh), o._p(wg), N, K, K, 1, o._p(img), img.numel(), e.st()), "pack_rows")
        return img
    e.image, e.pack_cols, e.pack_gated = image, pack_cols, pack_gated
    return e


def layer(rs, M, K, N, spread=0):
    x = rs.standard_normal((M, K)).astype(np.float32)
    wh = (rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32); bh = (rs.standard_normal(N) * 0.1).astype(np.float32)
    wg = (rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32); bg = (rs.standard_normal(N) * 0.1).astype(np.float32)
    if spread:
        sc = (10.0 ** rs.uniform(-spread, spread, K)).astype(np.float32)
        x = x * sc; wh = wh / sc; wg = wg / sc
    return x, wh, bh, wg, bg


@pytest.mark.parametrize("M,K,N", [(37, 52, 24), (130, 300, 300), (1000, 300, 300), (257, 40, 300), (5000, 304, 70), (2500, 300, 300),
                                   (25000, 300, 300)])
@pytest.mark.parametrize("spread", [0, 6])
def test_p6_gated_forward_over_the_transposed_image_holds_the_fp32_bar(env, M, K, N, spread):
    """evae_gated_dense_fwd_p6t: x as x^T's image read through the LDS transpose read, weights as the gated image: the fp64
    oracle at the fp32-MFMA kernel's tolerance and no worse than that kernel by more than a small factor, also with operands
    that span 10^+-spread per column."""
    rs = np.random.RandomState(M + K + spread)
    x, wh, bh, wg, bg = layer(rs, M, K, N, spread)
    y, _ = orc.gated_dense(*(a.astype(np.float64) for a in (x, wh, bh, wg, bg)))
    t = [dev(a) for a in (x, wh, bh, wg, bg)]
    ximg, _ = env.pack_cols(t[0], nks=env.lib.evae_p6_nks_rows(M))
    wimg = env.pack_gated(t[1], t[3])
    out = torch.empty((M, N), device="cuda"); s = torch.empty_like(out)
    env.chk(env.lib.evae_gated_dense_fwd_p6t(env.ops._p(ximg), env.lib.evae_p6_nks_rows(M), M, K, env.ops._p(wimg), env.ops._p(t[2]),
                                             env.ops._p(t[4]), N, env.ops._p(out), env.ops._p(s), env.st()), "fwd_p6t")
    env.ops.gemm_x6_configure(0, -1)
    out32 = env.ops.gated_dense(*t).cpu().numpy()
    env.ops.gemm_x6_configure(1, -1)
    e6, e32 = rel(out.cpu().numpy(), y), rel(out32, y)
    assert e6 < 2e-6, (e6, e32)
    assert e6 < 2.0 * e32 + 2e-7, (e6, e32)
    sg = 1.0 / (1.0 + np.exp(-(x.astype(np.float64) @ wg.astype(np.float64).T + bg)))
    assert np.abs(s.cpu().numpy() - sg).max() < 5e-6


@pytest.mark.parametrize("M1,M2,K,N", [(2560, 104, 300, 300), (600, 100, 784, 300), (25000, 100, 300, 300), (136, 37, 52, 70)])
def test_producers_leave_their_output_as_the_transposed_image(env, M1, M2, K, N):
    """evae_gated_dense_fwd_timg (fp32 / split-bf16 kernels: epilogue; thin launches: split-K finish): the image the launches
    write -- the first M1 rows by one, M2 more behind them by a second -- is bit for bit evae_p6_pack_cols of the fp32 output."""
    rs = np.random.RandomState(M1 + K)
    M = M1 + M2
    x, wh, bh, wg, bg = layer(rs, M, K, N)
    t = [dev(a) for a in (x, wh, bh, wg, bg)]
    nks = env.lib.evae_p6_nks_rows(M)
    img = env.image(N, nks)
    out = torch.empty((M, N), device="cuda"); s = torch.empty_like(out)
    lib, o = env.lib, env.ops
    for m0, mm in ((0, M1), (M1, M2)):
        nb = lib.evae_dense_fwd_workspace_bytes(mm, K, N, 1)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        env.chk(lib.evae_gated_dense_fwd_timg(C.c_void_p(t[0].data_ptr() + 4 * m0 * K), None, mm, K, K, o._p(t[1]), o._p(t[2]),
                                              o._p(t[3]), o._p(t[4]), N, C.c_void_p(out.data_ptr() + 4 * m0 * N), None,
                                              C.c_void_p(s.data_ptr() + 4 * m0 * N), o._p(img), nks, 0, m0, o._p(ws), nb, env.st()),
                "fwd_timg")
    ref = o.gated_dense(*t)
    assert torch.equal(out, ref) or rel(out.cpu().numpy(), ref.cpu().numpy()) < 1e-6
    want, _ = env.pack_cols(out, nks=nks)
    assert torch.equal(img, want)


@pytest.mark.parametrize("M1,M2,N,R", [(2560, 104, 300, 3000), (25000, 100, 300, 50000), (136, 38, 70, 200)])
def test_byte_store_layer_leaves_its_output_as_the_transposed_image(env, M1, M2, N, R):
    """evae_gated_dense_fwd_u8_timg: same output as evae_gated_dense_fwd_u8, image = evae_p6_pack_cols of it, bit for bit."""
    rs = np.random.RandomState(M1 + N)
    K, M = 784 if N == 300 else 48, M1 + M2
    q = (rs.randint(0, 256, (R, K)) * (rs.random_sample((R, K)) < 0.3)).astype(np.uint8)
    store = torch.zeros(R * K + 64, dtype=torch.uint8, device="cuda"); xs = store[:R * K].view(R, K); xs.copy_(torch.from_numpy(q))
    rows = dev(rs.randint(0, R, size=M).astype(np.int64))
    _, wh, bh, wg, bg = layer(rs, 1, K, N)
    t = [dev(a) for a in (wh, bh, wg, bg)]
    lib, o = env.lib, env.ops
    prep = torch.empty(lib.evae_dense_u8_prepared_bytes(N, K), dtype=torch.uint8, device="cuda")
    env.chk(lib.evae_dense_u8_prepare(o._p(t[0]), o._p(t[2]), N, K, o._p(prep), prep.numel(), env.st()), "prepare")
    nks = lib.evae_p6_nks_rows(M)
    img = env.image(N, nks)
    out = torch.empty((M, N), device="cuda"); s = torch.empty_like(out)
    out0 = torch.empty_like(out); s0 = torch.empty_like(out)
    env.chk(lib.evae_gated_dense_fwd_u8(o._p(xs), o._p(rows), M, K, K, 1.0 / 255.0, o._p(prep), o._p(t[1]), o._p(t[3]), N, o._p(out0),
                                        o._p(s0), env.st()), "fwd_u8")
    for m0, mm in ((0, M1), (M1, M2)):
        env.chk(lib.evae_gated_dense_fwd_u8_timg(o._p(xs), C.c_void_p(rows.data_ptr() + 8 * m0), mm, K, K, 1.0 / 255.0, o._p(prep),
                                                 o._p(t[1]), o._p(t[3]), N, C.c_void_p(out.data_ptr() + 4 * m0 * N),
                                                 C.c_void_p(s.data_ptr() + 4 * m0 * N), o._p(img), nks, 0, m0, env.st()), "fwd_u8_timg")
    assert torch.equal(out, out0) and torch.equal(s, s0)
    want, _ = env.pack_cols(out, nks=nks)
    assert torch.equal(img, want)


@pytest.mark.parametrize("M,H,Z", [(2560, 300, 40), (25000, 300, 40), (137, 64, 8), (100, 300, 40)])
def test_gate_fused_data_gradient_writes_the_image_of_dh_dg(env, M, H, Z):
    """evae_dense_bwd_data_timg (the heads' data gradient with encoder layer 2's gate derivative in its epilogue): fp32 (dh, dg)
    as evae_dense_bwd_data_wt, and [dh | dg]^T's image = evae_p6_pack_cols of them -- with and without the fp32 copy."""
    rs = np.random.RandomState(M + H)
    dy = dev((rs.standard_normal((M, Z)) * 0.1).astype(np.float32))
    w = dev((rs.standard_normal((Z, H)) * 0.1).astype(np.float32))
    a = dev(rs.standard_normal((M, H)).astype(np.float32)); s = dev(rs.random_sample((M, H)).astype(np.float32))
    lib, o = env.lib, env.ops
    ref = torch.empty((M, 2 * H), device="cuda")
    o._bwd_data(dy.data_ptr(), w, None, None, M, Z, Z, "cuda", out_prev=a, s_prev=s, out=ref, dg_ptr=ref.data_ptr() + 4 * H, ldo=2 * H)
    nks = lib.evae_p6_nks_rows(M)
    nb = lib.evae_dense_bwd_data_workspace_bytes(M, Z, H, 1)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    want, _ = env.pack_cols(ref, nks=nks)
    for with_fp32 in (True, False):
        img = env.image(2 * H, nks)
        got = torch.zeros((M, 2 * H), device="cuda")
        env.chk(lib.evae_dense_bwd_data_timg(o._p(dy), o._p(w), None, None, M, Z, Z, H, o._p(a), o._p(s),
                                             o._p(got) if with_fp32 else None, C.c_void_p(got.data_ptr() + 4 * H) if with_fp32 else None,
                                             2 * H, None, o._p(img), nks, 0, 0, o._p(ws), nb, env.st()), "bwd_data_timg")
        if with_fp32:
            assert torch.equal(got, ref)
        assert torch.equal(img, want), with_fp32


@pytest.mark.parametrize("M,H", [(2560, 300), (25000, 300), (1000, 64), (130, 40)])
def test_p6_data_gradient_over_the_transposed_image(env, M, H):
    """evae_dense_bwd_data_p6t: [dh2 | dg2] as its transpose's image, W2^T (banks stacked along the contraction) as an image, the
    gate derivative of the layer below in the epilogue: fp32 (dh1, dg1) against fp64; as byte-layer tile images: the same
    weight gradient as evae_dense_bwd_data_img's."""
    rs = np.random.RandomState(M + H)
    dq2 = (rs.standard_normal((M, 2 * H)) * 0.1).astype(np.float32)
    wh = (rs.standard_normal((H, H)) * 0.1).astype(np.float32); wg = (rs.standard_normal((H, H)) * 0.1).astype(np.float32)
    a1 = rs.standard_normal((M, H)).astype(np.float32); s1 = rs.random_sample((M, H)).astype(np.float32)
    v = dq2[:, :H].astype(np.float64) @ wh.astype(np.float64) + dq2[:, H:].astype(np.float64) @ wg.astype(np.float64)
    dh_ref = v * s1; dg_ref = v * a1 * (1.0 - s1.astype(np.float64))
    t = [dev(x) for x in (dq2, wh, wg, a1, s1)]
    lib, o = env.lib, env.ops
    nks = lib.evae_p6_nks_rows(M)
    dimg, _ = env.pack_cols(t[0], nks=nks)
    wimg, _ = env.pack_cols(t[1], t[2])               # rows = outputs k, contraction = [n of bank h | n of bank g]
    out = torch.empty((M, 2 * H), device="cuda")
    env.chk(lib.evae_dense_bwd_data_p6t(o._p(dimg), nks, M, 2 * H, o._p(wimg), H, o._p(t[3]), o._p(t[4]), o._p(out),
                                        C.c_void_p(out.data_ptr() + 4 * H), 2 * H, None, 0, 0, env.st()), "bwd_data_p6t")
    got = out.cpu().numpy()
    assert rel(got[:, :H], dh_ref) < 2e-6 and rel(got[:, H:], dg_ref) < 2e-6
    # no gate: the plain product
    plain = torch.empty((M, H), device="cuda")
    env.chk(lib.evae_dense_bwd_data_p6t(o._p(dimg), nks, M, 2 * H, o._p(wimg), H, None, None, o._p(plain), None, H, None, 0, 0, env.st()),
            "bwd_data_p6t(plain)")
    assert rel(plain.cpu().numpy(), v) < 2e-6
    if H == 300 and M % 8 == 0:
        # byte-layer images: the weight gradient of the first layer from them = from evae_dense_bwd_data_img's
        D, R = 784, 3000
        q = (rs.randint(0, 256, (R, D)) * (rs.random_sample((R, D)) < 0.3)).astype(np.uint8)
        store = torch.zeros(R * D + 64, dtype=torch.uint8, device="cuda"); xs = store[:R * D].view(R, D); xs.copy_(torch.from_numpy(q))
        rows = dev(rs.randint(0, R, size=M).astype(np.int64))
        nb = lib.evae_dense_bwd_weight_u8_workspace_bytes(M, 2 * H, D)
        off, nslab = C.c_size_t(0), C.c_int(0)
        env.chk(lib.evae_dense_bwd_weight_u8_images(M, 2 * H, D, C.byref(off), C.byref(nslab)), "images")
        res = []
        for route in (0, 1):
            ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
            img = ws.data_ptr() + off.value
            if route == 0:
                wsd = torch.zeros(lib.evae_dense_bwd_data_workspace_bytes(M, H, H, 2), dtype=torch.uint8, device="cuda")
                env.chk(lib.evae_dense_bwd_data_img(o._p(t[0]), o._p(t[1]), C.c_void_p(t[0].data_ptr() + 4 * H), o._p(t[2]), M, H, 2 * H, H,
                                                    o._p(t[3]), o._p(t[4]), C.c_void_p(img), nslab.value, 0, None, o._p(wsd), wsd.numel(),
                                                    env.st()), "bwd_data_img")
            else:
                env.chk(lib.evae_dense_bwd_data_p6t(o._p(dimg), nks, M, 2 * H, o._p(wimg), H, o._p(t[3]), o._p(t[4]), None, None, 0,
                                                    C.c_void_p(img), nslab.value, 0, env.st()), "bwd_data_p6t(images)")
            dw = torch.empty((2 * H, D), device="cuda"); db = torch.empty(2 * H, device="cuda")
            env.chk(lib.evae_dense_bwd_weight_u8(None, M, 2 * H, 2 * H, o._p(xs), o._p(rows), D, D, 1.0 / 255.0, o._p(dw), o._p(db),
                                                 o._p(ws), ws.numel(), env.st()), "bwd_weight_u8")
            res.append((dw.cpu().numpy(), db.cpu().numpy()))
        assert rel(res[1][0], res[0][0]) < 2e-6 and rel(res[1][1], res[0][1]) < 2e-6


@pytest.mark.parametrize("M,N,K", [(25100, 600, 300), (2568, 600, 300), (777, 130, 90), (100, 80, 300), (4099, 24, 68)])
def test_p6_weight_gradient_over_transposed_images(env, M, N, K):
    """evae_dense_bwd_weight_p6: dW = dy^T x and db = column sums of dy from the images of dy^T and x^T (+ the all-ones row
    behind x's columns), against fp64 and against the fp32-MFMA weight gradient."""
    rs = np.random.RandomState(M + N)
    dy = (rs.standard_normal((M, N)) * 0.01).astype(np.float32); x = rs.standard_normal((M, K)).astype(np.float32)
    dw_ref = dy.astype(np.float64).T @ x.astype(np.float64); db_ref = dy.astype(np.float64).sum(0)
    tdy, tx = dev(dy), dev(x)
    lib, o = env.lib, env.ops
    nks = lib.evae_p6_nks_rows(M)
    dimg, _ = env.pack_cols(tdy, nks=nks)
    ximg = env.image(K + 1, nks)
    env.chk(lib.evae_p6_pack_cols(o._p(tx), None, M, K, K, -1, nks, o._p(ximg), ximg.numel(), env.st()), "pack x")
    env.chk(lib.evae_p6_fill_row(o._p(ximg), nks, K, 1.0, 0, M, env.st()), "ones row")
    nb = lib.evae_dense_bwd_weight_p6_workspace_bytes(nks, N, K)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    dw = torch.empty((N, K), device="cuda"); db = torch.empty(N, device="cuda")
    env.chk(lib.evae_dense_bwd_weight_p6(o._p(dimg), o._p(ximg), nks, N, K, o._p(dw), o._p(db), o._p(ws), nb, env.st()), "bwd_weight_p6")
    o.gemm_x6_configure(0, -1)
    dw32, db32 = o._bwd_weight(tdy, tx, None, K)
    o.gemm_x6_configure(1, -1)
    e6, e32 = rel(dw.cpu().numpy(), dw_ref), rel(dw32.cpu().numpy(), dw_ref)
    assert e6 < 3e-6 and e6 < 2.0 * e32 + 3e-7, (e6, e32)
    assert rel(db.cpu().numpy(), db_ref) < 3e-6
    # without the bias gradient
    dw2 = torch.empty((N, K), device="cuda")
    env.chk(lib.evae_dense_bwd_weight_p6(o._p(dimg), o._p(ximg), nks, N, K, o._p(dw2), None, o._p(ws), nb, env.st()), "bwd_weight_p6")
    assert rel(dw2.cpu().numpy(), dw_ref) < 3e-6
