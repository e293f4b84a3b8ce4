"""GPU parity of the implicit-GEMM convolution kernels (csrc/evae_conv.hip) against torch's CPU float64
conv2d / autograd on the layer shapes of models/convHVAE_2level.py and models/fully_conv.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


CASES = [  # N, C, H, W, Co, k, stride, pad
    (5, 1, 28, 28, 32, 7, 1, 3), (5, 32, 28, 28, 32, 3, 2, 1), (4, 32, 14, 14, 64, 5, 1, 2),
    (4, 64, 14, 14, 64, 3, 2, 1), (6, 64, 7, 7, 6, 3, 1, 1), (3, 64, 28, 28, 1, 1, 1, 0),
    (2, 3, 16, 16, 48, 3, 2, 1), (2, 48, 8, 8, 96, 3, 2, 1), (3, 96, 4, 4, 1, 3, 1, 1), (2, 1, 9, 11, 5, 3, 2, 1),
    # channel counts that are no multiple of 32 on the channels-last GEMM path (a tap is padded to 32-channel slabs in the
    # K index only): fully_conv's 48-channel residual blocks, and odd small ones
    (3, 48, 16, 16, 48, 3, 1, 1), (2, 48, 32, 32, 48, 3, 1, 1), (2, 16, 8, 8, 32, 3, 1, 1), (2, 20, 9, 9, 8, 3, 1, 1),
    (2, 96, 16, 16, 48, 3, 1, 1), (2, 48, 8, 8, 3, 3, 1, 1),
]


@pytest.mark.parametrize("gated", [True, False])
@pytest.mark.parametrize("N,C,H,W,Co,k,s,p", CASES)
def test_conv2d_fwd_bwd_matches_torch(N, C, H, W, Co, k, s, p, gated, gemm_pipe):
    from evae import ops
    rs = np.random.RandomState(N + C + Co + k)
    x = torch.from_numpy(rs.standard_normal((N, C, H, W)).astype(np.float32))
    wh = torch.from_numpy((rs.standard_normal((Co, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32))
    wg = torch.from_numpy((rs.standard_normal((Co, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32))
    bh = torch.from_numpy((rs.standard_normal(Co) * 0.1).astype(np.float32))
    bg = torch.from_numpy((rs.standard_normal(Co) * 0.1).astype(np.float32))
    # float64 CPU reference
    xr, whr, wgr, bhr, bgr = [t.double().requires_grad_(True) for t in (x, wh, wg, bh, bg)]
    yr = F.conv2d(xr, whr, bhr, s, p)
    if gated:
        yr = yr * torch.sigmoid(F.conv2d(xr, wgr, bgr, s, p))
    gout = torch.from_numpy(rs.standard_normal(tuple(yr.shape)).astype(np.float32))
    yr.backward(gout.double())
    # HIP
    xd, whd, wgd, bhd, bgd = [t.cuda().requires_grad_(True) for t in (x, wh, wg, bh, bg)]
    if gated:
        y = ops.gated_conv2d(xd, whd, bhd, wgd, bgd, s, p)
    else:
        y = ops.conv2d(xd, whd, bhd, s, p)
    y.backward(gout.cuda())
    assert y.shape == yr.shape
    assert rel(y, yr) < 1e-5
    assert rel(xd.grad, xr.grad) < 1e-5
    assert rel(whd.grad, whr.grad) < 1e-5 and rel(bhd.grad, bhr.grad) < 1e-5
    if gated:
        assert rel(wgd.grad, wgr.grad) < 1e-5 and rel(bgd.grad, bgr.grad) < 1e-5


@pytest.mark.parametrize("N,C,H,W,k,per", [(3, 48, 32, 32, 3, None), (2, 96, 16, 16, 3, None), (5, 48, 16, 16, 3, "2"),
                                            (2, 16, 9, 7, 3, None), (2, 20, 8, 8, 5, None), (2, 64, 8, 8, 3, None)])
def test_residual_block_matches_torch(N, C, H, W, k, per, monkeypatch, gemm_pipe):
    """x + conv(ELU(x)) (models/fully_conv.py:13-23) through ops.res_block: output and all gradients against float64
    autograd, also with the tensors processed in passes of 2 images."""
    from evae import ops
    if per:
        monkeypatch.setenv("EVAE_CL_IMAGES_PER_PASS", per)
    rs = np.random.RandomState(C + H)
    x = torch.from_numpy((rs.standard_normal((N, C, H, W)) * 1.5).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((C, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32))
    b = torch.from_numpy((rs.standard_normal(C) * 0.1).astype(np.float32))
    xr, wr, br = [t.double().requires_grad_(True) for t in (x, w, b)]
    yr = xr + F.conv2d(F.elu(xr), wr, br, 1, k // 2)
    gout = torch.from_numpy(rs.standard_normal(tuple(yr.shape)).astype(np.float32))
    yr.backward(gout.double())
    xd, wd, bd = [t.cuda().requires_grad_(True) for t in (x, w, b)]
    assert ops.res_block_supported(xd, wd, 1, k // 2)
    y = ops.res_block(xd, wd, bd)
    y.backward(gout.cuda())
    assert rel(y, yr) < 1e-5 and rel(xd.grad, xr.grad) < 1e-5
    assert rel(wd.grad, wr.grad) < 1e-5 and rel(bd.grad, br.grad) < 1e-5


def test_fully_conv_block_module_uses_fused_path_and_matches_unfused():
    """models.fully_conv.block (weight-normed): the fused residual path against the module's own unfused composition,
    gradients with respect to the weight-norm parameters g and v included."""
    from models.fully_conv import block
    torch.manual_seed(3)
    m = block(48, 48).cuda()
    with torch.no_grad():
        m.conv1.weight_g.mul_(torch.rand_like(m.conv1.weight_g) + 0.5)
    x = torch.randn(4, 48, 16, 16, device="cuda")
    res = []
    for fused in (True, False):
        xi = x.clone().requires_grad_(True)
        m.zero_grad()
        y = m(xi) if fused else xi + m.f(xi)
        y.square().sum().backward()
        res.append([y.detach(), xi.grad] + [p.grad.clone() for p in m.conv1.parameters()])
    for a, b_ in zip(*res):
        assert rel(a, b_) < 2e-5


def test_conv2d_activations_and_modules():
    from utils.nn import Conv2d, GatedConv2d
    torch.manual_seed(0)
    x = torch.randn(4, 8, 10, 10)
    for act in (torch.nn.Sigmoid(), torch.nn.Hardtanh(-4.5, 0.0), None):
        m = Conv2d(8, 5, 3, 1, 1, activation=act)
        ref = m.conv.double()(x.double())
        ref = ref if act is None else act(ref)
        m = m.float().cuda()
        out = m(x.cuda())
        assert rel(out, ref) < 1e-5
    g = GatedConv2d(8, 6, 3, 2, 1)
    ref = g.h.double()(x.double()) * torch.sigmoid(g.g.double()(x.double()))
    g = g.float().cuda()
    assert rel(g(x.cuda()), ref) < 1e-5


@pytest.mark.parametrize("case", [(5, 32, 28, 28, 32, 3, 2, 1), (7, 64, 7, 7, 6, 3, 1, 1), (6, 32, 14, 14, 64, 5, 1, 2)])
def test_conv2d_channels_last_multi_pass(case, monkeypatch, gemm_pipe):
    """Tensors beyond 2 GiB are processed in passes over the images (31-bit buffer offsets); force 2 images per
    pass on small tensors and compare with the single-pass result."""
    from evae import ops
    N, C, H, W, Co, k, s, p = case
    rs = np.random.RandomState(11)
    x = torch.from_numpy(rs.standard_normal((N, C, H, W)).astype(np.float32)).cuda()
    ws_ = [torch.from_numpy((rs.standard_normal((Co, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)).cuda() for _ in range(2)]
    bs_ = [torch.from_numpy((rs.standard_normal(Co) * 0.1).astype(np.float32)).cuda() for _ in range(2)]
    gout = None
    res = []
    for per in (None, "2"):
        if per is None:
            monkeypatch.delenv("EVAE_CL_IMAGES_PER_PASS", raising=False)
        else:
            monkeypatch.setenv("EVAE_CL_IMAGES_PER_PASS", per)
        t = [a.clone().requires_grad_(True) for a in (x, ws_[0], bs_[0], ws_[1], bs_[1])]
        y = ops.gated_conv2d(t[0], t[1], t[2], t[3], t[4], s, p)
        if gout is None:
            gout = torch.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32)).cuda()
        y.backward(gout)
        res.append([y.detach()] + [a.grad for a in t])
    for a, b in zip(*res):
        assert rel(a, b) < 2e-6


@pytest.mark.parametrize("gated,C,Co,k,s,p,H", [(True, 32, 64, 5, 1, 2, 14), (True, 64, 64, 3, 2, 1, 14), (False, 96, 96, 3, 1, 1, 16),
                                                (True, 1, 32, 7, 1, 3, 28)])
def test_conv_layers_at_step_size_take_the_bf16_pipe_by_default(gated, C, Co, k, s, p, H):
    """The layer shapes of c3 / c5 over enough images that the DEFAULT launch policy picks the split-bf16 kernels (forward,
    data gradient, and the weight gradient where its tiles are filled): against float64 autograd on the first images (a
    convolution is independent per image) and against the fp32-MFMA kernels on the whole tensors."""
    from evae import ops
    N = 2200
    rs = np.random.RandomState(C + Co)
    x = torch.from_numpy(rs.standard_normal((N, C, H, H)).astype(np.float32)).cuda()
    ws_ = [torch.from_numpy((rs.standard_normal((Co, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)).cuda() for _ in range(2)]
    bs_ = [torch.from_numpy((rs.standard_normal(Co) * 0.1).astype(np.float32)).cuda() for _ in range(2)]
    gout = None
    res = []
    for pipe in (1, 0):
        ops.gemm_x6_configure(pipe, 2048)
        try:
            t = [a.clone().requires_grad_(True) for a in (x, ws_[0], bs_[0], ws_[1], bs_[1])]
            y = ops.gated_conv2d(t[0], t[1], t[2], t[3], t[4], s, p) if gated else ops.conv2d(t[0], t[1], t[2], s, p)
            if gout is None:
                gout = torch.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32)).cuda()
            y.backward(gout)
            res.append([y.detach()] + [a.grad for a in (t[:5] if gated else t[:3])])
        finally:
            ops.gemm_x6_configure(1, 2048)
    for a, b in zip(*res):
        assert rel(a, b) < 8e-6          # two fp32-accurate kernels, each within ~3e-6 of float64 at K = 800
    assert not torch.equal(res[0][0], res[1][0])                 # the two pipes round differently
    # float64 reference on the first images: output and data gradient (weight gradients sum over all images: checked above
    # against the fp32 kernel, which the small-shape tests pin to float64)
    n = 3
    xr = x[:n].double().cpu().requires_grad_(True)
    w64 = [w.double().cpu() for w in ws_]; b64 = [b.double().cpu() for b in bs_]
    yr = F.conv2d(xr, w64[0], b64[0], s, p)
    if gated:
        yr = yr * torch.sigmoid(F.conv2d(xr, w64[1], b64[1], s, p))
    yr.backward(gout[:n].double().cpu())
    assert rel(res[0][0][:n], yr) < 1e-5
    if C > 1:
        assert rel(res[0][1][:n], xr.grad) < 1e-5


def test_conv2d_patch_matrix_layer_many_images():
    """First layer of models/fully_conv.py at cache_z scale (3 -> 48 channels, 64x64, stride 2, thousands of images):
    the patch matrix is built in passes whose size must agree between the workspace query and the launch
    (regression: a pass larger than the workspace faulted).  Reference: torch's own GPU convolution."""
    from evae import ops
    torch.manual_seed(0)
    N = 4600
    x = torch.rand(N, 3, 64, 64, device="cuda")
    w = (torch.randn(48, 3, 3, 3, device="cuda") / 5).requires_grad_(True)
    b = torch.randn(48, device="cuda").requires_grad_(True)
    y = ops.conv2d(x, w, b, 2, 1)
    g = torch.randn_like(y)
    y.backward(g)
    wr = w.detach().clone().requires_grad_(True); br = b.detach().clone().requires_grad_(True)
    yr = F.conv2d(x, wr, br, 2, 1)
    yr.backward(g)
    assert rel(y, yr) < 1e-5 and rel(w.grad, wr.grad) < 2e-5 and rel(b.grad, br.grad) < 2e-5


G6_CONV_CASES = [   # tools/gen_goldens.py::G6_CONV_CASES
    ("gated", 1, 32, 7, 1, 3, 28, 28, 3, None), ("gated", 32, 32, 3, 2, 1, 28, 28, 2, None),
    ("gated", 32, 64, 5, 1, 2, 14, 14, 2, None), ("gated", 64, 6, 3, 1, 1, 7, 7, 3, None),
    ("gated", 3, 32, 3, 2, 1, 16, 12, 2, "elu"),
    ("plain", 64, 1, 1, 1, 0, 28, 28, 2, "sigmoid"), ("plain", 64, 3, 1, 1, 0, 16, 16, 2, "hardtanh"),
    ("plain", 32, 48, 3, 1, 1, 9, 9, 2, None),
]


@pytest.mark.parametrize("i", range(len(G6_CONV_CASES)))
def test_conv_modules_match_reference_golden(golden, i, gemm_pipe):
    """utils.nn.GatedConv2d / Conv2d (reference utils/nn.py:72-114) as modules: output and every gradient against the
    real reference's modules on the same weights and inputs (G6)."""
    from utils.nn import GatedConv2d, Conv2d
    g = golden("g6_conv_layers")
    kind, ci, co, k, st, pd, H, W, N, act = G6_CONV_CASES[i]
    acts = {None: None, "elu": torch.nn.ELU(), "sigmoid": torch.nn.Sigmoid(), "hardtanh": torch.nn.Hardtanh(-4.5, 0.)}
    rs = np.random.RandomState(600 + i)
    x = rs.standard_normal((N, ci, H, W)).astype(np.float32)
    sc = 1.0 / np.sqrt(ci * k * k)
    wh = (rs.standard_normal((co, ci, k, k)) * sc).astype(np.float32); bh = (rs.standard_normal(co) * 0.1).astype(np.float32)
    wg = (rs.standard_normal((co, ci, k, k)) * sc).astype(np.float32); bg = (rs.standard_normal(co) * 0.1).astype(np.float32)
    T = torch.from_numpy
    if kind == "gated":
        m = GatedConv2d(ci, co, k, st, pd, activation=acts[act])
        m.load_state_dict({"h.weight": T(wh), "h.bias": T(bh), "g.weight": T(wg), "g.bias": T(bg)})
    else:
        m = Conv2d(ci, co, k, st, pd, activation=acts[act])
        m.load_state_dict({"conv.weight": T(wh), "conv.bias": T(bh)})
    m = m.cuda()
    xt = T(x).cuda().requires_grad_(True)
    y = m(xt)
    gout = rs.standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(T(gout).cuda())

    def rel(a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    for key, arr, tol in (("y", y.detach().cpu().numpy(), 1e-5), ("dx", xt.grad.cpu().numpy(), 1e-4)):
        assert rel(arr.reshape(-1)[::5], g["c%d_%s" % (i, key)]) < tol, key        # every 5th element + the L2 norm are kept
        assert abs(np.linalg.norm(arr.astype(np.float64)) - float(g["c%d_%s_norm" % (i, key)])) <= tol * float(g["c%d_%s_norm" % (i, key)])
    for name, prm in m.named_parameters():
        assert rel(prm.grad.cpu().numpy(), g["c%d_d_%s" % (i, name)]) < 1e-4, name


# ---- the gated-convolution stack over pre-split pixel images (csrc/evae_conv_win.h, evae.ops.GatedConvStackFn) ----------------
_WIDE = ((32, 7, 1, 3), (32, 3, 2, 1), (64, 5, 1, 2), (64, 3, 2, 1), (6, 3, 1, 1))       # q(z2 | x) of models/convHVAE_2level.py
_NARROW = ((32, 3, 1, 1), (32, 3, 2, 1), (64, 3, 1, 1), (64, 3, 2, 1), (6, 3, 1, 1))     # x-branch of q(z1 | x, z2)


def _stack(table, seed):
    from utils.nn import GatedConv2d, GatedConvStack
    torch.manual_seed(seed)
    layers, c = [], 1
    for co, k, s, p in table:
        layers.append(GatedConv2d(c, co, k, s, p))
        c = co
    net = GatedConvStack(*layers)
    for prm in net.parameters():                     # lively gates and biases
        with torch.no_grad():
            prm.mul_(1.5).add_(0.02 * torch.randn_like(prm))
    return net


@pytest.mark.parametrize("table,N", [(_WIDE, 37), (_WIDE, 130), (_NARROW, 37)])
def test_gated_conv_stack_on_pixel_images_matches_float64(table, N):
    """Forward output and every parameter gradient of the image pipeline (layer 0 on the first-layer kernels, layers 1-4 on the
    window kernels: stride 1 and 2, 3 x 3 and 5 x 5, parity-planar and natural image rows, the gate derivative in the data gradients'
    epilogues, the weight gradients over pixel images, the 6-channel last layer zero-padded to 32 channels) against torch float64 on the CPU (reference
    utils/nn.py:72-97 chained as models/convHVAE_2level.py:21-46), and against the layer-by-layer HIP path."""
    from evae import ops
    net = _stack(table, 3)
    rs = np.random.RandomState(N)
    x = torch.from_numpy((rs.rand(N, 1, 28, 28) < 0.3).astype(np.float32))
    ref = _stack(table, 3).double()
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    h = x.double()
    for m in ref:
        h = F.conv2d(h, m.h.weight, m.h.bias, m.h.stride, m.h.padding) * torch.sigmoid(F.conv2d(h, m.g.weight, m.g.bias, m.g.stride, m.g.padding))
    gout = torch.from_numpy(rs.standard_normal(tuple(h.shape)).astype(np.float32))
    h.backward(gout.double())
    net = net.cuda()
    spec = [(m.h.weight, 1 if m.h.stride == (1, 1) else 2, m.h.padding[0]) for m in net]
    assert ops.conv_stack_depth((N, 1, 28, 28), spec) == 5, "all five layers run on the image pipeline (the 6-channel one zero-padded to 32)"
    old_min = ops.CONV_STACK_MIN_IMAGES
    outs, grads = [], []
    try:
        for stack_on in (True, False):
            ops.CONV_STACK_MIN_IMAGES = 1 if stack_on else 1 << 30
            net.zero_grad(set_to_none=True)
            y = net(x.cuda())
            y.backward(gout.cuda())
            torch.cuda.synchronize()
            outs.append(y.detach()); grads.append({k: p.grad.clone() for k, p in net.named_parameters()})
    finally:
        ops.CONV_STACK_MIN_IMAGES = old_min
    assert rel(outs[0], h) < 2e-5 and rel(outs[1], h) < 2e-5
    refg = dict(ref.named_parameters())
    for k in grads[0]:
        assert torch.isfinite(grads[0][k]).all(), k
        assert rel(grads[0][k], refg[k].grad) < 3e-5, (k, rel(grads[0][k], refg[k].grad), rel(grads[1][k], refg[k].grad))


def test_gated_conv_stack_without_gradients():
    """cache_z / evaluation: the stack under no_grad (no gates, no fp32 copies kept) equals the layer-by-layer path"""
    from evae import ops
    net = _stack(_WIDE, 5).cuda()
    x = (torch.rand(64, 1, 28, 28, device="cuda") < 0.3).float()
    old_min = ops.CONV_STACK_MIN_IMAGES
    try:
        with torch.no_grad():
            ops.CONV_STACK_MIN_IMAGES = 1
            a = net(x)
            ops.CONV_STACK_MIN_IMAGES = 1 << 30
            b = net(x)
    finally:
        ops.CONV_STACK_MIN_IMAGES = old_min
    assert rel(a, b) < 1e-5


@pytest.mark.parametrize("N,C,H,nblk", [(20, 48, 32, 3), (70, 96, 16, 2), (5, 48, 64, 2), (33, 16, 32, 1),
                                        # fully_conv on 28 x 28 inputs: 14 x 14 and 7 x 7 grids, pixel counts that are no multiple of 16 / 32 / 256
                                        (9, 48, 14, 2), (90, 96, 7, 2), (101, 48, 14, 1), (3, 96, 14, 1)])
def test_residual_block_run_on_pixel_images_matches_float64(N, C, H, nblk):
    """A run of residual blocks x + conv(ELU(x)) (reference models/fully_conv.py:13-23) through evae.ops.ResStackFn -- window kernels
    over pixel images, ELU image written by the block before, ELU' from the saved image, weight gradients over pixel images
    (48 channels: an odd number of channel groups) -- output and every gradient against torch float64 on the CPU."""
    from evae import ops
    rs = np.random.RandomState(N + C + H)
    x = torch.from_numpy((rs.standard_normal((N, C, H, H)) * 1.5).astype(np.float32))
    ws = [torch.from_numpy((rs.standard_normal((C, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)) for _ in range(nblk)]
    bs = [torch.from_numpy((rs.standard_normal(C) * 0.1).astype(np.float32)) for _ in range(nblk)]
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]; br = [b.double().requires_grad_(True) for b in bs]
    h = xr
    for w, b in zip(wr, br):
        h = h + F.conv2d(F.elu(h), w, b, 1, 1)
    gout = torch.from_numpy(rs.standard_normal(tuple(h.shape)).astype(np.float32))
    h.backward(gout.double())
    xd = x.cuda().requires_grad_(True)
    wd = [w.cuda().requires_grad_(True) for w in ws]; bd = [b.cuda().requires_grad_(True) for b in bs]
    old = ops.RES_STACK_MIN_PIXELS
    try:
        ops.RES_STACK_MIN_PIXELS = 1
        assert ops.res_stack_supported(xd, wd)
        y = ops.res_stack(xd, list(zip(wd, bd)))
        y.backward(gout.cuda())
    finally:
        ops.RES_STACK_MIN_PIXELS = old
    assert rel(y, h) < 2e-5 and rel(xd.grad, xr.grad) < 2e-5
    for k in range(nblk):
        assert rel(wd[k].grad, wr[k].grad) < 3e-5, (k, rel(wd[k].grad, wr[k].grad))
        assert rel(bd[k].grad, br[k].grad) < 3e-5, (k, rel(bd[k].grad, br[k].grad))


@pytest.mark.gpu
def test_weight_norm_set_matches_torch_weight_norm():
    """evae.ops.weight_norm_set (one launch for every weight-normed filter of a fully_conv network pass, reference
    models/fully_conv.py:18,41-58) against torch._weight_norm in float64: values and the gradients wrt every v and g; 40 filters (two
    calls of <= 32 inside), one of them unused downstream (zero gradient)."""
    from evae import ops
    torch.manual_seed(5)
    shapes = [(48, 3, 3, 3), (48, 48, 3, 3), (96, 48, 3, 3), (96, 96, 3, 3), (1, 96, 3, 3), (3, 48, 3, 3)] * 6 + [(7, 5, 1, 1)] * 4
    vs = [torch.randn(s, device="cuda", requires_grad=True) for s in shapes]
    gs = [(torch.rand((s[0], 1, 1, 1), device="cuda") + 0.5).requires_grad_(True) for s in shapes]
    ws = ops.weight_norm_set(list(zip(vs, gs)))
    cots = [torch.randn_like(w) for w in ws]
    sum((w * c).sum() for w, c in list(zip(ws, cots))[:-1]).backward()
    for k, (v, g, w, c) in enumerate(zip(vs, gs, ws, cots)):
        v64, g64 = v.detach().double().requires_grad_(True), g.detach().double().requires_grad_(True)
        w64 = torch._weight_norm(v64, g64, 0)
        assert (w.double() - w64).abs().max() <= 2e-6 * w64.abs().max()
        if k + 1 == len(ws):
            assert v.grad.abs().max() == 0 and g.grad.abs().max() == 0
            continue
        (w64 * c.double()).sum().backward()
        assert (v.grad.double() - v64.grad).abs().max() <= 1e-5 * v64.grad.abs().max()
        assert (g.grad.double() - g64.grad).abs().max() <= 1e-5 * g64.grad.abs().max() + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,H,Co,stride,elu,xgrad,layout", [
    (20, 48, 32, 3, 1, False, True, "nchw"),        # the output head of fully_conv (48 -> 3), gradient from the likelihood as NCHW planes
    (24, 3, 32, 48, 2, True, False, "nchw"),        # its first convolution: the data, stride 2, ELU behind
    (20, 48, 32, 96, 2, True, True, "cl"),          # 48 -> 96 stride 2 + ELU (weight gradient: one channel group per launch, variant 8)
    (5, 3, 64, 48, 2, True, False, "nchw"),         # fully_conv's first convolution at its own grid (64 -> 32: variant 7)
    (6, 96, 64, 48, 1, True, True, "cl"),           # 96 -> 48 on the 64 x 64 grid + ELU
    (33, 16, 16, 24, 1, False, True, "cl"),         # ragged last block, channels that are no multiple of 16 / 32
    (9, 1, 28, 48, 2, True, False, "nchw"),         # 9 x 14 x 14 = 1 764 output pixels: no multiple of 16 (a partial last image chunk of dy)
    (9, 32, 28, 48, 2, False, True, "cl"),
    (23, 48, 14, 96, 2, True, True, "cl"),          # 14 -> 7 (fully_conv on 28 x 28 inputs), 23 x 7 x 7 = 1 127 output pixels
    (23, 96, 14, 48, 1, True, True, "up"),          # 7 -> 14 behind nn.Upsample(2)
    (37, 48, 14, 1, 1, False, True, "cl"),          # its one-channel head
    (40, 1, 32, 96, 1, True, True, "up"),           # the decoder's first convolution: ONE channel, nn.Upsample(2) in front (16 -> 32)
    (7, 96, 64, 48, 1, True, True, "up"),           # its second: 96 -> 48 behind nn.Upsample(2) (32 -> 64)
])
def test_plain_convolution_on_pixel_images_matches_float64(N, C, H, Co, stride, elu, xgrad, layout):
    """evae.ops.plain_conv (3 x 3 'same' convolution, stride 1 / 2, optional fused ELU: the convolutions outside fully_conv's residual
    runs, reference models/fully_conv.py:41-58) against torch conv2d (+ ELU) in float64: output and the gradients wrt input, weight
    and bias, the gradient handed over as NCHW planes or channels-last."""
    from evae import ops
    torch.manual_seed(N + C)
    up = layout == "up"
    x = torch.randn(N, C, H // 2 if up else H, H // 2 if up else H, device="cuda")
    if layout == "cl":
        x = x.contiguous(memory_format=torch.channels_last)
    fwd_only = xgrad is None
    x.requires_grad_(bool(xgrad))
    w = (torch.randn(Co, C, 3, 3, device="cuda") * 0.1).requires_grad_(not fwd_only)
    b = (torch.randn(Co, device="cuda") * 0.3).requires_grad_(not fwd_only)
    assert ops.plain_conv_supported(x, w, stride, 1, upsample=up)
    y = ops.plain_conv(x, w, b, stride, elu=elu, upsample=up)
    x64, w64, b64 = (t.detach().double().requires_grad_(not fwd_only) for t in (x, w, b))
    r = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x64, scale_factor=2) if up else x64, w64, b64, stride=stride, padding=1)
    if elu:
        r = torch.nn.functional.elu(r)
    assert y.shape == r.shape
    assert (y.double() - r).abs().max() <= 2e-6 * r.abs().max()
    if fwd_only:
        assert not ops.plain_conv_supported(x, w.detach().requires_grad_(True), stride, 1)
        return
    g = torch.randn(r.shape, device="cuda")
    if layout == "cl":
        g = g.contiguous(memory_format=torch.channels_last)
    y.backward(g)
    r.backward(g.double())
    assert (w.grad.double() - w64.grad).abs().max() <= 3e-6 * w64.grad.abs().max()
    assert (b.grad.double() - b64.grad).abs().max() <= 3e-6 * b64.grad.abs().max()
    if xgrad:
        assert (x.grad.double() - x64.grad).abs().max() <= 3e-6 * x64.grad.abs().max()
