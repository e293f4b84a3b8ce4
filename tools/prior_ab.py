#!/usr/bin/env python3
"""A/B of the two exemplar-prior forward kernels (direct-difference VALU vs matrix-core) on the same inputs."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
    import torch
    from evae import ops
    torch.manual_seed(0)
    out = {}
    cases = [(100, 25000, 40, True), (5000, 50000, 40, False), (37, 1000, 40, True), (300, 777, 24, True),
             (5000, 3000, 64, False), (128, 128, 8, False), (129, 129, 40, True)]
    if os.environ.get("PRIOR_CASES"):
        cases = [tuple(int(v) for v in c.split(",")[:3]) + (c.split(",")[3] == "1",) for c in os.environ["PRIOR_CASES"].split(";")]
    for (B, C, Z, masked) in cases:
        z = torch.randn(B, Z, device="cuda") * 1.5
        c = torch.randn(C, Z, device="cuda")
        lv = torch.randn(Z, device="cuda") * 0.3 - 1.0
        zi = torch.randint(0, C, (B,), device="cuda") if masked else None
        ci = torch.arange(C, device="cuda") if masked else None
        m, s, n, _ = ops.prior_lse_fwd(z, c, lv, zi, ci)
        lp, lse = ops.prior_merge(m, s, n, C)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); ops.prior_lse_fwd(z, c, lv, zi, ci); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
        out[(B, C, Z, masked)] = (lp.double().cpu(), n.cpu(), min(ts))
    torch.save(out, sys.argv[2])
else:
    import torch
    subprocess.run([sys.executable, __file__, "child", "/tmp/prior_mfma.pt"], env=dict(os.environ), check=True)
    subprocess.run([sys.executable, __file__, "child", "/tmp/prior_valu.pt"], env=dict(os.environ, EVAE_PRIOR_VALU="1"), check=True)
    a, b = torch.load("/tmp/prior_mfma.pt"), torch.load("/tmp/prior_valu.pt")
    for k in a:
        d = (a[k][0] - b[k][0]).abs()
        print(k, "max |dlogp| %.3e  rel %.3e  nmask equal %s   mfma %.1f us  valu %.1f us" % (
            d.max(), (d / b[k][0].abs()).max(), bool((a[k][1] == b[k][1]).all()), a[k][2], b[k][2]))
