#!/usr/bin/env python3
"""Timing of evae_pairdist_topk at the c2 / c5 shapes: screening path (default) vs the exact fp64 scan
(EVAE_TOPK_EXACT_SCAN=1, child process), plus the number of candidates the screening keeps per query."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd")); sys.path.insert(1, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as gi
from evae import ops
dev = torch.device("cuda"); torch.manual_seed(0)
tag = "exact scan" if os.environ.get("EVAE_TOPK_EXACT_SCAN") else "screening "
for name, B, N, Z, k, clustered in (("c2 random", 100, 25000, 40, 10, False), ("c2 clustered", 100, 25000, 40, 10, True),
                                    ("c5 random", 64, 100000, 256, 10, False), ("c5 clustered", 64, 100000, 256, 10, True)):
    if clustered:
        z, c = gi.clustered_latents(1, B, N, Z)
        q, c = torch.from_numpy(z).to(dev), torch.from_numpy(c).to(dev)
    else:
        q = torch.randn(B, Z, device=dev); c = torch.randn(N, Z, device=dev)
    for _ in range(3):
        ops.pairdist_topk(q, c, k)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); ops.pairdist_topk(q, c, k); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    extra = ""
    if not os.environ.get("EVAE_TOPK_EXACT_SCAN"):
        ws = ops._ws.get(("topk", 0))
        # candidate counters live in the workspace (layout of evae_topk_screen.hip): cn [N], qn [ldt], cnmax, tmin, thr, cnt
        al = lambda x: (x + 255) // 256 * 256
        ldt = (B + 63) // 64 * 64; nt = (N + 127) // 128
        off = al(N * 4) + al(ldt * 4) + 256 + al(nt * ldt * 4) + al(ldt * 4)
        cnt = ws[off:off + 4 * B].view(torch.int32)
        extra = "  candidates/query: mean %.0f max %d" % (cnt.float().mean().item(), cnt.max().item())
    print("%s %-13s B=%3d N=%6d z=%3d k=%d: %8.1f us (cache %.0f MB -> %.0f GB/s)%s"
          % (tag, name, B, N, Z, k, ts[len(ts) // 2], N * Z * 4 / 1e6, N * Z * 4 / ts[len(ts) // 2] / 1e3, extra))
if not os.environ.get("EVAE_TOPK_EXACT_SCAN"):
    subprocess.run([sys.executable, __file__], env=dict(os.environ, EVAE_TOPK_EXACT_SCAN="1"))
