#!/usr/bin/env python3
"""evae_pairdist_topk at c5 size (100 x 100 000 x 256, k = 10) and c2 size under the stream kernel's ablation knob
(EVAE_TS_ABL, read once per process: run once per value)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd")); sys.path.insert(1, os.path.join(ROOT, "tests"))
import torch
from evae import ops
dev = torch.device("cuda"); torch.manual_seed(0)
for name, B, N, Z, k in (("c5", 100, 100000, 256, 10), ("c2", 100, 25000, 40, 10)):
    q = torch.randn(B, Z, device=dev); c = torch.randn(N, Z, device=dev)
    for _ in range(5):
        ops.pairdist_topk(q, c, k)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); ops.pairdist_topk(q, c, k); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print("abl=%s %s: median %.1f us  min %.1f" % (os.environ.get("EVAE_TS_ABL", "0"), name, ts[len(ts) // 2], ts[0]))
