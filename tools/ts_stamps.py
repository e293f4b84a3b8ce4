#!/usr/bin/env python3
"""Time stamps of the stream top-K kernel's phases (EVAE_TS_ABL=64: thread 0 of blocks 0 and 100 writes wall_clock64 at
start / queries loaded / end of each tile's main loop / epilogues done / flushed+flag / barrier passed / lists in LDS / ... )."""
import os, sys
os.environ["EVAE_TS_ABL"] = str(int(os.environ.get("EVAE_TS_ABL", "0")) | 64)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
import torch
from evae import ops
dev = torch.device("cuda"); torch.manual_seed(0)
for name, B, N, Z, k in (("c5", 100, 100000, 256, 10), ("c2", 100, 25000, 40, 10)):
    q = torch.randn(B, Z, device=dev); c = torch.randn(N, Z, device=dev)
    for _ in range(5):
        ops.pairdist_topk(q, c, k)
    torch.cuda.synchronize()
    ws = [v for kk, v in ops._ws.items() if kk[0] == "topk"][0]
    st = ws[(256 + 1024) * 4:(256 + 1024) * 4 + 512].view(torch.int64).cpu().tolist()
    for blk in (0, 1):
        t = [x for x in st[32 * blk:32 * blk + 32]]
        n = next((i for i, x in enumerate(t[1:], 1) if x < t[0]), len(t))
        print(name, "block", 0 if blk == 0 else 100, " ".join("%.1f" % ((x - t[0]) / 100.0) for x in t[:n]), "us (100 MHz clock)")
