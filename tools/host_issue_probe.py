#!/usr/bin/env python3
"""Host time to ISSUE the launches of a fully_conv residual run (6 blocks, forward + backward) against the GPU time they take:
is an eager c5 step waiting for Python?   python tools/host_issue_probe.py [images] [channels] [grid]"""
import sys, time, torch
sys.path.insert(0, "exemplar-vae_amd")
from evae import ops
N, C, H = (int(v) for v in (sys.argv[1:4] + ["100", "48", "32"][len(sys.argv) - 1:]))
x = torch.randn(N, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
ws = [(torch.randn(C, C, 3, 3, device="cuda") * 0.05).requires_grad_(True) for _ in range(6)]
bs = [torch.zeros(C, device="cuda", requires_grad=True) for _ in range(6)]
g = torch.randn(N, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)


def once():
    y = ops.res_stack(x, list(zip(ws, bs)))
    y.backward(g)


for _ in range(5):
    once()
torch.cuda.synchronize()
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    once()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("residual run of 6 blocks, %d x %d x %d x %d: host issue %.3f ms, GPU done after %.3f ms per forward + backward (~44 launches)"
      % (N, C, H, H, 1e3 * (t1 - t0) / reps, 1e3 * (t2 - t0) / reps))
