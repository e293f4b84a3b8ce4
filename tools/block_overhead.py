"""models.fully_conv.block, fused residual path on/off: time per call at launch-bound (1 image) and step-sized shapes."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "exemplar-vae_amd"))
from models.fully_conv import block

for C_, N, hw in ((48, 1, 8), (48, 200, 32), (96, 200, 16), (48, 1000, 32), (96, 1000, 16)):
    m = block(C_, C_).cuda()
    x = torch.randn(N, C_, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.randn(N, C_, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    for v in ("1", "0", "1", "0"):
        os.environ["EVAE_RESBLOCK"] = v
        for it in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(200):
                y = m(x)
            torch.cuda.synchronize(); tf = (time.perf_counter() - t) / 200
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(200):
                y = m(x); y.backward(g)
            torch.cuda.synchronize(); tb = (time.perf_counter() - t) / 200
        print("C=%d N=%d %dx%d fused=%s fwd %.1f us fwd+bwd %.1f us" % (C_, N, hw, hw, v, tf * 1e6, tb * 1e6), flush=True)
