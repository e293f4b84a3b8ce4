"""Phase stamps of block 0 of prior_train_kernel (build with -DEVAE_PT_STAMPS=1): tools/prior_train_probe.py [B C z]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "exemplar-vae_amd"))
import numpy as np, torch
from evae import ops
B, C, Z = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (100, 25000, 40)
g = torch.Generator(device="cuda").manual_seed(1)
z = torch.randn(B, Z, device="cuda", generator=g); c = torch.randn(C, Z, device="cuda", generator=g)
lv = torch.full((Z,), -1.0, device="cuda")
zi = torch.randint(0, 50000, (B,), device="cuda"); ci = torch.randint(0, 50000, (C,), device="cuda")
out = None
names = ["start", "queries staged+centred", "S", "partial row", "merge done (this block)", "token visible", "token in LDS", "P in LDS", "T/U", "end"]
acc = np.zeros(9)
n = 0
for it in range(30):
    out = ops.prior_train_step(z, c, lv, zi, ci, C, 0.5, out=out)
    torch.cuda.synchronize()
    st = ops.prior_train_state(torch.device("cuda", 0)).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    t = st[16:25]
    if it >= 10:
        acc += ((t - t[0]) & 0xFFFFFFFF) / 100.0; n += 1
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(50):
    ops.prior_train_step(z, c, lv, zi, ci, C, 0.5, out=out, phase=1)
e1.record(); torch.cuda.synchronize()
print("B=%d C=%d z=%d: kernel %.1f us back to back; block 0 stamps (us from start):" % (B, C, Z, e0.elapsed_time(e1) * 20.0))
for i in range(9):
    print("  %-28s %7.2f" % (["start", "queries staged+centred", "S", "partial row written", "merge part done", "token visible", "P in LDS", "T/U done", "end"][i], acc[i] / n))
