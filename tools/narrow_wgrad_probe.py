import os, sys
sys.path.insert(0, "/root/repo/exemplar-vae_amd")
import torch
from evae import ops, _lib
lib = _lib.load(); p, st = ops._p, ops._stream
M, N, K = 25100, 40, 300
dm = torch.randn(M, N, device="cuda") * 0.01; a1 = torch.randn(M, K, device="cuda")
dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
nb = lib.evae_dense_bwd_weight_workspace_bytes(M, N, K); w = torch.zeros(nb, dtype=torch.uint8, device="cuda")
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / n
ph = lambda k: _lib.check(lib.evae_dense_bwd_weight_phased(p(dm), M, N, N, p(a1), None, K, K, p(dw), p(db), 0, p(w), nb, k, st()), "x")
print("narrow=%s  kernel %.1f us, finish %.1f us, ws %.1f MB" % (os.environ.get("EVAE_WGRAD_NARROW", "1"), t(lambda: ph(1)), t(lambda: ph(2)), nb / 1e6))
