"""Stand-alone timing of the byte store's step head with and without the control-block hand-over (evae_batch_prologue_u8_step)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "exemplar-vae_amd"))
from evae import ops
dev = torch.device("cuda:0")
N, B, D, Z = 50000, 100, 784, 40
data = torch.randint(0, 256, (N + B, D), dtype=torch.uint8, device=dev)
words = int(os.environ.get('WORDS', 2 * 19968 + 4 * 25000 + 2 * B + 64))
words += words & 1
s0 = torch.zeros(words, dtype=torch.int64, device=dev); s1 = torch.zeros_like(s0); ctl = torch.zeros_like(s0)
o_idx, o_seed = (19968 + B, 19968 + 2 * B) if words > 30000 else (0, B)
for s in (s0, s1):
    s[o_idx:o_seed] = torch.randint(0, N, (B,), device=dev)
    s[o_seed] = 1234
state = torch.zeros(2, dtype=torch.int32, device=dev)
x = torch.zeros(B, D, device=dev); eps = torch.zeros(B, Z, device=dev); stage = data[N:]
idx = s0[o_idx:o_seed]; seed = s0[o_seed:o_seed + 2]
from evae import _lib
lib = _lib.load()
H = 300
wh = torch.randn(H, D, device=dev); wg = torch.randn(H, D, device=dev)
prep = torch.empty(lib.evae_dense_u8_prepared_bytes(H, D), dtype=torch.uint8, device=dev)
wm = torch.randn(Z, H, device=dev); w2h = torch.randn(H, H, device=dev); w2g = torch.randn(H, H, device=dev)
b1 = torch.empty(lib.evae_dense_bwd_data_wt_bytes(Z, H, 1), dtype=torch.uint8, device=dev)
b2 = torch.empty(lib.evae_dense_bwd_data_wt_bytes(H, H, 2), dtype=torch.uint8, device=dev)
PREP = (wh, wg, prep, [(wm, None, b1), (w2h, w2g, b2)] if os.environ.get('JOBS', '1') == '1' else []) if os.environ.get("PREP", "1") == "1" else None
def t(job, n=200):
    for _ in range(10):
        ops.batch_prologue_u8(data, idx, True, seed, 255.0, x, stage, eps, prepare=PREP, ctl_job=job)
    torch.cuda.synchronize()      # (back-to-back launches: the figure is the larger of host issue and device time -- rocprofv3 for the latter)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        ops.batch_prologue_u8(data, idx, True, seed, 255.0, x, stage, eps, prepare=PREP, ctl_job=job)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("plain    %.1f us" % t(None))
print("handover %.1f us" % t((s0, s1, ctl, state, o_idx, o_seed)))
print("state", state.tolist(), "ctl == s0", bool(torch.equal(ctl, s0)))
state[0] = 1
ops.batch_prologue_u8(data, idx, True, seed, 255.0, x, stage, eps, prepare=PREP, ctl_job=(s0, s1, ctl, state, o_idx, o_seed)); print("ctl == s1", bool(torch.equal(ctl, s1)))
