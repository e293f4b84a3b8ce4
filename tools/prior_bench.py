#!/usr/bin/env python3
"""Exemplar-prior forward at the IWAE sizes (HIP events around the whole entry point, median of N calls).  GPU box only.
  c2: S = 5000 importance samples x 50 000 exemplars, z = 40  (matrix-core kernel of evae_prior.hip)
  c5: S = 5000 x 100 000 exemplars, z = 256                   (GEMM + log-sum-exp epilogue, evae_prior_gemm.hip)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
import torch
from evae import ops

PEAK = 157.3


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    torch.manual_seed(0)
    out = []
    for tag, S, C, Z in (("c2_iwae", 5000, 50000, 40), ("c2_iwae_4img", 20000, 50000, 40), ("c5_iwae", 5000, 100000, 256),
                         ("c5_eval_batch", 100, 100000, 256)):
        if only and only != tag:
            continue
        mu = torch.randn(1, Z, device="cuda")
        z = mu + 0.3 * torch.randn(S, Z, device="cuda")
        c = torch.randn(C, Z, device="cuda")
        lv = torch.full((Z,), -0.5, device="cuda")
        us = timeit(lambda: ops.prior_lse_fwd(z, c, lv), n=n)
        fl = 2.0 * S * C * Z
        rec = {"case": tag, "S": S, "C": C, "z": Z, "us": round(us, 1), "tflops": round(fl / us / 1e6, 2),
               "frac_fp32_mfma": round(fl / us / 1e6 / PEAK, 4)}
        print(json.dumps(rec), flush=True)
        out.append(rec)


if __name__ == "__main__":
    main()
