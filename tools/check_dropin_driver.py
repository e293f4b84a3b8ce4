#!/usr/bin/env python3
"""Build container only: run the REFERENCE's own driver (density_estimation.py, read-only at /root/reference, not copied)
on top of THIS build's `models` / `utils` packages -- north_star: "drops into density_estimation.py unchanged".
There is no GPU here, so the run must get through argument parsing, dataset loading (local IDX files), model and
optimizer construction and into the first training step, where this build's ops fail loudly on a non-CUDA tensor
(no CPU fallback).  Anything else -- an ImportError, a missing function or argument, a different return arity -- is a gap
in the drop-in surface."""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, runpy
sys.path.insert(0, PKG)
sys.argv = ["density_estimation.py", "--dataset_name", "dynamic_mnist", "--prior", "exemplar_prior", "--model_name", "vae",
            "--number_components", "50", "--training_set_size", "100", "--batch_size", "20", "--test_batch_size", "10",
            "--epochs", "1", "--S", "10", "--seed", "1"]
try:
    runpy.run_path("/root/reference/density_estimation.py", run_name="__main__")
except Exception as e:
    import utils, models
    print("PACKAGES", utils.__file__, models.__file__)
    print("STOPPED", type(e).__name__, str(e).splitlines()[0][:200])
'''
with tempfile.TemporaryDirectory() as d:
    raw = os.path.join(d, "datasets", "dynamic_mnist", "MNIST", "raw"); os.makedirs(raw)
    rs = np.random.RandomState(0)
    def idx(name, arr):
        with open(os.path.join(raw, name), "wb") as f:
            f.write(struct.pack(">HBB", 0, 8, arr.ndim) + struct.pack(">" + "I" * arr.ndim, *arr.shape) + arr.tobytes())
    idx("train-images-idx3-ubyte", rs.randint(0, 256, (140, 28, 28)).astype(np.uint8)); idx("train-labels-idx1-ubyte", rs.randint(0, 10, 140).astype(np.uint8))
    idx("t10k-images-idx3-ubyte", rs.randint(0, 256, (30, 28, 28)).astype(np.uint8)); idx("t10k-labels-idx1-ubyte", rs.randint(0, 10, 30).astype(np.uint8))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg")
    r = subprocess.run([sys.executable, "-c", "PKG = %r\n" % os.path.join(ROOT, "exemplar-vae_amd") + CHILD], cwd=d, env=env,
                       capture_output=True, text=True)
    tail = [l for l in r.stdout.splitlines() if l.startswith(("PACKAGES", "STOPPED"))]
    print("\n".join(tail) if tail else r.stdout[-1500:] + r.stderr[-3000:])
    ok = any("STOPPED EvaeError" in l or "CUDA" in l for l in tail) and any(ROOT in l for l in tail if l.startswith("PACKAGES"))
    print("drop-in surface OK: the reference driver reached this build's first device op" if ok else "GAP")
    sys.exit(0 if ok else 1)
