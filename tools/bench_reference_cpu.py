#!/usr/bin/env python3
"""Build container only: time the REAL reference (read-only at /root/reference, imported as is) on this container's CPU
cores, on the synthetic workloads of SURVEY.md 8(d) / BASELINE.md 3, and write the NUMBERS to bench/ref_cpu_container.json.
The reference never travels to the GPU box; bench.py's cpu_baseline leg there times the numpy oracle instead.

    python tools/bench_reference_cpu.py [timed_iterations]
"""
import json, os, sys, time, types, warnings
from argparse import Namespace

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, os.path.join(ROOT, "tests"))
for name in ("torchvision", "torchvision.datasets", "wget"):
    sys.modules.setdefault(name, types.ModuleType(name))
import numpy as np
import torch
import golden_inputs as gi
from models.VAE import VAE                     # noqa: E402  (the reference's)
from utils.optimizer import AdamNormGrad       # noqa: E402

warnings.simplefilter("ignore")
torch.set_num_threads(os.cpu_count())
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N, B = 50000, 100
data = torch.from_numpy(gi.binary_images(0, N))
dataset = torch.utils.data.TensorDataset(data, torch.arange(N).reshape(-1, 1), torch.arange(N) % 10)


def args_for(C, approximate=False):
    return Namespace(prior="exemplar_prior", input_type="binary", input_size=[1, 28, 28], hidden_size=300, z1_size=40, z2_size=40,
                     model_name="vae", device="cpu", number_components=C, training_set_size=N, approximate_prior=approximate,
                     approximate_k=10, no_mask=False, no_attention=False, same_variational_var=False, use_logit=False, lambd=1e-4,
                     bottleneck=6, dataset_name="dynamic_mnist", continuous=False)


def timed(fn, warm, iters):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t0) / iters


out = {"host": {"cores": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__},
       "protocol": "reference imported from /root/reference, synthetic binary 28x28 data (tests/golden_inputs.binary_images(0, 50000)), "
                   "B = 100, warm-up then %d timed iterations" % ITERS, "results": {}}

for tag, C, approx in (("train_step_exact_C1000", 1000, False), ("train_step_exact_C25000", 25000, False)):
    torch.manual_seed(0)
    a = args_for(C, approx)
    model = VAE(a); model.train()
    opt = AdamNormGrad(model.parameters(), lr=5e-4)
    state = {"i": 0}

    def step():
        s = (state["i"] % (N // B)) * B; state["i"] += 1
        x = torch.bernoulli(data[s:s + B]); idx = torch.arange(s, s + B).reshape(-1, 1)
        opt.zero_grad()
        loss, RE, KL = model.calculate_loss((x, idx), 0.5, average=True, dataset=dataset)   # body of utils/training.py:31-40
        loss.backward()
        opt.step()
    sec = timed(step, 3 if C > 5000 else 10, ITERS if C > 5000 else 3 * ITERS)
    out["results"][tag] = {"ms_per_step": round(1e3 * sec, 2), "images_per_s": round(B / sec, 1)}
    print(tag, out["results"][tag], flush=True)

# evaluation: one ELBO batch against all 50 000 exemplars, and the prior of one IWAE test image (S = 5000)
torch.manual_seed(0)
a = args_for(25000)
model = VAE(a); model.eval()
with torch.no_grad():
    cz, clv = model.cache_z(dataset)
    emb = (cz, clv, torch.arange(N))
    x = data[:B]
    sec = timed(lambda: model.calculate_loss((x, None), average=False, exemplars_embedding=emb), 1, 5)
    out["results"]["eval_elbo_batch_vs_50000"] = {"ms": round(1e3 * sec, 1)}
    print("eval", out["results"]["eval_elbo_batch_vs_50000"], flush=True)
    xs = data[:1].expand(5000, 784).contiguous()
    sec = timed(lambda: model.calculate_loss((xs, None), exemplars_embedding=emb), 0, 2)
    out["results"]["iwae_one_test_image_S5000_vs_50000"] = {"s": round(sec, 2)}
    print("iwae", out["results"]["iwae_one_test_image_S5000_vs_50000"], flush=True)

json.dump(out, open(os.path.join(ROOT, "bench", "ref_cpu_container.json"), "w"), indent=1)
print("wrote bench/ref_cpu_container.json")
