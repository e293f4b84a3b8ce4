#!/usr/bin/env python3
"""Eager timing of one training step for the BASELINE.json configurations (parity-test cases, not bench lines).
usage: config_bench.py c1|c2|c2a|c3|c4|c5 [exemplars] [steps]   (c2a = c2 with the approximate kNN prior)"""
import os, sys, time
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd")); sys.path.insert(1, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import golden_inputs as gi

which = sys.argv[1]
CFG = {
    "c1": dict(model_name="vae", C=1000, N=50000, input_size=[1, 28, 28], input_type="binary", z=40),
    "c2": dict(model_name="vae", C=25000, N=50000, input_size=[1, 28, 28], input_type="binary", z=40),
    "c2a": dict(model_name="vae", C=25000, N=50000, input_size=[1, 28, 28], input_type="binary", z=40, approximate=True),
    "c3": dict(model_name="convhvae_2level", C=25000, N=50000, input_size=[1, 28, 28], input_type="binary", z=40),
    "c4": dict(model_name="hvae_2level", C=11500, N=23000, input_size=[1, 28, 28], input_type="binary", z=40),
    "c5": dict(model_name="single_conv", C=100000, N=100000, input_size=[3, 64, 64], input_type="continuous", z=256,
               approximate=True),
}[which]
C = int(sys.argv[2]) if len(sys.argv) > 2 else CFG["C"]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
B = 100
args = Namespace(prior="exemplar_prior", input_type=CFG["input_type"], input_size=CFG["input_size"], hidden_size=300,
                 z1_size=CFG["z"], z2_size=CFG["z"], model_name=CFG["model_name"], device="cuda", number_components=C,
                 training_set_size=CFG["N"], approximate_prior=bool(CFG.get("approximate")), approximate_k=10,
                 no_mask=False, no_attention=False, same_variational_var=False, use_logit=False, lambd=1e-4,
                 bottleneck=1, dataset_name="dynamic_mnist", continuous=False, batch_size=B, dynamic_binarization=False,
                 warmup=100, S=5000)
from utils.utils import importing_model
from utils.optimizer import AdamNormGrad
torch.manual_seed(0)
model = importing_model(args)(args).cuda()
opt = AdamNormGrad(model.parameters(), lr=5e-4)
D = int(np.prod(args.input_size))
N = CFG["N"]
if CFG["input_type"] == "binary":
    data = torch.from_numpy(gi.binary_images(0, N, D))
else:
    data = ((torch.randint(0, 256, (N, D)).float() + 0.5) / 256)
ds = torch.utils.data.TensorDataset(data, torch.arange(N).reshape(-1, 1), torch.zeros(N))
dev_data = model.resident_data(ds)
model.train()
cache = None
if args.approximate_prior:
    with torch.no_grad():
        t0 = time.perf_counter(); cache = tuple(model.cache_z(ds)); torch.cuda.synchronize()
        print("cache_z(%d): %.1f ms" % (N, 1e3 * (time.perf_counter() - t0)))


def step(i):
    s = (i * B) % (N - B)
    x = dev_data[s:s + B]; idx = torch.arange(s, s + B, device="cuda").reshape(-1, 1)
    opt.zero_grad()
    loss, RE, KL = model.calculate_loss((x, idx), 0.5, average=True, cache=cache, dataset=ds)
    loss.backward()
    opt.step()
    return loss


if os.environ.get("GRAPH"):       # replay the whole step from a hipGraph (evae/graph.py) instead of launching eagerly
    from evae.graph import GraphedTrainStep
    runner = GraphedTrainStep(model, opt, ds, B, False)
    if cache is not None:
        cache = runner.set_cache(cache)

    def step(i):      # noqa: F811
        s = (i * B) % (N - B)
        return runner(dev_data[s:s + B], torch.arange(s, s + B, device="cuda").reshape(-1, 1), 0.5)[0]
    for i in range(4):
        step(i)

for i in range(2):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    l = step(2 + i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("%s %s C=%d: %.2f ms/step -> %.0f images/s  (loss %.3f, peak mem %.1f GB)"
      % (which, CFG["model_name"], C, 1e3 * dt, B / dt, l.item(), torch.cuda.max_memory_allocated() / 2**30))
