"""c5 eager steps one by one: wall ms (synchronised), re-encoded exemplars, segments the caching allocator took from the driver.
python tools/c5_steps.py [steps]"""
import os
import sys
import time
from argparse import Namespace

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
from utils.utils import importing_model      # noqa: E402
from utils.optimizer import AdamNormGrad     # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
if os.environ.get("C5_THREADS"):
    torch.set_num_threads(int(os.environ["C5_THREADS"]))
torch.manual_seed(14); torch.cuda.manual_seed(14)
B, n_train = 100, 100000
args = Namespace(prior="exemplar_prior", input_type="continuous", input_size=[3, 64, 64], hidden_size=300, z1_size=256, z2_size=40,
                 model_name="single_conv", device=str(dev), number_components=n_train, training_set_size=n_train,
                 approximate_prior=True, approximate_k=10, no_mask=False, no_attention=False, same_variational_var=False,
                 use_logit=False, lambd=1e-4, bottleneck=1, dataset_name="celeba", continuous=True, batch_size=B,
                 dynamic_binarization=False, warmup=100, S=5000, shard_exemplars=False, shard_batch=False)
model = importing_model(args)(args).to(dev)
opt = AdamNormGrad(model.parameters(), lr=5e-4)
data_dev = (torch.randint(0, 256, (n_train, 3 * 64 * 64), device=dev, dtype=torch.int16).float() + 0.5) / 256
dataset = torch.utils.data.TensorDataset(data_dev, torch.arange(n_train).reshape(-1, 1), torch.zeros(n_train))
data_dev = model.resident_data(dataset)
idx_all = torch.arange(n_train, device=dev).reshape(-1, 1)
model.train()
with torch.no_grad():
    cache = tuple(model.cache_z(dataset))
counts = []
orig = model.q_z


def q_z(x, prior=False, rows=None, **kw):
    if prior and rows is not None:
        counts.append(int(rows.numel()))
    return orig(x, prior=prior, rows=rows, **kw)


model.q_z = q_z
if os.environ.get("C5_GC") == "0":
    import gc
    gc.disable()
SYNC = os.environ.get("C5_SYNC") == "1"
seg = lambda: torch.cuda.memory_stats()["segment.all.allocated"]
for i in range(steps):
    s_ = (i * B) % (n_train - B)
    torch.cuda.synchronize(); t0 = time.perf_counter(); g0 = seg()
    opt.zero_grad()
    loss, _, _ = model.calculate_loss((data_dev[s_:s_ + B], idx_all[s_:s_ + B]), 0.5, average=True, cache=cache, dataset=dataset)
    if SYNC:
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss.backward()
    if SYNC:
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("step %2d  %.2f ms (host: fwd %.1f bwd %.1f opt %.1f, then wait %.1f)  exemplars %d  new segments %d  reserved %.2f GB"
          % (i, (t4 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, counts[-1], seg() - g0,
             torch.cuda.memory_reserved() / 2 ** 30), flush=True)
