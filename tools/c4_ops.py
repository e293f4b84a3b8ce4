"""Which ATen operators does one eager c4 (hvae_2level) training step still launch, and from where?  torch.profiler with stacks:
python tools/c4_ops.py"""
import os, sys
sys.argv = [sys.argv[0], "c4", "11500", "3"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_bench.py")).read().split("def step(i):")[0])
from torch.profiler import profile, ProfilerActivity


def step(i):
    s = (i * B) % (N - B)
    x = dev_data[s:s + B]; idx = torch.arange(s, s + B, device="cuda").reshape(-1, 1)
    opt.zero_grad()
    loss, RE, KL = model.calculate_loss((x, idx), 0.5, average=True, cache=cache, dataset=ds)
    loss.backward()
    opt.step()


for i in range(3):
    step(i)
torch.cuda.synchronize()
import collections, traceback
from torch.utils._python_dispatch import TorchDispatchMode
by = collections.Counter()
LAUNCHING = ("add", "add_", "mul", "mul_", "copy_", "fill_", "zero_", "cat", "sum", "sub", "neg", "div", "clone", "normal_", "_to_copy",
             "index_select", "index_copy_", "arange", "where", "lt", "gt", "bitwise_and", "contiguous", "mean", "expand_copy")


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in LAUNCHING:
            fr = [f for f in traceback.extract_stack() if "exemplar-vae_amd" in f.filename]
            where = "%s:%d" % (fr[-1].filename.split("exemplar-vae_amd/")[-1], fr[-1].lineno) if fr else "(engine / no repo frame)"
            shp = tuple(args[0].shape) if len(args) and hasattr(args[0], "shape") else ()
            by[(name, where, shp)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    step(3)
torch.cuda.synchronize()
for (n, where, shp), c in sorted(by.items(), key=lambda t: (t[0][1], t[0][0])):
    print("%3d  %-14s %-50s %s" % (c, n, where, shp))
