#!/usr/bin/env python3
"""Build container only: a checkpoint written by THIS build's utils.utils.save_model is loaded by the REFERENCE's
utils.utils.load_model into the reference's model and optimizer (the other direction is a committed fixture,
tests/golden/g12_checkpoint.pth, see tools/gen_goldens.py::g12).  Two processes, because both trees name their
packages `models` / `utils`."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MINE = r'''
import sys, os
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd")); sys.path.insert(1, os.path.join(ROOT, "tests")); sys.path.insert(2, os.path.join(ROOT, "oracle"))
import torch, smoke_case
from models.VAE import VAE
from utils.optimizer import AdamNormGrad
from utils.utils import load_model, save_model
args = smoke_case.vae_args(input_size=[1, 8, 8], hidden_size=16, z1_size=8, z2_size=8, number_components=10, training_set_size=50, device="cpu")
model = VAE(args); opt = AdamNormGrad(model.parameters(), lr=5e-4)
load_model(os.path.join(ROOT, "tests", "golden", "g12_checkpoint.pth"), model, opt)
save_model(OUT + ".tmp", OUT, {'epoch': 9, 'state_dict': model.state_dict(), 'optimizer': opt.state_dict(), 'best_loss': 2.5, 'e': 1})
'''
REF = r'''
import sys, os, types
sys.path.insert(0, "/root/reference"); sys.dont_write_bytecode = True
for name in ("torchvision", "torchvision.datasets", "wget"): sys.modules.setdefault(name, types.ModuleType(name))
from argparse import Namespace
import numpy as np, torch
from models.VAE import VAE
from utils.optimizer import AdamNormGrad
from utils.utils import load_model
args = Namespace(prior="exemplar_prior", input_type="binary", input_size=[1, 8, 8], hidden_size=16, z1_size=8, z2_size=8, model_name="vae",
                 device="cpu", number_components=10, training_set_size=50, approximate_prior=False, approximate_k=10, no_mask=False,
                 no_attention=False, same_variational_var=False, use_logit=False, lambd=1e-4, bottleneck=6, dataset_name="dynamic_mnist", continuous=False)
model = VAE(args); opt = AdamNormGrad(model.parameters(), lr=5e-4)
ck = load_model(OUT, model, opt)
g = np.load(os.path.join(ROOT, "tests", "golden", "g12_checkpoint.npz"))
assert ck["epoch"] == 9 and ck["e"] == 1
for n, p in model.named_parameters():
    assert np.array_equal(p.detach().numpy(), g["loaded_" + n]), n
    assert np.array_equal(opt.state[p]["exp_avg"].numpy(), g["loaded_m_" + n]), n
    assert np.array_equal(opt.state[p]["exp_avg_sq"].numpy(), g["loaded_v_" + n]), n
    assert int(opt.state[p]["step"]) == 2
print("reference loaded this build's checkpoint: %d tensors, optimizer state identical" % len(list(model.parameters())))
'''
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "mine.pth")
    head = "ROOT = %r; OUT = %r\n" % (ROOT, out)
    for code in (MINE, REF):
        r = subprocess.run([sys.executable, "-c", head + code], capture_output=True, text=True)
        if r.returncode:
            print(r.stdout[-2000:], r.stderr[-3000:]); sys.exit(1)
    print(r.stdout.strip().splitlines()[-1])
