#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
import torch
from evae import ops
dev = torch.device("cuda"); torch.manual_seed(0)
q = torch.randn(100, 40, device=dev); c = torch.randn(25000, 40, device=dev)
q5 = torch.randn(64, 256, device=dev); c5 = torch.randn(100000, 256, device=dev)
for _ in range(5):
    ops.pairdist_topk(q, c, 10); ops.pairdist_topk(q5, c5, 10)
torch.cuda.synchronize()
