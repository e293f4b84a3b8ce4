"""one eager hvae_2level training step (+ a second one) with full tracebacks: tools/micro/hvae_step_debug.py [C]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from argparse import Namespace
from utils.utils import importing_model
from evae import ops
import golden_inputs as gi
C = int(sys.argv[1]) if len(sys.argv) > 1 else 500
N, B = 2 * C, 100
args = Namespace(prior="exemplar_prior", input_type="binary", input_size=[1, 28, 28], hidden_size=300, z1_size=40, z2_size=40,
                 model_name="hvae_2level", device="cuda", number_components=C, training_set_size=N, approximate_prior=False,
                 approximate_k=10, no_mask=False, no_attention=False, same_variational_var=False, use_logit=False, lambd=1e-4,
                 bottleneck=1, dataset_name="dynamic_mnist", continuous=False, batch_size=B, dynamic_binarization=False, warmup=100, S=5000)
model = importing_model(args)(args).cuda(); model.train()
data = torch.from_numpy(gi.binary_images(9, N))
dataset = torch.utils.data.TensorDataset(data, torch.arange(N).reshape(-1, 1), torch.zeros(N))
for it in range(2):
    try:
        model.zero_grad()
        loss, RE, KL = model.calculate_loss((data[:B].cuda(), torch.arange(B).reshape(-1, 1).cuda()), 0.5, average=True, dataset=dataset)
        with ops.deferred_wgrads(loss):
            loss.backward()
        torch.cuda.synchronize()
        print("step", it, float(loss), sum(float(p.grad.double().norm()) for p in model.parameters() if p.grad is not None))
    except Exception:
        traceback.print_exc()
