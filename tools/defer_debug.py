import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import golden_inputs as gi, smoke_case
from evae import ops
from utils.utils import importing_model
model_name, C = sys.argv[1], int(sys.argv[2])
B, N = 100, 2 * C + 300
data = torch.from_numpy(gi.binary_images(9, N))
dataset = torch.utils.data.TensorDataset(data, torch.arange(N).reshape(-1, 1), torch.zeros(N))
args = smoke_case.vae_args(model_name=model_name, number_components=C, training_set_size=N, batch_size=B)
torch.manual_seed(21)
model = importing_model(args)(args).cuda(); model.train(); model._use_fused = False
x = data[:B].cuda(); idx = torch.arange(B, device="cuda").reshape(-1, 1)
g = torch.Generator(device="cuda")
names = {p.data_ptr(): k for k, p in model.named_parameters()}
grads = []
for mode in ("plain", "learn", "defer", "plain2"):
    g.manual_seed(5); model._eps_generator = g
    torch.manual_seed(77)
    model.zero_grad(set_to_none=True)
    loss, RE, KL = model.calculate_loss((x, idx), 0.5, average=True, dataset=dataset)
    if mode.startswith("plain"):
        loss.backward()
    else:
        with ops.deferred_wgrads(loss if mode == 'defer' else None):
            loss.backward()
            print(mode, "jobs:", [(names.get(j[7], "?"), j[1], j[2], j[4]) for j in (ops._DEFER[0] or {"jobs": []})["jobs"]])
    torch.cuda.synchronize()
    grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
for i, m in enumerate(("learn", "defer", "plain2")):
    bad = [(k, float((grads[0][k] - grads[i + 1][k]).abs().max())) for k in grads[0] if not torch.equal(grads[0][k], grads[i + 1][k])]
    print(m, "mismatches vs plain:", bad[:12])
