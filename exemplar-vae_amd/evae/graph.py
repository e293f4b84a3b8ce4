"""hipGraph capture of one whole training step (reference utils/training.py:27-46: binarise, loss, backward,
optimizer) so that a step costs ONE graph launch on the host instead of ~90 kernel launches plus the Python
between them.  At the 25 000-exemplar configuration a step is < 2 ms of GPU work; with the exemplars sharded
over 8 GPUs it is a few hundred microseconds, far below what eager launching can feed.

What varies between steps lives in static device buffers that are refreshed before each replay:
  the batch (images + dataset indices), the exemplar indices (still drawn by the CPU generator exactly like
  reference models/BaseModel.py:245), beta, and AdamNormGrad's bias-corrected step size.  eps and the dynamic
  binarisation come from the CUDA generator inside the graph (torch registers it with the capture, so every
  replay advances the Philox offset).  RCCL collectives of the sharded prior are captured too.
"""
import torch

from . import shard


class GraphedTrainStep:
    def __init__(self, model, optimizer, dataset, batch_size, dynamic_binarization, warmup_steps=3):
        a = model.args
        self.model, self.opt, self.dataset = model, optimizer, dataset
        self.B = int(batch_size)
        self.binarize = bool(dynamic_binarization)
        dev = torch.device(a.device)
        D = int(torch.tensor(a.input_size).prod().item())
        self.x_in = torch.zeros((self.B, D), device=dev)
        self.idx_in = torch.zeros((self.B, 1), dtype=torch.int64, device=dev)
        self.beta = torch.ones((), device=dev)
        self.ex_idx = torch.zeros(a.number_components, dtype=torch.int64, device=dev)
        self._ex_host = torch.zeros(a.number_components, dtype=torch.int64)   # pageable on purpose, see _refresh
        self.out = None
        self.graph = None
        self.warmup_steps = warmup_steps
        self._calls = 0

    # the body that gets captured
    def _body(self):
        x = torch.bernoulli(self.x_in) if self.binarize else self.x_in
        self.opt.zero_grad(set_to_none=True)      # backward then installs the fused node's gradient buffers
        loss, RE, KL = self.model.calculate_loss((x, self.idx_in), self.beta, average=True, dataset=self.dataset)
        loss.backward()
        self.opt.step(_captured=True)
        return torch.stack((loss.detach(), -RE.detach(), KL.detach()))

    def _refresh(self, data, indices, beta):
        self.x_in.copy_(data.reshape(self.B, -1))
        self.idx_in.copy_(indices.reshape(self.B, 1))
        self.beta.fill_(float(beta))
        # same CPU-generator draw, with replacement, as the reference (models/BaseModel.py:245)
        a = self.model.args
        torch.randint(low=0, high=a.training_set_size, size=(a.number_components,), out=self._ex_host)
        # pageable source: the runtime stages it before returning, so the next draw cannot race the copy
        self.ex_idx.copy_(self._ex_host)
        self.opt.advance_graph_step()

    def __call__(self, data, indices, beta):
        """One training step; returns a device tensor (loss, -RE, KL) valid until the next call."""
        if self.graph is None and self._calls == 0:
            self.opt.enable_graph_mode()
        self.model._exemplar_indices_override = self.ex_idx
        try:
            self._refresh(data, indices, beta)
            if self.graph is None:
                # eager warm-up steps on a side stream (workspaces, attributes, RCCL channels), then capture
                if self._calls < self.warmup_steps:
                    s = torch.cuda.Stream()
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        out = self._body()
                    torch.cuda.current_stream().wait_stream(s)
                    self._calls += 1
                    return out
                torch.cuda.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.out = self._body()
                # capture does not execute: replay once for this call's step
            self.graph.replay()
            self._calls += 1
            return self.out
        finally:
            self.model._exemplar_indices_override = None
