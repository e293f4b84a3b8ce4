"""hipGraph capture of one whole training step (reference utils/training.py:27-46: binarise, loss, backward,
optimizer) so that a step costs ONE graph launch on the host instead of ~90 kernel launches plus the Python
between them.  At the 25 000-exemplar configuration a step is < 2 ms of GPU work; with the exemplars sharded
over 8 GPUs it is a few hundred microseconds, far below what eager launching can feed.

What varies between steps lives in static device buffers that are refreshed before each replay:
  the batch (images + dataset indices), the exemplar indices (still drawn by the CPU generator exactly like
  reference models/BaseModel.py:245), beta, and AdamNormGrad's bias-corrected step size (one upload for the
  scalars, pinned double-buffered staging, no fill kernels).  eps and the dynamic
  binarisation come from the CUDA generator inside the graph (torch registers it with the capture, so every
  replay advances the Philox offset).  RCCL collectives of the sharded prior are captured too.
"""
import torch

from . import ops, shard


class GraphedTrainStep:
    def __init__(self, model, optimizer, dataset, batch_size, dynamic_binarization, warmup_steps=3):
        a = model.args
        self.model, self.opt, self.dataset = model, optimizer, dataset
        self.B = int(batch_size)
        self.binarize = bool(dynamic_binarization)
        dev = torch.device(a.device)
        D = int(torch.tensor(a.input_size).prod().item())
        C = int(a.number_components)
        self.x_in = torch.zeros((self.B, D), device=dev)
        self.idx_in = torch.zeros((self.B, 1), dtype=torch.int64, device=dev)
        # gather list of the fused step: [this rank's exemplar indices | the staging rows of the batch]; only the head
        # changes between steps, so the captured graph needs no arange/cat
        self.lo, self.hi = shard.bounds(C) if model._sharded() else (0, C)
        _, n_data = model.resident_data_ext(dataset, self.B)
        self.rows = torch.zeros((self.hi - self.lo) + self.B, dtype=torch.int64, device=dev)
        self.rows[self.hi - self.lo:] = torch.arange(n_data, n_data + self.B, device=dev)
        # per-step scalars (beta, Adam step size per group) travel in ONE small upload; host staging is pinned and
        # double-buffered, an event per buffer says when its upload has been consumed
        self.ngroups = len(optimizer.param_groups)
        self.scal = torch.zeros(1 + self.ngroups, device=dev)
        self.beta = self.scal[0:1].reshape(())
        self._h_scal = [torch.zeros(1 + self.ngroups).pin_memory() for _ in range(2)]
        self._h_idx = [torch.zeros(C, dtype=torch.int64).pin_memory() for _ in range(2)]
        self._h_ev = [torch.cuda.Event() for _ in range(2)]
        self._one = torch.ones((), device=dev)
        self.out = torch.zeros(3, device=dev)       # (loss, -RE, KL) of the last step
        self.totals = torch.zeros(3, device=dev)    # running sums since reset_totals()
        self.graph = None
        self.failed = False
        self.warmup_steps = warmup_steps
        self._calls = 0

    def reset_totals(self):
        self.totals.zero_()

    # the body that gets captured
    def _body(self):
        x = torch.bernoulli(self.x_in) if self.binarize else self.x_in
        self.opt.zero_grad(set_to_none=True)      # backward then installs the fused node's gradient buffers
        loss, RE, KL = self.model.calculate_loss((x, self.idx_in), self.beta, average=True, dataset=self.dataset)
        loss.backward(gradient=self._one)
        self.opt.step(_captured=True)
        ops.step_stats_add(loss.detach(), RE.detach(), KL.detach(), self.out, self.totals)
        return self.out

    def _refresh(self, data, indices, beta):
        k = self._calls & 1
        self._h_ev[k].synchronize()               # the upload issued two steps ago from this buffer is done
        self.x_in.copy_(data.reshape(self.B, -1), non_blocking=True)
        self.idx_in.copy_(indices.reshape(self.B, 1), non_blocking=True)
        # same CPU-generator draw, with replacement, as the reference (models/BaseModel.py:245)
        a = self.model.args
        hi_ = self._h_idx[k]
        torch.randint(low=0, high=a.training_set_size, size=(a.number_components,), out=hi_)
        self.rows[:self.hi - self.lo].copy_(hi_[self.lo:self.hi], non_blocking=True)
        hs = self._h_scal[k]
        hs[0] = float(beta)
        self.opt.advance_graph_step(host_out=hs[1:])
        self.scal.copy_(hs, non_blocking=True)
        self._h_ev[k].record()

    def __call__(self, data, indices, beta):
        """One training step; returns a device tensor (loss, -RE, KL) valid until the next call."""
        if self.graph is None and self._calls == 0:
            self.opt.enable_graph_mode(storage=[self.scal[1 + g:2 + g] for g in range(self.ngroups)])
        self.model._exemplar_indices_override = (self.rows, self.hi - self.lo)
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
                if self.failed:
                    self._body()                 # capture was refused once: keep stepping eagerly
                    self._calls += 1
                    return self.out
                torch.cuda.synchronize()
                main_stream = torch.cuda.current_stream()
                graph = torch.cuda.CUDAGraph()
                # with RCCL in the step its watchdog thread polls events while we capture: thread-local capture mode
                # keeps those calls from invalidating the capture
                mode = "thread_local" if shard.is_active() else "global"
                import os
                mode = os.environ.get("EVAE_CAPTURE_MODE", mode)
                try:
                    with torch.cuda.graph(graph, capture_error_mode=mode):
                        self._body()
                    self.graph = graph
                except Exception as e:           # an op of this model that cannot be captured: eager from here on
                    import sys
                    print("evae.graph: hipGraph capture of the training step failed (%s: %s); running eagerly"
                          % (type(e).__name__, str(e).splitlines()[0][:120]), file=sys.stderr)
                    self.failed = True
                    # torch.cuda.graph.__exit__ does not restore the stream when capture_end() itself raises: the current
                    # stream would stay the (invalidated, still "capturing") capture stream and every later launch fail
                    torch.cuda.set_stream(main_stream)
                    for _ in range(4):           # a failed capture can leave a sticky HIP error behind: drain it
                        try:
                            torch.cuda.synchronize()
                            break
                        except Exception:
                            pass
                    self._body()
                    self._calls += 1
                    return self.out
                # capture does not execute: replay once for this call's step
            self.graph.replay()
            self._calls += 1
            return self.out
        finally:
            self.model._exemplar_indices_override = None
