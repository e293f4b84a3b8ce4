"""AdamNormGrad with the reference's constructor and state layout (reference utils/optimizer.py:7-80;
state keys 'step', 'exp_avg', 'exp_avg_sq'), one fused multi-tensor HIP launch pair per step.

Two ways to run a step:
  * eager (`step()`): the bias-corrected step size is computed on the host and passed by value;
  * inside a captured hipGraph (`evae.graph.GraphedTrainStep`): kernel arguments are frozen at capture, so
    the step size lives in a device scalar that `advance_graph_step()` updates before every replay.
With torch.distributed active the gradients are averaged over ranks first (the per-tensor norm needs the
reduced gradient)."""
import torch
from torch.optim import Optimizer

from evae import ops, shard


class AdamNormGrad(Optimizer):
    """Adam on per-tensor L2-normalised gradients: g <- g / (||g||_2 + 1e-7)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._tables = {}
        self._graph_step_size = None      # device scalars (one per param group) while graph-captured

    def _init_state(self, p):
        state = self.state[p]
        if len(state) == 0:
            state['step'] = 0
            state['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return state

    # ---- graph mode ---------------------------------------------------------------------------------
    def enable_graph_mode(self, storage=None):
        """Call before capturing a step: the captured launches read the step size from device memory
        (`storage`: caller-owned 1-element device tensors, one per param group)."""
        dev = self.param_groups[0]['params'][0].device
        self._graph_step_size = storage if storage is not None else [torch.zeros(1, device=dev) for _ in self.param_groups]

    def learn_members(self, tables):
        """After an EAGER step of the runner that owns `tables`: the parameters that received a gradient are the participants
        of its captured step (the reference skips parameters without one, utils/optimizer.py:50-57, and keeps no state for
        them -- e.g. the unused BatchNorm2d of fully_conv's blocks).  Returns True when the participants of every group share
        ONE step count, which is what a captured launch (one device step size per group) can express; a resumed checkpoint
        whose participants disagree has to be stepped eagerly (the caller's fallback)."""
        ok = True
        for gi, group in enumerate(self.param_groups):
            members = [p for p in group['params'] if p.grad is not None]
            tables[("members", gi)] = members
            ok = ok and len({self._init_state(p)['step'] for p in members}) <= 1
        return ok

    def advance_graph_step(self, host_out=None, tables=None):
        """Before each replay: bump the step counters of the parameters that take part in the captured step (`tables`: the
        runner-owned dict; its ("members", gi) lists come from learn_members / the capture) and upload the bias-corrected
        step size -- or, with `host_out`, write it there for the caller to upload.  The captured launches apply ONE step size
        per group, so the participants of a group must share one step count; nothing is mutated when they do not."""
        plan = []
        for gi, group in enumerate(self.param_groups):
            members = (tables or {}).get(("members", gi))
            if members is None:
                raise RuntimeError("AdamNormGrad.advance_graph_step: the participants of the captured step are unknown; run one "
                                   "eager step and learn_members() first")
            steps = {self._init_state(p)['step'] for p in members}
            if len(steps) > 1:
                raise RuntimeError("AdamNormGrad: parameters of one group reached the captured step with different step "
                                   "counts %s; step them eagerly" % sorted(steps))
            plan.append((gi, group, members, (steps.pop() + 1) if steps else None))
        for gi, group, members, step in plan:
            if step is None:
                continue
            for p in members:
                self.state[p]['step'] = step
            beta1, beta2 = group['betas']
            v = ops.adam_step_size(step, group['lr'], beta1, beta2)
            if host_out is not None:
                host_out[gi] = v
            else:   # fill kernel (value travels as a kernel argument: no host buffer to race with later steps)
                ((tables or {}).get("step_size") or self._graph_step_size)[gi].fill_(v)

    def finish_capture(self, tables=None):
        """After the capture of a step that contained step(_captured=True): upload the pointer tables of its launches."""
        src = self._tables if tables is None else tables
        ops.adam_flush_tables([v for k, v in src.items() if isinstance(k, tuple) and k and k[0] != "members"])

    @torch.no_grad()
    def step(self, closure=None, _captured=False, _tables=None, _stats=None):
        """`_tables`: a dict owned by the caller's captured graph -- every graph keeps its own device pointer tables (its own
        gradient buffers) and, under "step_size", its own device step-size scalars, so that capturing a second step on the
        same optimizer cannot redirect the first one's replays.  `_stats` = (loss, re, kl, step3, totals3) device tensors: the
        LAST launch of this step also records the step's statistics (evae.ops.step_stats_add's work); returns True through
        self._stats_done when it did."""
        tables = self._tables if _tables is None else _tables
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if shard.is_active():
            # the per-tensor norm needs the globally reduced gradient: reduce first, then normalise
            shard.allreduce_grads([p for group in self.param_groups for p in group['params']])
        self._stats_done = False
        launches = []          # (args, kwargs) of every multi-tensor launch of this step, issued below: the last one takes _stats
        for gi, group in enumerate(self.param_groups):
            beta1, beta2 = group['betas']
            by_step = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                state = self._init_state(p)
                if not _captured:
                    state['step'] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                by_step.setdefault(state['step'] if not _captured else 1, []).append((p, g, state))
            if _captured:
                tables[("members", gi)] = [p for items in by_step.values() for p, _, _ in items]
            for step, items in by_step.items():
                launches.append((([p.data for p, _, _ in items], [g for _, g, _ in items],
                                  [s['exp_avg'] for _, _, s in items],
                                  [s['exp_avg_sq'] for _, _, s in items],
                                  step, group['lr'], beta1, beta2, group['eps'], group['weight_decay']),
                                 dict(table_cache=tables.setdefault((gi, len(items), _captured), {}),
                                      step_size_dev=(tables.get("step_size") or self._graph_step_size)[gi] if _captured else None)))
        for i, (a, kw) in enumerate(launches):
            if _stats is not None and i == len(launches) - 1:
                kw["stats"] = _stats
                self._stats_done = True
            ops.adam_normgrad_step(*a, **kw)
        return loss
