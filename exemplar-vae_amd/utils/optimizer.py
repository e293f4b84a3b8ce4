"""AdamNormGrad with the reference's constructor and state layout (reference utils/optimizer.py:7-80;
state keys 'step', 'exp_avg', 'exp_avg_sq'), one fused multi-tensor HIP launch pair per step."""
import torch
from torch.optim import Optimizer

from evae import ops, shard


class AdamNormGrad(Optimizer):
    """Adam on per-tensor L2-normalised gradients: g <- g / (||g||_2 + 1e-7)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if shard.is_active():
            # the per-tensor norm needs the globally reduced gradient: reduce first, then normalise
            shard.allreduce_grads([p for group in self.param_groups for p in group['params']])
        for gi, group in enumerate(self.param_groups):
            beta1, beta2 = group['betas']
            by_step = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                state['step'] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                by_step.setdefault(state['step'], []).append((p, g, state))
            for step, items in by_step.items():
                ops.adam_normgrad_step([p.data for p, _, _ in items], [g for _, g, _ in items],
                                       [s['exp_avg'] for _, _, s in items],
                                       [s['exp_avg_sq'] for _, _, s in items],
                                       step, group['lr'], beta1, beta2, group['eps'], group['weight_decay'],
                                       table_cache=self._tables.setdefault((gi, len(items)), {}))
        return loss
