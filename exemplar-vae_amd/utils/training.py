"""One training epoch with the reference's signature (reference utils/training.py:5-51).

Differences that do not change results: losses are accumulated on the device and read back once per
epoch instead of three .item() syncs per step (training.py:42-44), and the exemplar images never leave
HBM (the model keeps a device-resident copy of dataset.tensors[0])."""
import torch

from evae import hostcpu, shard
from evae import ops as _ops
from evae.graph import GraphedTrainStep


def set_beta(args, epoch):
    if args.warmup == 0:
        return 1.
    return min(1. * epoch / args.warmup, 1.)


def train_one_epoch(epoch, args, train_loader, model, optimizer):
    hostcpu.limit_host_threads()          # once per process: this rank's share of the host cores (evae/hostcpu.py)
    model.train()
    beta = set_beta(args, epoch)
    print('beta: {}'.format(beta))
    if args.approximate_prior is True:
        with torch.no_grad():
            cache = tuple(model.cache_z(train_loader.dataset))
    else:
        cache = None
    totals = None
    graphed = _graphed_step(args, model, optimizer, train_loader)
    if graphed is not None:
        graphed.reset_totals()
        if cache is not None:
            cache = graphed.set_cache(cache)      # the captured launches refresh the cache in place, in static buffers
    nstep = 0
    for data, indices, target in train_loader:
        nstep += 1
        if graphed is None and model._sharded() and nstep % shard.REPLICA_CHECK_EVERY == 0:
            shard.check_replicas(model.parameters())      # replica mode's run-time guard (evae/shard.py)
        if graphed is not None and len(data) == graphed.B:
            graphed(data, indices, beta)     # one hipGraph launch per step; sums accumulate in graphed.totals
            continue
        data, indices = data.to(args.device), indices.to(args.device)
        x = torch.bernoulli(data) if args.dynamic_binarization else data
        optimizer.zero_grad()
        loss, RE, KL = model.calculate_loss((x, indices), beta, average=True, cache=cache,
                                            dataset=train_loader.dataset)
        with _ops.deferred_wgrads(loss):         # thin layers' weight gradients grouped behind the backward pass (evae/ops.py)
            loss.backward()
        optimizer.step()
        with torch.no_grad():
            step_vals = torch.stack((loss.detach(), -RE.detach(), KL.detach()))
            totals = step_vals if totals is None else totals + step_vals
            if cache is not None:
                cache = (cache[0].detach(), cache[1].detach())
    if graphed is not None:
        totals = graphed.totals.clone() if totals is None else totals + graphed.totals
    _ops.prior_train_check()       # the one-launch prior's co-residency guard (reads back; the .tolist() below synchronises anyway)
    train_loss, train_re, train_kl = (totals / len(train_loader)).tolist()
    return train_loss, train_re, train_kl


def _graphed_step(args, model, optimizer, train_loader):
    """The captured-step runner for this (model, optimizer, dataset), or None when the configuration is not
    the graph-capturable one (fused `vae` exact-prior path on a GPU) or args.use_hip_graph is False."""
    if not getattr(args, 'use_hip_graph', True) or not str(args.device).startswith('cuda'):
        return None
    a = model.args
    # capturable: the exact exemplar prior, and the approximate (cache + top-k) one on a single device with the leave-one-out
    # mask on -- its exemplar union lives in a fixed list of B * k slots with masked repeats instead of a data-dependent
    # `unique` (models/BaseModel.py::get_approximate_nearest_exemplars; dense encoders only: a convolutional encoder would
    # re-encode B * k images where `unique` leaves far fewer).  The `vae` model runs the one-node fused step inside
    # the graph (exact prior), every other case its modular autograd path
    ok = a.prior == 'exemplar_prior' and (a.approximate_prior is False or
                                          (a.no_mask is False and not model._sharded() and not model._is_conv()))
    if not ok:
        return None
    # the runner keeps the optimizer and the dataset alive, so their ids cannot be recycled while it is cached
    key = (id(optimizer), id(train_loader.dataset), train_loader.batch_size, bool(args.dynamic_binarization))
    cache = model.__dict__.setdefault('_graphed_steps', {})
    if key not in cache:
        cache[key] = GraphedTrainStep(model, optimizer, train_loader.dataset, train_loader.batch_size,
                                      args.dynamic_binarization)
    return cache[key]
