"""Sharding of the exemplar prior across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The C exemplar slots of a
step are split into R contiguous, possibly uneven shards; every rank encodes only its shard (the
dominant cost, 1/R each) and holds the full batch latents (the B-row batch path is replicated: same
weights, same eps, same CPU-generator exemplar indices on every rank).

Forward exchange: ONE all-gather of the packed per-row partials (max, sumexp, nmask) -- 3*B floats per
rank, latency-bound, so a single one-shot collective instead of a MAX-then-SUM pair -- followed by the
local merge kernel (evae_prior_merge): M = max_r m_r, lse = M + log sum_r s_r e^{m_r - M},
logprior = lse - log(C - sum_r nmask_r).  This is the all-reduce of partial log-sum-exps.

Backward: w_ij is recomputed from the GLOBAL lse; dcentres stays local, dz [B x z] and dlogvar [z] are
sum-all-reduced.  Parameter gradients are then averaged over ranks (allreduce_grads, called by
AdamNormGrad.step) -- dcentres is pre-scaled by R so that the average equals
(replicated batch-path gradient) + sum_r (shard-r exemplar-path gradient), i.e. the single-GPU gradient.

Data-parallel batches (args.shard_batch, evae/fused_vae.py): every rank trains on its OWN batch (global batch R*B,
gradients averaged as usual) while the exemplars stay sharded.  Each rank then scores the latents of ALL ranks
against its shard -- all-gather of z [R*B x z] (and the dataset indices for the leave-one-out mask), the same number
of pairs as B queries against all C exemplars -- the partials are all-gathered and every rank merges its own B
rows; backward: all-gather of (lse, upstream coefficient), shard-side backward over all R*B queries, sum-all-reduce
of dz [R*B x z] of which every rank keeps its rows.  dcentres / dlogvar are then complete sums over all queries, so
the mean all-reduce of the parameter gradients gives exactly the gradient of the global-batch mean loss (no
rescaling): tests/test_gpu_sharded.py checks 2 ranks x B against 1 process x 2B.
"""
import os

import torch
import torch.distributed as dist

from . import ops


class ShardedEmbedding(tuple):
    """(centres_local, logvar_local, indices_local) plus the global exemplar count."""

    def __new__(cls, items, total):
        obj = super().__new__(cls, items)
        obj.sharded_total = int(total)
        return obj


# EVAE_SHARD_FORCE=1: the sharded code path (collectives included) also in a process group of ONE rank -- how the one leased GPU of the
# build box executes the RCCL branch (tests/test_gpu_sharded.py::test_rccl_group_of_one_rank_*); nothing else sets it
FORCE = [os.environ.get("EVAE_SHARD_FORCE") == "1"]


def is_active():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE[0])


def world():
    return (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)


def bounds(n, rank=None, world_size=None):
    """[lo, hi) of shard `rank` when n items are split into world_size contiguous shards whose sizes
    differ by at most one (the first n % R shards get the extra item; shards may be empty)."""
    if rank is None or world_size is None:
        rank, world_size = world()
    q, r = divmod(int(n), int(world_size))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


# ---- evaluation over a sharded latent cache (SURVEY 8e "cached / eval mode: contiguous row blocks of the [N x z] cache") -----------
# Every rank scores the SAME queries (test images, importance samples) against ITS rows of the cache and the packed partials
# (max, sumexp, nmask) [3 x S] are all-gathered and merged (ShardedPriorLogP's forward; 60 KB at S = 5000).  The queries are
# sampled on every rank, so the ranks must draw the same noise: replicated_noise() gives the model a generator seeded with a
# value rank 0 broadcasts once per evaluation loop (nothing is consumed from any rank's global generators).
_NOISE_CALLS = [0]


def synced_generator(device, group=None):
    """A torch.Generator on `device` with the same seed on every rank: rank 0's torch.initial_seed() mixed with a per-process
    call counter (loops called in the same order on every rank count alike), broadcast as one int64."""
    _NOISE_CALLS[0] += 1
    seed = (int(torch.initial_seed()) * 6364136223846793005 + 1442695040888963407 * _NOISE_CALLS[0]) & 0x7FFFFFFFFFFFFFFF
    t = torch.tensor([seed], dtype=torch.int64, device=device)
    if is_active():
        dist.broadcast(t, src=0, group=group)
    g = torch.Generator(device=device)
    g.manual_seed(int(t.item()))
    return g


class replicated_noise:
    """with replicated_noise(model, embedding): ... -- while the embedding is a ShardedEmbedding, model._draw_eps draws from a
    generator that is seeded identically on every rank (models/BaseModel.py::_draw_eps); a no-op for a plain tuple, and for a
    model whose _draw_eps was replaced on the instance (tests injecting their own noise)."""

    def __init__(self, model, embedding, device=None, group=None):
        self.model = model
        self.on = getattr(embedding, 'sharded_total', None) is not None and is_active()
        self.device, self.group = device, group

    def __enter__(self):
        if self.on:
            dev = self.device if self.device is not None else self.model.args.device
            self.prev = getattr(self.model, '_eps_generator', None)
            self.model._eps_generator = synced_generator(torch.device(dev), self.group)
        return self

    def __exit__(self, *exc):
        if self.on:
            self.model._eps_generator = self.prev
        return False


def shard_rows(n, group=None):
    """[lo, hi) of this rank's contiguous block of n cache rows (uneven blocks allowed, see bounds)."""
    if not is_active():
        return 0, int(n)
    return bounds(n, dist.get_rank(group), dist.get_world_size(group))


def _all_gather_flat(t, group=None):
    """One all-gather of a contiguous tensor -> [R x *t.shape] (flat 1-D buffers: accepted by RCCL and gloo)."""
    R = dist.get_world_size(group)
    out = torch.empty(R * t.numel(), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.reshape(-1), group=group)
    return out.reshape((R,) + tuple(t.shape))


def gather_partials(m, s, n, group=None):
    """All-gather the packed [3 x B] partials of every rank -> ([R x B], [R x B], [R x B])."""
    packed = torch.stack((m, s, n)).contiguous()
    out = _all_gather_flat(packed, group)                     # [R x 3 x B]
    return out[:, 0].contiguous(), out[:, 1].contiguous(), out[:, 2].contiguous()


def gather_topk(val, idx, group=None):
    """All-gather per-shard top-k candidates ([B x k] values and GLOBAL indices) -> [R x B x k] each."""
    return _all_gather_flat(val.contiguous(), group), _all_gather_flat(idx.contiguous(), group)


def sharded_topk(q, cache_local, k, index_base, sqrt=False, group=None):
    """Exact global top-k over a row-sharded cache: local top-k, one all-gather, merge kernel."""
    if cache_local.shape[0] >= k:
        idx, val = ops.pairdist_topk(q, cache_local, k, sqrt=sqrt, index_base=index_base)
    else:                                            # shard smaller than k (or empty): pad with empties
        B = q.shape[0]
        val = torch.full((B, k), float('inf'), device=q.device)
        idx = torch.full((B, k), -1, dtype=torch.int64, device=q.device)
        n = cache_local.shape[0]
        if n > 0:
            i2, v2 = ops.pairdist_topk(q, cache_local, n, sqrt=sqrt, index_base=index_base)
            val[:, :n] = v2
            idx[:, :n] = i2
    v, i = gather_topk(val, idx, group)
    return ops.topk_merge(v, i)


class ShardedPriorLogP(torch.autograd.Function):
    """log p(z_i) under the exemplar mixture whose exemplars are sharded over the ranks of `group`."""

    @staticmethod
    def forward(ctx, z, centres_local, log_var_row, z_idx, c_idx_local, c_total, group=None):
        m, s, n, _ = ops.prior_lse_fwd(z, centres_local, log_var_row, z_idx, c_idx_local)
        gm, gs, gn = gather_partials(m, s, n, group)
        lp, lse = ops.prior_merge(gm, gs, gn, c_total)
        ctx.save_for_backward(z, centres_local, log_var_row, lse)
        ctx.misc = (z_idx, c_idx_local, group)
        return lp

    @staticmethod
    def backward(ctx, g):
        z, centres_local, log_var_row, lse = ctx.saved_tensors
        z_idx, c_idx_local, group = ctx.misc
        dz, dc, dlv = ops.prior_lse_bwd(z, centres_local, log_var_row, z_idx, c_idx_local, lse, g.contiguous())
        packed = torch.cat((dz.reshape(-1), dlv.reshape(-1)))
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)          # one 16 KB collective
        dz = packed[:dz.numel()].reshape(dz.shape)
        dlv = packed[dz.numel():].reshape(log_var_row.shape)
        R = dist.get_world_size(group)
        return dz, dc * float(R), dlv, None, None, None, None


_FLAT = [None, None]      # the flat buffer the last fused step wrote all its parameter gradients into, and the slice to reduce


def register_flat_grads(flat, lo=None, hi=None):
    """evae/fused_vae.py allocates every parameter gradient of a step as a view of ONE buffer; when autograd installs
    those views as .grad (zero_grad(set_to_none=True) before the backward), allreduce_grads reduces the buffer in place:
    one collective and one scale, no flatten / unflatten copies.
    [lo, hi): the part of the buffer that differs between the ranks.  With the batch REPLICATED (same images, same noise on
    every rank) only the encoder q(z | .) sees the rank's own exemplar shard; the decoder's and the log-variance head's
    gradients come from the batch rows alone and are the same numbers on every rank (deterministic kernels, identical inputs),
    and the prior's dlogvar has already been sum-reduced with dz -- so the collective carries 2.65 MB instead of 4.47 MB."""
    _FLAT[0] = flat
    _FLAT[1] = None if lo is None else (int(lo), int(hi))


def allreduce_grads(params, group=None):
    """Average the gradients of `params` over ranks with ONE flat all-reduce (a few MB: 4.5 MB for vae,
    9.7 MB for hvae -- one bucket, so one collective per step)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = _FLAT[0]
    if flat is not None:
        base = flat.untyped_storage().data_ptr()
        if all(g.untyped_storage().data_ptr() == base for g in grads):
            part = flat if _FLAT[1] is None else flat[_FLAT[1][0]:_FLAT[1][1]]
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
            part.div_(dist.get_world_size(group))
            return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(dist.get_world_size(group))
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


# ---- replica guard -------------------------------------------------------------------------------------------------------------
# With the batch replicated, allreduce_grads reduces only the encoder's slice of the gradient buffer: every other parameter is
# ASSUMED to receive the same bits on every rank (identical inputs, deterministic kernels, identical devices and seeds).  Nothing
# in a step would notice if that ever stopped being true -- the replicas' decoders and Adam moments would drift apart silently
# (ADVICE r03).  check_replicas is the run-time guard: a two-float fingerprint (sum and sum of squares, fp64) of the given
# parameters, MAX- and MIN-reduced in one small collective; any difference raises.  Called every REPLICA_CHECK_EVERY steps by
# the captured-step runner and the eager training loop (outside any capture: it reads back).
REPLICA_CHECK_EVERY = 64


def check_replicas(params, group=None, what="parameters"):
    if not is_active():
        return True
    ps = [p.detach() for p in params if p is not None]
    if not ps:
        return True
    f = torch.zeros(2, dtype=torch.float64, device=ps[0].device)
    for p in ps:
        d = p.double()
        f[0] += d.sum()
        f[1] += (d * d).sum()
    both = torch.cat((f, -f))
    dist.all_reduce(both, op=dist.ReduceOp.MAX, group=group)       # max(f) and -min(f) in one collective
    hi, lo = both[:2], -both[2:]
    if not bool(torch.equal(hi, lo)):
        raise RuntimeError("evae.shard: the replicas' %s differ between ranks (fingerprint range %s .. %s): replica mode needs "
                           "identical seeds, devices and deterministic kernels -- set EVAE_REDUCE_ALL=1 to all-reduce every "
                           "gradient instead" % (what, lo.tolist(), hi.tolist()))
    return True
