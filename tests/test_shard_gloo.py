"""world_size-2 gloo tests (CPU) of the multi-GPU host logic in evae/shard.py: uneven shard bounds, the
one-shot all-gather of packed partial log-sum-exps, the top-k candidate gather and the flat gradient
all-reduce.  The partials themselves come from the oracle here (the HIP kernels need a GPU; the same merge
is exercised on a GPU with logical shards in tests/test_gpu_kernels.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "exemplar-vae_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import evae_oracle as orc
        import golden_inputs as gi
        from evae import shard
        B, C, zd = 24, 1001, 40                       # odd C: shards of 501 / 500
        z, c = gi.clustered_latents(3, B, C, zd)
        zi, ci = gi.mask_indices(4, B, C, 300)
        lv = np.full(zd, -0.8, np.float32)
        lo, hi = shard.bounds(C)
        assert (lo, hi) == ((0, 501) if rank == 0 else (501, 1001))
        assert shard.is_active() and shard.world() == (rank, world)
        # 1. partial log-sum-exp exchange
        m, s, n = orc.prior_partials(z, zi, c[lo:hi], lv, ci[lo:hi], True)
        gm, gs, gn = shard.gather_partials(torch.from_numpy(m), torch.from_numpy(s), torch.from_numpy(n))
        assert gm.shape == (world, B)
        merged = orc.prior_merge(gm.numpy(), gs.numpy(), gn.numpy(), C)
        full = orc.log_p_z(z, zi, c, lv[None], ci, test=False)
        err = float(np.abs(merged - full).max() / np.abs(full).max())
        # 2. top-k candidates with global indices
        d = orc.pairwise_distance(z, c[lo:hi])
        v, i = orc.topk_smallest(d, 10)
        gv, gi_ = shard.gather_topk(torch.from_numpy(v), torch.from_numpy(i + lo))
        allv = gv.permute(1, 0, 2).reshape(B, -1).numpy(); alli = gi_.permute(1, 0, 2).reshape(B, -1).numpy()
        order = np.lexsort((alli, allv), axis=1)[:, :10]
        got = np.take_along_axis(alli, order, axis=1)
        want = orc.topk_smallest(orc.pairwise_distance(z, c), 10)[1]
        topk_ok = bool(np.array_equal(got, want))
        # 3. flat gradient all-reduce (mean)
        ps = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2))]
        ps[0].grad = torch.full((5, 3), float(rank + 1)); ps[1].grad = torch.arange(7.) * (rank + 1)
        shard.allreduce_grads(ps)
        grads_ok = bool(torch.allclose(ps[0].grad, torch.full((5, 3), 1.5)) and
                        torch.allclose(ps[1].grad, torch.arange(7.) * 1.5) and ps[2].grad is None)
        q.put((rank, err, topk_ok, grads_ok))
    finally:
        dist.destroy_process_group()


def test_bounds_cover_and_are_contiguous():
    sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
    from evae import shard
    for n, R in ((25000, 8), (11500, 8), (3, 8), (0, 4), (7, 1)):
        cuts = [shard.bounds(n, r, R) for r in range(R)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1
    assert shard.bounds(11500, 0, 8) == (0, 1438) and shard.bounds(11500, 7, 8) == (10063, 11500)


def test_two_rank_exchange_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, topk_ok, grads_ok in res:
        assert err < 2e-6, (rank, err)
        assert topk_ok and grads_ok
