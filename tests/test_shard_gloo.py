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


def _worker4(rank, world, port, q):
    for p in (os.path.join(ROOT, "exemplar-vae_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import evae_oracle as orc
        import golden_inputs as gi
        from evae import ops, shard
        T = torch.from_numpy

        # the oracle behind the autograd node's three kernels: the collective logic of ShardedPriorLogP runs unchanged
        def fwd(z, c, lv, zi, ci):
            masked = zi is not None and ci is not None
            m, s, n = orc.prior_partials(z.numpy(), None if zi is None else zi.numpy(), c.numpy(), lv.numpy(),
                                         None if ci is None else ci.numpy(), masked)
            return T(m), T(s), T(n), None

        def merge(m, s, n, c_total):
            lp = orc.prior_merge(m.numpy(), s.numpy(), n.numpy(), c_total)
            mm = m.numpy().max(0)
            lse = mm + np.log((s.numpy() * np.exp(m.numpy() - mm)).sum(0))
            return T(lp.astype(np.float32)), T(lse.astype(np.float32))

        def bwd(z, c, lv, zi, ci, lse, g):
            # per-shard gradient with the GLOBAL lse: w_ij = exp(p_ij - lse_i)
            z64, c64, lv64 = z.numpy().astype(np.float64), c.numpy().astype(np.float64), lv.numpy().astype(np.float64)
            Cn = len(c64)
            dz = np.zeros_like(z64); dc = np.zeros_like(c64); dlv = np.zeros_like(lv64)
            if Cn > 0:
                diff = z64[:, None, :] - c64[None, :, :]
                p = -0.5 * (lv64 + np.log(2 * np.pi)).sum() - 0.5 * (diff ** 2 * np.exp(-lv64)).sum(-1)
                if zi is not None and ci is not None:
                    p[zi.numpy().reshape(-1, 1) == ci.numpy().reshape(1, -1)] = -np.inf
                w = np.exp(p - lse.numpy().astype(np.float64)[:, None]) * g.numpy().astype(np.float64)[:, None]
                dz = -(w[:, :, None] * diff * np.exp(-lv64)).sum(1)
                dc = (w[:, :, None] * diff * np.exp(-lv64)).sum(0)
                dlv = (w[:, :, None] * (-0.5 + 0.5 * diff ** 2 * np.exp(-lv64))).sum((0, 1))
            return T(dz.astype(np.float32)), T(dc.astype(np.float32)), T(dlv.astype(np.float32))
        # the test seam lives HERE, not in the product: the device-kernel wrappers of evae.ops are replaced in this worker process
        ops.prior_lse_fwd, ops.prior_merge, ops.prior_lse_bwd = fwd, merge, bwd
        out = []
        for C in (3, 9):                 # C = 3 over 4 ranks: shards of 1, 1, 1 and an EMPTY one; C = 9: 3, 2, 2, 2
            B, zd = 6, 5
            z, c = gi.clustered_latents(40 + C, B, C, zd)
            zi, ci = gi.mask_indices(41, B, C, 50)
            ci[0] = zi[0, 0]                                   # one leave-one-out hit
            lv = np.linspace(-0.9, -0.2, zd).astype(np.float32)
            gout = np.random.RandomState(3).standard_normal(B).astype(np.float32)
            lo, hi = shard.bounds(C)
            zt = T(z).requires_grad_(True); ct = T(c[lo:hi].copy()).requires_grad_(True); lt = T(lv).requires_grad_(True)
            lp = shard.ShardedPriorLogP.apply(zt, ct, lt, T(zi), T(ci[lo:hi].copy()), C)
            (lp * T(gout)).sum().backward()
            dz, dc, dlv, _ = orc.prior_grads(z.astype(np.float64), zi, c.astype(np.float64), lv.astype(np.float64), ci, True,
                                             gout.astype(np.float64))
            full = orc.log_p_z(z, zi, c, lv[None], ci, test=False)
            e_lp = float(np.abs(lp.detach().numpy() - full).max() / np.abs(full).max())
            e_dz = float(np.abs(zt.grad.numpy() - dz).max() / max(np.abs(dz).max(), 1e-30))
            e_dlv = float(np.abs(lt.grad.numpy() - dlv).max() / max(np.abs(dlv).max(), 1e-30))
            # dcentres comes back scaled by the world size (the parameter-gradient all-reduce averages over ranks)
            e_dc = 0.0 if hi == lo else float(np.abs(ct.grad.numpy() / world - dc[lo:hi]).max() / max(np.abs(dc).max(), 1e-30))
            out.append((C, hi - lo, e_lp, e_dz, e_dlv, e_dc))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_four_rank_sharded_prior_with_uneven_and_empty_shards_gloo():
    """ShardedPriorLogP forward and backward over FOUR gloo ranks with C = 3 exemplars (one rank holds an empty shard) and
    C = 9 (uneven shards): the all-gather of packed partial log-sum-exps, the merge, the sum-all-reduce of dz / dlogvar and
    the world-size scaling of dcentres, with the oracle standing in for the three device kernels."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = {}
    for rank, out in res:
        for C, n_local, e_lp, e_dz, e_dlv, e_dc in out:
            sizes.setdefault(C, {})[rank] = n_local
            assert e_lp < 2e-6 and e_dz < 2e-5 and e_dlv < 2e-5 and e_dc < 2e-5, (rank, C, e_lp, e_dz, e_dlv, e_dc)
    assert sorted(sizes[3].values()) == [0, 1, 1, 1] and sorted(sizes[9].values()) == [2, 2, 2, 3]
