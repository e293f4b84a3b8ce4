"""gloo tests (CPU, world size 2 and 4) of the EVALUATION path over a row-sharded latent cache (SURVEY.md 8e "cached / eval
mode: contiguous row blocks of the [N x z] cache"; reference utils/evaluation.py:15-41,56-59,72-103, utils/knn_on_latent.py:4-9):

  * utils.evaluation.load_all_pseudo_input -> models.BaseModel.cache_z_shard (this rank's rows only, uneven blocks),
  * utils.evaluation.evaluate_loss / calculate_likelihood through BaseModel.log_p_z -> evae.shard.ShardedPriorLogP (one
    all-gather of the packed [3 x S] partial log-sum-exps per call) with the SAME importance samples on every rank
    (shard.replicated_noise: ranks are seeded DIFFERENTLY here, rank 0's broadcast seed must win),
  * utils.knn_on_latent.find_nearest_neighbors on a row block (local top-20, all-gather, exact merge),

each equal to the un-sharded computation on the same noise.  The model is a test double whose layers are the numpy oracle's
(the HIP kernels need a GPU: tests/test_gpu_sharded.py runs the same comparison through the real model, two ranks on one
device); the methods under test -- cache_z_shard, log_p_z, importance_sample_losses, _draw_eps -- are BaseModel's own."""
import os
import sys
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_TRAIN, N_TEST, D, H, Z, S = 203, 10, 64, 32, 8, 40      # 203 rows over 4 ranks: blocks of 51, 51, 51, 50


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "exemplar-vae_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import contextlib, io
        import evae_oracle as orc
        from evae import ops, shard
        from models.BaseModel import BaseModel
        from utils import evaluation, knn_on_latent
        T = torch.from_numpy

        # ---- the oracle behind the device kernels of the prior and the top-K ----
        def fwd(z, c, lv, zi, ci):
            m, s, n = orc.prior_partials(z.numpy(), None, c.numpy(), lv.numpy(), None, False)
            return T(m), T(s), T(n), None

        def merge(m, s, n, c_total):
            lp = orc.prior_merge(m.numpy(), s.numpy(), n.numpy(), c_total)
            return T(lp.astype(np.float32)), None

        def topk(qq, cache, k, sqrt, base):
            if cache.shape[0] == 0:
                return (torch.full((len(qq), k), -1, dtype=torch.int64), torch.full((len(qq), k), float("inf")))
            dd = orc.pairdist_direct_f64(qq.numpy(), cache.numpy())
            dd = (np.sqrt(dd) if sqrt else dd).astype(np.float32)
            kk = min(k, cache.shape[0])
            v, i = orc.topk_smallest(dd, kk)
            vv = np.full((len(qq), k), np.inf, np.float32); ii = np.full((len(qq), k), -1, np.int64)
            vv[:, :kk] = v; ii[:, :kk] = i + base
            return T(ii), T(vv)

        def topk_merge(v, i):
            R, B, k = v.shape
            allv = v.permute(1, 0, 2).reshape(B, -1).numpy(); alli = i.permute(1, 0, 2).reshape(B, -1).numpy()
            order = np.lexsort((alli, allv), axis=1)[:, :k]
            return T(np.take_along_axis(alli, order, axis=1)), T(np.take_along_axis(allv, order, axis=1))
        # the test seam lives HERE, not in the product: the device-kernel wrappers of evae.ops are replaced in this worker process
        ops.prior_lse_fwd, ops.prior_merge = fwd, merge
        ops.pairdist_topk = lambda qq, cache, k, sqrt=False, index_base=0, **kw: topk(qq, cache, k, sqrt, index_base)
        ops.topk_merge = topk_merge

        class FullPrior:          # stands in for ops.PriorLogP (the un-sharded device kernel)
            @staticmethod
            def apply(z, centers, lv_row, zi, ci):
                lp = orc.log_p_z(z.numpy(), None, centers.numpy(), lv_row.numpy()[None], None, test=True)
                return T(lp.astype(np.float32))
        ops.PriorLogP = FullPrior

        class Double:
            """`vae` with the oracle's layers; everything the evaluation loops call."""
            cache_z_shard = BaseModel.cache_z_shard
            log_p_z = BaseModel.log_p_z
            importance_sample_losses = BaseModel.importance_sample_losses
            _draw_eps = BaseModel._draw_eps

            def __init__(self):
                self.p = orc.vae_init_params(np.random.RandomState(123), D=D, H=H, Z=Z)
                self.args = Namespace(prior="exemplar_prior", device="cpu", z1_size=Z, use_logit=False, shard_exemplars=True,
                                      model_name="vae", batch_size=7, no_mask=False)
                self.training = False

            def eval(self):
                self.training = False

            def resident_data(self, dataset):
                return dataset.tensors[0]

            def q_z(self, x, prior=False):
                m, lv, _ = orc.vae_q_z(self.p, x.numpy(), prior=prior)
                return T(m.astype(np.float32)), T(np.ascontiguousarray(lv, dtype=np.float32))

            def cache_z(self, dataset):
                return self.q_z(dataset.tensors[0], prior=True)

            def calculate_loss(self, x, beta=1., average=False, exemplars_embedding=None, cache=None, dataset=None):
                x, _ = x
                mu, lv = self.q_z(x)
                z = mu + self._draw_eps(mu) * torch.exp(0.5 * lv)
                x_mean, _ = orc.vae_p_x(self.p, z.numpy())
                RE = T(orc.log_bernoulli(x.numpy(), x_mean, axis=1).astype(np.float32))
                log_q = T(orc.log_normal_diag(z.numpy(), mu.numpy(), lv.numpy(), axis=1).astype(np.float32))
                KL = log_q - self.log_p_z((z, None), exemplars_embedding)
                return -RE + beta * KL, RE, KL

        rs = np.random.RandomState(5)
        train = (rs.random_sample((N_TRAIN, D)) < 0.3).astype(np.float32)
        test = (rs.random_sample((N_TEST, D)) < 0.3).astype(np.float32)
        ds_train = torch.utils.data.TensorDataset(T(train), torch.arange(N_TRAIN).reshape(-1, 1), torch.zeros(N_TRAIN))
        ds_test = torch.utils.data.TensorDataset(T(test), torch.zeros(N_TEST))
        loader = torch.utils.data.DataLoader(ds_test, batch_size=4)
        model = Double()
        torch.manual_seed(1000 + rank)                   # ranks disagree on purpose: the broadcast seed must make them agree
        with contextlib.redirect_stdout(io.StringIO()):
            # reference: the full cache on this rank, noise from the generator the sharded loops will build (same counter)
            emb_full = evaluation.load_all_pseudo_input(Namespace(prior="exemplar_prior", shard_exemplars=False), model, ds_train)
            assert not hasattr(emb_full, "sharded_total")
            shard._NOISE_CALLS[0] = 0
            model._eps_generator = shard.synced_generator(torch.device("cpu"))
            elbo_full = evaluation.evaluate_loss(model.args, model, loader, exemplars_embedding=emb_full)
            model._eps_generator = shard.synced_generator(torch.device("cpu"))
            ll_full = evaluation.calculate_likelihood(model.args, model, loader, S=S, exemplars_embedding=emb_full)
            model._eps_generator = None
            # sharded: every rank holds its row block only
            emb = evaluation.load_all_pseudo_input(model.args, model, ds_train)
            lo, hi = shard.bounds(N_TRAIN)
            assert emb.sharded_total == N_TRAIN and emb[0].shape[0] == hi - lo and int(emb[2][0]) == lo
            shard._NOISE_CALLS[0] = 0
            elbo_sh = evaluation.evaluate_loss(model.args, model, loader, exemplars_embedding=emb)
            ll_sh = evaluation.calculate_likelihood(model.args, model, loader, S=S, exemplars_embedding=emb)
            assert getattr(model, "_eps_generator", None) is None            # restored
        # kNN over the row-sharded training latents
        zq = model.q_z(T(test), prior=True)[0]
        z_block = knn_on_latent._posterior_means(model, T(train), model.args.batch_size, sharded=True)
        n_used = (N_TRAIN // model.args.batch_size) * model.args.batch_size
        nn_sh = knn_on_latent.find_nearest_neighbors(zq, z_block, None).numpy()
        z_all = model.q_z(T(train[:n_used]), prior=True)[0]
        dd = np.sqrt(orc.pairdist_direct_f64(zq.numpy(), z_all.numpy())).astype(np.float32)
        nn_full = orc.topk_smallest(dd, 20)[1]
        q.put((rank, elbo_full, elbo_sh, ll_full, ll_sh, bool(np.array_equal(nn_sh, nn_full)), hi - lo,
               int(z_block.shape[0]), int(z_block.shard_base)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_evaluation_matches_the_full_cache_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-30)
    for rank, elbo_full, elbo_sh, ll_full, ll_sh, knn_ok, n_local, n_block, base in res:
        for a, b in zip(elbo_sh, elbo_full):
            assert rel(a, b) < 1e-6, (rank, elbo_sh, elbo_full)
        assert rel(ll_sh, ll_full) < 1e-6, (rank, ll_sh, ll_full)
        assert knn_ok
    # every rank reports the same numbers (same noise everywhere), and the row blocks are the uneven contiguous split
    assert all(r[2] == res[0][2] and r[4] == res[0][4] for r in res)
    assert sum(r[6] for r in res) == N_TRAIN and max(r[6] for r in res) - min(r[6] for r in res) <= 1
    assert sum(r[7] for r in res) == (N_TRAIN // 7) * 7
