"""Two ranks sharing one MI355X (gloo backend over CUDA tensors, since RCCL refuses duplicate devices):
the sharded exemplar prior (evae/shard.py + evae/fused_vae.py, shard_exemplars=True) must reproduce the
single-process training trajectory: same losses, same parameters after several AdamNormGrad steps."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, C, N, STEPS = 32, 501, 2000, 4          # odd C: shards of 251 / 250


def _run(rank, world, port, q, fused):
    global C, N
    dedup = isinstance(fused, str) and fused.endswith("dedup")
    if dedup:                    # enough draws with replacement for the per-shard distinct-row tables to switch on (r06)
        C, N = 5001, 4000
        fused = True if fused == "dedup" else fused[:-len("_dedup")]
        if world == 1:           # the single process encodes EVERY draw, as the reference does (models/BaseModel.py:243-254)
            os.environ["EVAE_DEDUP_EAGER"] = "0"
    for p in (os.path.join(ROOT, "exemplar-vae_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import evae_oracle as orc
    import golden_inputs as gi
    import smoke_case
    from utils.optimizer import AdamNormGrad
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        data = gi.binary_images(5, N)
        dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
        cache = None
        if fused == "approximate":       # the cache + top-K prior with the candidate list sharded (BASELINE.json configs[4] style)
            args = smoke_case.vae_args(number_components=C, training_set_size=N, batch_size=B, shard_exemplars=world > 1,
                                       approximate_prior=True, approximate_k=7)
            model, _ = smoke_case.build_model(torch, np, orc, args)
        elif fused == "hvae_2level":       # BASELINE.json configs[3]: the hierarchical model over a sharded exemplar set
            from utils.utils import importing_model
            args = smoke_case.vae_args(model_name="hvae_2level", number_components=C, training_set_size=N, batch_size=B,
                                       shard_exemplars=world > 1)
            torch.manual_seed(5)
            model = importing_model(args)(args).cuda()
        else:
            args = smoke_case.vae_args(number_components=C, training_set_size=N, batch_size=B, shard_exemplars=world > 1)
            model, _ = smoke_case.build_model(torch, np, orc, args)
            model._use_fused = fused
        model.train()
        opt = AdamNormGrad(model.parameters(), lr=5e-4)
        torch.manual_seed(11); torch.cuda.manual_seed(11)        # identical eps / exemplar draws on every rank
        losses = []
        if fused == "approximate":
            with torch.no_grad():
                cache = tuple(model.cache_z(dataset))
        for it in range(STEPS):
            xb = torch.from_numpy(data[it * B:(it + 1) * B]).cuda()
            ib = torch.arange(it * B, (it + 1) * B).reshape(-1, 1).cuda()
            opt.zero_grad()
            loss, RE, KL = model.calculate_loss((xb, ib), 0.7, average=True, dataset=dataset, cache=cache)
            loss.backward()
            opt.step()
            losses.append(loss.item())
            if cache is not None:
                cache = (cache[0].detach(), cache[1].detach())
        out = {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}
        if cache is not None:
            out["__cache__"] = cache[0].cpu().numpy()
        if dedup and world > 1:  # the tables were in use: this rank's draws name fewer distinct rows than the 92 % bar
            lo, hi = __import__("evae.shard", fromlist=["bounds"]).bounds(C)
            dd = model._dedup_draws(torch.randint(0, N, (hi - lo,)))
            assert dd is not None and dd[0].numel() < 0.92 * (hi - lo)
        q.put((rank, losses, out))
    finally:
        if world > 1:
            dist.destroy_process_group()


def _spawn(world, fused):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000) + {True: 7, False: 0, "approximate": 19, "dedup": 23, "hvae_2level_dedup": 29}.get(fused, 13)
    procs = [ctx.Process(target=_run, args=(r, world, port, q, fused)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("fused", [True, False, "hvae_2level", "approximate", "dedup", "hvae_2level_dedup"])
def test_two_rank_sharded_training_matches_single(fused):
    single = _spawn(1, fused)[0]
    double = _spawn(2, fused)
    # the replicas must stay bit-identical to each other: the fused step only all-reduces the encoder's gradients (the other
    # tensors' gradients come from the replicated batch rows alone), which is only right while every rank computes the same numbers
    for kk in double[0][2]:
        assert np.array_equal(double[0][2][kk], double[1][2][kk]), kk

    def rel(a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    for rank, losses, params in double:
        assert rel(losses, single[1]) < 1e-5, (rank, losses, single[1])
        for k in params:
            assert rel(params[k], single[2][k]) < 3e-4, (rank, k)   # summation order differs between 1 and 2 shards; Adam on normalised gradients moves every weight by ~lr per step whatever the gradient magnitude, so rounding-level differences in near-zero gradient entries show up at the 1e-4 level (the losses above agree to 1e-5)


# ---- data-parallel batches over sharded exemplars (args.shard_batch): 2 ranks x B images == 1 process x 2B images ----
def _run_dp(rank, world, port, q):
    for p in (os.path.join(ROOT, "exemplar-vae_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import evae_oracle as orc
    import golden_inputs as gi
    import smoke_case
    from utils.optimizer import AdamNormGrad
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        data = gi.binary_images(5, N)
        dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(N).reshape(-1, 1), torch.zeros(N))
        GB = 2 * B                                                   # global batch
        lb = GB // world                                             # this process's share
        args = smoke_case.vae_args(number_components=C, training_set_size=N, batch_size=lb, shard_exemplars=world > 1,
                                   shard_batch=world > 1)
        model, _ = smoke_case.build_model(torch, np, orc, args)
        model.train()
        opt = AdamNormGrad(model.parameters(), lr=5e-4)
        torch.manual_seed(11)                                        # identical exemplar draws in every process
        losses = []
        for it in range(STEPS):
            lo = it * GB + rank * lb
            xb = torch.from_numpy(data[lo:lo + lb]).cuda()
            ib = torch.arange(lo, lo + lb).reshape(-1, 1).cuda()
            eps = np.random.RandomState(100 + it).standard_normal((GB, 40)).astype(np.float32)[rank * lb:(rank + 1) * lb]
            model._draw_eps = lambda like, e=eps: torch.from_numpy(e).to(like.device)
            opt.zero_grad()
            loss, RE, KL = model.calculate_loss((xb, ib), 0.7, average=True, dataset=dataset)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        out = {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}
        if world > 1:
            # the fused step's gradients must be views of ONE buffer, so that their averaging is a single in-place collective
            from evae import shard
            base = shard._FLAT[0].untyped_storage().data_ptr()
            out["__flat__"] = np.asarray([float(all(p.grad.untyped_storage().data_ptr() == base for p in model.parameters()))])
        q.put((rank, losses, out))
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_two_rank_data_parallel_batches_match_single_process_global_batch():
    ctx = mp.get_context("spawn")

    def spawn(world):
        q = ctx.Queue()
        port = 29900 + (os.getpid() % 1000)
        procs = [ctx.Process(target=_run_dp, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        return res
    single = spawn(1)[0]
    double = spawn(2)

    def rel(a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    mean_losses = np.mean([np.asarray(l) for _, l, _ in double], axis=0)   # global-batch mean = mean of the rank means
    assert rel(mean_losses, single[1]) < 1e-5, (mean_losses, single[1])
    for rank, _, params in double:
        assert params.pop("__flat__")[0] == 1.0
        for k in params:
            assert rel(params[k], single[2][k]) < 3e-4, (rank, k)


# ---- evaluation over a row-sharded latent cache (SURVEY 8e cached / eval mode; reference utils/evaluation.py:15-41,56-103) ----
def _run_eval(rank, world, port, q, model_name):
    for p in (os.path.join(ROOT, "exemplar-vae_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import contextlib, io
    import evae_oracle as orc
    import golden_inputs as gi
    import smoke_case
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from evae import shard
        from utils import evaluation, knn_on_latent
        from utils.utils import importing_model
        Ntr, Nte, S = 2003, 24, 200                       # odd N: row blocks of 1002 / 1001
        data = gi.binary_images(5, Ntr)
        test = gi.binary_images(6, Nte)
        dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(Ntr).reshape(-1, 1), torch.arange(Ntr) % 10)
        test_ds = torch.utils.data.TensorDataset(torch.from_numpy(test), torch.arange(Nte) % 10)
        loader = torch.utils.data.DataLoader(test_ds, batch_size=8)
        args = smoke_case.vae_args(model_name=model_name, number_components=500, training_set_size=Ntr, batch_size=8,
                                   shard_exemplars=world > 1)
        torch.manual_seed(5); torch.cuda.manual_seed(5)
        model = importing_model(args)(args).cuda()
        model.eval()
        torch.manual_seed(21 + 100 * rank)                # the ranks' own generators disagree: rank 0's broadcast seed decides
        shard._NOISE_CALLS[0] = 0
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            emb = evaluation.load_all_pseudo_input(args, model, dataset)
            if world > 1:
                lo, hi = shard.bounds(Ntr)
                assert emb.sharded_total == Ntr and emb[0].shape[0] == hi - lo
                elbo = evaluation.evaluate_loss(args, model, loader, exemplars_embedding=emb)
                ll = evaluation.calculate_likelihood(args, model, loader, S=S, exemplars_embedding=emb)
            else:                                         # one process: the same generators the sharded loops build
                model._eps_generator = shard.synced_generator(torch.device("cuda"))
                elbo = evaluation.evaluate_loss(args, model, loader, exemplars_embedding=emb)
                model._eps_generator = shard.synced_generator(torch.device("cuda"))
                ll = evaluation.calculate_likelihood(args, model, loader, S=S, exemplars_embedding=emb)
                model._eps_generator = None
            knn = {str(k): [] for k in (3, 7)}
            tl = torch.utils.data.DataLoader(dataset, batch_size=8)
            knn_on_latent.report_knn_on_latent(tl, loader, loader, model, "", knn, args, val=True)
            zq = model.q_z(torch.from_numpy(test).cuda(), prior=True)[0]
            zr = knn_on_latent._posterior_means(model, torch.from_numpy(data).cuda(), 8, sharded=world > 1)
            nn_idx = knn_on_latent.find_nearest_neighbors(zq, zr, None).cpu().numpy()
        q.put((rank, elbo, ll, nn_idx, knn))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("model_name", ["vae", "hvae_2level"])
def test_two_rank_sharded_evaluation_matches_single(model_name):
    """load_all_pseudo_input -> cache_z_shard, evaluate_loss, calculate_likelihood (IWAE) and the latent kNN with the cache split
    over two ranks == the single-process values on the same noise (ELBO / LogL to 1e-6, neighbour lists bit-exact)."""
    ctx = mp.get_context("spawn")

    def spawn(world):
        qq = ctx.Queue()
        port = 30300 + (os.getpid() % 1000) + (3 if model_name == "vae" else 11)
        procs = [ctx.Process(target=_run_eval, args=(r, world, port, qq, model_name)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted([qq.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        return res
    single = spawn(1)[0]
    double = spawn(2)
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-30)
    for rank, elbo, ll, nn_idx, knn in double:
        for a, b in zip(elbo, single[1]):
            assert rel(a, b) < 1e-6, (rank, elbo, single[1])
        assert rel(ll, single[2]) < 1e-6, (rank, ll, single[2])
        assert np.array_equal(nn_idx, single[3])
        assert knn == single[4] and all(np.isfinite(v).all() for v in knn.values())


# ---- bench.py itself, two ranks on one device (the multi-GPU leg the driver launches at round end, as a test) ----
def _bench(args, world):
    """Run bench.py the way the driver does for --gpus N (torch.distributed.run, one rank per 'GPU'), both ranks on device 0
    over gloo (EVAE_BENCH_ONE_DEVICE: RCCL refuses duplicate devices); returns rank 0's JSON line."""
    import json
    import subprocess
    env = dict(os.environ, EVAE_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + args
    if world > 1:
        port = 29900 + (os.getpid() % 500)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + cmd[1:]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    return json.loads(lines[-1])


COMMON = ["--steps", "10", "--warmup", "2", "--exemplars", "2000", "--iwae-images", "0", "--cpu-baseline-steps", "0", "--probe-steps", "0",
          "--no-ramp", "--no-graph", "--no-dp-line"]


def test_bench_replica_mode_two_ranks_match_one_process():
    """`bench.py --gpus 2` (replicated batch, exemplars sharded): exactly three collectives ISSUED per step (counted at
    torch.distributed's entry points), both ranks contributed, the mean loss of 12 steps equals the single-process run's to
    1e-6, and the replicas' parameters are bit-identical after them (evae.shard.check_replicas)."""
    one = _bench(COMMON, 1)
    two = _bench(COMMON, 2)
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2
    assert two["collectives"]["count"] == 3 and two["collectives"]["issued_in_one_step"]["count"] == 3, two["collectives"]
    assert two["steps_in_mean_loss"] == one["steps_in_mean_loss"]
    assert abs(two["mean_loss_f64"] - one["mean_loss_f64"]) <= 1e-6 * abs(one["mean_loss_f64"]), (two["mean_loss_f64"], one["mean_loss_f64"])
    assert two["replicas_identical"] is True


def test_bench_data_parallel_mode_two_ranks():
    """`bench.py --gpus 2 --parallel dp`: own batch per rank, six collectives issued per step, weak scaling line."""
    two = _bench(COMMON + ["--parallel", "dp"], 2)
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["scaling"] == "weak"
    assert two["config"]["global_batch"] == 200
    assert two["collectives"]["issued_in_one_step"]["count"] == two["collectives"]["count"] == 6, two["collectives"]
    assert np.isfinite(two["mean_loss_f64"])


def test_bench_iwae_two_ranks_match_one_process():
    """`bench.py --gpus 2 --config iwae`: the latent cache row-sharded over two ranks; the two runs draw DIFFERENT importance
    samples (the sharded evaluator seeds a generator all ranks share), so the estimates agree to Monte-Carlo noise only -- the
    strict comparison with injected noise is test_two_rank_sharded_evaluation_matches_single."""
    flags = ["--config", "iwae", "--iwae-images", "8", "--steps", "50", "--cpu-baseline-steps", "0"]
    one = _bench(flags, 1)
    two = _bench(flags, 2)
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2
    assert abs(two["neg_log_px"] - one["neg_log_px"]) <= 0.02 * abs(one["neg_log_px"]), (two["neg_log_px"], one["neg_log_px"])


# ---- the RCCL branch itself, executed (VERDICT r05 missing #3): a process group of ONE rank on backend "nccl" with the sharded path
#      forced on (EVAE_SHARD_FORCE) -- init, the three collectives of a step issued eagerly in the warm-up steps, then CAPTURED into the
#      step's hipGraph (thread-local capture mode, evae/graph.py) and replayed.  Two ranks need two GPUs; the driver's 8-GPU run is the
#      only place those exist.
def _run_rccl_one(q, model_name, port, sharded):
    for p in (os.path.join(ROOT, "exemplar-vae_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    if sharded:
        os.environ["EVAE_SHARD_FORCE"] = "1"
    staged = model_name.endswith("_staged")
    if staged:           # the control block through the upload stream and the first launch's hand-over, as at 12 500 exemplars per rank
        os.environ["EVAE_CTL_DIRECT"] = "0"
        model_name = model_name[:-len("_staged")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import evae_oracle as orc
    import golden_inputs as gi
    import smoke_case
    from evae import shard
    from evae.graph import GraphedTrainStep
    from utils.optimizer import AdamNormGrad
    from utils.utils import importing_model
    torch.cuda.set_device(0)
    seen = []
    if sharded:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        assert shard.is_active() and dist.get_backend() == "nccl"
        for name in ("all_reduce", "all_gather_into_tensor"):
            def wrap(*a, _f=getattr(dist, name), _n=name, **kw):
                seen.append((_n, bool(torch.cuda.is_current_stream_capturing())))
                return _f(*a, **kw)
            setattr(dist, name, wrap)
    try:
        Bq, Cq, Nq = 32, 5001, 4000
        data = gi.binary_images(5, Nq)
        dataset = torch.utils.data.TensorDataset(torch.from_numpy(data), torch.arange(Nq).reshape(-1, 1), torch.zeros(Nq))
        args = smoke_case.vae_args(model_name=model_name, number_components=Cq, training_set_size=Nq, batch_size=Bq,
                                   shard_exemplars=sharded)
        if model_name == "vae":
            model, _ = smoke_case.build_model(torch, np, orc, args)
        else:
            torch.manual_seed(5)
            model = importing_model(args)(args).cuda()
        model.train()
        opt = AdamNormGrad(model.parameters(), lr=5e-4)
        torch.manual_seed(11); torch.cuda.manual_seed(11)
        runner = GraphedTrainStep(model, opt, dataset, Bq, False)
        losses = []
        for it in range(8):
            xb = torch.from_numpy(data[it * Bq:(it + 1) * Bq])
            ib = torch.arange(it * Bq, (it + 1) * Bq).reshape(-1, 1)
            losses.append(runner(xb, ib, 0.7)[0].item())
        torch.cuda.synchronize()
        info = {"captured": runner.graph is not None, "failed": bool(runner.failed), "dedup": runner.dedup is not None,
                "sharded": bool(model._sharded()), "seen": seen, "handover": bool(runner._handover and runner.by_index),
                "parity": (int(runner._ho_state[0]), runner._calls & 1), "counter": (int(runner.ctl[runner._o_seed + 1]), runner._calls - 1)}
        q.put((losses, {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}, info))
    finally:
        if sharded:
            dist.destroy_process_group()


@pytest.mark.parametrize("model_name", ["vae", "hvae_2level", "vae_staged"])
def test_rccl_group_of_one_rank_captures_and_replays_the_sharded_step(model_name):
    ctx = mp.get_context("spawn")
    out = []
    for sharded in (False, True):
        q = ctx.Queue()
        port = 30700 + (os.getpid() % 1000) + {"vae": 5, "vae_staged": 11}.get(model_name, 17)
        p = ctx.Process(target=_run_rccl_one, args=(q, model_name, port, sharded))
        p.start()
        try:
            out.append(q.get(timeout=420))
        finally:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        assert p.exitcode == 0
    (l0, p0, i0), (l1, p1, i1) = out
    assert i0["captured"] and not i0["failed"] and not i0["sharded"]
    assert i1["sharded"] and i1["captured"] and not i1["failed"], i1
    assert i1["dedup"], i1                                  # the per-shard distinct-row tables were on (5 001 draws from 4 000 rows)
    if model_name == "vae_staged":                          # (r06) the sharded step's first launch handed the control block over
        for i_ in (i0, i1):
            assert i_["handover"] and i_["parity"][0] == i_["parity"][1] and i_["counter"][0] == i_["counter"][1], i_
    else:
        assert not i1["handover"], i1
    names = [n for n, _ in i1["seen"]]
    # every step issues the partials' all-gather, the (dz, dlogvar) all-reduce and the gradient all-reduce; at least one step's worth
    # of them was issued INSIDE the capture (and then replayed five times)
    assert names.count("all_gather_into_tensor") >= 4 and names.count("all_reduce") >= 8, names
    captured = [n for n, c in i1["seen"] if c]
    assert captured.count("all_gather_into_tensor") >= 1 and captured.count("all_reduce") >= 2, i1["seen"]

    def rel(a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert np.isfinite(l1).all() and rel(l1, l0) < 1e-5, (l1, l0)
    for k in p0:
        assert rel(p1[k], p0[k]) < 3e-4, k
