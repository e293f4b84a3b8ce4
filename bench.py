#!/usr/bin/env python3
"""Headline benchmark: training images/sec of `vae` + exemplar prior, dynamic_mnist-shaped synthetic
data, 25 000 exemplars, exact prior, batch 100 (BASELINE.json configs[1]).

A step is the body of the reference's training loop (utils/training.py:27-46): binarise the batch,
forward, sample + encode the C exemplars, exemplar prior, backward, AdamNormGrad -- all through the
drop-in API (models.VAE.VAE.calculate_loss -> loss.backward() -> utils.optimizer.AdamNormGrad.step),
i.e. through libevae_hip.so.  With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU,
RCCL) the C exemplars are sharded across the ranks and the per-shard partial log-sum-exps are merged
(evae/shard.py, evae/fused_vae.py).  Default (--parallel dp): every rank trains on its OWN 100-image batch --
global batch 100 N, gradients averaged, each rank scores the queries of all ranks against its exemplar shard and
the partials return to their owners -- so the per-GPU batch is fixed ("weak" scaling) while the 25 000 exemplars
in total are split N ways.  --parallel replica keeps one replicated 100-image batch ("strong" scaling).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (the fp32-MFMA GEMM behind GatedDense), algorithmic flops per
                  launch / mean launch duration measured with HIP events inside the timed region
  cpu_baseline -- the numpy oracle's train step (oracle/evae_oracle.py) timed on the host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
sys.path.insert(1, os.path.join(ROOT, "tests"))          # golden_inputs: the synthetic-data generator

import numpy as np
import torch
import torch.distributed as dist

PEAK_FP32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 MFMA
B, C, N_TRAIN, D, H, Z = 100, 25000, 50000, 784, 300, 40


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--exemplars", type=int, default=C)
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--parallel", choices=("dp", "replica"), default="dp",
                    help="with --gpus N > 1: 'dp' = every rank trains on its own 100-image batch (global batch 100 N) "
                         "against the exemplar set sharded over the ranks (weak scaling in the batch); 'replica' = the "
                         "same 100-image batch replicated on every rank, only the exemplars sharded (strong scaling)")
    ap.add_argument("--probe-warmup", type=int, default=20,
                    help="untimed eager steps in front of the probe steps (clock ramp after the host pause)")
    ap.add_argument("--probe-steps", type=int, default=20,
                    help="eager steps run after the timed region to time the dominant kernel with HIP events")
    ap.add_argument("--iwae-images", type=int, default=64,
                    help="test images for the IWAE test log p(x) leg (S=5000 samples each vs all 50 000 exemplars); 0 disables")
    ap.add_argument("--cpu-baseline-steps", type=int, default=80,
                    help="oracle steps timed for cpu_baseline (0 disables)")
    return ap.parse_args()


def model_args(device, n_exemplars, sharded, shard_batch=False):
    from argparse import Namespace
    return Namespace(prior="exemplar_prior", input_type="binary", input_size=[1, 28, 28], hidden_size=H,
                     z1_size=Z, z2_size=Z, model_name="vae", device=device, number_components=n_exemplars,
                     training_set_size=N_TRAIN, approximate_prior=False, approximate_k=10, no_mask=False,
                     no_attention=False, same_variational_var=False, use_logit=False, lambd=1e-4,
                     bottleneck=6, dataset_name="dynamic_mnist", continuous=False, batch_size=B,
                     dynamic_binarization=True, warmup=100, S=5000, shard_exemplars=sharded, shard_batch=shard_batch)


def gated_flops(M, K, N):
    return 2.0 * M * K * 2 * N


def cpu_baseline(steps):
    """The oracle's restatement of the same training step on the host cores (numpy + its BLAS threads).  More BLAS
    threads are not faster on these hosts (64 threads: 170-180 images/s, 16 threads: 560-630 on the 2 x 64-core box), so
    a short calibration picks the thread count and `cores` reports the one used."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import evae_oracle as orc
    import golden_inputs as gi
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                   # no control over the BLAS pool: time it as it comes
        threadpool_limits = None
    rs = np.random.RandomState(0)
    data = gi.binary_images(0, N_TRAIN)
    state = {"p": orc.vae_init_params(np.random.RandomState(123)), "opt": {}, "s": 0}

    def one_step():
        s = state["s"]; state["s"] += 1
        bidx = np.arange(s * B, (s + 1) * B).reshape(-1, 1) % N_TRAIN
        x = data[bidx[:, 0]]
        eps = rs.standard_normal((B, Z)).astype(np.float32)
        ex_idx = rs.randint(0, N_TRAIN, size=(C,))
        t0 = time.perf_counter()
        orc.vae_train_step(state["p"], state["opt"], x, bidx, eps, data[ex_idx], ex_idx, beta=0.5)
        return time.perf_counter() - t0

    one_step()                                          # warms the BLAS threads / page faults
    threads, note = os.cpu_count(), ""
    if threadpool_limits is not None:
        best = None
        for nt in sorted({min(os.cpu_count(), n) for n in (64, 32, 16, 8)}, reverse=True):
            with threadpool_limits(limits=nt):
                dt = min(one_step(), one_step())
            if best is None or dt < best[0]:
                best = (dt, nt)
        threads = best[1]
        note = "; BLAS threads chosen by a 2-step calibration over 64/32/16/8"
    t_tot = 0.0
    ctx = threadpool_limits(limits=threads) if threadpool_limits is not None else None
    if ctx is not None:
        ctx.__enter__()
    try:
        for _ in range(steps):
            t_tot += one_step()
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    return {"value": round(B * steps / t_tot, 2), "unit": "images/sec", "cores": threads,
            "kind": "port",
            "sample": "%d steps of the numpy oracle's vae train step (B=%d, C=%d, N=%d) on %d BLAS threads of a %d-CPU host%s"
                      % (steps, B, C, N_TRAIN, threads, os.cpu_count(), note)}


def capture_probe():
    """Child process of a multi-GPU run: capture an all-reduce and an all-gather into a hipGraph, replay, check."""
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = 0 if os.environ.get("EVAE_BENCH_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if os.environ.get("EVAE_BENCH_ONE_DEVICE") == "1":
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
    x = torch.full((256,), float(rank + 1), device=dev)
    out = torch.empty(256 * world, device=dev)
    y = x.clone()
    dist.all_reduce(y); dist.all_gather_into_tensor(out, x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y.copy_(x)
        dist.all_reduce(y)
        dist.all_gather_into_tensor(out, x)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ok = abs(float(y[0].item()) - world * (world + 1) / 2.0) < 1e-3 and abs(float(out[-1].item()) - world) < 1e-3
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


def main():
    if "--capture-probe" in sys.argv:
        capture_probe()
        return
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # test hook (tests/test_gpu_sharded.py style): several ranks on ONE GPU over gloo, since RCCL refuses duplicate devices
    one_device = os.environ.get("EVAE_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    n_ex = a.exemplars

    # Can this RCCL build capture collectives into a hipGraph (and replay them without hanging)?  A capture that fails
    # half-way cannot be recovered from in-process (the collective library's internal streams and torch's RNG state stay
    # in capture mode), so the question is put to a throw-away child process per rank, under a timeout; the ranks then
    # agree on the verdict.
    if world > 1 and not a.no_graph:
        import subprocess
        env = dict(os.environ, MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 17))
        try:
            rc = subprocess.run([sys.executable, os.path.abspath(__file__), "--capture-probe"], env=env, timeout=90,
                                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
        except Exception:
            rc = 1
        flag = torch.tensor([1.0 if rc == 0 else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() < 0.5:
            if rank == 0:
                print("bench: RCCL collectives cannot be captured into a hipGraph here; launching eagerly", file=sys.stderr)
            a.no_graph = True

    import golden_inputs as gi
    from evae import ops
    from models.VAE import VAE
    from utils.optimizer import AdamNormGrad
    from utils.training import set_beta

    # synthetic dynamic_mnist-shaped training set: binary 28x28, N=50 000, seed 0
    data = torch.from_numpy(gi.binary_images(0, N_TRAIN))
    dataset = torch.utils.data.TensorDataset(data, torch.arange(N_TRAIN).reshape(-1, 1), torch.arange(N_TRAIN) % 10)
    dp = world > 1 and a.parallel == "dp"
    args = model_args("cuda:%d" % local_rank, n_ex, sharded=world > 1, shard_batch=dp)
    torch.manual_seed(14)                    # same weights and (CPU-generator) exemplar draws on every rank
    torch.cuda.manual_seed(14)
    model = VAE(args).to(dev)
    if dp:
        torch.cuda.manual_seed(14 + rank)    # data-parallel batches: every rank its own eps / binarisation stream
    opt = AdamNormGrad(model.parameters(), lr=5e-4)
    data_dev = model.resident_data(dataset)  # one upload; exemplar gathers read HBM from here on
    idx_all = torch.arange(N_TRAIN, device=dev).reshape(-1, 1)
    idx_host = torch.arange(N_TRAIN).reshape(-1, 1)      # what a DataLoader over the training set hands out
    beta = set_beta(args, 50)
    model.train()
    nb = N_TRAIN // B
    loss_acc = torch.zeros((), device=dev)

    def batch_start(i):                      # data-parallel: rank r takes the r-th batch of every group of `world`
        return (((i * world + rank) if dp else i) % nb) * B

    def eager_step(i):
        s = batch_start(i)
        x = torch.bernoulli(data_dev[s:s + B])                     # dynamic binarisation (training.py:31)
        opt.zero_grad()
        loss, RE, KL = model.calculate_loss((x, idx_all[s:s + B]), beta, average=True, dataset=dataset)
        loss.backward()
        opt.step()
        loss_acc.add_(loss.detach())

    from evae.graph import GraphedTrainStep
    graphed = None if a.no_graph else GraphedTrainStep(model, opt, dataset, B, True)

    state = {"graphed": graphed}

    def step(i):
        g = state["graphed"]
        if g is None:
            return eager_step(i)
        s = batch_start(i)
        try:
            out = g(data_dev[s:s + B], idx_host[s:s + B], beta)     # one hipGraph launch (after 3 eager warm-ups)
        except Exception as e:                                      # capture refused (e.g. by the collective
            if g.graph is not None and g._calls > g.warmup_steps + 1:  # library): keep measuring, eagerly
                raise
            print("bench: hipGraph capture failed (%s); continuing with eager launches" % type(e).__name__,
                  file=sys.stderr)
            state["graphed"] = None
            model._exemplar_indices_override = None
            torch.cuda.synchronize()
            return eager_step(i)
        # the step's losses are accumulated in g.totals by the captured graph itself

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    g_ = state["graphed"]
    loss_sum = float(loss_acc.item()) + (float(g_.totals[0].item()) if g_ is not None else 0.0)
    # Per-kernel timing of the dominant kernel: HIP-event pairs around every GatedDense forward launch.
    # Event pairs cannot be read back from inside a replayed graph, so the same steps are run eagerly right
    # after the timed region (identical kernels, identical shapes); the rocprofv3 summary of the whole
    # command (profiles/) reports the same average for this kernel.
    graphed = state["graphed"]
    # the clocks sag during the host pause above and take ~12 eager steps (30 ms) to come back: untimed steps first,
    # otherwise the event pairs time the launch at a lower clock than the timed region (and rocprof's trace of it) ran at
    for i in range(a.probe_warmup if graphed is not None else 0):
        eager_step(a.warmup + a.steps + i)
    ops.PROBE = {"gated_dense_fwd": []}
    for i in range(a.probe_steps if graphed is not None else 0):
        eager_step(a.warmup + a.steps + a.probe_warmup + i)
    fence()
    probe, ops.PROBE = ops.PROBE, None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = loss_sum / (a.warmup + a.steps)

    # roofline of the dominant launch: gemm_kernel<KC,KC,EPI_GATED>, encoder layer 1 (one launch per step)
    ev = probe["gated_dense_fwd"]
    durs_ms = [s.elapsed_time(e) for s, e, _, _ in ev]
    if os.environ.get("EVAE_BENCH_DUMP_PROBE"):
        print("probe us per launch:", [round(1e3 * d / r, 1) for d, (_, _, _, r) in zip(durs_ms, ev)], file=sys.stderr)
    flops = [f for _, _, f, _ in ev]
    nlaunch = sum(r for _, _, _, r in ev)
    roof = None
    if durs_ms:
        achieved = sum(flops) / (sum(durs_ms) * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_gated_dense.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"bound": "mfma",
                "kernel": "evae::gemm_kernel<true, true, 1, true, 128, 8, 0, true> -- GatedDense forward of encoder layer 1 "
                          "([C+B] gathered rows x 784 -> 2 x 300, gate fused; the row-gathered variant is a symbol of its own)",
                "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                "launches": nlaunch, "avg_launch_us": round(1e3 * sum(durs_ms) / nlaunch, 2),
                "flops_per_launch": round(sum(flops) / nlaunch)}

    # second half of BASELINE.json's metric: test log p(x) (IWAE, S = 5000, all N_train exemplars as the prior)
    iwae = None
    if world == 1 and a.iwae_images > 0:
        from utils.evaluation import calculate_likelihood
        test = torch.from_numpy(gi.binary_images(2, a.iwae_images))
        test_ds = torch.utils.data.TensorDataset(test, torch.zeros(len(test)))
        loader = torch.utils.data.DataLoader(test_ds, batch_size=100)
        import contextlib, io
        model.eval()
        with torch.no_grad():
            cz, clv = model.cache_z(dataset)
            emb = (cz, clv, torch.arange(len(cz)))
            with contextlib.redirect_stdout(io.StringIO()):
                calculate_likelihood(args, model, torch.utils.data.DataLoader(
                    torch.utils.data.TensorDataset(test[:2], torch.zeros(2)), batch_size=2), S=args.S, exemplars_embedding=emb)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ll = calculate_likelihood(args, model, loader, S=args.S, exemplars_embedding=emb)
                torch.cuda.synchronize()
                t_ll = time.perf_counter() - t1
        model.train()
        iwae = {"neg_log_px": round(ll, 3), "images": a.iwae_images, "S": args.S, "exemplars": N_TRAIN,
                "ms_per_image": round(1e3 * t_ll / a.iwae_images, 3),
                "note": "utils.evaluation.calculate_likelihood on synthetic test images after the benchmark's training steps"}

    gb = B * world if dp else B              # images per step over all ranks
    if rank == 0:
        out = {
            "metric": "training images/sec", "value": round(gb * a.steps / dt, 1), "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 4),
            "higher_is_better": True, "scaling": "strong" if (world > 1 and not dp) else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "vae + exemplar_prior, dynamic_mnist-shaped binary 28x28, N=%d, batch %d per GPU, "
                                   "%d exemplars in total, exact prior (BASELINE.json configs[1])" % (N_TRAIN, B, n_ex),
                       "global_batch": gb, "exemplars": n_ex,
                       "parallelism": ("single GPU" if world == 1 else
                                       ("dp%d (own %d-image batch per rank) x exemplar-shard x%d, partial-LSE exchange over RCCL"
                                        % (world, B, world)) if dp else
                                       ("replicated batch, exemplar-shard x%d" % world)),
                       "launch": "eager" if graphed is None else "hipGraph replay of the whole step"},
            "mean_loss": round(final_loss, 4),
            "roofline": roof,
            "test_log_px": iwae,
            "cpu_baseline": None,
        }
        if world == 1 and a.cpu_baseline_steps > 0:
            out["cpu_baseline"] = cpu_baseline(a.cpu_baseline_steps)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
