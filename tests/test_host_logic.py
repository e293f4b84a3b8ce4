"""CPU-only checks of the host-side mirror of the reference interface: class / function names, constructor
arguments, state_dict keys (so reference checkpoints load), the model-name mapping and the loop helpers."""
import inspect
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

import evae_oracle as orc
import smoke_case


def args_for(model_name, **kw):
    return smoke_case.vae_args(device="cpu", model_name=model_name, **kw)


def test_vae_state_dict_names_and_shapes():
    from models.VAE import VAE
    m = VAE(args_for("vae"))
    sd = m.state_dict()
    assert list(sd.keys()) == orc.VAE_PARAM_NAMES
    ref = orc.vae_init_params(np.random.RandomState(0))
    for k, v in sd.items():
        assert tuple(v.shape) == ref[k].shape, k


@pytest.mark.parametrize("name,module,n_entries,n_params", [
    ("hvae_2level", "models.HVAE_2level", 55, 2431025),
    ("convhvae_2level", "models.convHVAE_2level", 99, 2210862)])
def test_hierarchical_models_match_reference_parameter_counts(name, module, n_entries, n_params):
    mod = __import__(module, fromlist=["VAE"])
    m = mod.VAE(args_for(name))
    assert len(m.state_dict()) == n_entries            # SURVEY.md section 8a, probed from the reference
    assert sum(p.numel() for p in m.parameters()) == n_params
    for key in ("q_z1_layers_x.0.h.weight", "p_z1_layers_z2.1.g.bias", "prior_log_variance"):
        assert key in m.state_dict()


def test_fully_conv_state_dict_entries():
    from models.fully_conv import VAE
    m = VAE(args_for("single_conv", input_size=[3, 64, 64], input_type="continuous", bottleneck=1, z1_size=256))
    sd = m.state_dict()
    assert len(sd) == 289 and sum(p.numel() for p in m.parameters()) == 1341087
    assert any(k.endswith("weight_g") for k in sd) and any("normalization" in k for k in sd)


def test_importing_model_mapping():
    from utils.utils import importing_model
    import models.VAE, models.HVAE_2level, models.convHVAE_2level, models.fully_conv
    assert importing_model(Namespace(model_name="vae")) is models.VAE.VAE
    assert importing_model(Namespace(model_name="hvae_2level")) is models.HVAE_2level.VAE
    assert importing_model(Namespace(model_name="convhvae_2level")) is models.convHVAE_2level.VAE
    assert importing_model(Namespace(model_name="single_conv")) is models.fully_conv.VAE
    with pytest.raises(Exception):
        importing_model(Namespace(model_name="nope"))


def test_reference_signatures_are_kept():
    from models.BaseModel import BaseModel
    from utils import distributions, knn_on_latent, training, evaluation
    from utils.nn import GatedDense, NonLinear, GatedConv2d, Conv2d
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(BaseModel.calculate_loss) == ["self", "x", "beta", "average", "exemplars_embedding", "cache", "dataset"]
    assert sig(BaseModel.log_p_z) == ["self", "z", "exemplars_embedding", "sum", "test"]
    assert sig(BaseModel.log_p_z_exemplar) == ["self", "z", "z_indices", "exemplars_embedding", "test"]
    assert sig(BaseModel.cache_z) == ["self", "dataset", "prior", "cuda"]
    assert sig(BaseModel.get_exemplar_set) == ["self", "z_mean", "z_log_var", "dataset", "cache", "x_indices"]
    assert sig(BaseModel.get_approximate_nearest_exemplars) == ["self", "z", "cache", "dataset"]
    assert sig(BaseModel.q_z)[:3] == ["self", "x", "prior"]
    assert sig(distributions.pairwise_distance) == ["z", "means"]
    assert sig(distributions.log_normal_diag_vectorized) == ["x", "mean", "log_var"]
    assert sig(knn_on_latent.find_nearest_neighbors) == ["z_val", "z_train", "z_train_log_var"]
    assert sig(knn_on_latent.report_knn_on_latent) == ["train_loader", "val_loader", "test_loader", "model", "dir",
                                                       "knn_dictionary", "args", "val"]
    assert sig(training.train_one_epoch) == ["epoch", "args", "train_loader", "model", "optimizer"]
    assert sig(evaluation.evaluate_loss) == ["args", "model", "loader", "dataset", "exemplars_embedding"]
    assert sig(evaluation.calculate_likelihood) == ["args", "model", "loader", "S", "exemplars_embedding"]
    assert sig(GatedDense.__init__) == ["self", "input_size", "output_size", "activation", "no_attention"]
    assert sig(NonLinear.__init__) == ["self", "input_size", "output_size", "bias", "activation"]
    assert sig(GatedConv2d.__init__)[:6] == ["self", "input_channels", "output_channels", "kernel_size", "stride", "padding"]
    assert sig(Conv2d.__init__)[-1] == "bias"


def test_set_beta_and_optimizer_state_layout():
    from utils.training import set_beta
    from utils.optimizer import AdamNormGrad
    a = Namespace(warmup=100)
    assert set_beta(a, 50) == 0.5 and set_beta(a, 500) == 1.0 and set_beta(Namespace(warmup=0), 3) == 1.0
    p = torch.nn.Parameter(torch.zeros(3))
    opt = AdamNormGrad([p], lr=5e-4)
    assert opt.defaults["betas"] == (0.9, 0.999) and opt.defaults["eps"] == 1e-8 and opt.defaults["weight_decay"] == 0


def test_captured_step_bookkeeping_skips_parameters_without_a_gradient():
    """a resumed checkpoint whose unused parameters (fully_conv's BatchNorm2d) carry no state, or another
    step count, must neither crash the captured step nor get optimizer state the reference would not keep
    (reference utils/optimizer.py:50-57 skips p.grad is None)."""
    from utils.optimizer import AdamNormGrad
    used, unused = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2))
    opt = AdamNormGrad([used, unused], lr=5e-4)
    st = opt._init_state(used)
    st['step'] = 5                                    # a resumed, trained parameter; `unused` has no state at all
    tables = {}
    with pytest.raises(RuntimeError):                 # participants unknown: refuse instead of walking every parameter
        opt.advance_graph_step(host_out=[0.0], tables=tables)
    used.grad = torch.ones(3)
    assert opt.learn_members(tables) is True and tables[("members", 0)] == [used]
    out = [0.0]
    opt.advance_graph_step(host_out=out, tables=tables)
    assert opt.state[used]['step'] == 6 and len(opt.state[unused]) == 0
    assert out[0] == pytest.approx(5e-4 * (1 - 0.999 ** 6) ** 0.5 / (1 - 0.9 ** 6))
    # participants that disagree: reported by learn_members (the runner then steps eagerly), nothing mutated by advance
    unused.grad = torch.ones(2)
    assert opt.learn_members(tables) is False
    with pytest.raises(RuntimeError):
        opt.advance_graph_step(host_out=out, tables=tables)
    assert opt.state[used]['step'] == 6 and opt.state[unused]['step'] == 0


def test_he_initializer_statistics():
    from models.VAE import VAE
    torch.manual_seed(0)
    m = VAE(args_for("vae"))
    w = m.q_z_layers[0].h.weight
    assert abs(w.std().item() - (2.0 / 784) ** 0.5) < 2e-3       # reference utils/nn.py:12-14


def test_oracle_sharded_merge_is_associative():
    """Merging shard partials in any grouping gives the single-shard log-prior (fp32 noise only)."""
    import golden_inputs as gi
    z, c = gi.clustered_latents(5, 32, 1000, 40)
    zi, ci = gi.mask_indices(6, 32, 1000, 400)
    lv = np.full(40, -1.1, np.float32)
    full = orc.log_p_z(z, zi, c, lv[None], ci, test=False)
    for cuts in ([0, 1000], [0, 1, 999, 1000], [0, 125, 250, 375, 500, 625, 750, 875, 1000], [0, 0, 1000, 1000]):
        parts = [orc.prior_partials(z, zi, c[a:b], lv, ci[a:b], True) for a, b in zip(cuts[:-1], cuts[1:])]
        merged = orc.prior_merge([p[0] for p in parts], [p[1] for p in parts], [p[2] for p in parts], 1000)
        assert np.abs(merged - full).max() < 2e-5 * np.abs(full).max()


def _tiny_vae(device):
    from models.VAE import VAE
    from utils.optimizer import AdamNormGrad
    args = smoke_case.vae_args(input_size=[1, 8, 8], hidden_size=16, z1_size=8, z2_size=8, number_components=10,
                               training_set_size=50, device=device)
    model = VAE(args).to(device)
    return model, AdamNormGrad(model.parameters(), lr=5e-4)


def test_reference_checkpoint_loads_into_model_and_optimizer():
    """tests/golden/g12_checkpoint.pth was written by the REFERENCE's utils.utils.save_model (content of
    density_estimation.py:148-149) after two AdamNormGrad steps: utils.utils.load_model must restore the model and the
    optimizer state exactly as the reference's own load_model does (values recorded in g12_checkpoint.npz)."""
    import os
    from utils.utils import load_model, save_model
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(here, "g12_checkpoint.npz"))
    model, opt = _tiny_vae("cpu")
    ck = load_model(os.path.join(here, "g12_checkpoint.pth"), model, opt)
    assert ck["epoch"] == 7 and ck["e"] == 3 and ck["best_loss"] == 123.5
    assert [n for n, _ in model.named_parameters()] == list(g["names"])
    for n, p in model.named_parameters():
        st = opt.state[p]
        assert np.array_equal(p.detach().numpy(), g["loaded_" + n]), n
        assert np.array_equal(st["exp_avg"].numpy(), g["loaded_m_" + n]), n
        assert np.array_equal(st["exp_avg_sq"].numpy(), g["loaded_v_" + n]), n
        assert int(st["step"]) == int(g["loaded_step_" + n]) == 2
    # and the file this build writes has the same layout (keys, per-parameter state entries, param_groups fields)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        content = {'epoch': 8, 'state_dict': model.state_dict(), 'optimizer': opt.state_dict(), 'best_loss': 1.0, 'e': 0}
        save_model(os.path.join(d, "c.tmp"), os.path.join(d, "c.pth"), content)
        mine = torch.load(os.path.join(d, "c.pth"))
    ref = torch.load(os.path.join(here, "g12_checkpoint.pth"))
    assert set(mine.keys()) == set(ref.keys())
    assert list(mine["state_dict"].keys()) == list(ref["state_dict"].keys())
    assert set(mine["optimizer"].keys()) == set(ref["optimizer"].keys())
    assert set(mine["optimizer"]["param_groups"][0].keys()) >= {"lr", "betas", "eps", "weight_decay", "params"}
    assert mine["optimizer"]["param_groups"][0]["params"] == ref["optimizer"]["param_groups"][0]["params"]
    for k in ref["optimizer"]["state"]:
        assert set(mine["optimizer"]["state"][k].keys()) == set(ref["optimizer"]["state"][k].keys()) == {"step", "exp_avg", "exp_avg_sq"}


def test_load_data_pipeline_matches_reference_golden():
    """utils.load_data.base_load_data.load_dataset == the reference's on the same raw arrays and numpy seed: split,
    value range (three variants), fixed-seed binarisation of validation / test, dataset tuples (tools/gen_goldens.py::g14)."""
    import os
    from types import SimpleNamespace
    from utils.load_data.base_load_data import base_load_data
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g14_load_data.npz"))
    T = torch.from_numpy

    class stub(base_load_data):
        def obtain_data(self):
            return (SimpleNamespace(data=T(g["raw_train"]), train_labels=T(g["lab_train"])),
                    SimpleNamespace(data=T(g["raw_test"]), test_labels=T(g["lab_test"])))
    variants = {"dyn": dict(input_type="binary", dynamic_binarization=True, continuous=False, use_logit=False),
                "grey": dict(input_type="gray", dynamic_binarization=False, continuous=True, use_logit=False),
                "logit": dict(input_type="gray", dynamic_binarization=False, continuous=False, use_logit=True)}
    for tag, kw in variants.items():
        a = Namespace(dataset_name="dynamic_mnist", input_size=[1, 8, 8], training_set_size=250, batch_size=32, test_batch_size=20,
                      use_training_data_init=0, number_components=10, lambd=1e-4, **kw)
        np.random.seed(141)
        tr, va, te, a2 = stub(a, no_binarization=(tag != "dyn")).load_dataset()
        got = dict(zip(("x_train", "idx", "y_train"), tr.dataset.tensors))
        got.update(zip(("x_val", "y_val"), va.dataset.tensors)); got.update(zip(("x_test", "y_test"), te.dataset.tensors))
        for k, v in got.items():
            ref = g[tag + "_" + k]
            assert v.numpy().dtype == ref.dtype and np.array_equal(v.numpy(), ref), (tag, k)
        assert str(g[tag + "_input_type"]) == a2.input_type
        assert [tr.batch_size, va.batch_size, te.batch_size] == list(g[tag + "_batch"])
        assert tr.dataset.tensors[1].shape == (250, 1) and tr.dataset.tensors[1].dtype == torch.int64


def test_load_dataset_dispatch_reads_local_idx_files(tmp_path, monkeypatch):
    """utils.load_data.data_loader_instances.load_dataset('dynamic_mnist') from IDX files on disk (no torchvision, no
    network): args fields as the reference sets them, loaders of the documented shapes; a missing dataset fails loudly."""
    import struct
    from utils.load_data.data_loader_instances import load_dataset
    raw = tmp_path / "datasets" / "dynamic_mnist" / "MNIST" / "raw"
    raw.mkdir(parents=True)
    rs = np.random.RandomState(3)

    def idx(name, arr):
        with open(raw / name, "wb") as f:
            f.write(struct.pack(">HBB", 0, 8, arr.ndim) + struct.pack(">" + "I" * arr.ndim, *arr.shape) + arr.tobytes())
    idx("train-images-idx3-ubyte", rs.randint(0, 256, (120, 28, 28)).astype(np.uint8))
    idx("train-labels-idx1-ubyte", rs.randint(0, 10, 120).astype(np.uint8))
    idx("t10k-images-idx3-ubyte", rs.randint(0, 256, (30, 28, 28)).astype(np.uint8))
    idx("t10k-labels-idx1-ubyte", rs.randint(0, 10, 30).astype(np.uint8))
    monkeypatch.chdir(tmp_path)
    a = Namespace(dataset_name="dynamic_mnist", continuous=False, use_logit=False, lambd=1e-4, batch_size=16, test_batch_size=10,
                  training_set_size=None, use_training_data_init=0, number_components=5)
    tr, va, te, a = load_dataset(a, training_num=100)
    assert (a.input_size, a.input_type, a.dynamic_binarization, a.training_set_size) == ([1, 28, 28], "binary", True, 100)
    x, i, y = tr.dataset.tensors
    assert x.shape == (100, 784) and x.dtype == torch.float32 and float(x.max()) <= 1.0 and i.shape == (100, 1)
    assert va.dataset.tensors[0].shape == (20, 784) and te.dataset.tensors[0].shape == (30, 784)
    assert set(np.unique(te.dataset.tensors[0].numpy())) <= {0.0, 1.0}          # evaluation splits binarised once
    a.dataset_name = "fashion_mnist"
    with pytest.raises(FileNotFoundError):
        load_dataset(a)
    a.dataset_name = "imagenet"
    with pytest.raises(Exception, match="Wrong name of the dataset"):
        load_dataset(a)


def test_state_dict_names_and_shapes_of_every_architecture():
    """Names, order and shapes of model.state_dict() for every architecture x input geometry the reference's datasets
    produce (dumped from the reference's own constructors, tools/gen_goldens.py::g18): a checkpoint of either tree loads
    into the other."""
    import json, os
    from utils.utils import importing_model
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g18_state_dict_shapes.json")
    spec = json.load(open(path))
    assert len(spec) == 9
    for key, d in spec.items():
        name, ds = key.split("|")
        args = smoke_case.vae_args(model_name=name, dataset_name=ds, input_size=d["input_size"], input_type=d["input_type"],
                                   bottleneck=d["bottleneck"], z1_size=d["z1_size"], continuous=(d["input_type"] != "binary"),
                                   device="cpu", rs_blocks=4)
        model = importing_model(args)(args)
        got = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        assert got == d["entries"], key


def test_host_thread_budget_follows_the_cgroup_quota(monkeypatch, tmp_path):
    """evae.hostcpu: the intra-op pool is bounded by min(8, affinity, cgroup quota, cores / local ranks); an explicit
    OMP_NUM_THREADS stands (r03: a 256-thread pool under a 16-CPU quota froze the c5 step for 25-80 ms every few steps)."""
    import builtins
    import torch
    from evae import hostcpu
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            f = tmp_path / "cpu.max"
            f.write_text("300000 100000\n")
            return real_open(f, *a, **k)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)), raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    assert hostcpu.cpu_budget() == 3
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    assert hostcpu.cpu_budget() == 1
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    before = torch.get_num_threads()
    try:
        monkeypatch.delenv("OMP_NUM_THREADS", raising=False); monkeypatch.delenv("EVAE_HOST_THREADS", raising=False)
        torch.set_num_threads(max(before, 6))
        hostcpu._DONE[0] = False
        assert hostcpu.limit_host_threads() == 3 and torch.get_num_threads() == 3
        assert hostcpu.limit_host_threads() == 3                   # once per process
        torch.set_num_threads(6)
        hostcpu._DONE[0] = False
        monkeypatch.setenv("OMP_NUM_THREADS", "6")
        assert hostcpu.limit_host_threads() == 6                   # the user's setting stands
    finally:
        hostcpu._DONE[0] = True
        torch.set_num_threads(before)


def test_host_dedup_of_an_exemplar_draw():
    """evae_host_dedup (host side of the captured step, r04): the distinct rows of a draw with replacement (reference
    models/BaseModel.py:245) in first-occurrence order, every draw's position among them, one draw per distinct row, the
    multiplicities; padding behind the distinct rows; refusal when they do not fit."""
    import ctypes as C
    from evae import _lib
    lib = _lib.load()
    for n, N, seed in ((25000, 50000, 0), (1000, 50000, 1), (64, 8, 2), (1, 1, 3)):
        rs = np.random.RandomState(seed)
        d = torch.from_numpy(rs.randint(0, N, n).astype(np.int64))
        cap = min(n, N) + 5
        rows = torch.full((cap,), -7, dtype=torch.int64); inv = torch.zeros(n, dtype=torch.int64)
        rep = torch.full((cap,), -7, dtype=torch.int64); mult = torch.full((cap,), -7.0)
        p = lambda t: C.c_void_p(t.data_ptr())
        for _ in range(2):             # (the stamp table is kept between calls)
            U = lib.evae_host_dedup(p(d), n, N, cap, p(rows), p(inv), p(rep), p(mult))
        dn, r, iv, rp, m = d.numpy(), rows.numpy(), inv.numpy(), rep.numpy(), mult.numpy()
        uniq, first = np.unique(dn, return_index=True)
        assert U == len(uniq)
        assert np.array_equal(r[:U], dn[np.sort(first)])                    # first-occurrence order
        assert np.array_equal(r[iv], dn) and np.array_equal(iv[rp[:U]], np.arange(U))
        assert np.array_equal(m[:U], np.bincount(dn, minlength=N)[r[:U]].astype(np.float32))
        assert (r[U:] == r[0]).all() and (rp[U:] == 0).all() and (m[U:] == 0).all()
        if U > 1:
            assert lib.evae_host_dedup(p(d), n, N, U - 1, p(rows), p(inv), p(rep), p(mult)) == -1
    bad = torch.tensor([0, 9], dtype=torch.int64)
    assert lib.evae_host_dedup(C.c_void_p(bad.data_ptr()), 2, 5, 4, p(rows), p(inv), p(rep), p(mult)) == -1
