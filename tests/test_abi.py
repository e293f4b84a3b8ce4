"""CPU-only: libevae_hip.so loads (no GPU needed to dlopen) and exports every symbol that
include/evae_hip.h declares; the ctypes table in evae/_lib.py covers exactly that set."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "evae_hip.h")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(evae_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    syms = header_symbols()
    for must in ("evae_prior_lse_fwd", "evae_prior_merge", "evae_prior_lse_bwd", "evae_pairdist_topk",
                 "evae_topk_merge", "evae_gated_dense_fwd", "evae_linear_fwd", "evae_dense_bwd_data",
                 "evae_dense_bwd_weight", "evae_adam_normgrad_step", "evae_version", "evae_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from evae import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail("libevae_hip.so missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.evae_version() == 1


def test_ctypes_table_matches_header():
    from evae import _lib
    assert sorted(_lib.SIGNATURES.keys()) == header_symbols()
    _lib.load()


def test_product_has_no_cpu_fallback():
    """Ops refuse CPU tensors instead of silently computing elsewhere, and nothing under the package
    imports the oracle."""
    import torch
    from evae import ops, _lib
    with pytest.raises(_lib.EvaeError):
        ops.prior_lse_fwd(torch.zeros(2, 4), torch.zeros(3, 4), torch.zeros(4))
    with pytest.raises(_lib.EvaeError):
        ops.pairdist_topk(torch.zeros(2, 4), torch.zeros(30, 4), 3)
    pkg = os.path.join(ROOT, "exemplar-vae_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert "evae_oracle" not in text and "import oracle" not in text, os.path.join(dirpath, f)
