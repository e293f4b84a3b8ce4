#!/usr/bin/env python3
"""Find reads of uninitialised device memory: run test functions with torch.empty / empty_like / Tensor.new_empty returning NaN-filled
float tensors (a fresh process usually sees zero pages there, a long one whatever was freed before).
  python tools/poison_empty.py tests/test_gpu_model.py::test_configuration_matrix_trains_one_step[11] ...   (pytest node ids)"""
import sys
import torch
_e, _el = torch.empty, torch.empty_like


def _poison(t):
    if t.is_cuda and t.is_floating_point() and t.numel() > 0:
        t.fill_(float("nan"))
    return t


torch.empty = lambda *a, **k: _poison(_e(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_el(*a, **k))
import pytest
sys.exit(pytest.main(["-x", "-q", "-m", "gpu"] + sys.argv[1:]))
