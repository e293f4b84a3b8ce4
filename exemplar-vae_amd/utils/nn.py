"""Layer modules with the reference's names, constructor arguments and state_dict keys
(reference utils/nn.py:12-114), computing through the HIP dense kernels (evae.ops).

GatedDense / NonLinear / Linear run on the fp32-MFMA GEMM of libevae_hip.so with the bias, activation and
gate fused into the epilogue.  GatedConv2d / Conv2d / HipConv2d run as channels-last convolutions on the same GEMM kernel
(csrc/evae_conv_cl.hip: a K-slab is 32 channels of one filter tap, so the im2col gather is the dense tile with a per-slab
offset; one pass over x computes both filter banks of a gated layer and applies the gate in the epilogue; any channel
count >= 16 that is a multiple of 4) -- thin first layers through a patch matrix, data gradients into 1/3-channel inputs
through the NCHW implicit-GEMM kernels of csrc/evae_conv.hip."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from evae import ops


def he_init(m):
    """N(0, sqrt(2 / fan_in)) on m.weight  (reference utils/nn.py:12-14)."""
    m.weight.data.normal_(0, float(np.sqrt(2.0 / m.in_features)))


def xavier_init(m):
    m.weight.data.normal_(0, float(np.sqrt(2.0 / (m.in_features + m.out_features))))


def normal_init(m, mean=0., std=0.01):
    m.weight.data.normal_(mean, std)


def _act_code(activation):
    """Map an nn activation module onto an epilogue code; None if it has to run as a separate op."""
    if activation is None:
        return ops.ACT_NONE, 0.0, 0.0
    if isinstance(activation, nn.Sigmoid):
        return ops.ACT_SIGMOID, 0.0, 0.0
    if isinstance(activation, nn.Hardtanh):
        return ops.ACT_HARDTANH, float(activation.min_val), float(activation.max_val)
    return None


def dense(x, linear_module, activation=None, rows=None):
    """activation(linear_module(x)) through evae_linear_fwd; x may be row-gathered by `rows`."""
    x2 = x if x.dim() == 2 else x.reshape(-1, x.shape[-1])
    code = _act_code(activation)
    if code is None:
        y = ops.linear(x2, linear_module.weight, linear_module.bias, rows=rows)
        y = activation(y)
    else:
        y = ops.linear(x2, linear_module.weight, linear_module.bias, code[0], code[1], code[2], rows=rows)
    if rows is None and x.dim() != 2:
        y = y.reshape(*x.shape[:-1], y.shape[-1])
    return y


class HipLinear(nn.Linear):
    """torch.nn.Linear parameters (same state_dict keys), forward on the HIP GEMM."""

    def forward(self, x, rows=None):
        return dense(x, self, None, rows=rows)


class NonLinear(nn.Module):
    def __init__(self, input_size, output_size, bias=True, activation=None):
        super().__init__()
        self.activation = activation
        self.linear = nn.Linear(int(input_size), int(output_size), bias=bias)

    def forward(self, x, rows=None):
        return dense(x, self.linear, self.activation, rows=rows)


class GatedDense(nn.Module):
    """h(x) * sigmoid(g(x)); with no_attention=True the reference degenerates to ReLU(h(x)) and builds
    no gate (utils/nn.py:44-69)."""

    def __init__(self, input_size, output_size, activation=None, no_attention=False):
        super().__init__()
        self.activation = activation
        self.no_attention = no_attention
        self.sigmoid = nn.Sigmoid()
        self.h = nn.Linear(input_size, output_size)
        if no_attention is False:
            self.g = nn.Linear(input_size, output_size)
        else:
            self.activation = nn.ReLU()

    def forward(self, x, rows=None, x_scale=None):
        if self.no_attention is False and self.activation is None:
            x2 = x if x.dim() == 2 else x.reshape(-1, x.shape[-1])
            leaf = ops.active_leaf_stream()
            if leaf is not None and rows is None and x_scale is None and torch.is_grad_enabled() and self.h.weight.requires_grad:
                return ops.gated_dense_split(x2, self.h.weight, self.h.bias, self.g.weight, self.g.bias, leaf)
            # x_scale: x is the uint8 image store (models/BaseModel.py::resident_u8), pixel = byte * x_scale
            return ops.gated_dense(x2, self.h.weight, self.h.bias, self.g.weight, self.g.bias, rows=rows, x_scale=x_scale)
        h = dense(x, self.h, None, rows=rows)
        if self.activation is not None:
            h = self.activation(h)
        if self.no_attention is False:
            return h * dense(x, self.g, self.sigmoid, rows=rows)
        return h


class GatedConv2d(nn.Module):
    """act(h(x)) * sigmoid(g(x)) with two convolutions sharing the input (utils/nn.py:72-97).
    Like the reference, no_attention=True cannot run (no `g` is built there either)."""

    def __init__(self, input_channels, output_channels, kernel_size, stride, padding, dilation=1,
                 activation=None, no_attention=False):
        super().__init__()
        self.no_attention = no_attention
        self.activation = activation
        self.sigmoid = nn.Sigmoid()
        self.h = nn.Conv2d(input_channels, output_channels, kernel_size, stride, padding, dilation)
        if no_attention is False:
            self.g = nn.Conv2d(input_channels, output_channels, kernel_size, stride, padding, dilation)
        else:
            self.activation = nn.ELU()

    def forward(self, x):
        assert self.h.dilation == (1, 1) and self.h.groups == 1
        if self.activation is None:
            # both filter banks in one implicit GEMM over x, gate applied in its epilogue
            return ops.gated_conv2d(x, self.h.weight, self.h.bias, self.g.weight, self.g.bias,
                                    self.h.stride, self.h.padding)
        h = self.activation(ops.conv2d(x, self.h.weight, self.h.bias, self.h.stride, self.h.padding))
        return h * ops.conv2d(x, self.g.weight, self.g.bias, self.g.stride, self.g.padding, ops.ACT_SIGMOID)


class GatedConvStack(nn.Sequential):
    """nn.Sequential of GatedConv2d layers (same children, same state_dict keys) whose leading layers run as ONE operator over
    pre-split pixel images when the batch is large (evae.ops.GatedConvStackFn: the exemplar rows of a training step, cache_z);
    small batches and anything the image kernels do not take go layer by layer as before."""

    def forward(self, x):
        mods = list(self)
        if (ops.CONV_STACK_ON and x.is_cuda and x.dim() == 4 and x.shape[0] >= ops.CONV_STACK_MIN_IMAGES and not x.requires_grad
                and all(isinstance(m, GatedConv2d) and m.no_attention is False and m.activation is None and m.h.dilation == (1, 1)
                        and m.h.groups == 1 for m in mods)):
            spec = [(m.h.weight, ops._int1(m.h.stride), ops._int1(m.h.padding)) for m in mods]
            b = ops.conv_stack_depth(tuple(x.shape), spec)
            if b:
                h = ops.gated_conv_stack(x, [(m.h.weight, m.h.bias, m.g.weight, m.g.bias, m.h.stride, m.h.padding) for m in mods[:b]])
                for m in mods[b:]:
                    h = m(h)
                return h
        return super().forward(x)


class Conv2d(nn.Module):
    def __init__(self, input_channels, output_channels, kernel_size, stride, padding, dilation=1,
                 activation=None, bias=True):
        super().__init__()
        self.activation = activation
        self.conv = nn.Conv2d(input_channels, output_channels, kernel_size, stride, padding, dilation, bias=bias)

    def forward(self, x):
        c = self.conv
        assert c.dilation == (1, 1) and c.groups == 1
        code = _act_code(self.activation)
        if code is None:
            return self.activation(ops.conv2d(x, c.weight, c.bias, c.stride, c.padding))
        return ops.conv2d(x, c.weight, c.bias, c.stride, c.padding, code[0], code[1], code[2])


class HipConv2d(nn.Conv2d):
    """torch.nn.Conv2d parameters / state_dict keys (weight-norm hooks included), forward on the HIP kernels."""

    def forward(self, x):
        assert self.dilation == (1, 1) and self.groups == 1 and self.padding_mode == 'zeros'
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding)
