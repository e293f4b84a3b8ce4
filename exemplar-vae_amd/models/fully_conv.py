"""Fully-convolutional weight-normed residual VAE, model_name='single_conv' (reference
models/fully_conv.py:7-81), same submodule names / state_dict entries (including the BatchNorm2d the
reference's residual block builds but never calls)."""
import os

import torch
import torch.nn as nn
from torch.nn.utils import weight_norm

from models.AbsModel import AbsModel
from evae import ops
from utils.nn import HipConv2d


class block(nn.Module):
    """x + conv(ELU(x)) with a weight-normed 3x3 convolution."""

    def __init__(self, input_size, output_size, stride=1, kernel=3, padding=1):
        super().__init__()
        self.normalization = nn.BatchNorm2d(input_size)       # present in the state_dict, unused (as in the reference)
        self.conv1 = weight_norm(HipConv2d(input_size, output_size, kernel_size=kernel, stride=stride,
                                           padding=padding, bias=True))
        self.activation = torch.nn.ELU()
        self.f = torch.nn.Sequential(self.activation, self.conv1)

    def forward(self, x):
        conv = self.conv1
        if x.is_cuda and os.environ.get("EVAE_RESBLOCK", "1") != "0" and ops.res_block_supported(x, conv.weight_v, conv.stride, conv.padding):
            for hook in conv._forward_pre_hooks.values():      # weight_norm: weight = g * v / ||v|| (differentiable)
                hook(conv, (x,))
            return ops.res_block(x, conv.weight, conv.bias)
        return x + self.f(x)


class _Seq(nn.Sequential):
    """nn.Sequential (same children, same state_dict keys) that hands every run of consecutive residual blocks to ONE operator over
    pre-split pixel images (evae.ops.ResStackFn) when the tensor is large enough to fill the machine."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, block) and x.is_cuda and os.environ.get("EVAE_RESBLOCK", "1") != "0":
                j = i
                while j < len(mods) and isinstance(mods[j], block):
                    j += 1
                run = mods[i:j]
                if ops.res_stack_supported(x, [b.conv1.weight_v for b in run]):
                    for b in run:                        # weight_norm: weight = g * v / ||v|| (differentiable, torch's own hook)
                        for hook in b.conv1._forward_pre_hooks.values():
                            hook(b.conv1, (x,))
                    x = ops.res_stack(x, [(b.conv1.weight, b.conv1.bias) for b in run])
                    i = j
                    continue
            x = m(x)
            i += 1
        return x


def _wn_conv(cin, cout, stride=1):
    return weight_norm(HipConv2d(in_channels=cin, out_channels=cout, kernel_size=3, stride=stride, padding=1))


class VAE(AbsModel):
    def __init__(self, args):
        super().__init__(args)

    def create_model(self, args, train_data_size=None):
        self.train_data_size = train_data_size
        self.cs = 48
        self.bottleneck = self.args.bottleneck
        cs, c_in = self.cs, self.args.input_size[0]
        self.q_z_layers = _Seq(
            _wn_conv(c_in, cs, 2), nn.ELU(), *[block(cs, cs) for _ in range(6)],
            _wn_conv(cs, cs * 2, 2), nn.ELU(), *[block(cs * 2, cs * 2) for _ in range(6)])
        self.q_z_mean = _wn_conv(cs * 2, self.bottleneck)
        self.q_z_logvar = _wn_conv(cs * 2, self.bottleneck)
        self.p_x_layers = _Seq(
            nn.Upsample(scale_factor=2), _wn_conv(self.bottleneck, cs * 2), nn.ELU(),
            *[block(cs * 2, cs * 2) for _ in range(6)],
            nn.Upsample(scale_factor=2), _wn_conv(cs * 2, cs), nn.ELU(), *[block(cs, cs) for _ in range(6)])
        if self.args.input_type == 'binary':
            self.p_x_mean = nn.Sequential(HipConv2d(cs, c_in, kernel_size=3, stride=1, padding=1), nn.Sigmoid())
        elif self.args.input_type in ('gray', 'continuous'):
            self.p_x_mean = _wn_conv(cs, c_in)
            self.p_x_logvar = HipConv2d(cs, c_in, kernel_size=3, stride=1, padding=1)
