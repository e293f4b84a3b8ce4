"""Gated-convolutional two-level HVAE.  Architecture and submodule names (= state_dict keys) follow reference
models/convHVAE_2level.py:9-97; the network is described by the tables below and instantiated in table order, which
is also the order the reference draws its initial weights in.  Dense parts run on the HIP GEMM, the gated convolutions
go through utils.nn.GatedConv2d (channels-last implicit GEMM)."""
import numpy as np
import torch.nn as nn

from models.AbsHModel import BaseHModel
from utils.nn import Conv2d, GatedConv2d, GatedConvStack, GatedDense, NonLinear

# flattened size of the 6-channel feature map the two conv encoders end in, per dataset family
_ENCODER_FEATURES = {'freyfaces': 210, 'cifar10': 384, 'svhn': 384}
_ENCODER_FEATURES_DEFAULT = 294          # 28 x 28 inputs: 6 x 7 x 7
_DENSE_WIDTH = 300
_DECODER_CHANNELS = 64

# (out_channels, kernel, stride, padding) of the five gated convolutions of an encoder
_ENCODER_WIDE = ((32, 7, 1, 3), (32, 3, 2, 1), (64, 5, 1, 2), (64, 3, 2, 1), (6, 3, 1, 1))      # q(z2 | x)
_ENCODER_NARROW = ((32, 3, 1, 1), (32, 3, 2, 1), (64, 3, 1, 1), (64, 3, 2, 1), (6, 3, 1, 1))    # x-branch of q(z1 | x, z2)


def _clip(lo, hi):
    return nn.Hardtanh(min_val=lo, max_val=hi)


class VAE(BaseHModel):
    def __init__(self, args):
        super().__init__(args)

    def _conv_stack(self, c_in, table):
        layers = []
        for c_out, k, s, p in table:
            layers.append(GatedConv2d(c_in, c_out, k, s, p, no_attention=self.args.no_attention))
            c_in = c_out
        return GatedConvStack(*layers)

    def _dense_stack(self, *widths, plain=False):
        kw = {} if plain else {'no_attention': self.args.no_attention}
        return nn.Sequential(*[GatedDense(a, b, **kw) for a, b in zip(widths[:-1], widths[1:])])

    def _gaussian_heads(self, prefix, width, zdim):
        setattr(self, prefix + '_mean', NonLinear(width, zdim, activation=None))
        setattr(self, prefix + '_logvar', NonLinear(width, zdim, activation=_clip(-6., 2.)))

    def create_model(self, args):
        self.h_size = _ENCODER_FEATURES.get(args.dataset_name, _ENCODER_FEATURES_DEFAULT)
        feat, fc, ch = self.h_size, _DENSE_WIDTH, _DECODER_CHANNELS
        colours, n_pix = self.args.input_size[0], int(np.prod(self.args.input_size))
        z1, z2 = self.args.z1_size, self.args.z2_size

        self.q_z_layers = self._conv_stack(colours, _ENCODER_WIDE)
        self._gaussian_heads('q_z', feat, z2)

        self.q_z1_layers_x = self._conv_stack(colours, _ENCODER_NARROW)
        self.q_z1_layers_z2 = self._dense_stack(z2, feat, plain=True)
        self.q_z1_layers_joint = self._dense_stack(2 * feat, fc, plain=True)
        self._gaussian_heads('q_z1', fc, z1)

        self.p_z1_layers_z2 = self._dense_stack(z2, fc, fc)
        self._gaussian_heads('p_z1', fc, z1)

        self.p_x_layers_z1 = self._dense_stack(z1, fc)
        self.p_x_layers_z2 = self._dense_stack(z2, fc)
        self.p_x_layers_joint_pre = self._dense_stack(2 * fc, n_pix)
        self.p_x_layers_joint = self._conv_stack(colours, ((ch, 3, 1, 1),) * 4)

        kind = self.args.input_type                       # 1x1 output convolutions
        if kind == 'binary':
            self.p_x_mean = Conv2d(ch, 1, 1, 1, 0, activation=nn.Sigmoid())
        elif kind in ('gray', 'continuous', 'pca'):
            self.p_x_mean = Conv2d(ch, 1 if kind == 'pca' else colours, 1, 1, 0)
            self.p_x_logvar = Conv2d(ch, colours, 1, 1, 0, activation=_clip(-4.5, 0.))

    def forward(self, x):
        return super().forward(x.view(-1, *self.args.input_size))
