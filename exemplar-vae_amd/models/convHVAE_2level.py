"""Gated-convolutional two-level HVAE (reference models/convHVAE_2level.py:9-97), same submodule names.
Dense parts run on the HIP GEMM; the gated convolutions go through utils.nn.GatedConv2d."""
import numpy as np
import torch.nn as nn

from models.AbsHModel import BaseHModel
from utils.nn import Conv2d, GatedConv2d, GatedDense, NonLinear


def _ht(lo=-6., hi=2.):
    return nn.Hardtanh(min_val=lo, max_val=hi)


class VAE(BaseHModel):
    def __init__(self, args):
        super().__init__(args)

    def create_model(self, args):
        if args.dataset_name == 'freyfaces':
            self.h_size = 210
        elif args.dataset_name in ('cifar10', 'svhn'):
            self.h_size = 384
        else:
            self.h_size = 294
        fc = 300
        c_in, na = self.args.input_size[0], args.no_attention
        z1, z2 = self.args.z1_size, self.args.z2_size

        def enc(first_k, first_p, mid_k, mid_p):
            return nn.Sequential(
                GatedConv2d(c_in, 32, first_k, 1, first_p, no_attention=na),
                GatedConv2d(32, 32, 3, 2, 1, no_attention=na),
                GatedConv2d(32, 64, mid_k, 1, mid_p, no_attention=na),
                GatedConv2d(64, 64, 3, 2, 1, no_attention=na),
                GatedConv2d(64, 6, 3, 1, 1, no_attention=na))

        # q(z2 | x)
        self.q_z_layers = enc(7, 3, 5, 2)
        self.q_z_mean = NonLinear(self.h_size, z2, activation=None)
        self.q_z_logvar = NonLinear(self.h_size, z2, activation=_ht())
        # q(z1 | x, z2)
        self.q_z1_layers_x = enc(3, 1, 3, 1)
        self.q_z1_layers_z2 = nn.Sequential(GatedDense(z2, self.h_size))
        self.q_z1_layers_joint = nn.Sequential(GatedDense(2 * self.h_size, fc))
        self.q_z1_mean = NonLinear(fc, z1, activation=None)
        self.q_z1_logvar = NonLinear(fc, z1, activation=_ht())
        # p(z1 | z2)
        self.p_z1_layers_z2 = nn.Sequential(GatedDense(z2, fc, no_attention=na), GatedDense(fc, fc, no_attention=na))
        self.p_z1_mean = NonLinear(fc, z1, activation=None)
        self.p_z1_logvar = NonLinear(fc, z1, activation=_ht())
        # p(x | z1, z2)
        self.p_x_layers_z1 = nn.Sequential(GatedDense(z1, fc, no_attention=na))
        self.p_x_layers_z2 = nn.Sequential(GatedDense(z2, fc, no_attention=na))
        self.p_x_layers_joint_pre = nn.Sequential(
            GatedDense(2 * fc, int(np.prod(self.args.input_size)), no_attention=na))
        self.p_x_layers_joint = nn.Sequential(
            GatedConv2d(c_in, 64, 3, 1, 1, no_attention=na), GatedConv2d(64, 64, 3, 1, 1, no_attention=na),
            GatedConv2d(64, 64, 3, 1, 1, no_attention=na), GatedConv2d(64, 64, 3, 1, 1, no_attention=na))
        if self.args.input_type == 'binary':
            self.p_x_mean = Conv2d(64, 1, 1, 1, 0, activation=nn.Sigmoid())
        elif self.args.input_type in ('gray', 'continuous'):
            self.p_x_mean = Conv2d(64, c_in, 1, 1, 0)
            self.p_x_logvar = Conv2d(64, c_in, 1, 1, 0, activation=_ht(-4.5, 0.))
        elif self.args.input_type == 'pca':
            self.p_x_mean = Conv2d(64, 1, 1, 1, 0)
            self.p_x_logvar = Conv2d(64, c_in, 1, 1, 0, activation=_ht(-4.5, 0.))

    def forward(self, x):
        x = x.view(-1, self.args.input_size[0], self.args.input_size[1], self.args.input_size[2])
        return super().forward(x)
