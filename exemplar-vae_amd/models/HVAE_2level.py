"""MLP two-level HVAE (reference models/HVAE_2level.py:11-66), same submodule names."""
import numpy as np
import torch
import torch.nn as nn

from models.AbsHModel import BaseHModel
from utils.nn import GatedDense, HipLinear, NonLinear


def _ht():
    return nn.Hardtanh(min_val=-6., max_val=2.)


class VAE(BaseHModel):
    def __init__(self, args):
        super().__init__(args)

    def create_model(self, args):
        print("create_model")
        self.args = args
        d_in, hid = int(np.prod(self.args.input_size)), self.args.hidden_size
        z1, z2 = self.args.z1_size, self.args.z2_size
        # q(z2 | x)
        self.q_z_layers = nn.Sequential(GatedDense(d_in, hid), GatedDense(hid, hid))
        self.q_z_mean = HipLinear(hid, z2)
        if args.same_variational_var:
            self.q_z_logvar = torch.nn.Parameter(torch.randn((1)))
        else:
            self.q_z_logvar = NonLinear(hid, z2, activation=_ht())
        # q(z1 | x, z2)
        self.q_z1_layers_x = nn.Sequential(GatedDense(d_in, hid))
        self.q_z1_layers_z2 = nn.Sequential(GatedDense(z2, hid))
        self.q_z1_layers_joint = nn.Sequential(GatedDense(2 * hid, hid))
        self.q_z1_mean = HipLinear(hid, z1)
        self.q_z1_logvar = NonLinear(hid, z1, activation=_ht())
        # p(z1 | z2)
        self.p_z1_layers_z2 = nn.Sequential(GatedDense(z2, hid), GatedDense(hid, hid))
        self.p_z1_mean = HipLinear(hid, z1)
        self.p_z1_logvar = NonLinear(hid, z1, activation=_ht())
        # p(x | z1, z2)
        self.p_x_layers_z1 = nn.Sequential(GatedDense(z1, hid))
        self.p_x_layers_z2 = nn.Sequential(GatedDense(z2, hid))
        self.p_x_layers_joint = nn.Sequential(GatedDense(2 * hid, hid))
