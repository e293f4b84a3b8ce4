"""MLP VAE 784 -> 300 -> 300 -> 40 / 40 -> 300 -> 300 -> 784 (reference models/VAE.py:11-30), same
submodule names and state_dict keys."""
import numpy as np
import torch
import torch.nn as nn

from models.AbsModel import AbsModel
from utils.nn import GatedDense, HipLinear, NonLinear


class VAE(AbsModel):
    def __init__(self, args):
        super().__init__(args)

    def create_model(self, args, train_data_size=None):
        self.train_data_size = train_data_size
        d_in, hid, zd = int(np.prod(self.args.input_size)), self.args.hidden_size, self.args.z1_size
        na = self.args.no_attention
        self.q_z_layers = nn.Sequential(GatedDense(d_in, hid, no_attention=na), GatedDense(hid, hid, no_attention=na))
        self.q_z_mean = HipLinear(hid, zd)
        if args.same_variational_var:
            self.q_z_logvar = torch.nn.Parameter(torch.randn((1)))
        else:
            self.q_z_logvar = NonLinear(hid, zd, activation=nn.Hardtanh(min_val=-6., max_val=2.))
        self.p_x_layers = nn.Sequential(GatedDense(zd, hid, no_attention=na), GatedDense(hid, hid, no_attention=na))
