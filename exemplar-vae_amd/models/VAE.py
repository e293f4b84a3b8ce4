"""Fully-connected VAE, MNIST-sized by default: 784 -> 300 -> 300 -> 40 and back.  Widths and submodule names
(= state_dict keys) follow reference models/VAE.py:11-30; the output heads come from models.BaseModel."""
import numpy as np
import torch
import torch.nn as nn

from models.AbsModel import AbsModel
from utils.nn import GatedDense, HipLinear, NonLinear


class VAE(AbsModel):
    def __init__(self, args):
        super().__init__(args)

    def _gated_pair(self, n_in, width):
        plain = self.args.no_attention
        return nn.Sequential(GatedDense(n_in, width, no_attention=plain), GatedDense(width, width, no_attention=plain))

    def create_model(self, args, train_data_size=None):
        self.train_data_size = train_data_size
        n_pix, width, zdim = int(np.prod(self.args.input_size)), self.args.hidden_size, self.args.z1_size
        self.q_z_layers = self._gated_pair(n_pix, width)
        self.q_z_mean = HipLinear(width, zdim)
        # log-variance of q: one learnt scalar shared by all inputs, or a clipped linear head
        self.q_z_logvar = (torch.nn.Parameter(torch.randn((1))) if args.same_variational_var
                           else NonLinear(width, zdim, activation=nn.Hardtanh(min_val=-6., max_val=2.)))
        self.p_x_layers = self._gated_pair(zdim, width)
