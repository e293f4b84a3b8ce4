"""Fully-connected two-level HVAE.  Widths and submodule names (= state_dict keys) follow reference
models/HVAE_2level.py:11-66; the output heads p_x_mean / p_x_logvar are created by models.BaseModel.  Modules are
instantiated in the order of `_trunks` / `_heads` interleaved as below, the order the reference draws weights in."""
import numpy as np
import torch
import torch.nn as nn

from models.AbsHModel import BaseHModel
from utils.nn import GatedDense, HipLinear, NonLinear


def _gated(*widths):
    return nn.Sequential(*[GatedDense(a, b) for a, b in zip(widths[:-1], widths[1:])])


def _logvar_head(width, zdim):
    return NonLinear(width, zdim, activation=nn.Hardtanh(min_val=-6., max_val=2.))


class VAE(BaseHModel):
    def __init__(self, args):
        super().__init__(args)

    def create_model(self, args):
        print("create_model")
        self.args = args
        n_pix, w = int(np.prod(args.input_size)), args.hidden_size
        z1, z2 = args.z1_size, args.z2_size

        # q(z2 | x): two gated layers; the log-variance is one shared scalar when same_variational_var is set
        self.q_z_layers = _gated(n_pix, w, w)
        self.q_z_mean = HipLinear(w, z2)
        self.q_z_logvar = torch.nn.Parameter(torch.randn((1))) if args.same_variational_var else _logvar_head(w, z2)

        # q(z1 | x, z2): one gated layer per input, one joint layer over their concatenation
        for name, widths in (('q_z1_layers_x', (n_pix, w)), ('q_z1_layers_z2', (z2, w)),
                             ('q_z1_layers_joint', (2 * w, w))):
            setattr(self, name, _gated(*widths))
        self.q_z1_mean, self.q_z1_logvar = HipLinear(w, z1), _logvar_head(w, z1)

        # p(z1 | z2)
        self.p_z1_layers_z2 = _gated(z2, w, w)
        self.p_z1_mean, self.p_z1_logvar = HipLinear(w, z1), _logvar_head(w, z1)

        # p(x | z1, z2): same shape as q(z1 | .) with the latents as inputs
        for name, widths in (('p_x_layers_z1', (z1, w)), ('p_x_layers_z2', (z2, w)),
                             ('p_x_layers_joint', (2 * w, w))):
            setattr(self, name, _gated(*widths))
