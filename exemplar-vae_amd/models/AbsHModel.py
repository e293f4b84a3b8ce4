"""Two-level hierarchical VAE base (reference models/AbsHModel.py:8-107): exemplar prior on z2,
Gaussian p(z1|z2)."""
import numpy as np
import torch

from models.BaseModel import BaseModel
from utils.distributions import log_normal_diag


class BaseHModel(BaseModel):
    def __init__(self, args):
        super().__init__(args)

    def kl_loss(self, latent_stats, exemplars_embedding, dataset, cache, x_indices):
        z1_q, z1_q_mean, z1_q_logvar, z2_q, z2_q_mean, z2_q_logvar, z1_p_mean, z1_p_logvar = latent_stats
        if exemplars_embedding is None and self.args.prior == 'exemplar_prior':
            exemplars_embedding = self.get_exemplar_set(z2_q_mean, z2_q_logvar, dataset, cache, x_indices)
        z1s, z2s = self.args.z1_size, self.args.z2_size
        log_p_z1 = log_normal_diag(z1_q.view(-1, z1s), z1_p_mean.view(-1, z1s), z1_p_logvar.view(-1, z1s), dim=1)
        log_q_z1 = log_normal_diag(z1_q.view(-1, z1s), z1_q_mean.view(-1, z1s), z1_q_logvar.view(-1, z1s), dim=1)
        log_p_z2 = self.log_p_z(z=(z2_q, x_indices), exemplars_embedding=exemplars_embedding)
        log_q_z2 = log_normal_diag(z2_q.view(-1, z2s), z2_q_mean.view(-1, z2s), z2_q_logvar.view(-1, z2s), dim=1)
        return -(log_p_z1 + log_p_z2 - log_q_z1 - log_q_z2)

    def generate_x_from_z(self, z, with_reparameterize=True):
        z1_mean, z1_logvar = self.p_z1(z)
        z1 = self.reparameterize(z1_mean, z1_logvar) if with_reparameterize else z1_mean
        xs, _ = self.p_x(z1.view(-1, self.args.z1_size), z.view(-1, self.args.z2_size))
        return xs

    def p_z1(self, z2):
        h = self.p_z1_layers_z2(z2)
        return self.p_z1_mean(h), self.p_z1_logvar(h)

    def q_z1(self, x, z2):
        hx = self.q_z1_layers_x(x)
        if self.args.model_name == 'convhvae_2level':
            hx = hx.reshape(hx.size(0), -1)      # conv outputs may be channels-last tensors: reshape keeps the logical (c, y, x) order
        hz = self.q_z1_layers_z2(z2)
        h = self.q_z1_layers_joint(torch.cat((hx, hz), 1))
        return self.q_z1_mean(h), self.q_z1_logvar(h)

    def p_x(self, z1, z2, x=None):
        h = torch.cat((self.p_x_layers_z1(z1), self.p_x_layers_z2(z2)), 1)
        conv = 'convhvae_2level' in self.args.model_name
        if conv:
            h = self.p_x_layers_joint_pre(h)
            h = h.view(-1, self.args.input_size[0], self.args.input_size[1], self.args.input_size[2])
        h_decoder = self.p_x_layers_joint(h)
        x_mean = self.p_x_mean(h_decoder)
        d_in = int(np.prod(self.args.input_size))
        if conv:
            x_mean = x_mean.reshape(-1, d_in)
        if self.args.input_type == 'binary':
            x_logvar = 0.
        else:
            x_mean = torch.clamp(x_mean, min=0. + 1. / 512., max=1. - 1. / 512.)
            x_logvar = self.p_x_logvar(h_decoder)
            if conv:
                x_logvar = x_logvar.reshape(-1, d_in)
        return x_mean, x_logvar

    def forward(self, x):
        z2_q_mean, z2_q_logvar = self.q_z(x)
        z2_q = self.reparameterize(z2_q_mean, z2_q_logvar)
        z1_q_mean, z1_q_logvar = self.q_z1(x, z2_q)
        z1_q = self.reparameterize(z1_q_mean, z1_q_logvar)
        z1_p_mean, z1_p_logvar = self.p_z1(z2_q)
        x_mean, x_logvar = self.p_x(z1_q, z2_q)
        return x_mean, x_logvar, (z1_q, z1_q_mean, z1_q_logvar, z2_q, z2_q_mean, z2_q_logvar, z1_p_mean, z1_p_logvar)
