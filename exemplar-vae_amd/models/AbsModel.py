"""One-level VAE on top of BaseModel (reference models/AbsModel.py:8-49)."""
import numpy as np
import torch

from models.BaseModel import BaseModel
from utils.distributions import log_normal_diag


class AbsModel(BaseModel):
    def __init__(self, args):
        super().__init__(args)

    def kl_loss(self, latent_stats, exemplars_embedding, dataset, cache, x_indices):
        z_q, z_q_mean, z_q_logvar = latent_stats
        if exemplars_embedding is None and self.args.prior == 'exemplar_prior':
            exemplars_embedding = self.get_exemplar_set(z_q_mean, z_q_logvar, dataset, cache, x_indices)
        log_p_z = self.log_p_z(z=(z_q, x_indices), exemplars_embedding=exemplars_embedding)
        log_q_z = log_normal_diag(z_q, z_q_mean, z_q_logvar, dim=1)
        return -(log_p_z - log_q_z)

    def generate_x_from_z(self, z, with_reparameterize=True):
        generated_x, _ = self.p_x(z)
        if getattr(self.args, 'use_logit', False) is True:
            return self.logit_inverse(generated_x)
        return generated_x

    def p_x(self, z):
        if 'conv' in self.args.model_name:
            z = z.reshape(-1, self.bottleneck, self.args.input_size[1] // 4, self.args.input_size[1] // 4)
        h = self.p_x_layers(z)
        x_mean = self.p_x_mean(h)
        d_in = int(np.prod(self.args.input_size))
        if self.args.input_type == 'binary':
            x_logvar = torch.zeros(1, d_in)
        elif self.args.use_logit is False:
            x_mean = torch.clamp(x_mean, min=0. + 1. / 512., max=1. - 1. / 512.)
            x_logvar = self.decoder_logstd * x_mean.new_ones(size=x_mean.shape)
        else:
            # the reference leaves x_logvar unbound here (AbsModel.py:36-42); same failure, clearer message
            raise UnboundLocalError("AbsModel.p_x: continuous input with use_logit=True has no x_logvar")
        return x_mean.reshape(-1, d_in), x_logvar.reshape(-1, d_in)

    def forward(self, x, label=0, num_categories=10):
        z_q_mean, z_q_logvar = self.q_z(x)
        z_q = self.reparameterize(z_q_mean, z_q_logvar)
        x_mean, x_logvar = self.p_x(z_q)
        return x_mean, x_logvar, (z_q, z_q_mean, z_q_logvar)
