"""Two-latent-layer VAE (z2 under the exemplar prior, z1 | z2 Gaussian): composition of the sub-networks a concrete
model declares (models.HVAE_2level, models.convHVAE_2level).  Behavioural contract: reference
models/AbsHModel.py:8-107 -- method names, argument order, the 8-tuple of latent statistics and the flat
[B x D] decoder outputs are what models.BaseModel and utils.evaluation rely on."""
import numpy as np
import torch

from models.BaseModel import BaseModel
from utils.distributions import log_normal_diag

_CLAMP_LO, _CLAMP_HI = 1.0 / 512.0, 1.0 - 1.0 / 512.0


class BaseHModel(BaseModel):
    def __init__(self, args):
        super().__init__(args)

    def _is_conv_hvae(self):
        return 'convhvae_2level' in self.args.model_name

    # ---- conditionals ----------------------------------------------------------------------------------------------
    def p_z1(self, z2):
        """p(z1 | z2): mean, log-variance"""
        trunk = self.p_z1_layers_z2(z2)
        return self.p_z1_mean(trunk), self.p_z1_logvar(trunk)

    def q_z1(self, x, z2):
        """q(z1 | x, z2): the image branch and the z2 branch are concatenated before the joint layer"""
        img = self.q_z1_layers_x(x)
        if self.args.model_name == 'convhvae_2level':
            img = img.reshape(img.size(0), -1)   # conv features may be channels-last tensors: reshape keeps (c, y, x) order
        joint = self.q_z1_layers_joint(torch.cat((img, self.q_z1_layers_z2(z2)), dim=1))
        return self.q_z1_mean(joint), self.q_z1_logvar(joint)

    def p_x(self, z1, z2, x=None):
        """p(x | z1, z2) -> (mean, log-variance); log-variance is the scalar 0 for binary data"""
        feat = torch.cat((self.p_x_layers_z1(z1), self.p_x_layers_z2(z2)), dim=1)
        conv, D = self._is_conv_hvae(), int(np.prod(self.args.input_size))
        if conv:                                   # dense pre-layer -> image -> gated conv stack
            c, hh, ww = self.args.input_size
            feat = self.p_x_layers_joint_pre(feat).view(-1, c, hh, ww)
        top = self.p_x_layers_joint(feat)
        flat = (lambda t: t.reshape(-1, D)) if conv else (lambda t: t)
        mean = flat(self.p_x_mean(top))
        if self.args.input_type == 'binary':
            return mean, 0.
        return mean.clamp(min=_CLAMP_LO, max=_CLAMP_HI), flat(self.p_x_logvar(top))

    def generate_x_from_z(self, z, with_reparameterize=True):
        mu1, lv1 = self.p_z1(z)
        z1 = self.reparameterize(mu1, lv1) if with_reparameterize else mu1
        return self.p_x(z1.view(-1, self.args.z1_size), z.view(-1, self.args.z2_size))[0]

    # ---- objective -------------------------------------------------------------------------------------------------
    def kl_loss(self, latent_stats, exemplars_embedding, dataset, cache, x_indices):
        """[log q(z1|x,z2) - log p(z1|z2)] + [log q(z2|x) - log p(z2)], one value per row"""
        z1, q1_mu, q1_lv, z2, q2_mu, q2_lv, p1_mu, p1_lv = latent_stats
        emb = exemplars_embedding
        if emb is None and self.args.prior == 'exemplar_prior':
            emb = self.get_exemplar_set(q2_mu, q2_lv, dataset, cache, x_indices)
        d1, d2 = self.args.z1_size, self.args.z2_size
        rows1 = lambda t: t.view(-1, d1)
        rows2 = lambda t: t.view(-1, d2)
        kl_z1 = (log_normal_diag(rows1(z1), rows1(q1_mu), rows1(q1_lv), dim=1)
                 - log_normal_diag(rows1(z1), rows1(p1_mu), rows1(p1_lv), dim=1))
        kl_z2 = (log_normal_diag(rows2(z2), rows2(q2_mu), rows2(q2_lv), dim=1)
                 - self.log_p_z(z=(z2, x_indices), exemplars_embedding=emb))
        return kl_z1 + kl_z2

    def forward(self, x):
        q2_mu, q2_lv = self.q_z(x)
        z2 = self.reparameterize(q2_mu, q2_lv)
        q1_mu, q1_lv = self.q_z1(x, z2)
        z1 = self.reparameterize(q1_mu, q1_lv)
        p1_mu, p1_lv = self.p_z1(z2)
        mean, logvar = self.p_x(z1, z2)
        return mean, logvar, (z1, q1_mu, q1_lv, z2, q2_mu, q2_lv, p1_mu, p1_lv)
