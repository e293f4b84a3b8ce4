"""Single-latent-layer VAE: how the pieces defined by a concrete model (q_z layers, p_x layers, heads) combine into
KL term, decoder call and forward pass.  Behavioural contract: reference models/AbsModel.py:8-49 (same method
names, argument order and return conventions, so models.VAE / models.fully_conv and the evaluation code work unchanged)."""
import numpy as np
import torch

from models.BaseModel import BaseModel
from utils.distributions import log_normal_diag

_CLAMP_LO, _CLAMP_HI = 1.0 / 512.0, 1.0 - 1.0 / 512.0      # continuous means are kept off 0 / 1 (one half grey level)


class AbsModel(BaseModel):
    def __init__(self, args):
        super().__init__(args)

    # ---- decoder -------------------------------------------------------------------------------------------------
    def _flat_dim(self):
        return int(np.prod(self.args.input_size))

    def _decoder_input(self, z):
        """conv decoders consume the latent as a [bottleneck x H/4 x W/4] map"""
        if 'conv' not in self.args.model_name:
            return z
        side = self.args.input_size[1] // 4
        return z.reshape(-1, self.bottleneck, side, side)

    def p_x(self, z):
        """-> (mean [B x D], log-variance [B x D] or [1 x D] zeros for binary data)"""
        mean = self.p_x_mean(self.p_x_layers(self._decoder_input(z)))
        if hasattr(self.p_x_layers, 'clear_heads'):
            self.p_x_layers.clear_heads()
        D = self._flat_dim()
        kind = self.args.input_type
        if kind == 'binary':
            logvar = torch.zeros(1, D)
        elif self.args.use_logit is False:
            mean = mean.clamp(min=_CLAMP_LO, max=_CLAMP_HI)
            logvar = self.decoder_logstd.expand_as(mean)        # one value: the likelihood kernel reduces its gradient itself
        else:
            # undefined in the reference as well (it falls through with no log-variance bound); fail with a message
            raise UnboundLocalError("AbsModel.p_x: continuous input with use_logit=True has no x_logvar")
        return mean.reshape(-1, D), logvar.reshape(-1, D)

    def generate_x_from_z(self, z, with_reparameterize=True):
        x, _ = self.p_x(z)
        return self.logit_inverse(x) if getattr(self.args, 'use_logit', False) is True else x

    # ---- objective -------------------------------------------------------------------------------------------------
    def kl_loss(self, latent_stats, exemplars_embedding, dataset, cache, x_indices):
        """log q(z|x) - log p(z), one value per row"""
        z, mu, logvar = latent_stats
        emb = exemplars_embedding
        if emb is None and self.args.prior == 'exemplar_prior':
            emb = self.get_exemplar_set(mu, logvar, dataset, cache, x_indices)
        prior_term = self.log_p_z(z=(z, x_indices), exemplars_embedding=emb)
        posterior_term = log_normal_diag(z, mu, logvar, dim=1)
        return posterior_term - prior_term

    def importance_sample_losses(self, data, S, exemplars_embedding):
        """q(z|x) is the same for the S copies of an image: encode the g images once, expand mean / log-variance,
        then sample, decode and score the g * S rows as calculate_loss does (beta = 1, per-row values)."""
        flat = data.reshape(data.size(0), -1)
        mu, logvar = self.q_z(flat)
        mu, logvar = mu.repeat_interleave(S, dim=0), logvar.repeat_interleave(S, dim=0)
        z = self.reparameterize(mu, logvar)
        x_mean, x_logvar = self.p_x(z)
        RE = self.reconstruction_loss(flat.repeat_interleave(S, dim=0), x_mean, x_logvar)
        KL = self.kl_loss((z, mu, logvar), exemplars_embedding, None, None, None)
        return -RE + KL

    def forward(self, x, label=0, num_categories=10):
        mu, logvar = self.q_z(x)
        z = self.reparameterize(mu, logvar)
        return (*self.p_x(z), (z, mu, logvar))
