"""Two-latent-layer VAE (z2 under the exemplar prior, z1 | z2 Gaussian): composition of the sub-networks a concrete
model declares (models.HVAE_2level, models.convHVAE_2level).  Behavioural contract: reference
models/AbsHModel.py:8-107 -- method names, argument order, the 8-tuple of latent statistics and the flat
[B x D] decoder outputs are what models.BaseModel and utils.evaluation rely on."""
import os

import numpy as np
import torch

from models.BaseModel import BaseModel
from utils.distributions import log_normal_diag

# training step of the dense 2-level model on two streams (calculate_loss below); EVAE_HVAE_TWO_STREAM=0: one stream
_TWO_STREAM = os.environ.get("EVAE_HVAE_TWO_STREAM", "1") != "0"
def _two_stream_conv():     # the convolutional 2-level model too (r05: c3 31.4 -> 29.9 ms); read per call so that a test can switch it
    return os.environ.get("EVAE_HVAE_TWO_STREAM_CONV", "1") != "0"

# ... with each pair of heads + its sample + its log-density as one Function, and the loss assembly as one (evae.ops.HeadsReparamFn,
# ElboFn: ~40 launches fewer per step); EVAE_HVAE_FUSED_HEADS=0: the separate modules
_FUSED_HEADS = os.environ.get("EVAE_HVAE_FUSED_HEADS", "1") != "0"

# EVAE_HVAE_LEAF_STREAM=1: the thin layers' weight gradients as autograd nodes on a third stream (evae.ops.gated_dense_split).  Off:
# measured 0.861 -> 0.998 ms at c4 -- every cross-stream edge of a replayed graph costs the chain ~10 us on this runtime (r03)
_LEAF_STREAM = os.environ.get("EVAE_HVAE_LEAF_STREAM", "0") != "0"

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

    def _sample_heads(self, trunk, mean_head, logvar_head, zdim):
        """(z, mean, log-variance, log q(z | .)) of a pair of heads on `trunk`: the sample of reparameterize() (same noise draw)
        and log_normal_diag(z, mean, logvar) of kl_loss -- through ONE Function when the heads are a Linear and a
        Hardtanh-clamped Linear (every model of the reference: models/HVAE_2level.py:23-24,36-37)."""
        from evae import ops
        from utils.nn import HipLinear, NonLinear
        if (_FUSED_HEADS and trunk.is_cuda and trunk.dim() == 2 and isinstance(mean_head, HipLinear)
                and isinstance(logvar_head, NonLinear) and isinstance(logvar_head.activation, torch.nn.Hardtanh)):
            eps = self._draw_eps(torch.empty((trunk.shape[0], zdim), device=trunk.device, dtype=torch.float32))
            act = logvar_head.activation
            return ops.heads_reparam(trunk, mean_head.weight, mean_head.bias, logvar_head.linear.weight, logvar_head.linear.bias,
                                     eps, act.min_val, act.max_val)
        mu, lv = mean_head(trunk).view(-1, zdim), logvar_head(trunk).view(-1, zdim)
        z = self.reparameterize(mu, lv).view(-1, zdim)
        return z, mu, lv, log_normal_diag(z, mu, lv, dim=1)

    def _density_heads(self, trunk, mean_head, logvar_head, zq, zdim):
        """(mean, log-variance, log N(zq | mean, exp(logvar))) of a pair of heads on `trunk` -- p(z1 | z2) and its density term"""
        from evae import ops
        from utils.nn import HipLinear, NonLinear
        if (_FUSED_HEADS and trunk.is_cuda and trunk.dim() == 2 and isinstance(mean_head, HipLinear)
                and isinstance(logvar_head, NonLinear) and isinstance(logvar_head.activation, torch.nn.Hardtanh)):
            act = logvar_head.activation
            return ops.heads_density(trunk, mean_head.weight, mean_head.bias, logvar_head.linear.weight, logvar_head.linear.bias,
                                     zq.view(-1, zdim), act.min_val, act.max_val)
        mu, lv = mean_head(trunk).view(-1, zdim), logvar_head(trunk).view(-1, zdim)
        return mu, lv, log_normal_diag(zq.view(-1, zdim), mu, lv, dim=1)

    def _two_stream_path(self, x, x_indices, exemplars_embedding, dataset):
        a = self.args
        return (_TWO_STREAM and self.training and a.prior == 'exemplar_prior' and a.approximate_prior is False
                and exemplars_embedding is None and dataset is not None and x_indices is not None and x.is_cuda
                and torch.is_grad_enabled() and (not self._is_conv() or _two_stream_conv()) and not self._sharded())

    def calculate_loss(self, x, beta=1., average=False, exemplars_embedding=None, cache=None, dataset=None):
        """Training step with the exact exemplar prior on one device (reference models/BaseModel.py:54-77 over AbsHModel.py:13-106):
        the same modules and the same arithmetic as the one-stream path, issued on TWO streams -- the batch rows' path (eleven
        thin dense layers, ~150 launches of a few microseconds each with their backward) on a side stream, the exemplar rows'
        encoder q(z2 | .) and the prior on the caller's stream; they meet at z2 (forward) and at log p(z2).  Autograd runs
        every node's backward on the stream of its forward, so the two halves of the backward pass overlap the same way.
        r03: c4 (11 500 exemplars) 1.23 -> see DESIGN section 4; the exemplar side was waiting behind the thin launches."""
        xx, x_indices = x
        if not self._two_stream_path(xx, x_indices, exemplars_embedding, dataset):
            return super().calculate_loss(x, beta, average, exemplars_embedding, cache, dataset)
        from evae import ops
        main = torch.cuda.current_stream()
        side = ops.model_side_stream(xx.device)        # registered: what runs there takes its own kernel workspaces
        side.wait_stream(main)
        d1, d2 = self.args.z1_size, self.args.z2_size
        leaf = ops.model_leaf_stream(xx.device) if _LEAF_STREAM else None
        with torch.cuda.stream(side), ops.leaf_branch(leaf):
            # forward() of this class, in its order (the two reparameterize calls draw z2's noise, then z1's), minus p(z1 | z2)
            xin = xx.view(-1, *self.args.input_size) if self._is_conv() else xx
            conv = self._is_conv()
            flat = (lambda t: t.reshape(t.size(0), -1)) if conv else (lambda t: t)       # conv features -> (c, y, x) rows, as q_z / q_z1 do
            z2, q2_mu, q2_lv, log_q2 = self._sample_heads(flat(self.q_z_layers(xin)), self.q_z_mean, self.q_z_logvar, d2)
            z2_ready = torch.cuda.Event(); z2_ready.record()
            joint = self.q_z1_layers_joint(torch.cat((flat(self.q_z1_layers_x(xin)), self.q_z1_layers_z2(z2)), dim=1))       # q_z1()
            z1, q1_mu, q1_lv, log_q1 = self._sample_heads(joint, self.q_z1_mean, self.q_z1_logvar, d1)
            z1_ready = torch.cuda.Event(); z1_ready.record()
            x_mean, x_logvar = self.p_x(z1, z2)
            x_flat = xx.reshape(xx.shape[0], -1) if xx.dim() != 2 else xx
            RE = self.reconstruction_loss(x_flat, x_mean, x_logvar)
        # this stream: the exemplar rows' encoder (independent of the batch: it starts at once), then what hangs off z2 alone --
        # the prior and the p(z1 | z2) branch (its eight launches and their backward come off the longer chain)
        emb = self.get_exemplar_set(q2_mu, q2_lv, dataset, cache, x_indices)
        main.wait_event(z2_ready)
        z2.record_stream(main)
        log_prior = self.log_p_z(z=(z2.view(-1, d2), x_indices), exemplars_embedding=emb)
        trunk1 = self.p_z1_layers_z2(z2)                     # p_z1(): trunk, then the two heads (with the density of z1)
        main.wait_event(z1_ready)
        z1.record_stream(main)
        p1_mu, p1_lv, log_p1 = self._density_heads(trunk1, self.p_z1_mean, self.p_z1_logvar, z1, d1)
        lp_ready = torch.cuda.Event(); lp_ready.record()
        log_prior.record_stream(side); log_p1.record_stream(side)
        with torch.cuda.stream(side):
            side.wait_event(lp_ready)
            if _FUSED_HEADS:
                loss, RE, KL = ops.elbo(RE, log_q1, log_p1, beta, average, log_q2, log_prior)
            else:
                KL = (log_q1 - log_p1) + (log_q2 - log_prior)           # (the grouping of kl_loss)
                loss = -RE + beta * KL
                if average:
                    loss, RE, KL = torch.mean(loss), torch.mean(RE), torch.mean(KL)
        main.wait_stream(side)
        for t in (loss, RE, KL):
            t.record_stream(main)
        return loss, RE, KL

    def forward(self, x):
        q2_mu, q2_lv = self.q_z(x)
        z2 = self.reparameterize(q2_mu, q2_lv)
        q1_mu, q1_lv = self.q_z1(x, z2)
        z1 = self.reparameterize(q1_mu, q1_lv)
        p1_mu, p1_lv = self.p_z1(z2)
        mean, logvar = self.p_x(z1, z2)
        return mean, logvar, (z1, q1_mu, q1_lv, z2, q2_mu, q2_lv, p1_mu, p1_lv)
