"""Fully-convolutional weight-normed residual VAE, model_name='single_conv' (reference
models/fully_conv.py:7-81), same submodule names / state_dict entries (including the BatchNorm2d the
reference's residual block builds but never calls)."""
import os

import torch
import torch.nn as nn
from torch.nn.utils import weight_norm

from models.AbsModel import AbsModel
from evae import ops
from utils.nn import HipConv2d


class block(nn.Module):
    """x + conv(ELU(x)) with a weight-normed 3x3 convolution."""

    def __init__(self, input_size, output_size, stride=1, kernel=3, padding=1):
        super().__init__()
        self.normalization = nn.BatchNorm2d(input_size)       # present in the state_dict, unused (as in the reference)
        self.conv1 = weight_norm(HipConv2d(input_size, output_size, kernel_size=kernel, stride=stride,
                                           padding=padding, bias=True))
        self.activation = torch.nn.ELU()
        self.f = torch.nn.Sequential(self.activation, self.conv1)

    def forward(self, x):
        conv = self.conv1
        if x.is_cuda and os.environ.get("EVAE_RESBLOCK", "1") != "0" and ops.res_block_supported(x, conv.weight_v, conv.stride, conv.padding):
            for hook in conv._forward_pre_hooks.values():      # weight_norm: weight = g * v / ||v|| (differentiable)
                hook(conv, (x,))
            return ops.res_block(x, conv.weight, conv.bias)
        return x + self.f(x)


def _is_wn(m):
    return isinstance(m, HipConv2d) and hasattr(m, "weight_v")


class _Seq(nn.Sequential):
    """nn.Sequential (same children, same state_dict keys) that (a) normalises the weights of ALL its weight-normed convolutions (and
    of the `heads` that read its output) in one launch (evae.ops.weight_norm_set) instead of one hook launch per layer, and (b) hands
    every run of consecutive residual blocks to ONE operator over pre-split pixel images (evae.ops.ResStackFn) when the tensor is
    large enough to fill the machine."""

    heads = ()                                             # weight-normed convolutions applied to this stack's output

    def clear_heads(self):
        """Drop this pass's head weights that no head fetched (q_z(prior=True) never calls q_z_logvar, so every training
        step left a non-leaf, grad-requiring tensor -- and through it the whole weight-norm graph -- on the module: copy.deepcopy raised
        'only graph leaves support deepcopy', torch.save pickled it, a later head call could have used a stale weight)."""
        self.__dict__.pop("_head_w", None)

    def __getstate__(self):
        st = self.__dict__.copy()
        st.pop("_head_w", None)                            # never part of a copy / a pickle
        return st

    def head_weight(self, m):
        """the weight of head `m` from this pass's set (None: not computed -> the module's own hook does it)"""
        return self._head_w.pop(id(m), None) if getattr(self, "_head_w", None) else None

    def forward(self, x):
        mods = list(self)
        fused = x.is_cuda and os.environ.get("EVAE_RESBLOCK", "1") != "0"
        wn = {}
        self._head_w = {}
        # (the set only where launches count -- the pixel threshold of the residual-stack operator: the seeded golden models of
        # tests/test_gpu_model.py (G9 / G21, 4 images) are conditioned so badly that one ulp in every weight_g moves their gradients by
        # 1e-4 .. 1e-2 (tools/wn_cmp.py); they keep torch's own per-layer kernel, the yardstick their tolerances were set with)
        if fused and os.environ.get("EVAE_WN_SET", "1") != "0" and x.shape[0] * x.shape[2] * x.shape[3] >= ops.RES_STACK_MIN_PIXELS // 4:
            convs = [m.conv1 if isinstance(m, block) else m for m in mods if isinstance(m, block) or _is_wn(m)]
            convs = [c for c in convs if _is_wn(c)] + [h for h in self.heads if _is_wn(h)]
            if convs:
                ws = ops.weight_norm_set([(c.weight_v, c.weight_g) for c in convs])
                wn = {id(c): w for c, w in zip(convs, ws)}
                self._head_w = {id(h): wn[id(h)] for h in self.heads if id(h) in wn}

        def weight(conv):                                  # weight_norm: weight = g * v / ||v|| (differentiable)
            w = wn.get(id(conv))
            if w is None:
                for hook in conv._forward_pre_hooks.values():
                    hook(conv, (x,))
                w = conv.weight
            return w

        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, block) and fused:
                j = i
                while j < len(mods) and isinstance(mods[j], block):
                    j += 1
                run = mods[i:j]
                if ops.res_stack_supported(x, [b.conv1.weight_v for b in run]):
                    x = ops.res_stack(x, [(weight(b.conv1), b.conv1.bias) for b in run])
                    i = j
                    continue
                if ops.res_block_supported(x, m.conv1.weight_v, m.conv1.stride, m.conv1.padding):
                    x = ops.res_block(x, weight(m.conv1), m.conv1.bias)
                    i += 1
                    continue
            if (fused and isinstance(m, nn.Upsample) and m.scale_factor in (2, 2.0) and m.mode == "nearest" and i + 1 < len(mods) and _is_wn(mods[i + 1])
                    and os.environ.get("EVAE_PLAIN_CONV", "1") != "0"):
                c = mods[i + 1]
                w = weight(c)
                if ops.plain_conv_supported(x, w, c.stride, c.padding, upsample=True):      # the upsampling happens in the image pack
                    elu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ELU)
                    x = ops.plain_conv(x, w, c.bias, c.stride, elu=elu, upsample=True)
                    i += 3 if elu else 2
                    continue
            if fused and _is_wn(m):
                w = weight(m)
                elu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ELU)
                if os.environ.get("EVAE_PLAIN_CONV", "1") != "0" and ops.plain_conv_supported(x, w, m.stride, m.padding):
                    x = ops.plain_conv(x, w, m.bias, m.stride, elu=elu)            # (the ELU behind it in its epilogue)
                    i += 2 if elu else 1
                    continue
                x = ops.conv2d(x, w, m.bias, m.stride, m.padding)
            else:
                x = m(x)
            i += 1
        return x


def _wn_conv(cin, cout, stride=1):
    return weight_norm(HipConv2d(in_channels=cin, out_channels=cout, kernel_size=3, stride=stride, padding=1))


class _HeadConv(HipConv2d):
    """A weight-normed convolution applied to a _Seq's output: takes its weight from that stack's one-launch set when there is one
    (no hook launch), else behaves like any weight-normed HipConv2d."""

    def __call__(self, x):
        stack = self.__dict__.get("_stack")
        w = stack.head_weight(self) if stack is not None else None
        if w is not None:
            if os.environ.get("EVAE_PLAIN_CONV", "1") != "0" and ops.plain_conv_supported(x, w, self.stride, self.padding):
                return ops.plain_conv(x, w, self.bias, self.stride)
            return ops.conv2d(x, w, self.bias, self.stride, self.padding)
        return super().__call__(x)


def _wn_head(stack, cin, cout):
    m = weight_norm(_HeadConv(in_channels=cin, out_channels=cout, kernel_size=3, stride=1, padding=1))
    m.__dict__["_stack"] = stack                                   # (not a submodule: no state_dict entry, no cycle in .modules())
    stack.heads = tuple(stack.heads) + (m,)
    return m


class VAE(AbsModel):
    def __init__(self, args):
        super().__init__(args)

    def create_model(self, args, train_data_size=None):
        self.train_data_size = train_data_size
        self.cs = 48
        self.bottleneck = self.args.bottleneck
        cs, c_in = self.cs, self.args.input_size[0]
        self.q_z_layers = _Seq(
            _wn_conv(c_in, cs, 2), nn.ELU(), *[block(cs, cs) for _ in range(6)],
            _wn_conv(cs, cs * 2, 2), nn.ELU(), *[block(cs * 2, cs * 2) for _ in range(6)])
        self.q_z_mean = _wn_head(self.q_z_layers, cs * 2, self.bottleneck)
        self.q_z_logvar = _wn_head(self.q_z_layers, cs * 2, self.bottleneck)
        self.p_x_layers = _Seq(
            nn.Upsample(scale_factor=2), _wn_conv(self.bottleneck, cs * 2), nn.ELU(),
            *[block(cs * 2, cs * 2) for _ in range(6)],
            nn.Upsample(scale_factor=2), _wn_conv(cs * 2, cs), nn.ELU(), *[block(cs, cs) for _ in range(6)])
        if self.args.input_type == 'binary':
            self.p_x_mean = nn.Sequential(HipConv2d(cs, c_in, kernel_size=3, stride=1, padding=1), nn.Sigmoid())
        elif self.args.input_type in ('gray', 'continuous'):
            self.p_x_mean = _wn_head(self.p_x_layers, cs, c_in)
            self.p_x_logvar = HipConv2d(cs, c_in, kernel_size=3, stride=1, padding=1)
