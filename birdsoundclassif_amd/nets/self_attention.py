"""Self-attention pyramid (reference nets/self_attention.py:10-82), HIP forward.

state_dict keys: `attn.attention_modules.{i}.{query,key,value,final_projection}.{weight,bias}` for the
top-n levels; the lower levels hold nn.Identity.  One level = five fp32-MFMA GEMM launches + one row
softmax: [Q|K|V] projection (one GEMM, N = 3d), S = QK^T * (1/denom) (NT), softmax rows, ctx = P V (NN),
final projection with the residual `fm + attn(fm)` fused in its epilogue; the backward is hand-written
(nets/functional.py:Attention).
"""
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from . import _prep, functional as Fn

# A level that the reference passes through nn.Identity comes out as fm + fm (self_attention.py:69,76,
# SURVEY Appendix C-1).  Instead of materialising 2*fm the pyramid hands (fm, 2.0) to the FPN, whose
# lateral 1x1 convolution applies the factor exactly in its epilogue (power of two).
Scaled = namedtuple('Scaled', ['tensor', 'factor'])
# Evaluation mode, pyramid straight in front of the plain FPN: an attention level is handed over unfinished.  Its value is
# fm + ctx W_o^T + b_o, but the one consumer, the FPN's lateral 1x1 (W_l, b_l), is linear too:
#   * `lateral=False`: (fm, ctx, W_o, b_o) -- the lateral takes the projection into its own weights (fpn.FPN.forward,
#     _prep.lateral_of_projection): 2 C d multiply-adds per token become 2 p d (p = 384 lateral channels, C = 1024 / 2048, d = C / 2);
#   * `lateral=True` (the pyramid was given the laterals): `ctx` is already W_l W_o times the module's context, [.., p] -- the VALUE
#     projection was composed with W_o and W_l (_prep.value_of_lateral), so values, P V and the hand-over are p wide instead of d and no
#     projection GEMM is left at all: the lateral adds `ctx` through its residual input.
Projected = namedtuple('Projected', ['tensor', 'ctx', 'wo', 'bo', 'lateral'])
# The same composition with a gradient to come: the level is not computed at all here -- the FPN runs module + lateral as ONE tape node
# (functional.AttnLateral) on (fm, this module, 1 / denominator).
Deferred = namedtuple('Deferred', ['tensor', 'module', 'inv'])


class SelfAttention(nn.Module):
    """reference self_attention.py:10-56.  `downscale_factor` / `position_encoding` are the `pyramid_top_n_attn ==
    n_levels` configuration: the input gets + 0.5 * frequency encoding, and with a downscale factor the module returns the
    down-and-up interpolated INPUT -- the attention the reference computes there is discarded (:52-54, SURVEY Appendix
    C-6), so it is not computed here."""

    def __init__(self, input_dim, inner_dim, downscale_factor=1, position_encoding=False):
        super().__init__()
        self.query = nn.Linear(input_dim, inner_dim)
        self.key = nn.Linear(input_dim, inner_dim)
        self.value = nn.Linear(input_dim, inner_dim)
        self.final_projection = nn.Linear(inner_dim, input_dim)
        self.inner_dim = inner_dim
        self.downscale_factor = downscale_factor
        self.position_encoding = position_encoding
        self._pe = {}

    def _half_pe(self, h, w, c, device, sign):
        """+-0.5 * one_dimension_positional_encoding(h, c) as an NHWC table [h, w, c] (self_attention.py:28-31)."""
        key = (h, w, c, str(device), sign)
        if key not in self._pe:
            from .position_encoding import one_dimension_positional_encoding
            t = (0.5 * sign) * one_dimension_positional_encoding(h, c)
            self._pe[key] = t[:, None, :].expand(h, w, c).contiguous().to(device)
        return self._pe[key]

    def forward(self, inpt, residual=True, defer_projection=False):
        """inpt NHWC [B,h,w,C] -> NHWC [B,h,w,C] = (inpt +) module(inpt); `defer_projection` (no gradient, plain level): `Projected`."""
        B, h, w, Cc = inpt.shape
        if not residual:
            raise NotImplementedError('SelfAttention without the SAPyramid residual is not on the hot path')
        x = inpt
        if self.position_encoding:
            x = Fn.AddConst.apply(inpt, self._half_pe(h, w, Cc, inpt.device, 1.0))
        if self.downscale_factor > 1:
            f = self.downscale_factor
            small = Fn.UpsampleAdd.apply(x, None, h // f, w // f)
            return Fn.Add.apply(inpt, Fn.UpsampleAdd.apply(small, None, h, w))
        L, d = h * w, self.inner_dim
        if L % 32:
            raise NotImplementedError(f'attention over {h}x{w} tokens: the P.V GEMM needs H*W % 32 == 0')
        inv = float(np.float32(1.0) / np.float32(np.round(np.sqrt(d), 2)))            # self_attention.py:47
        if isinstance(defer_projection, nn.Conv2d) and not self.position_encoding and torch.is_grad_enabled() and d % 32 == 0 \
                and defer_projection.weight.shape[0] % 32 == 0 and defer_projection.weight.shape[1] == Cc and Cc % 32 == 0:
            return Deferred(inpt, self, inv)
        if defer_projection and not self.position_encoding and not torch.is_grad_enabled() and d % 32 == 0:
            wv, bv, lat = self.value.weight, self.value.bias, isinstance(defer_projection, nn.Conv2d)
            if lat:                                    # the FPN lateral that will read this level
                if defer_projection.weight.shape[0] % 32 or defer_projection.weight.shape[1] != Cc:
                    raise ValueError('the lateral handed to SelfAttention does not read this level')
                wv, bv = _prep.value_of_lateral(defer_projection.weight, self.final_projection.weight, wv, bv)
            _, _, cx, _ = Fn.attention_context(x.view(B, L, Cc), self.query.weight, self.query.bias, self.key.weight, self.key.bias, wv, bv, inv)
            return Projected(inpt, cx.view(B, h, w, -1), self.final_projection.weight, self.final_projection.bias, lat)
        out = Fn.Attention.apply(x.view(B, L, Cc), self.query.weight, self.query.bias, self.key.weight, self.key.bias,
                                 self.value.weight, self.value.bias, self.final_projection.weight,
                                 self.final_projection.bias, inv).view(B, h, w, Cc)        # = x + attention(x)
        if self.position_encoding:                     # the pyramid's residual is the ORIGINAL map, not map + encoding
            out = Fn.AddConst.apply(out, self._half_pe(h, w, Cc, inpt.device, -1.0))
        return out


class SAPyramid(nn.Module):

    def __init__(self, channels, top_n):
        super().__init__()
        if top_n == len(channels):                     # self_attention.py:63-66
            self.attention_modules = nn.ModuleDict({str(i): SelfAttention(cn, cn, max(1, 2 ** (3 - i)), True)
                                                    for (i, cn) in enumerate(channels)})
        else:
            self.attention_modules = nn.ModuleDict({
                str(i): SelfAttention(cn, cn // 2) if (i >= (len(channels) - top_n)) else nn.Identity()
                for (i, cn) in enumerate(channels)})

    def forward(self, x, defer_projection=False):
        """x: bottom-up list of NHWC maps -> list of `fm + module(fm)`; identity levels as Scaled(fm, 2.0); with `defer_projection`
        (True, or the FPN's laterals {str(level): nn.Conv2d}: the caller is the plain FPN, evaluation mode) attention levels as `Projected`."""
        out = []
        for i, fm in enumerate(x):
            m = self.attention_modules[str(i)]
            dp = defer_projection[str(i)] if isinstance(defer_projection, (dict, nn.ModuleDict)) else defer_projection
            out.append(Scaled(fm, 2.0) if isinstance(m, nn.Identity) else m(fm, residual=True, defer_projection=dp))
        return out


def materialize(levels):
    """Scaled(fm, f) -> f * fm: for consumers that cannot fold the factor into a GEMM epilogue (the heads, when the
    pyramid runs AFTER the FPN: --fpn_first / --sandwich_attn)."""
    def one(l):
        if isinstance(l, Scaled):
            return Fn.Scale.apply(l.tensor, l.factor)
        if isinstance(l, Deferred):
            raise ValueError('a level deferred to the FPN has no value of its own (only functional.AttnLateral computes it, lateral included)')
        if isinstance(l, Projected):
            if l.lateral:
                raise ValueError('a level handed over in its lateral form has no value of its own (only the FPN lateral it was made for reads it)')
            B, h, w, Cc = l.tensor.shape
            return ops.linear(l.ctx.reshape(B * h * w, -1), l.wo.detach(), l.bo.detach(), residual=l.tensor.reshape(B * h * w, Cc)).view(B, h, w, Cc)
        return l
    return [one(l) for l in levels]


def build_sa_layers(args, channels):
    """reference self_attention.py:79-82: a pair of pyramids for --sandwich_attn (before and after the FPN)."""
    if type(channels) == tuple:
        return nn.ModuleList([SAPyramid(cns, args.pyramid_top_n_attn) for cns in channels])
    return SAPyramid(channels, args.pyramid_top_n_attn)
