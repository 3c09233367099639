"""Self-attention pyramid (reference nets/self_attention.py:10-82), HIP forward.

state_dict keys: `attn.attention_modules.{i}.{query,key,value,final_projection}.{weight,bias}` for the
top-n levels; the lower levels hold nn.Identity.  One level = five fp32-MFMA GEMM launches + one row
softmax: [Q|K] projection (one GEMM, N = 2d), V^T projection (bias per row), S = QK^T * (1/denom),
softmax rows, ctx = P V, final projection with the residual `fm + attn(fm)` fused in its epilogue.
"""
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from . import _prep

# A level that the reference passes through nn.Identity comes out as fm + fm (self_attention.py:69,76,
# SURVEY Appendix C-1).  Instead of materialising 2*fm the pyramid hands (fm, 2.0) to the FPN, whose
# lateral 1x1 convolution applies the factor exactly in its epilogue (power of two).
Scaled = namedtuple('Scaled', ['tensor', 'factor'])


class SelfAttention(nn.Module):

    def __init__(self, input_dim, inner_dim, downscale_factor=1, position_encoding=False):
        super().__init__()
        if downscale_factor != 1 or position_encoding:
            raise NotImplementedError('pyramid_top_n_attn == n_levels variant (Appendix C-6) is outside the hot-path scope')
        self.query = nn.Linear(input_dim, inner_dim)
        self.key = nn.Linear(input_dim, inner_dim)
        self.value = nn.Linear(input_dim, inner_dim)
        self.final_projection = nn.Linear(inner_dim, input_dim)
        self.inner_dim = inner_dim

    def forward(self, inpt, residual=True):
        """inpt NHWC [B,h,w,C] -> NHWC [B,h,w,C] = (inpt +) attention(inpt)."""
        B, h, w, Cc = inpt.shape
        L, d = h * w, self.inner_dim
        if L % 32:
            raise NotImplementedError(f'attention over {h}x{w} tokens: the P.V GEMM needs H*W % 32 == 0')
        x2d = inpt.view(B * L, Cc)
        wqk = _prep.cat_rows('qk_w', self.query.weight, self.key.weight)
        bqk = _prep.cat_rows('qk_b', self.query.bias, self.key.bias)
        qk = ops.linear(x2d, wqk, bqk)                                              # [B*L, 2d]
        vt = ops.bgemm_nt(self.value.weight.detach(), inpt.view(B, L, Cc), shift=self.value.bias.detach(),
                          shift_per_row=True)                                          # [B, d, L]
        inv = float(np.float32(1.0) / np.float32(np.round(np.sqrt(d), 2)))            # self_attention.py:47
        s = torch.empty((B, L, L), device=inpt.device, dtype=torch.float32)
        ops.gemm_conv(qk, qk[:, d:], s, B=1, H=L, W=1, Cin=d, N=L, x_ld=2 * d, w_ld=2 * d, groups=B,
                      x_gs=L * 2 * d, w_gs=L * 2 * d, y_gs=L * L, alpha=inv)
        ops.softmax_rows_(s.view(B * L, L))
        ctx = ops.bgemm_nt(s, vt)                                                       # [B, L, d]
        out = ops.linear(ctx.view(B * L, d), self.final_projection.weight.detach(),
                         self.final_projection.bias.detach(), residual=x2d if residual else None)
        return out.view(B, h, w, Cc)


class SAPyramid(nn.Module):

    def __init__(self, channels, top_n):
        super().__init__()
        if top_n == len(channels):
            raise NotImplementedError('pyramid_top_n_attn == n_levels variant is outside the hot-path scope')
        self.attention_modules = nn.ModuleDict({
            str(i): SelfAttention(cn, cn // 2) if (i >= (len(channels) - top_n)) else nn.Identity()
            for (i, cn) in enumerate(channels)})

    def forward(self, x):
        """x: bottom-up list of NHWC maps -> list of `fm + module(fm)`; identity levels as Scaled(fm, 2.0)."""
        out = []
        for i, fm in enumerate(x):
            m = self.attention_modules[str(i)]
            out.append(Scaled(fm, 2.0) if isinstance(m, nn.Identity) else m(fm, residual=True))
        return out


def build_sa_layers(args, channels):
    if type(channels) == tuple:
        raise NotImplementedError('--sandwich_attn is outside the hot-path scope')
    return SAPyramid(channels, args.pyramid_top_n_attn)
