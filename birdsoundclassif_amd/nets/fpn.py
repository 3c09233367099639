"""Feature pyramid (reference nets/fpn.py:120-146), HIP forward.

state_dict keys: `fpn.pt_wise.{i}` (bottom-up) and `fpn.out_convs.{i}` ('0' applied to the COARSEST
level, reference fpn.py:137-145 / SURVEY Appendix C-2).  The 3x3 384->256 output convolutions carry 70 %
of the detector's forward FLOPs; they run on the fp32-MFMA implicit-GEMM kernel.
"""
import torch
import torch.nn as nn

from .. import ops
from . import _prep, functional as Fn
from .layers import DepthwiseSepConv2d
from .self_attention import Deferred, Projected, Scaled, materialize


class FusionModule(nn.Module):
    """reference fpn.py:9-30: ReLU'd learnable weights, `num / (sum + 1e-4)`, then a depthwise-separable block."""

    def __init__(self, n_ends, cn):
        super().__init__()
        self.weights = nn.Parameter(torch.ones(n_ends), requires_grad=True)
        self.conv = DepthwiseSepConv2d(cn, cn)
        self.act = nn.ReLU()

    def forward(self, inputs):
        x2 = inputs[2] if len(inputs) == 3 else None
        return self.conv(Fn.WeightedSum.apply(inputs[0], inputs[1], x2, self.weights))


class Rescale(nn.Module):
    """reference fpn.py:33-44: bilinear (align_corners) resize to the target level, then a 1x1 convolution when the
    channel counts differ."""

    def __init__(self, in_cn, out_cn):
        super().__init__()
        self.up = nn.Upsample(scale_factor=2)              # parameter-free member of the reference, unused there too
        if in_cn != out_cn:
            self.pt_wise = nn.Conv2d(in_cn, out_cn, 1)

    def forward(self, x, out_size):
        out = Fn.UpsampleAdd.apply(x, None, int(out_size[0]), int(out_size[1]))
        if hasattr(self, 'pt_wise'):
            out = Fn.conv(out, self.pt_wise.weight, bias=self.pt_wise.bias)
        return out


class BiFPNLayer(nn.Module):
    """reference fpn.py:47-100; `channels` bottom-up, NHWC maps."""

    def __init__(self, channels, output_channels=None):
        super().__init__()
        n = len(channels)
        self.rescalings_td = nn.ModuleDict({str(i + 1): Rescale(in_cn, out_cn)
                                            for i, (out_cn, in_cn) in enumerate(zip(channels[:-1], channels[1:]))})
        self.rescalings_bu = nn.ModuleDict({str(i): Rescale(in_cn, out_cn)
                                            for i, (in_cn, out_cn) in enumerate(zip(channels[:-1], channels[1:]))})
        self.fusions_td = nn.ModuleDict({str(i + 1): FusionModule(2, cn) for i, cn in enumerate(channels[1:-1])})
        self.fusions_bu = nn.ModuleDict({str(i): FusionModule(2 if i in [0, n - 1] else 3, cn) for i, cn in enumerate(channels)})
        if output_channels is not None:
            self.out_pt_wise_convs = nn.ModuleDict({str(i): nn.Conv2d(cn, output_channels, 1) for i, cn in enumerate(channels)})

    def forward(self, inputs):
        n = len(inputs)
        sizes = [tuple(e.shape[1:3]) for e in inputs]
        td_out = inputs[-1]
        td = [td_out]
        for i in range(n - 2, 0, -1):                                     # top-down
            td_out = self.fusions_td[str(i)]([inputs[i], self.rescalings_td[str(i + 1)](td_out, sizes[i])])
            td.insert(0, td_out)
        td.insert(0, self.rescalings_td['1'](td_out, sizes[0]))
        bu_out = self.fusions_bu['0']([inputs[0], td[0]])                 # bottom-up
        bu = [bu_out]
        for i in range(1, n - 1):
            bu_out = self.fusions_bu[str(i)]([inputs[i], td[i], self.rescalings_bu[str(i - 1)](bu_out, sizes[i])])
            bu.append(bu_out)
        bu.append(self.fusions_bu[str(n - 1)]([inputs[-1], self.rescalings_bu[str(n - 2)](bu_out, sizes[-1])]))
        if hasattr(self, 'out_pt_wise_convs'):
            bu = [Fn.conv(b, c.weight, bias=c.bias) for b, c in ((bu[i], self.out_pt_wise_convs[str(i)]) for i in range(n))]
        return bu


class BiFPN(nn.Module):
    """reference fpn.py:103-115 (`--fpn bifpn`)."""

    def __init__(self, n_layers, channels, out_cn):
        super().__init__()
        self.layers = nn.ModuleList([BiFPNLayer(channels, out_cn if i == (n_layers - 1) else None) for i in range(n_layers)])

    def forward(self, x):
        x = materialize(x)                      # doubled identity levels of SAPyramid: no lateral GEMM to absorb the factor
        for layer in self.layers:
            x = layer(x)
        return x


class FPN(nn.Module):

    def __init__(self, channels, p_cn, out_cn):
        super().__init__()
        self.pt_wise = nn.ModuleDict({str(i): nn.Conv2d(cn, p_cn, 1) for i, cn in enumerate(channels)})
        self.out_convs = nn.ModuleDict({str(i): nn.Conv2d(p_cn, out_cn, 3, padding=1) for i in range(len(channels))})

    def forward(self, x, lazy_strides=None):
        """x: bottom-up list of NHWC maps (or Scaled(map, factor)) -> bottom-up list of NHWC [B,h,w,out_cn].
        `lazy_strides` {level: stride}: that output map is only read by a 3x3 / stride convolution (the RPN's convolution of
        the level) and by the RoI pooling: it is computed on demand (Fn.conv), pixels nobody reads stay unwritten."""
        # The top-down merge `bilinear(coarser) + lateral` (reference fpn.py:143-144) happens in the epilogue of the
        # lateral 1x1 GEMM, so the lateral maps are never written and re-read on their own.
        outs, merged = [], None
        for i in range(len(x) - 1, -1, -1):                       # coarsest level first
            fm = x[i]
            c = self.pt_wise[str(i)]
            if isinstance(fm, Deferred):
                # training: attention module + lateral (+ merge) as one tape node, composed (functional.AttnLateral)
                m = fm.module
                merged = Fn.AttnLateral.apply(fm.tensor, m.query.weight, m.query.bias, m.key.weight, m.key.bias, m.value.weight, m.value.bias,
                                              m.final_projection.weight, m.final_projection.bias, c.weight, c.bias, merged, fm.inv)
                oc_ = self.out_convs[str(len(x) - 1 - i)]
                outs.insert(0, Fn.conv(merged, oc_.weight, bias=oc_.bias, kh=3, kw=3, pad=1, lazy_stride=(lazy_strides or {}).get(i),
                                       accept_stash=i > 0))
                continue
            if isinstance(fm, Projected):
                # evaluation mode: the level is fm + ctx W_o^T + b_o with the projection still to do -- the lateral takes it into its own
                # weights: W_l fm + shift, then (W_l W_o) ctx + that + the top-down merge
                wc, sh = _prep.lateral_of_projection(c.weight, c.bias, fm.wo, fm.bo)
                if fm.lateral:                         # ctx is already W_l W_o ctx: the lateral itself, + ctx through the residual input
                    merged = ops.conv2d(fm.tensor, _prep.krsc(c.weight), shift=sh, residual=fm.ctx, up=merged)
                else:
                    y1 = ops.conv2d(fm.tensor, _prep.krsc(c.weight), shift=sh)
                    merged = ops.conv2d(fm.ctx, wc, residual=y1, up=merged)
                outs.insert(0, Fn.conv(merged, self.out_convs[str(len(x) - 1 - i)].weight, bias=self.out_convs[str(len(x) - 1 - i)].bias,
                                       kh=3, kw=3, pad=1, lazy_stride=(lazy_strides or {}).get(i), accept_stash=i > 0))
                continue
            t, alpha = (fm.tensor, fm.factor) if isinstance(fm, Scaled) else (fm, 1.0)
            # the lateral of a demand-driven level is itself only evaluated where that level's output convolution reads it;
            # the finest level has no finer level that would read `merged`
            oc = self.out_convs[str(len(x) - 1 - i)]
            lz = (lazy_strides or {}).get(i) if i == 0 and Fn.lazy3x3_ok(t.shape[1], t.shape[2], c.weight.shape[0], oc.weight) else None
            merged = Fn.conv(t, c.weight, bias=c.bias, alpha=alpha, up=merged, lazy_stride=lz)
            # (i > 0: `merged` is also the `up` of the next finer lateral, whose backward pass runs first and hands its share of the
            # gradient over: Fn._STASH)
            outs.insert(0, Fn.conv(merged, oc.weight, bias=oc.bias, kh=3, kw=3, pad=1,
                                   lazy_stride=(lazy_strides or {}).get(i), accept_stash=i > 0))
        return outs

def build_fpn(args, channels):
    if args.fpn == 'bifpn':
        return BiFPN(args.n_bifpn_layers, channels, args.out_fpn_chan)
    if args.fpn != 'fpn':
        raise ValueError(f'not supported {args.fpn}')
    return FPN(channels, args.fpn_p_chan, args.out_fpn_chan)
