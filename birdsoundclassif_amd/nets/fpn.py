"""Feature pyramid (reference nets/fpn.py:120-146), HIP forward.

state_dict keys: `fpn.pt_wise.{i}` (bottom-up) and `fpn.out_convs.{i}` ('0' applied to the COARSEST
level, reference fpn.py:137-145 / SURVEY Appendix C-2).  The 3x3 384->256 output convolutions carry 70 %
of the detector's forward FLOPs; they run on the fp32-MFMA implicit-GEMM kernel.
"""
import torch.nn as nn

from . import functional as Fn
from .self_attention import Scaled


class FPN(nn.Module):

    def __init__(self, channels, p_cn, out_cn):
        super().__init__()
        self.pt_wise = nn.ModuleDict({str(i): nn.Conv2d(cn, p_cn, 1) for i, cn in enumerate(channels)})
        self.out_convs = nn.ModuleDict({str(i): nn.Conv2d(p_cn, out_cn, 3, padding=1) for i in range(len(channels))})

    def forward(self, x):
        """x: bottom-up list of NHWC maps (or Scaled(map, factor)) -> bottom-up list of NHWC [B,h,w,out_cn]."""
        # The top-down merge `bilinear(coarser) + lateral` (reference fpn.py:143-144) happens in the epilogue of the
        # lateral 1x1 GEMM, so the lateral maps are never written and re-read on their own.
        outs, merged = [], None
        for i in range(len(x) - 1, -1, -1):                       # coarsest level first
            fm = x[i]
            t, alpha = (fm.tensor, fm.factor) if isinstance(fm, Scaled) else (fm, 1.0)
            c = self.pt_wise[str(i)]
            merged = Fn.conv(t, c.weight, bias=c.bias, alpha=alpha, up=merged)
            oc = self.out_convs[str(len(x) - 1 - i)]
            outs.insert(0, Fn.conv(merged, oc.weight, bias=oc.bias, kh=3, kw=3, pad=1))
        return outs

def build_fpn(args, channels):
    if args.fpn != 'fpn':
        raise ValueError(f'not supported {args.fpn}: BiFPN is outside the hot-path scope (SURVEY.md §8f)')
    return FPN(channels, args.fpn_p_chan, args.out_fpn_chan)
