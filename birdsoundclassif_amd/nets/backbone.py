"""Backbone modules (reference nets/backbone.py), HIP forward.

Same classes / constructor arguments / state_dict keys as the reference: `backbone.0.init_conv.*`,
`backbone.0.body.<torchvision ResNet-50 names>`.  torchvision is not a dependency: the ResNet-50 v1.5
definition (public architecture, reference backbone.py:131 `getattr(torchvision.models, name)`) is
stated here as parameter containers, and the forward is a chain of fp32-MFMA implicit-GEMM launches with
FrozenBatchNorm affine + ReLU + residual fused into the epilogues.  Activations are NHWC.
"""
import torch
from torch import nn

from .. import ops
from . import _prep, functional as Fn
from .position_encoding import build_position_encoding

bcbk_channels = {'resnet': {'2': 64, '3': 256, '4': 512, '5': 1024, '6': 2048}}
_RESNET_LAYERS = {'resnet50': [3, 4, 6, 3], 'resnet101': [3, 4, 23, 3], 'resnet152': [3, 8, 36, 3]}


class FrozenBatchNorm2d(nn.Module):
    """Fixed statistics + affine (reference backbone.py:26-62, eps = 1e-5); never run on its own: its
    (scale, shift) are folded into the producing convolution's epilogue."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer('weight', torch.ones(n))
        self.register_buffer('bias', torch.zeros(n))
        self.register_buffer('running_mean', torch.zeros(n))
        self.register_buffer('running_var', torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        state_dict.pop(prefix + 'num_batches_tracked', None)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def affine(self):
        return _prep.bn_affine(self.weight, self.bias, self.running_mean, self.running_var, 1e-5, frozen=True)


class _Bottleneck(nn.Module):
    """ResNet v1.5 bottleneck (stride on the 3x3)."""

    def __init__(self, inplanes, planes, stride, downsample, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.downsample = downsample
        self.stride = stride
        # set by _ResNetBody: is the input a ReLU output / does the output feed anything besides the next block
        self.mask_input, self.mask_gy = False, True

    def forward(self, x):
        if torch.is_grad_enabled() and x.dim() == 4 and self.conv1.weight.shape[0] % 32 == 0:
            ds = self.downsample
            sd, bd = ds[1].affine() if ds is not None else (None, None)
            return Fn.Bottleneck.apply(x, self.conv1.weight, self.conv2.weight, self.conv3.weight,
                                       ds[0].weight if ds is not None else None, *self.bn1.affine(), *self.bn2.affine(),
                                       *self.bn3.affine(), sd, bd, self.stride, self.mask_input, self.mask_gy)
        s, b = self.bn1.affine()
        o = Fn.conv(x, self.conv1.weight, scale=s, shift=b, act=ops.ACT_RELU)
        s, b = self.bn2.affine()
        if self.stride == 1 and Fn._winograd_ok(o, self.conv2.weight, 3, 3, 1, 1):
            # >= 128 channels: Winograd F(2x2,3x3) with FrozenBN + ReLU in the output transform (same as Fn.Bottleneck)
            o = ops.conv3x3_winograd(o, _prep.wino23(self.conv2.weight), b, scale=s, relu=True)
        else:
            o = Fn.conv(o, self.conv2.weight, scale=s, shift=b, kh=3, kw=3, stride=self.stride, pad=1, act=ops.ACT_RELU)
        if self.downsample is not None:
            s, b = self.downsample[1].affine()
            idt = Fn.conv(x, self.downsample[0].weight, scale=s, shift=b, stride=self.stride)
        else:
            idt = x
        s, b = self.bn3.affine()
        return Fn.conv(o, self.conv3.weight, scale=s, shift=b, residual=idt, act=ops.ACT_RELU)


class _ResNetBody(nn.Module):
    """conv1/bn1/relu/maxpool/layer1..4 with torchvision's names; returns the 5 taps the reference takes
    with IntermediateLayerGetter (backbone.py:82-85): relu, layer1..layer4."""

    def __init__(self, layers, norm_layer, dilation=False):
        """`dilation` (reference backbone.py:129-131: torchvision's `replace_stride_with_dilation=[False, False, True]`): layer4 keeps
        layer3's resolution -- its first block runs with stride 1 (3x3 dilation 1, like torchvision: the first block uses the
        PREVIOUS dilation), the following blocks with 3x3 / dilation 2 / padding 2.  Same parameters, same state_dict."""
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.dilation = bool(dilation)
        inplanes = 64
        for li, (planes, n, stride) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 2, 1 if dilation else 2)), start=1):
            blocks = []
            for bi in range(n):
                st = stride if bi == 0 else 1
                ds = None
                if bi == 0 and (st != 1 or inplanes != planes * 4):
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=st, bias=False), norm_layer(planes * 4))
                blocks.append(_Bottleneck(inplanes, planes, st, ds, norm_layer))
                if dilation and li == 4 and bi > 0:                   # state_dict / repr parity with torchvision's module
                    blocks[-1].conv2.dilation, blocks[-1].conv2.padding = (2, 2), (2, 2)
                blocks[-1].mask_input = not (li == 1 and bi == 0)      # layer1.0 reads the max-pool output
                blocks[-1].mask_gy = bi == n - 1                       # the layer output is a tap (several consumers)
                inplanes = planes * 4
            setattr(self, f'layer{li}', nn.Sequential(*blocks))

    def forward(self, x, init_conv=None):
        """x: NHWC image; with `init_conv` (1 -> 3 channels) the stem is ONE differentiable op (Fn.Stem)."""
        s, b = self.bn1.affine()
        if init_conv is not None:
            x = Fn.Stem.apply(x, init_conv.weight, init_conv.bias, self.conv1.weight, s, b)
        else:
            x = Fn.conv(x, self.conv1.weight, scale=s, shift=b, kh=7, kw=7, stride=2, pad=3, act=ops.ACT_RELU)
        taps = [x]
        x = Fn.MaxPool.apply(x, True)              # the stem output is a ReLU output
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(self, f'layer{li}')):
                if self.dilation and li == 4 and bi == 1:
                    # the dilated blocks: a 3x3 / dilation-2 / pad-2 convolution is the ordinary 3x3 on the four parity classes of the
                    # pixels, and the rest of a bottleneck is point-wise -> run the ordinary blocks on the space-to-batch form
                    x = Fn.SpaceToBatch2.apply(x, False)
                x = blk(x)
            if self.dilation and li == 4 and len(self.layer4) > 1:
                x = Fn.SpaceToBatch2.apply(x, True)
            taps.append(x)
        return taps


class BackboneBase(nn.Module):

    def __init__(self, backbone, name, in_channels, train_backbone):
        super().__init__()
        for _, parameter in backbone.named_parameters():
            if not train_backbone:
                parameter.requires_grad_(False)
        self.body = backbone
        self.num_channels = list(bcbk_channels['resnet'].values())
        if in_channels != 3:
            self.init_conv = nn.Conv2d(in_channels, 3, 1)
        self.in_channels = in_channels
        self.strides = [2 ** (i + 1) for i in range(len(self.num_channels))]

    def forward(self, x):
        """x: NHWC [B,H,W,in_channels] -> list of 5 NHWC maps (reference backbone.py:110-113)."""
        if hasattr(self, 'init_conv'):
            if self.in_channels != 1:
                raise NotImplementedError('init_conv is implemented for 1 input channel (reference default)')
            return self.body(x, self.init_conv)
        return self.body(x)


class Backbone(BackboneBase):
    """ResNet backbone with frozen BatchNorm (reference backbone.py:116-132)."""

    def __init__(self, name, in_channels, train_backbone, dilation, norm_layer_name):
        if name not in _RESNET_LAYERS:
            raise ValueError(f'not supported {name}: the accelerated path implements {sorted(_RESNET_LAYERS)}')
        if norm_layer_name != 'frozen_batchnorm':
            raise NotImplementedError('only --norm_layer_backbone frozen_batchnorm (reference default) is implemented')
        super().__init__(_ResNetBody(_RESNET_LAYERS[name], FrozenBatchNorm2d, dilation=dilation), name, in_channels, train_backbone)


class Joiner(nn.Sequential):
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)

    def forward(self, x):
        """-> (list of 5 NHWC feature maps, None).  The per-level sine encodings of the reference
        (backbone.py:139-148) are unused by the default config and not evaluated."""
        return self[0](x), None


def build_backbone(args):
    position_embedding = build_position_encoding(args)
    train_backbone = args.lr_backbone > 0
    backbone = Backbone(args.backbone, args.inpt_channels, train_backbone, args.dilation, args.norm_layer_backbone)
    model = Joiner(backbone, position_embedding)
    model.num_channels = backbone.num_channels
    model.strides = backbone.strides
    setattr(args, 'n_layers', len(backbone.num_channels))
    return model
