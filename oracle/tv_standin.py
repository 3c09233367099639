"""TEST INFRASTRUCTURE ONLY — minimal in-memory `torchvision` stand-in.

torchvision is not installed in the build container, yet every
`nbm_model.nets.*` module of the reference imports it
(reference nets/backbone.py:5-8, nets/util/box_ops.py:6).  This file provides
exactly the three symbols the reference touches so that the reference package
can be imported *in this container* to validate `oracle/` and to generate the
fixtures under `tests/golden/`:

  * torchvision.models.resnet50(norm_layer=..., replace_stride_with_dilation=...)
      -- public ResNet-50 v1.5 architecture (stride on the 3x3 of each
         bottleneck), state_dict names identical to torchvision's.
  * torchvision.models._utils.IntermediateLayerGetter
  * torchvision.ops.boxes.box_area

Written from the published architecture; nothing here is shipped in the
product path and nothing here is copied from the reference.
"""
import sys
import types
from collections import OrderedDict

import torch
from torch import nn


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample, norm_layer, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation,
                               dilation=dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class _ResNet(nn.Module):
    def __init__(self, layers, norm_layer=None, replace_stride_with_dilation=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        rsd = replace_stride_with_dilation or [False, False, False]
        self._norm = norm_layer
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make(64, layers[0], 1, False)
        self.layer2 = self._make(128, layers[1], 2, rsd[0])
        self.layer3 = self._make(256, layers[2], 2, rsd[1])
        self.layer4 = self._make(512, layers[3], 2, rsd[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)

    def _make(self, planes, blocks, stride, dilate):
        prev_dil = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                               self._norm(planes * 4))
        mods = [_Bottleneck(self.inplanes, planes, stride, ds, self._norm, prev_dil)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            mods.append(_Bottleneck(self.inplanes, planes, 1, None, self._norm, self.dilation))
        return nn.Sequential(*mods)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet50(**kw):
    kw.pop('weights', None), kw.pop('pretrained', None)
    return _ResNet([3, 4, 6, 3], **kw)


class IntermediateLayerGetter(nn.ModuleDict):
    """Runs children in registration order, collects the named taps, stops after the last."""

    def __init__(self, model, return_layers):
        left = dict(return_layers)
        kept = OrderedDict()
        for name, mod in model.named_children():
            kept[name] = mod
            left.pop(name, None)
            if not left:
                break
        super().__init__(kept)
        self.return_layers = dict(return_layers)

    def forward(self, x):
        out = OrderedDict()
        for name, mod in self.items():
            x = mod(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


def box_area(boxes):
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def install():
    """Register the stand-in as `torchvision` in sys.modules (no-op if a real one is importable)."""
    if 'torchvision' in sys.modules:
        return
    tv = types.ModuleType('torchvision')
    models = types.ModuleType('torchvision.models')
    _utils = types.ModuleType('torchvision.models._utils')
    ops = types.ModuleType('torchvision.ops')
    boxes = types.ModuleType('torchvision.ops.boxes')
    models.resnet50 = resnet50
    _utils.IntermediateLayerGetter = IntermediateLayerGetter
    models._utils = _utils
    boxes.box_area = box_area
    ops.boxes = boxes
    tv.models, tv.ops = models, ops
    for m in (tv, models, _utils, ops, boxes):
        sys.modules[m.__name__] = m
