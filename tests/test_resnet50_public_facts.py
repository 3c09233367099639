"""CPU: the ResNet-50 definitions this build owns -- `oracle/tv_standin.py` (used to import the reference and to make the
golden taps) and the product's `_ResNetBody` -- pinned to PUBLIC facts about torchvision's ResNet-50 v1.5 (torchvision
is absent from this image and unpinned in the reference, backbone.py:8,131): parameter count,
state_dict key names and shapes, the stride sitting on the 3x3 convolution, block counts, tap shapes at 375 x 1024.
M2 is therefore "public-architecture pinned", not reference pinned."""
import torch

from oracle import tv_standin


def _conv_bn_params(sd):
    return sum(v.numel() for k, v in sd.items()
               if not k.startswith('fc.') and 'running_' not in k and 'num_batches_tracked' not in k)


CANON = {   # a sample of torchvision.models.resnet50().state_dict() (public, stable since 0.3)
    'conv1.weight': (64, 3, 7, 7), 'bn1.weight': (64,), 'bn1.running_var': (64,),
    'layer1.0.conv1.weight': (64, 64, 1, 1), 'layer1.0.conv2.weight': (64, 64, 3, 3),
    'layer1.0.conv3.weight': (256, 64, 1, 1), 'layer1.0.downsample.0.weight': (256, 64, 1, 1),
    'layer1.0.downsample.1.bias': (256,), 'layer1.2.bn3.weight': (256,),
    'layer2.0.conv1.weight': (128, 256, 1, 1), 'layer2.0.downsample.0.weight': (512, 256, 1, 1),
    'layer2.3.conv2.weight': (128, 128, 3, 3), 'layer3.0.conv2.weight': (256, 256, 3, 3),
    'layer3.5.conv3.weight': (1024, 256, 1, 1), 'layer4.0.downsample.0.weight': (2048, 1024, 1, 1),
    'layer4.2.conv3.weight': (2048, 512, 1, 1), 'layer4.2.bn3.running_mean': (2048,),
}


def check_body(mod, sd, with_fc):
    assert _conv_bn_params(sd) == 23_508_032                      # 25 557 032 - fc (2 049 000)
    for k, shape in CANON.items():
        assert tuple(sd[k].shape) == shape, k
    assert [len(getattr(mod, f'layer{i}')) for i in range(1, 5)] == [3, 4, 6, 3]
    assert sum(1 for k in sd if k.endswith('conv1.weight') or k.endswith('conv2.weight') or k.endswith('conv3.weight')
               or k.endswith('downsample.0.weight')) == 53
    assert not any('layer1.1.downsample' in k or 'layer2.1.downsample' in k for k in sd)
    # v1.5: the stride of a down-sampling bottleneck is on its 3x3 convolution (v1 had it on the first 1x1)
    for li in (2, 3, 4):
        blk = getattr(mod, f'layer{li}')[0]
        assert blk.conv1.stride == (1, 1) and blk.conv2.stride == (2, 2) and blk.downsample[0].stride == (2, 2)
        assert all(b.conv2.stride == (1, 1) for b in list(getattr(mod, f'layer{li}'))[1:])
    assert mod.layer1[0].conv2.stride == (1, 1) and mod.conv1.stride == (2, 2) and mod.conv1.padding == (3, 3)
    if with_fc:
        assert tuple(sd['fc.weight'].shape) == (1000, 2048) and sum(v.numel() for k, v in sd.items()
                                                                     if 'running_' not in k and 'num_batches' not in k) == 25_557_032


def test_stand_in_is_torchvision_resnet50_v15():
    m = tv_standin.resnet50()
    sd = m.state_dict()
    check_body(m, sd, with_fc=True)
    assert len(sd) == 320                                          # 53 conv + 53 x 5 BatchNorm entries + fc weight / bias
    # taps the reference takes with IntermediateLayerGetter (backbone.py:82-85) at the configured image size
    body = tv_standin.IntermediateLayerGetter(m, {'relu': '0', 'layer1': '1', 'layer2': '2', 'layer3': '3', 'layer4': '4'})
    with torch.no_grad():
        taps = body(torch.zeros(1, 3, 375, 1024))
    assert [tuple(t.shape[1:]) for t in taps.values()] == [(64, 188, 512), (256, 94, 256), (512, 47, 128),
                                                           (1024, 24, 64), (2048, 12, 32)]
    assert 'avgpool' not in dict(body.named_children()) and 'fc' not in dict(body.named_children())


def test_product_body_has_the_same_public_layout():
    from birdsoundclassif_amd.nets.backbone import FrozenBatchNorm2d, _ResNetBody
    body = _ResNetBody([3, 4, 6, 3], FrozenBatchNorm2d)
    sd = body.state_dict()
    check_body(body, sd, with_fc=False)
    ref = {k: tuple(v.shape) for k, v in tv_standin.resnet50(norm_layer=FrozenBatchNorm2d).state_dict().items()
           if not k.startswith('fc.')}
    assert {k: tuple(v.shape) for k, v in sd.items()} == ref       # same keys, same shapes, no num_batches_tracked
    assert list(sd) == list(ref)                                   # same ORDER (checkpoints load positionally nowhere,
                                                                   # but the flat optimizer buffers follow it)
