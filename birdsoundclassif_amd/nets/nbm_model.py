"""NbmModel / SetCriterion / initialize_model / build (reference nets/nbm_model.py), HIP forward.

The module tree, and therefore every `state_dict` key and shape, is the reference's:
a released `{'checkpoints': state_dict}` file loads unchanged through `initialize_model`.
"""
import os

import numpy as np
import torch
from torch import nn

from .backbone import build_backbone
from .fpn import build_fpn
from . import functional as Fn
from .self_attention import build_sa_layers, materialize
from .head import build_head

DEFER_PROJECTION_TRAIN = os.environ.get('NBM_DEFER_PROJECTION_TRAIN', '1') != '0'
DEFER_PROJECTION = os.environ.get('NBM_DEFER_PROJECTION', '1') != '0'      # evaluation mode: attention's final projection folded into the FPN laterals


class NbmModel(nn.Module):
    """reference nbm_model.py:22-80."""

    def __init__(self, args, backbone, attn, fpn, head):
        super().__init__()
        self.args = args
        self.backbone = backbone
        self.attn = attn
        self.fpn = fpn
        self.head = head

    def _lazy_strides(self):
        """{pyramid level: stride of the RPN's depthwise 3x3 on that level (layers.py:62-65)} for the FPN output maps that are
        computed on demand: default topology only (FPN outputs read by the RPN and the RoI pooling, nothing else).
        Level 0 (stride 8): three 2x2 tiles out of four are never read.  Level 1 (stride 4): every tile holds a pattern pixel, so
        the listed F(2x2,3x3) tiles gained nothing there (round 2: 10.1 -> 8.5 ms at B = 64, eaten by the RoI phase); with the
        pattern pixels going through the cell transforms (csrc/cellwino.hip: 25 plane products per 4x4 cell instead of the 64 of
        its four tiles) it pays: detect step 81.4 -> 75.6 ms at B = 64.  NBM_LAZY_LEVEL1=0 keeps level 1 dense."""
        from .. import ondemand
        a = self.args
        if getattr(a, 'fpn_first', False) or getattr(a, 'sandwich_attn', False) or getattr(a, 'fpn', 'fpn') != 'fpn':
            return None
        st = a.anchor_stride / 2
        if not (st >= 6 and st == int(st)):
            return None
        out = {0: int(st)}
        if ondemand.CELL_FWD and os.environ.get('NBM_LAZY_LEVEL1', '1') != '0' and int(st) % 2 == 0 and int(st) // 2 >= 3:
            out[1] = int(st) // 2
        return out

    def _fpn_nhwc(self, samples, lazy=False):
        if samples.dim() != 4 or samples.shape[1] != self.args.inpt_channels:
            raise ValueError(f'expected [B,{self.args.inpt_channels},H,W], got {tuple(samples.shape)}')
        x = samples.permute(0, 2, 3, 1).contiguous()
        token = Fn.stash_reset(self)           # gradient hand-over registrations are per forward pass, owned by this model (functional._PASS)
        features, _ = self.backbone(x)
        if torch.is_grad_enabled() and features[-1].requires_grad:
            # data-parallel training: when autograd hands the gradient of the LAST tap to the backbone, every non-backbone gradient is
            # final -- the exchange of that flat buffer starts there, beside the backbone's backward kernels (train.DP_OVERLAP)
            from .. import train as _train
            if _train.exchange_armed():
                features[-1].register_hook(_train.backbone_boundary_hook)
        if getattr(self.args, 'add_posenc', False):                       # nbm_model.py:45-46
            pe = self.backbone[1]
            features = [Fn.AddConst.apply(f, pe(f)) for f in features]
        if getattr(self.args, 'fpn_first', False):                        # nbm_model.py:47-52
            return materialize(self.attn(self.fpn(features)))
        if getattr(self.args, 'sandwich_attn', False):
            return materialize(self.attn[1](self.fpn(self.attn[0](features))))
        # evaluation mode, plain FPN: the attention levels' final projection is folded into the FPN's laterals (self_attention.Projected)
        # (with a gradient to come: module + lateral as one composed tape node, functional.AttnLateral -- NBM_DEFER_PROJECTION_TRAIN=0: off)
        defer = type(self.fpn).__name__ == 'FPN' and DEFER_PROJECTION and (not torch.is_grad_enabled() or DEFER_PROJECTION_TRAIN)
        levels = self.attn(features, defer_projection=self.fpn.pt_wise) if defer else self.attn(features)
        if lazy and self._lazy_strides():
            out = self.fpn(levels, lazy_strides=self._lazy_strides())
        else:
            out = self.fpn(levels)
        Fn.fpn_out_register(out, token or None)  # the early backward pass of the RPN branch parks its gradients by these maps (train.step)
        return out

    def forward_first_stage(self, samples, host_work=None, lazy=False):
        """samples [B,1,H,W] f32 on the GPU -> {'rois','rpn_cls_scores','rpn_bbox_reg','fpn_out'} (nbm_model.py:39-54).
        Tensors are NCHW-shaped views of NHWC storage.  `host_work(rpn_cls_scores, rpn_bbox_reg)` (optional callable) runs after
        every kernel of the first stage has been queued and before the host waits for the RoI count, i.e. hidden behind the GPU work.
        `lazy=False` (default): every map of 'fpn_out' is dense, like the reference's.  `lazy=True` (what this package's own
        `train.step` and `detect` pass): 'fpn_out'[0] (and [1]) hold only the pixels their consumers read -- the RPN pattern now, the
        tiles under the RoI windows once a RoI pooling (`forward_second_stage`, any number of times, any RoIs) runs on THIS tensor or a
        full view of it; every other pixel is unwritten memory, so a clone / slice / arithmetic on the map is meaningless.  Without a
        gradient to come (evaluation) not even the RPN pattern is formed: the RPN's first block is composed with the map's own
        convolution; `ondemand.pattern_materialize(map)` forms the pattern pixels for a caller that wants to read them."""
        fpn_out = self._fpn_nhwc(samples, lazy=lazy)
        rois, cls, reg = self.head.forward_first_stage([f.permute(0, 3, 1, 2) for f in fpn_out], host_work)
        return {'rois': rois, 'rpn_cls_scores': cls, 'rpn_bbox_reg': reg,
                'fpn_out': [f.permute(0, 3, 1, 2) for f in fpn_out]}

    def forward_second_stage(self, fpn_pyramid_out, rois, nms_thresh=None, min_score=None, training=None):
        """nbm_model.py:56-64."""
        if training is None:
            training = self.training
        nms_thresh = 0.3 if nms_thresh is None else nms_thresh
        min_score = 0.5 if min_score is None else min_score
        outputs = self.head.forward_second_stage(fpn_pyramid_out, rois, nms_thresh, min_score, training)
        if training:
            bbox_reg, bbox_classes = outputs
            return {'bbox_reg': bbox_reg, 'bbox_classes': bbox_classes}
        return outputs

    @torch.no_grad()
    def detect(self, samples, nms_thresh=0.3, min_score=0.5, independent=False):
        """Sync-free eval forward: -> (det [B,50,6] rows {class,x1,y1,x2,y2,score} sorted by (class, score desc),
        n_det int32 [B]), both on the device.  Used by bulk inference; `forward` wraps it.
        `independent=False`: the reference's semantics for ONE model call on this batch -- the proposal counts are coupled over
        the batch (pre / post-NMS top-N = min over the images, layers.py:287, nets_utils.py:236).  `independent=True`: every
        image is a batch of its own, i.e. the result of B model calls with one image each (what the reference CLI does with B
        single-window files); the counts live in int32 [B] device tensors."""
        fpn_out = self._fpn_nhwc(samples, lazy=True)
        rois, _, n_roi, _, _, _ = self.head.forward_first_stage_device(fpn_out, independent=independent)
        return self.head.fast_rcnn.detect_device(fpn_out, rois, n_roi, nms_thresh, min_score)

    def forward(self, samples, nms_thresh=0.3, min_score=0.5):
        """-> list[B] of {'1'..'num_classes': {'bbox_coord','scores'}} (nbm_model.py:66-80).  Eval only."""
        if self.training:
            raise RuntimeError('NbmModel.forward is the inference entry point; call .eval() first '
                               '(training goes through forward_first_stage / forward_second_stage)')
        det, n_det = self.detect(samples, nms_thresh, min_score)
        return self.head.fast_rcnn.dets_to_dicts(det, n_det, self.args.num_classes)


def initialize_model(model, path=None, train=True):
    """reference nbm_model.py:325-341 (keeps only checkpoint keys that exist in the model)."""
    if path is not None:
        model_dict = model.state_dict()
        state_dict = torch.load(path, map_location='cpu', weights_only=False)
        state_dict = {k: v for k, v in state_dict['checkpoints'].items() if k in model_dict}
        model_dict.update(state_dict)
        model.load_state_dict(model_dict)
    if train:
        model.train()
    else:
        model.eval()
    return model


def build(args, train=True):
    """reference nbm_model.py:344-381 -> (NbmModel, SetCriterion)."""
    from .criterion import SetCriterion
    device = torch.device(args.device)
    backbone = build_backbone(args)
    if getattr(args, 'fpn_first', False):                                 # nbm_model.py:349-354
        attn_channels = [args.out_fpn_chan] * len(backbone.num_channels)
    elif getattr(args, 'sandwich_attn', False):
        attn_channels = (backbone.num_channels, [args.out_fpn_chan] * len(backbone.num_channels))
    else:
        attn_channels = backbone.num_channels
    attn = build_sa_layers(args, attn_channels)
    fpn = build_fpn(args, backbone.num_channels)
    head = build_head(args)
    model = NbmModel(args, backbone, attn, fpn, head).to(device)
    model = initialize_model(model, train=train)
    weight_dict = {'first_class_loss': args.fs_cls_loss_coef, 'first_regression_loss': args.fs_reg_loss_coef,
                   'sec_class_loss': args.sec_cls_loss_coef, 'sec_regression_loss': args.sec_reg_loss_coef,
                   'first_neg_class_loss': args.fs_neg_cls_loss_coef, 'sec_neg_class_loss': args.sec_neg_cls_loss_coef}
    criterion = SetCriterion(args, weight_dict)
    criterion.to(device)
    return model, criterion
