"""SetCriterion (reference nets/nbm_model.py:83-226): the two-stage Faster-R-CNN losses.

The losses are reductions over the 16 sampled anchors / RoIs per image: index/gather bookkeeping on a few hundred
values, evaluated with the same tensor expressions as the reference; their inputs (`rpn_cls_scores`, `rpn_bbox_reg`,
`bbox_reg`, `bbox_classes`) and every gradient that leaves them are produced / consumed by the HIP kernels.
The two negative-step quirks of the reference are reproduced in closed form.
"""
import numpy as np
import torch
from torch import nn

from .targets import AnchorTargetLayer, ProposalTargetLayer


def smooth_l1(a, b):
    """reference nets_utils.py:275-281 (beta = 1)."""
    d = (a - b).abs()
    big = d >= 1
    return (~big).float() * 0.5 * d ** 2 + big.float() * (d - 0.5)


class SetCriterion(nn.Module):

    def __init__(self, args, weight_dict):
        super().__init__()
        self.config = args
        self.weight_dict = weight_dict
        self.anchor_target_layer = AnchorTargetLayer(args)
        self.proposal_target_layer = ProposalTargetLayer(args)

    def first_stage_loss(self, labels_pred, bbox_reg, gt_bbox=None, lengths=None, neg_sample=False):
        """labels_pred [B,2A',h,w] softmaxed, bbox_reg [B,4A',h,w] (reference nbm_model.py:102-164)."""
        cfg = self.config
        if neg_sample:
            B = len(labels_pred)
            p = labels_pred.permute(0, 2, 3, 1).reshape(B, -1, 2)
            # reference: sort by objectness, take the top 320, then a broadcasting gather that keeps only the
            # top-1 anchor of every image, through BOTH of its probabilities (Appendix C-15a)
            top = p[..., 1].argmax(dim=1)
            top_p = p[torch.arange(B, device=p.device), top]
            return {'first_neg_class_loss': (-torch.log(top_p)).mean()}
        assert gt_bbox is not None and lengths is not None
        labels, reg_targets = self.anchor_target_layer(gt_bbox, lengths, device=labels_pred.device)
        p = labels_pred.permute(0, 2, 3, 1).reshape(-1, 2)
        lab = labels.permute(0, 2, 3, 1).flatten()
        keep = torch.nonzero(lab != -1)[:, 0]
        p, lab = p[keep], lab[keep]
        class_loss = (-torch.log(p[torch.arange(len(p), device=p.device), lab])).sum() * (1 / len(p))
        r = bbox_reg.permute(0, 2, 3, 1).reshape(-1, 4)[keep]
        t = reg_targets.permute(0, 2, 3, 1).reshape(-1, 4)[keep]
        regression_loss = (smooth_l1(r, t) * (lab == 1).float()[:, None]).sum()
        if regression_loss > 0:
            regression_loss = regression_loss * (4 / (lab > 0).sum())
        return {'first_class_loss': class_loss, 'first_regression_loss': regression_loss}

    @torch.no_grad()
    def generate_all_rois(self, *args, **kwargs):
        rois, bbox_targets, labels = self.proposal_target_layer(*args, **kwargs)
        return {'rois': rois, 'bbox_targets': bbox_targets, 'labels': labels}

    def second_stage_loss(self, bbox_reg, bbox_classes, bbox_targets=None, labels=None, neg_sample=False):
        """reference nbm_model.py:171-217."""
        cfg = self.config
        if neg_sample:
            # the reference's n x n gather has the value and gradient of mean(-log p[:,0]) (Appendix C-15b)
            return {'sec_neg_class_loss': (-torch.log(bbox_classes[:, 0])).mean()}
        assert bbox_targets is not None and labels is not None
        B, nb, nc = len(bbox_targets), cfg.rcnn_batch_size, cfg.num_classes
        t = bbox_targets.view(B * nb, 4 * (nc + 1))
        lab = labels.flatten().long()
        ar = torch.arange(len(lab), device=lab.device)
        pg = bbox_classes[ar, lab]
        if cfg.focal_loss:
            class_loss = (-(1 - pg).pow(1.5) * torch.log(pg)).mean()
        else:
            class_loss = (-torch.log(pg)).sum() * (1 / (B * nb))
        mask = torch.zeros_like(bbox_reg)
        for i in range(4):
            mask[ar, i + lab * 4] = 1
        mask[:, 0:4] = 0
        regression_loss = (mask * smooth_l1(bbox_reg, t)).sum()
        if regression_loss > 0:
            regression_loss = regression_loss * (4 / (lab > 0).sum())
        return {'sec_class_loss': class_loss, 'sec_regression_loss': regression_loss}

    @torch.no_grad()
    def loss_cardinality(self, outputs, targets):
        return {'cardinality_error': (outputs.argmax(-1) != 0).sum().item() - (targets != 0).sum().item()}
