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
        self._pre = None
        self._pre_loss = None
        self._pinned = {}

    def _upload(self, name, arr, dev):
        """Small host array -> device through a persistent pinned staging buffer, asynchronously: a pageable `.to(device)` is a
        stream-ordered blocking copy, i.e. the host would wait for every kernel queued in front of it.  The staging buffer of a
        name is reused every step, so every copy out of it is followed by an event, and the host waits for THAT event (the copy
        of the previous step: long done, the wait returns at once) before it overwrites the buffer -- on the soft-failure paths of a
        step no other host sync separates two uses."""
        if dev.type != 'cuda':
            return torch.from_numpy(arr).to(dev)
        t = torch.from_numpy(np.ascontiguousarray(arr))
        slot = self._pinned.get(name)
        if slot is None or slot[0].numel() < t.numel() or slot[0].dtype != t.dtype:
            if slot is not None and slot[1] is not None:
                slot[1].synchronize()               # the old buffer's last copy must be out before the buffer is dropped
            slot = self._pinned[name] = [torch.empty((max(t.numel(), 8192),), dtype=t.dtype).pin_memory(), None]
        buf, ev = slot
        if ev is not None:
            ev.synchronize()
        view = buf[:t.numel()].view(t.shape)
        view.copy_(t)
        out = view.to(dev, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        return out

    def precompute_first_stage_loss(self, labels_pred, bbox_reg, gt_bbox, lengths):
        """Positive step: anchor targets (host) AND the first-stage loss kernels, queued behind the first-stage forward while the
        GPU is still executing it -- the host does not wait for anything here (pinned uploads), so when it blocks on the RoI count
        afterwards the loss is already computed instead of costing 8 ms of idle GPU behind the sync.  `first_stage_loss` returns
        the result."""
        self._pre_loss = self.first_stage_loss(labels_pred, bbox_reg, gt_bbox, lengths, False)

    @torch.no_grad()
    def start_anchor_targets(self, gt_bbox, lengths, device):
        """Top of a positive step (train.step), BEFORE the forward pass is queued: the arithmetic half of the AnchorTargetLayer -- IoU
        of every inside anchor with the image's boxes, best box, thresholds, best-anchor ties (nbm_anchor_targets) -- on a side stream
        of its own, results on their way to pinned host memory.  It depends on the labels only, so it neither waits for the queued
        GPU work of the previous step nor delays this one; `first_stage_loss` hands the result to the layer, which then only draws."""
        from .. import ops
        self._pre_anchor = None
        device = torch.device(device)
        if device.type != 'cuda' or not len(lengths) or min(lengths) < 1 or max(lengths) > 256:
            return
        layer = self.anchor_target_layer
        gt_c = np.ascontiguousarray(gt_bbox.detach().float().cpu().numpy(), dtype=np.float32)
        B, G = len(lengths), max(lengths)
        gt_pad = np.full((B, G, 4), -1, dtype=np.float32)
        lens = np.asarray(lengths)
        gt_pad[np.arange(G)[None, :] < lens[:, None]] = gt_c
        side = self.__dict__.get('_side_stream')
        if side is None:
            side = self.__dict__['_side_stream'] = torch.cuda.Stream(device=device, priority=-1)
        anc = self.__dict__.get('_anchors_dev')
        if anc is None or anc.device != device:
            anc = self.__dict__['_anchors_dev'] = layer.anchors.to(device).contiguous()
            torch.cuda.current_stream(device).synchronize()           # once: the constant is there before the side stream reads it
        with torch.cuda.stream(side):
            lab, amx, flag = ops.anchor_targets(anc, self._upload('agt_pad', gt_pad, device), self._upload('an_gt', lens.astype(np.int32), device),
                                                self.config.rpn_neg_label, self.config.rpn_pos_label)
            host = []
            for name, t in (('alab', lab), ('aamx', amx), ('aflag', flag)):
                buf = self._pinned.get('d2h_' + name)
                if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                    buf = self._pinned['d2h_' + name] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
                buf.copy_(t, non_blocking=True)
                host.append(buf)
            ev = torch.cuda.Event()
            ev.record(side)
        self._pre_anchor = (host, ev, (lab, amx, flag), [int(v) for v in lengths])       # the device tensors stay alive until consumed

    def _take_anchor_pre(self, lengths):
        pre, self._pre_anchor = getattr(self, '_pre_anchor', None), None
        if pre is None or pre[3] != [int(v) for v in lengths]:
            return None
        pre[1].synchronize()                     # a few hundred microseconds of GPU work queued before the forward pass: long done
        return tuple(t.numpy() for t in pre[0])

    def precompute_first_stage_targets(self, gt_bbox, lengths):
        """Run the AnchorTargetLayer (host, NumPy RNG) ahead of the forward pass so that it overlaps with the GPU work
        still queued from the previous step; `first_stage_loss` consumes the result.  Same RNG call order as computing
        it inside the loss (anchor targets are always drawn before proposal targets)."""
        self._pre = self.anchor_target_layer(gt_bbox, lengths, device='cpu', pre=self._take_anchor_pre(lengths))

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
        if self._pre_loss is not None:
            out, self._pre_loss = self._pre_loss, None
            return out
        assert gt_bbox is not None and lengths is not None
        dev = labels_pred.device
        if self._pre is not None:
            (labels, reg_targets), self._pre = self._pre, None
        else:
            labels, reg_targets = self.anchor_target_layer(gt_bbox, lengths, device='cpu',      # host (NumPy RNG draws); the IoU half
                                                           pre=self._take_anchor_pre(lengths))  # may come from the device
        lab_np = labels.permute(0, 2, 3, 1).reshape(-1).numpy()
        keep_np = np.nonzero(lab_np != -1)[0]
        lab_k = lab_np[keep_np]
        n_keep, n_pos = len(keep_np), int((lab_k > 0).sum())
        keep = self._upload('keep', keep_np, dev)
        lab = self._upload('lab', lab_k, dev)
        t = self._upload('t', reg_targets.permute(0, 2, 3, 1).reshape(-1, 4)[torch.from_numpy(keep_np)].numpy(), dev)
        p = labels_pred.permute(0, 2, 3, 1).reshape(-1, 2)[keep]
        class_loss = (-torch.log(p.gather(1, lab[:, None])[:, 0])).sum() * (1 / n_keep)
        r = bbox_reg.permute(0, 2, 3, 1).reshape(-1, 4)[keep]
        regression_loss = (smooth_l1(r, t) * (lab == 1).float()[:, None]).sum()
        if n_pos > 0:            # reference: `if regression_loss > 0` -- equivalent without a device sync
            regression_loss = regression_loss * (4 / n_pos)
        return {'first_class_loss': class_loss, 'first_regression_loss': regression_loss}

    @torch.no_grad()
    def precompute_proposal_iou(self, rois, gt_bbox, lengths):
        """Positive step, queued behind the proposal layer before the host waits for the RoI count: the arithmetic half of the
        ProposalTargetLayer (IoU of every proposal / ground-truth box with the image's boxes, best box per row: nbm_proposal_iou)
        on the device, and RoIs + results on their way to pinned host memory -- the host then only thresholds and draws
        (`generate_all_rois(..., use_precomputed=True)`, called by the step that queued this for the first rows of these very
        RoIs).  rois: [B,cap,4] device tensor of the proposal layer (rows beyond the RoI count are never looked at)."""
        from .. import ops
        self._pre_iou = None
        if rois.device.type != 'cuda' or rois.dim() != 3 or not len(lengths):
            return
        gt_c = np.ascontiguousarray(gt_bbox.detach().float().cpu().numpy(), dtype=np.float32)      # labels live on the host
        gt_pad, batched = self.proposal_target_layer.pad_gt(gt_c, lengths)
        if not batched:
            return
        dev = rois.device
        mx, asg = ops.proposal_iou(rois, self._upload('gt_pad', gt_pad, dev), self._upload('n_gt', np.asarray(lengths, dtype=np.int32), dev))
        host = []
        for name, t in (('rois', rois), ('mx', mx), ('asg', asg)):
            buf = self._pinned.get('d2h_' + name)
            if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                buf = self._pinned['d2h_' + name] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
            buf.copy_(t, non_blocking=True)
            host.append(buf)
        ev = torch.cuda.Event()
        ev.record()
        self._pre_iou = (rois.shape[0], rois.shape[1], host, ev)

    @torch.no_grad()
    def generate_all_rois(self, rois, *args, use_precomputed=False, **kwargs):
        """`use_precomputed`: `rois` are the first rows of the RoIs `precompute_proposal_iou` was called with in this step."""
        pre, self._pre_iou = getattr(self, '_pre_iou', None), None
        if use_precomputed and pre is not None and rois.dim() == 3 and rois.shape[0] == pre[0] and rois.shape[1] <= pre[1]:
            pre[3].synchronize()                 # long done: the host has read the RoI count, queued behind these copies
            kwargs['pre'] = (pre[2][0].numpy(), pre[2][1].numpy(), pre[2][2].numpy(), pre[1])
        rois, bbox_targets, labels = self.proposal_target_layer(rois, *args, **kwargs)
        # the labels were built on the host: keep that copy for second_stage_loss (a .cpu() there is a device sync)
        self._labels_host = (labels, self.proposal_target_layer.last_labels_host) if labels is not None else None
        return {'rois': rois, 'bbox_targets': bbox_targets, 'labels': labels}

    def second_stage_loss(self, bbox_reg, bbox_classes, bbox_targets=None, labels=None, neg_sample=False):
        """reference nbm_model.py:171-217."""
        cfg = self.config
        if neg_sample:
            # the reference's n x n gather has the value and gradient of mean(-log p[:,0]) (Appendix C-15b)
            return {'sec_neg_class_loss': (-torch.log(bbox_classes[:, 0])).mean()}
        assert bbox_targets is not None and labels is not None
        B, nb, nc = len(bbox_targets), cfg.rcnn_batch_size, cfg.num_classes
        dev = bbox_reg.device
        t = bbox_targets.view(B * nb, 4 * (nc + 1))
        held = getattr(self, '_labels_host', None)
        if held is not None and held[0] is labels:
            lab_np = held[1].reshape(-1).astype(np.int64)                        # 16 labels per image, still on the host
        else:
            lab_np = labels.detach().flatten().cpu().numpy().astype(np.int64)
        n_fg = int((lab_np > 0).sum())
        lab = torch.from_numpy(lab_np).to(dev)
        pg = bbox_classes.gather(1, lab[:, None])[:, 0]
        if cfg.focal_loss:
            class_loss = (-(1 - pg).pow(1.5) * torch.log(pg)).mean()
        else:
            class_loss = (-torch.log(pg)).sum() * (1 / (B * nb))
        cols = torch.from_numpy((lab_np[:, None] * 4 + np.arange(4)[None, :])).to(dev)   # the 4 slots of the GT class
        fg = torch.from_numpy((lab_np > 0).astype(np.float32)).to(dev)[:, None]          # no objective for background
        regression_loss = (smooth_l1(bbox_reg.gather(1, cols), t.gather(1, cols)) * fg).sum()
        if n_fg > 0:
            regression_loss = regression_loss * (4 / n_fg)
        return {'sec_class_loss': class_loss, 'sec_regression_loss': regression_loss}

    @torch.no_grad()
    def loss_cardinality(self, outputs, targets):
        """Logging only (reference nbm_model.py:219-226).  A 0-dim DEVICE tensor: `.item()` here would make the host wait for the
        second-stage forward before it may queue the backward pass; whoever logs the value converts it (float() / int())."""
        n_tgt = int((targets != 0).sum().item()) if not targets.is_cuda else (targets != 0).sum()
        return {'cardinality_error': (outputs.argmax(-1) != 0).sum() - n_tgt}
