"""Faster-R-CNN head wiring (reference nets/head.py:9-42)."""
import torch
import torch.nn as nn

from .layers import RegionProposalNetwork, ProposalLayer, FastRCNN


class Faster_RCNN(nn.Module):

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.rpn = RegionProposalNetwork(args, args.n_layers, args.top_size)
        self.prop_layer = ProposalLayer(args, args.n_layers)
        self.fast_rcnn = FastRCNN(args)

    def forward(self, x, nms_thresh=0.3, min_score=0.5):
        rois, _, _ = self.forward_first_stage(x)
        return self.forward_second_stage(x, rois, nms_thresh, min_score)

    def forward_second_stage(self, *args, **kwargs):
        return self.fast_rcnn(*args, **kwargs)

    def forward_first_stage_device(self, fpn_nhwc, independent=False):
        """-> (rois [B,cap,4], roi_scores, n_roi device int32 [1] (or [B] with `independent`: every image a batch of its own),
        cls NHWC, reg NHWC, raw cls NHWC); no host sync."""
        cls, reg, cls_raw = self.rpn.forward_nhwc(fpn_nhwc)
        with torch.no_grad():
            rois, scores, n_roi = self.prop_layer.forward_device(cls.detach(), reg.detach(), independent=independent)
        return rois, scores, n_roi, cls, reg, cls_raw

    def forward_first_stage(self, fpn_pyramid_out, host_work=None):
        """fpn_pyramid_out: list of NCHW-shaped maps -> (rois [B,R,4] | empty, cls_scores, bbox_reg) (head.py:32-38)."""
        fm = [f.permute(0, 2, 3, 1).contiguous() for f in fpn_pyramid_out]
        rois, _, n_roi, cls, reg, _ = self.forward_first_stage_device(fm)
        # the RoI count goes to pinned memory NOW, with an event behind it: the host will wait for THAT, not for what `host_work`
        # queues after it (first-stage loss, early backward pass of the RPN branch: `n_roi.item()` waited for all of it -- 5 ms
        # during which the host could already build the proposal targets)
        pin = self.__dict__.get('_n_roi_pin')
        if pin is None:
            pin = self.__dict__['_n_roi_pin'] = torch.zeros((1,), dtype=torch.int32).pin_memory()
        ev = None
        if n_roi.is_cuda:
            pin.copy_(n_roi.view(-1)[:1], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        if host_work is not None:           # everything of the first stage is queued; the host is still ahead of the GPU here
            if getattr(host_work, 'wants_rois', False):          # device-side work on the proposals, queued before the sync below
                host_work(cls.permute(0, 3, 1, 2), reg.permute(0, 3, 1, 2), rois=rois)
            else:
                host_work(cls.permute(0, 3, 1, 2), reg.permute(0, 3, 1, 2))
        if ev is not None:
            ev.synchronize()
            n = int(pin.item())
        else:
            n = int(n_roi.item())
        if n == 0:
            print('Not enough possible RoIs, RPN failed')
            rois = torch.tensor([]).to(cls.device)
        else:
            rois = rois[:, :n].contiguous()
        return rois, cls.permute(0, 3, 1, 2), reg.permute(0, 3, 1, 2)


def build_head(args):
    return Faster_RCNN(args)
