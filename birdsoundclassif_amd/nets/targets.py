"""Training target layers (reference nets/layers.py:102-216 AnchorTargetLayer, :306-396 ProposalTargetLayer).

Host-side integer / sampling logic, exactly where the reference runs it (it moves labels to numpy and draws with the
global `np.random` stream, layers.py:187,196,370-376): a few hundred boxes per step, no arithmetic worth a kernel.
The NumPy call order (fg then bg, image by image) is the reference's, so a seeded run reproduces its samples.
Inputs may live on any device; outputs are returned on `device`.
"""
import numpy as np
import torch
import torch.nn as nn

from .util.nets_utils import generate_anchors_frcnn, get_anchor_shifts_frcnn


def box_iou_incl(a, b):
    """[N,4] x [M,4] -> [N,M] IoU with the inclusive-pixel (+1) convention (reference nets_utils.py:103-126)."""
    a, b = a[:, None, :], b[None, :, :]
    xi = (torch.minimum(a[..., 2], b[..., 2]) - torch.maximum(a[..., 0], b[..., 0]) + 1).clamp(min=0)
    yi = (torch.minimum(a[..., 3], b[..., 3]) - torch.maximum(a[..., 1], b[..., 1]) + 1).clamp(min=0)
    inter = xi * yi
    area_a = (a[..., 2] - a[..., 0] + 1) * (a[..., 3] - a[..., 1] + 1)
    area_b = (b[..., 2] - b[..., 0] + 1) * (b[..., 3] - b[..., 1] + 1)
    return inter / ((area_a + area_b) - inter)


def box_encode(anchors, bbox):
    """Regression targets of `bbox` relative to `anchors` (reference nets_utils.py:129-146)."""
    wa = anchors[:, 2] - anchors[:, 0] + 1
    ha = anchors[:, 3] - anchors[:, 1] + 1
    xa, ya = anchors[:, 0] + 0.5 * wa, anchors[:, 1] + 0.5 * ha
    w = bbox[:, 2] - bbox[:, 0] + 1
    h = bbox[:, 3] - bbox[:, 1] + 1
    x, y = bbox[:, 0] + 0.5 * w, bbox[:, 1] + 0.5 * h
    return torch.stack([(x - xa) / wa, (y - ya) / ha, torch.log(w / wa), torch.log(h / ha)], dim=1)


class AnchorTargetLayer(nn.Module):

    def __init__(self, config):
        super().__init__()
        self.config = config
        height, width = config.top_size
        base = generate_anchors_frcnn(base_size=config.base_size, ratios=config.ratios, scales=config.scales)
        allanc = (base + get_anchor_shifts_frcnn(width, height, config.anchor_stride)).reshape(-1, 4)
        self.A, self.K = len(base), height * width
        self.inds_inside = np.where((allanc[:, 0] >= 0) & (allanc[:, 1] >= 0) & (allanc[:, 2] < config.img_width) &
                                    (allanc[:, 3] < config.img_height))[0]
        self.anchors = torch.from_numpy(allanc[self.inds_inside].astype(np.float32))
        self.all_anchors = allanc

    def forward(self, gt_bbox, lengths, device=None):
        """-> (labels [B,A,h,w] int64 in {-1,0,1}, reg_targets [B,4A,h,w] f32)."""
        cfg = self.config
        device = gt_bbox.device if device is None else device
        gt = gt_bbox.detach().float().cpu()
        B, n_in = len(lengths), len(self.inds_inside)
        h, w = cfg.top_size
        ov = box_iou_incl(self.anchors, gt)
        labels = torch.full((B, n_in), -1, dtype=torch.int64)
        tgt = torch.zeros(B, n_in, 4)
        idx = np.cumsum([0] + list(lengths))
        for b, (i0, i1) in enumerate(zip(idx[:-1], idx[1:])):
            o = ov[:, i0:i1]
            mx, amx = o.max(dim=1)
            gmx, _ = o.max(dim=0)
            labels[b, mx < cfg.rpn_neg_label] = 0
            labels[b, mx >= cfg.rpn_pos_label] = 1
            if gmx.max().item() > 0:                                  # every GT gets its best anchor(s), ties included
                pos = torch.nonzero(gmx > 0)[:, 0]
                labels[b, torch.nonzero(o[:, pos] == gmx[pos])[:, 0]] = 1
            num_fg = int(cfg.rpn_fg_fraction * cfg.rpn_batchsize)
            fg = torch.nonzero(labels[b] == 1)[:, 0]
            if len(fg) > num_fg:
                labels[b, np.random.choice(fg.numpy(), len(fg) - num_fg, replace=False)] = -1
            num_bg = cfg.rpn_batchsize - int((labels[b] == 1).sum())
            bg = torch.nonzero(labels[b] == 0)[:, 0]
            if len(bg) > num_bg:
                labels[b, np.random.choice(bg.numpy(), len(bg) - num_bg, replace=False)] = -1
            tgt[b] = box_encode(self.anchors, gt[i0:i1][amx])
        tgt = labels.unsqueeze(2).clamp(min=0) * tgt
        all_l = torch.full((B, len(self.all_anchors)), -1, dtype=torch.int64)
        all_l[:, self.inds_inside] = labels
        all_t = torch.zeros(B, len(self.all_anchors), 4)
        all_t[:, self.inds_inside] = tgt
        return (all_l.view(B, h, w, self.A).permute(0, 3, 1, 2).to(device),
                all_t.view(B, h, w, self.A * 4).permute(0, 3, 1, 2).to(device))


class ProposalTargetLayer(nn.Module):

    def __init__(self, config):
        super().__init__()
        self.config = config

    def forward(self, rois, gt_bbox, bird_ids, lengths):
        """rois [B,R,4] -> (rois [B,16,4], bbox_targets [B,16,4(1+nc)], labels [B,16] float) on rois.device, or
        (None, None, None) when the batch cannot be filled (reference layers.py:359-364)."""
        cfg = self.config
        device = rois.device
        rois_c, gt_c, ids_c = rois.detach().float().cpu(), gt_bbox.detach().float().cpu(), bird_ids.detach().float().cpu()
        nc, nb = cfg.num_classes, cfg.rcnn_batch_size
        assert cfg.bg_threshold_hi <= cfg.fg_threshold
        out_r, out_t, out_l = [], [], []
        idx = np.cumsum([0] + list(lengths))
        for b, (i0, i1) in enumerate(zip(idx[:-1], idx[1:])):
            gt = gt_c[i0:i1]
            allr = torch.cat([rois_c[b], gt], dim=0) if gt.max().item() > -1 else rois_c[b]
            mx, asg = box_iou_incl(allr, gt).max(dim=-1)
            lab = ids_c[i0:i1][asg].clone()
            lab[mx < cfg.fg_threshold] = 0
            gta = gt[asg]
            fg = torch.nonzero(mx > cfg.fg_threshold)[:, 0]
            bg = torch.nonzero((cfg.bg_threshold_hi > mx) & (mx >= cfg.bg_threshold_lo))[:, 0]
            other = list(set(range(len(mx))) - set(bg.numpy()) - set(fg.numpy()))
            nfg = min(len(fg), int(cfg.rcnn_fg_prop * nb))
            if len(bg) + len(other) < nb - nfg:
                print(f'~~~~ NOT ENOUGH BG: {len(bg)} / IGNORED ROIS: {len(other)}, FILLING WITH POSITIVES: {len(fg)} ~~~~')
                if len(bg) + len(other) < nb - len(fg):
                    print('~~~~ IMPOSSIBLE TO FILL THE RCNN BATCH, NOT ENOUGH ROIS ~~~~')
                    return None, None, None
                nfg = max(nfg, nb - (len(bg) + len(other)))
            nbg = min(len(bg), nb - nfg)
            fgi = np.random.choice(fg.numpy(), nfg, replace=False)
            bgi = np.random.choice(bg.numpy(), nbg, replace=False)
            if len(fgi) + len(bgi) < nb:
                bgi = np.hstack([bgi, np.random.choice(other, nb - len(fgi) - len(bgi), replace=False)])
            keep = torch.from_numpy(np.hstack((fgi, bgi)).astype(np.int64))
            bl, br = lab[keep], allr[keep]
            t4 = box_encode(br, gta[keep])
            exp = torch.zeros(nb, 4 * (1 + nc))                       # one slot of 4 per class (nets_utils.py:248-259)
            li = bl.long()
            sel = torch.nonzero(li >= 1)[:, 0]
            for k in range(4):
                exp[sel, 4 * li[sel] + k] = t4[sel, k]
            out_r.append(br), out_t.append(exp), out_l.append(bl)
        return torch.stack(out_r).to(device), torch.stack(out_t).to(device), torch.stack(out_l).to(device)
