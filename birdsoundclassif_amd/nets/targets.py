"""Training target layers (reference nets/layers.py:102-216 AnchorTargetLayer, :306-396 ProposalTargetLayer).

Host-side integer / sampling logic, exactly where the reference runs it (it moves labels to numpy and draws with the
global `np.random` stream, layers.py:187,196,370-376): a few hundred boxes per step, no arithmetic worth a kernel.
Written on NumPy float32 arrays (same IEEE operations and order as the reference's torch-CPU float32 expressions, so
thresholds and ties resolve identically) because per-image torch calls cost more in dispatch than in work.
The NumPy RNG call order (fg then bg, image by image) is the reference's, so a seeded run reproduces its samples.
"""
import numpy as np
import torch
import torch.nn as nn

from .util.nets_utils import generate_anchors_frcnn, get_anchor_shifts_frcnn

_F = np.float32


def box_iou_incl(a, b):
    """[N,4] x [M,4] float32 -> [N,M] IoU with the inclusive-pixel (+1) convention (reference nets_utils.py:103-126)."""
    a, b = a[:, None, :], b[None, :, :]
    xi = np.maximum(np.minimum(a[..., 2], b[..., 2]) - np.maximum(a[..., 0], b[..., 0]) + _F(1), _F(0))
    yi = np.maximum(np.minimum(a[..., 3], b[..., 3]) - np.maximum(a[..., 1], b[..., 1]) + _F(1), _F(0))
    inter = xi * yi
    area_a = (a[..., 2] - a[..., 0] + _F(1)) * (a[..., 3] - a[..., 1] + _F(1))
    area_b = (b[..., 2] - b[..., 0] + _F(1)) * (b[..., 3] - b[..., 1] + _F(1))
    return inter / ((area_a + area_b) - inter)


def box_encode(anchors, bbox):
    """Regression targets of `bbox` relative to `anchors`, float32 (reference nets_utils.py:129-146)."""
    wa = anchors[:, 2] - anchors[:, 0] + _F(1)
    ha = anchors[:, 3] - anchors[:, 1] + _F(1)
    xa, ya = anchors[:, 0] + _F(0.5) * wa, anchors[:, 1] + _F(0.5) * ha
    w = bbox[:, 2] - bbox[:, 0] + _F(1)
    h = bbox[:, 3] - bbox[:, 1] + _F(1)
    x, y = bbox[:, 0] + _F(0.5) * w, bbox[:, 1] + _F(0.5) * h
    # torch.log on float32: evaluate through torch so the rounding is the reference's own
    tw = torch.log(torch.from_numpy(w / wa)).numpy()
    th = torch.log(torch.from_numpy(h / ha)).numpy()
    return np.stack([(x - xa) / wa, (y - ya) / ha, tw, th], axis=1).astype(_F)


def _draw(pop, k):
    """`np.random.choice(pop, k, replace=False)` without its argument checks (4 us of 12 per call, 3 calls per image): the legacy
    RandomState draws `permutation(len(pop))[:k]` for it, so values and stream position are the same (checked in
    tests/test_targets_host.py)."""
    assert k <= len(pop)
    return pop[np.random.permutation(len(pop))[:k]]


def _f32(t):
    return np.ascontiguousarray(t.detach().float().cpu().numpy(), dtype=_F)


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
        self.anchors_np = allanc[self.inds_inside].astype(_F)
        self.anchors = torch.from_numpy(self.anchors_np)
        self.all_anchors = allanc

    def _host_labels(self, o, cfg):
        """Labels of one image's inside anchors BEFORE the subsampling + first best box, from its IoU block o [n_in, n] (the
        reference's own arithmetic, layers.py:162-179)."""
        neg_t, pos_t = _F(cfg.rpn_neg_label), _F(cfg.rpn_pos_label)
        mx, amx = o.max(axis=1), o.argmax(axis=1)
        gmx = o.max(axis=0)
        lb = np.full((o.shape[0],), -1, dtype=np.int64)
        lb[mx < neg_t] = 0
        lb[mx >= pos_t] = 1
        if gmx.max() > 0:                                         # every GT keeps its best anchor(s), ties included
            pos = np.nonzero(gmx > 0)[0]
            lb[np.nonzero(o[:, pos] == gmx[pos])[0]] = 1
        return lb, amx

    def forward(self, gt_bbox, lengths, device=None, pre=None):
        """-> (labels [B,A,h,w] int64 in {-1,0,1}, reg_targets [B,4A,h,w] f32) on `device`.
        `pre` (optional, from `SetCriterion.start_anchor_targets`): (lab int8 [B,n_in], amx int16 [B,n_in], flag int32 [B]) host
        arrays -- the IoU / arg-max / threshold half already done on the device (nbm_anchor_targets); the draws stay here, in the
        reference's order (fg then bg, image by image)."""
        cfg = self.config
        device = gt_bbox.device if device is None else device
        gt = _f32(gt_bbox)
        B, n_in = len(lengths), len(self.inds_inside)
        h, w = cfg.top_size
        idx = np.cumsum([0] + list(lengths))
        if pre is not None and (pre[0].shape != (B, n_in) or len(pre[2]) != B):
            pre = None
        ov = box_iou_incl(self.anchors_np, gt) if pre is None else None      # [n_in, sum N]
        labels = np.full((B, n_in), -1, dtype=np.int64)
        amx_all = np.zeros((B, n_in), dtype=np.int64)
        num_fg = int(cfg.rpn_fg_fraction * cfg.rpn_batchsize)
        for b, (i0, i1) in enumerate(zip(idx[:-1], idx[1:])):
            if pre is not None and not pre[2][b]:
                lb, amx = pre[0][b].astype(np.int64), pre[1][b]
            else:                                                     # host arithmetic (no device results, or a degenerate box: NaN)
                lb, amx = self._host_labels(ov[:, i0:i1] if ov is not None else box_iou_incl(self.anchors_np, gt[i0:i1]), cfg)
            fg = np.nonzero(lb == 1)[0]
            if len(fg) > num_fg:
                lb[np.random.choice(fg, len(fg) - num_fg, replace=False)] = -1
            num_bg = cfg.rpn_batchsize - int((lb == 1).sum())
            bg = np.nonzero(lb == 0)[0]
            if len(bg) > num_bg:
                lb[np.random.choice(bg, len(bg) - num_bg, replace=False)] = -1
            labels[b] = lb
            amx_all[b] = amx + i0                                     # row of `gt` assigned to every anchor
        all_l = np.full((B, len(self.all_anchors)), -1, dtype=np.int64)
        all_l[:, self.inds_inside] = labels
        all_t = np.zeros((B, len(self.all_anchors), 4), dtype=_F)
        gw, gh = gt[:, 2] - gt[:, 0] + _F(1), gt[:, 3] - gt[:, 1] + _F(1)
        if len(gt) and bool((gw > 0).all() and (gh > 0).all()):
            # targets are label-masked (`max(label, 0) * target`): with proper boxes every target is finite, so only the <= 8
            # foreground anchors of an image carry a value -- encode those rows alone (the batch-wide encode of 15 089 x B rows was a
            # third of this layer's host time)
            bb, aa = np.nonzero(labels == 1)
            if len(bb):
                all_t[bb, self.inds_inside[aa]] = box_encode(self.anchors_np[aa], gt[amx_all[bb, aa]])
        else:
            # a degenerate box: 0 * inf = NaN must come out exactly where the reference produces it -- the full encode
            matched = gt[amx_all.reshape(-1)].reshape(B, n_in, 4)
            tgt = box_encode(np.tile(self.anchors_np, (B, 1)), matched.reshape(-1, 4)).reshape(B, n_in, 4)
            all_t[:, self.inds_inside] = np.maximum(labels, 0)[..., None].astype(_F) * tgt
        return (torch.from_numpy(all_l).view(B, h, w, self.A).permute(0, 3, 1, 2).to(device),
                torch.from_numpy(all_t).view(B, h, w, self.A * 4).permute(0, 3, 1, 2).to(device))


class ProposalTargetLayer(nn.Module):

    def __init__(self, config):
        super().__init__()
        self.config = config

    @staticmethod
    def pad_gt(gt_c, lengths):
        """-> (gt_pad [B,gmax,4] (-1 where there is no box), batched: every image has a real box -- the only case the batched IoU
        (host or device) reproduces; the reference's per-image special case `gt.max() > -1` takes the loop below otherwise)."""
        B, gmax = len(lengths), max(lengths)
        idx = np.cumsum([0] + list(lengths))
        gt_pad = np.full((B, gmax, 4), -1, dtype=_F)
        if min(lengths) > 0:                 # every image has rows: one masked assignment, one segmented maximum
            lens = np.asarray(lengths)
            gt_pad[np.arange(gmax)[None, :] < lens[:, None]] = gt_c
            batched = bool((np.maximum.reduceat(gt_c.max(axis=1), idx[:-1]) > -1).all())
            return gt_pad, batched
        for b, (i0, i1) in enumerate(zip(idx[:-1], idx[1:])):
            gt_pad[b, :i1 - i0] = gt_c[i0:i1]
        batched = all(gt_c[i0:i1].max() > -1 for i0, i1 in zip(idx[:-1], idx[1:]))
        return gt_pad, batched

    def forward(self, rois, gt_bbox, bird_ids, lengths, pre=None):
        """rois [B,R,4] -> (rois [B,16,4], bbox_targets [B,16,4(1+nc)], labels [B,16] float) on rois.device, or
        (None, None, None) when the batch cannot be filled (reference layers.py:359-364).
        `pre` (optional, from `SetCriterion.precompute_proposal_iou`): (rois_host [B,>=R,4], mx [B,cap+gmax], asg [B,cap+gmax], cap)
        -- the IoU / best-box arithmetic already done on the device (nbm_proposal_iou) and copied to the host with the RoIs."""
        cfg = self.config
        device = rois.device
        gt_c, ids_c = _f32(gt_bbox), _f32(bird_ids)
        nc, nb = cfg.num_classes, cfg.rcnn_batch_size
        assert cfg.bg_threshold_hi <= cfg.fg_threshold
        fg_t, lo_t, hi_t = _F(cfg.fg_threshold), _F(cfg.bg_threshold_lo), _F(cfg.bg_threshold_hi)
        B = len(lengths)
        R, gmax = rois.shape[1], max(lengths)
        idx = np.cumsum([0] + list(lengths))
        gt_pad, batched = self.pad_gt(gt_c, lengths)
        if pre is not None and batched:
            rois_h, mx_h, asg_h, cap = pre
            rois_c = rois_h[:, :R]
            mx_all = np.concatenate([mx_h[:, :R], mx_h[:, cap:cap + gmax]], axis=1)
            asg_all = np.concatenate([asg_h[:, :R], asg_h[:, cap:cap + gmax]], axis=1).astype(np.int64)
            all_pad = np.concatenate([rois_c, gt_pad], axis=1)
        else:
            rois_c = _f32(rois)
        out_t = np.zeros((B, nb, 4 * (1 + nc)), dtype=_F) if torch.device(device).type != 'cuda' else None
        if batched and pre is None:
            # IoU / best-GT assignment for the whole batch in one shot (GT padded to the largest count; padded columns can
            # never win the max), then the per-image sampling with the reference's RNG call order
            valid = np.arange(gmax)[None, :] < np.asarray(lengths)[:, None]
            all_pad = np.concatenate([rois_c, gt_pad], axis=1)                       # [B, R + gmax, 4]
            # [B, gmax, R + gmax] with the long axis innermost: the same fp32 operations per element as the reference's IoU
            # (nets_utils.py:103-126), 3x faster in NumPy than the [.., R + gmax, gmax] broadcast (inner loops of length gmax).
            # (torch's CPU kernels were tried: bit-identical too, but waking 32 threads for 1 M-element tensors made it 43 ms on
            # the GPU box.)  The training step does this on the device instead (`pre`).
            a0, a1, a2, a3 = (np.ascontiguousarray(all_pad[:, None, :, k]) for k in range(4))
            g0, g1, g2, g3 = (np.ascontiguousarray(gt_pad[:, :, None, k]) for k in range(4))
            xi = np.minimum(a2, g2)
            xi -= np.maximum(a0, g0)
            xi += _F(1)
            np.maximum(xi, _F(0), out=xi)
            yi = np.minimum(a3, g3)
            yi -= np.maximum(a1, g1)
            yi += _F(1)
            np.maximum(yi, _F(0), out=yi)
            inter = xi
            inter *= yi
            area_a = (a2 - a0 + _F(1)) * (a3 - a1 + _F(1))
            area_g = (g2 - g0 + _F(1)) * (g3 - g1 + _F(1))
            den = area_a + area_g
            den -= inter
            with np.errstate(divide='ignore', invalid='ignore'):
                ov_all = inter / den
            ov_all[~np.broadcast_to(valid[:, :, None], ov_all.shape)] = -1
            mx_all, asg_all = ov_all.max(axis=1), ov_all.argmax(axis=1)
        if batched:
            # thresholds for the whole batch; the GT columns an image does not have are neither foreground nor background (NaN)
            n_all = R + np.asarray(lengths)
            mx_m = mx_all.copy()
            mx_m[np.arange(R + gmax)[None, :] >= n_all[:, None]] = np.nan
            with np.errstate(invalid='ignore'):
                fg_mask = mx_m > fg_t
                bg_mask = (hi_t > mx_m) & (mx_m >= lo_t)
            keep_all = np.zeros((B, nb), dtype=np.int64)
            # the candidate lists of ALL images from two nonzero() calls (row-major: already grouped by image, ascending inside)
            fg_cols, bg_cols = np.nonzero(fg_mask)[1], np.nonzero(bg_mask)[1]
            fg_end, bg_end = np.cumsum(fg_mask.sum(axis=1)), np.cumsum(bg_mask.sum(axis=1))
        else:
            out_r = np.zeros((B, nb, 4), dtype=_F)
            out_l = np.zeros((B, nb), dtype=_F)
            gt_keep = np.zeros((B, nb, 4), dtype=_F)
        for b, (i0, i1) in enumerate(zip(idx[:-1], idx[1:])):
            if batched:
                n_rows = int(n_all[b])
                fg = fg_cols[(fg_end[b - 1] if b else 0):fg_end[b]]
                bg = bg_cols[(bg_end[b - 1] if b else 0):bg_end[b]]
            else:
                gt = gt_c[i0:i1]
                allr = np.concatenate([rois_c[b], gt], axis=0) if gt.max() > -1 else rois_c[b]
                ov = box_iou_incl(allr, gt)
                mx, asg = ov.max(axis=-1), ov.argmax(axis=-1)
                n_rows = len(mx)
                lab = ids_c[i0:i1][asg].copy()
                lab[mx < fg_t] = 0
                gta = gt[asg]
                fg = np.nonzero(mx > fg_t)[0]
                bg = np.nonzero((hi_t > mx) & (mx >= lo_t))[0]
            n_other = n_rows - len(bg) - len(fg)           # fg and bg are disjoint; the set itself is built only if drawn from
            nfg = min(len(fg), int(cfg.rcnn_fg_prop * nb))
            if len(bg) + n_other < nb - nfg:
                print(f'~~~~ NOT ENOUGH BG: {len(bg)} / IGNORED ROIS: {n_other}, FILLING WITH POSITIVES: {len(fg)} ~~~~')
                if len(bg) + n_other < nb - len(fg):
                    print('~~~~ IMPOSSIBLE TO FILL THE RCNN BATCH, NOT ENOUGH ROIS ~~~~')
                    return None, None, None
                nfg = max(nfg, nb - (len(bg) + n_other))
            nbg = min(len(bg), nb - nfg)
            fgi = _draw(fg, nfg)
            bgi = _draw(bg, nbg)
            if len(fgi) + len(bgi) < nb:
                other = np.asarray(list(set(range(n_rows)) - set(bg) - set(fg)))   # reference order: Python set iteration
                bgi = np.hstack([bgi, _draw(other, nb - len(fgi) - len(bgi))])
            if batched:
                keep_all[b, :len(fgi)] = fgi
                keep_all[b, len(fgi):] = bgi
            else:
                keep = np.hstack((fgi, bgi)).astype(np.int64)
                out_r[b], out_l[b] = allr[keep], lab[keep]
                gt_keep[b] = gta[keep]
        if batched:            # labels / boxes / assigned GT of the kept rows, gathered for the whole batch at once
            ids_pad = np.zeros((B, gmax), dtype=_F)
            ids_pad[np.arange(gmax)[None, :] < np.asarray(lengths)[:, None]] = ids_c
            bi = np.arange(B)[:, None]
            asg_k, mx_k = asg_all[bi, keep_all], mx_all[bi, keep_all]
            out_l = ids_pad[bi, asg_k]
            out_l[mx_k < fg_t] = 0
            out_r = np.ascontiguousarray(all_pad[bi, keep_all])
            gt_keep = gt_pad[bi, asg_k]
        # one encode for the whole batch (a torch.log call per image costs more in dispatch than in work)
        t4 = box_encode(out_r.reshape(-1, 4), gt_keep.reshape(-1, 4)).reshape(B, nb, 4)
        li = out_l.astype(np.int64)
        self.last_labels_host = out_l
        if torch.device(device).type == 'cuda':
            # the [B, 16, 4 (1 + nc)] target tensor is 99 % zeros (4.9 MB at B = 128): its 4 values per foreground row go up as
            # [B, 16, 4] and are scattered into their class slot on the device (one slot of 4 per class, nets_utils.py:248-259)
            t4[li < 1] = 0
            t4_d, li_d = self._upload('t4', t4, device), self._upload('li', li, device)
            tgt = torch.zeros((B * nb, 1 + nc, 4), device=device, dtype=torch.float32)
            tgt[torch.arange(B * nb, device=device), li_d.view(-1)] = t4_d.view(-1, 4)
            return self._upload('r', out_r, device), tgt.view(B, nb, 4 * (1 + nc)), self._upload('l', out_l, device)
        bsel, rsel = np.nonzero(li >= 1)                               # one slot of 4 per class (nets_utils.py:248-259)
        for k in range(4):
            out_t[bsel, rsel, 4 * li[bsel, rsel] + k] = t4[bsel, rsel, k]
        return torch.from_numpy(out_r).to(device), torch.from_numpy(out_t).to(device), torch.from_numpy(out_l).to(device)

    def _upload(self, name, arr, device):
        """Host array -> device through a persistent pinned buffer, asynchronously: a pageable `.to(device)` is stream-ordered AND
        blocks the host, i.e. the host would sit behind the kernels already queued (the early backward pass of the RPN branch)
        instead of launching the second stage.  The buffer is reused every step: the previous copy was consumed a step ago."""
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if torch.device(device).type != 'cuda':
            return t.to(device)
        pin = self.__dict__.setdefault('_pinned', {})
        buf = pin.get(name)
        if buf is None or buf.numel() < t.numel() or buf.dtype != t.dtype:
            buf = pin[name] = torch.empty((max(t.numel(), 1024),), dtype=t.dtype).pin_memory()
        view = buf[:t.numel()].view(t.shape)
        view.copy_(t)
        return view.to(device, non_blocking=True)
