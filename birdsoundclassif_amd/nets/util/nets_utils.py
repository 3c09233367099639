"""Functional helpers of the detector (reference nets/util/nets_utils.py).  Anchor generation and the
config plumbing are host-side integer work; `nms` runs the device bitmask NMS."""
import numpy as np
import torch
import torch.nn as nn

IMG_SIZE = (375, 1024)


def generate_anchors_frcnn(base_size, ratios, scales):
    """15 base anchors, scale-major / ratio-minor, int-truncated (reference nets_utils.py:35-49)."""
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales)
    side = np.sqrt(float(base_size) * float(base_size))
    wh = np.stack([np.sqrt(ratios), 1.0 / np.sqrt(ratios)], axis=1) * side                 # [R, 2]
    wh = (wh.reshape(1, -1) * scales[:, None]).reshape(-1, 2)
    return (np.concatenate([-wh / 2, wh / 2], axis=1) + int(base_size / 2)).astype(int)


def get_anchor_shifts_frcnn(width, height, anchor_stride):
    """[K,1,4] shifts, x fastest (reference nets_utils.py:52-59)."""
    sx = np.arange(0, width) * anchor_stride
    sy = np.arange(0, height) * anchor_stride
    xy = np.stack([np.tile(sx, len(sy)), np.repeat(sy, len(sx))], axis=1)
    return np.concatenate([xy, xy], axis=1).reshape(-1, 1, 4)


def weight_init(m):
    """reference nets_utils.py:149-156."""
    classname = m.__class__.__name__
    if classname.find('BatchNorm') != -1:
        m.weight.data.normal_(0.0, 0.02)
    if (classname.find('Linear') != -1) & (classname.find('LinearLayer') == -1):
        nn.init.kaiming_normal_(m.weight)
    if (classname.find('Conv2d') != -1) and (classname.find('DepthwiseSepConv2d') == -1):
        nn.init.kaiming_normal_(m.weight)


def collate_fn(list_batch):
    """reference nets_utils.py:159-166."""
    lengths = [len(elt[2]) for elt in list_batch]
    img_batch = torch.stack([e[0] for e in list_batch])
    neg_img_batch = torch.stack([e[1] for e in list_batch])
    bb_coord_batch = torch.cat([e[2] for e in list_batch], dim=0)
    bird_ids = torch.cat([e[3] for e in list_batch])
    return [img_batch, neg_img_batch, bb_coord_batch, bird_ids, lengths]


def bool_parser(string):
    return string.lower() != 'false'


def train_test_split(length, val_prop):
    indices = np.arange(length)
    np.random.shuffle(indices)
    cut = int(val_prop * length)
    return indices[cut:], indices[:cut]


def setattr_others(args):
    """Derived config fields (reference nets_utils.py:405-416)."""
    if args.n_ratios == 3:
        setattr(args, 'ratios', [0.5, 1, 2])
    elif args.n_ratios == 5:
        setattr(args, 'ratios', [0.2, 0.5, 1, 2, 5])
    if 'vgg' in args.backbone:
        setattr(args, 'n_layers', 4)
        setattr(args, 'top_size', (23, 64))
    else:
        setattr(args, 'n_layers', 5)
        setattr(args, 'top_size', (24, 64))
    setattr(args, 'scales', 2 ** np.arange(args.n_layers))


def nms(bbox_pred, scores, nms_thresh=0.7, post_nms_topN=300, return_idx=False):
    """Greedy NMS in the given order with the batch-coupled truncation (reference nets_utils.py:210-245),
    on the device bitmask kernel.  bbox_pred [B,N,4], scores [B,N] (any device; moved to the GPU)."""
    from ... import ops
    dev = bbox_pred.device if bbox_pred.is_cuda else torch.device('cuda')
    B, N = scores.shape
    cap = max(64, (N + 63) // 64 * 64)
    if cap > 4096:
        raise NotImplementedError('device NMS handles up to 4096 boxes per image')
    bx = torch.zeros((B, cap, 4), device=dev, dtype=torch.float32)
    sc = torch.zeros((B, cap), device=dev, dtype=torch.float32)
    bx[:, :N], sc[:, :N] = bbox_pred.to(dev), scores.to(dev)
    n_in = torch.full((1,), N, device=dev, dtype=torch.int32)
    post = min(int(post_nms_topN), cap)
    rois, rs, n_out = ops.nms_batched(bx, sc, n_in, nms_thresh, post)
    n = int(n_out.item())
    out = (rois[:, :n].to(bbox_pred.device), rs[:, :n].to(bbox_pred.device))
    if return_idx:
        # keep lists are recovered from the workspace-free outputs by matching positions (order preserving)
        raise NotImplementedError('return_idx: use ops.nms_batched directly')
    return out


# --------------------------------------------------------------------------- annotations and evaluation metrics
def read_annot_file(annot_path):
    """Audacity spectral-label file -> list of [time line, frequency line] pairs (reference nets_utils.py:419-430)."""
    with open(annot_path, 'r') as f:
        lines = f.readlines()
    return [[lines[i], lines[i + 1]] for i in range(0, len(lines) - 1, 2)]


def format_single_annot(annot, pix_precision_y=33.3, pix_precision_x=0.002993197278911565, low_freq=500, h_pix=375):
    """('t0\\tt1\\tspecies', '\\\\\\tf0\\tf1') -> (species, [x1, y1, x2, y2]) in spectrogram pixels of the WHOLE file
    (reference nets_utils.py:433-440): seconds / DT and (Hz - 500) / 33.3, rounded half to even, y clipped to the image."""
    t0, t1, spec = annot[0].replace('\n', '').split('\t')
    f0, f1 = annot[1].replace('\n', '').replace('\\\t', '').split('\t')
    return spec, [np.round(float(t0) / pix_precision_x), np.round((float(f0) - low_freq) / pix_precision_y).clip(min=0),
                  np.round(float(t1) / pix_precision_x), np.round((float(f1) - low_freq) / pix_precision_y).clip(max=h_pix - 1)]


def format_txt_annots(annot_path):
    """-> {species: [[x1,y1,x2,y2], ...]} (reference nets_utils.py:443-451)."""
    out = {}
    for annot in read_annot_file(annot_path):
        spec, coords = format_single_annot(annot)
        out.setdefault(spec, []).append(coords)
    return out


def _iou_plus1(a, b):
    """[K,4] x [N,4] -> [K,N] IoU, inclusive-pixel (+1) convention (reference `bbox_overlap`, nets_utils.py:103-126)."""
    a, b = np.asarray(a, dtype=np.float64).reshape(-1, 4), np.asarray(b, dtype=np.float64).reshape(-1, 4)
    iw = (np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]) + 1).clip(min=0)
    ih = (np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]) + 1).clip(min=0)
    inter = iw * ih
    area_a = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    area_b = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    return inter / (area_a[:, None] + area_b[None, :] - inter)


_RECALL_EDGES = np.arange(0, 1.1, 0.1)          # the reference's `pd.cut` edges, float artefacts included


def calculate_ap(kind):
    """Rows already sorted by decreasing confidence, `kind[i]` in {'TP','FP','FN'} -> (AP, recall), the reference's
    definition (nets_utils.py:509-534): running precision with the denominator clipped at TP+FP, running recall,
    precision interpolated as the max over rows of EQUAL recall, rows binned into 10 right-closed recall intervals,
    AP = (sum over non-empty bins of the bin's mean interpolated precision) / 10.  AP = -1 when nothing was predicted."""
    kind = np.asarray(kind)
    n_tp, n_fp, n_fn = int((kind == 'TP').sum()), int((kind == 'FP').sum()), int((kind == 'FN').sum())
    recall_total = n_tp / max(1, n_tp + n_fn)
    if n_tp + n_fp == 0:
        return -1, recall_total
    hits = np.cumsum(kind == 'TP')
    precision = hits / np.arange(1, len(kind) + 1).clip(max=n_tp + n_fp)
    recall = hits / max(1, n_tp + n_fn)
    _, group = np.unique(recall, return_inverse=True)
    best = np.full(group.max() + 1, -np.inf)
    np.maximum.at(best, group, precision)
    prec_interp = best[group]
    bins = np.searchsorted(_RECALL_EDGES, recall, side='left') - 1
    bins[recall == _RECALL_EDGES[0]] = 0                                   # include_lowest
    ok = (bins >= 0) & (bins < 10)
    ap = sum(prec_interp[ok & (bins == b)].mean() for b in np.unique(bins[ok])) / 10
    return ap, recall_total


def compute_AP_scores(outputs, filter_sp=None):
    """outputs: list of (detections {species: {'bbox_coord', 'scores'}}, ground truth {species: [[x1,y1,x2,y2], ...]})
    per file -> {'AP', 'mAP', 'Rec', 'mRec'} (reference nets_utils.py:454-506, IoU threshold 0.5).  Like the reference:
    a predicted box is TP when its best IoU with a same-species GT box is >= 0.5 (several boxes may claim one GT box),
    GT boxes of species that were not predicted at all in the file count as FN, unmatched GT boxes of predicted species
    do not."""
    species, iou, scores = [], [], []
    for output, annots in outputs:
        for spec in output:
            sc = np.asarray(output[spec]['scores'], dtype=np.float64).reshape(-1)
            if spec in annots:
                best = _iou_plus1(np.asarray(output[spec]['bbox_coord'], dtype=np.float32), annots[spec]).max(axis=1)
            else:
                best = np.zeros(len(sc))
            species += [spec] * len(sc)
            iou += best.tolist()
            scores += sc.tolist()
    # FN rows are appended per file after that file's predictions in the reference; order among score-0 rows is immaterial
    for output, annots in outputs:
        for spec in annots:
            if spec not in output:
                species += [spec] * len(annots[spec])
                iou += [0.0] * len(annots[spec])
                scores += [0.0] * len(annots[spec])
    if not species:
        return {'AP': 0, 'mAP': 0, 'Rec': 0, 'mRec': 0}
    species, iou, scores = np.array(species, dtype=object), np.array(iou), np.array(scores)
    kind = np.where(scores == 0, 'FN', np.where(iou >= 0.5, 'TP', 'FP'))
    order = np.argsort(-scores, kind='stable')
    species, kind = species[order], kind[order]
    if filter_sp is not None:
        sel = np.array([s in filter_sp for s in species], dtype=bool)
        species, kind = species[sel], kind[sel]
    AP, Rec = calculate_ap(kind)
    per = [calculate_ap(kind[species == s]) for s in sorted(set(species.tolist()))]
    mAP = np.array([a for a, _ in per if a > -1]).mean()
    mRec = np.array([r for _, r in per]).mean()
    return {'AP': AP, 'mAP': mAP, 'Rec': Rec, 'mRec': mRec}
