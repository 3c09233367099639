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
