"""Host half of the ProposalTargetLayer (reference layers.py:306-396) after the IoU moved to the device: the draws and the
`pre` hand-over must reproduce the all-host path exactly (CPU only; the device kernel is checked in test_gpu_ops.py)."""
import numpy as np
import torch

from birdsoundclassif_amd import synth
from birdsoundclassif_amd.nets import targets
from birdsoundclassif_amd.train import default_args


def test_draw_is_numpy_choice_without_replacement():
    rng = np.random.RandomState(3)
    for trial in range(300):
        n = int(rng.randint(0, 700))
        k = int(rng.randint(0, min(n, 16) + 1))
        pop = np.sort(rng.choice(5000, n, replace=False)).astype(np.int64)
        np.random.seed(trial)
        a, sa = np.random.choice(pop, k, replace=False), np.random.get_state()
        np.random.seed(trial)
        b, sb = targets._draw(pop, k), np.random.get_state()
        assert np.array_equal(a, b)
        assert sa[0] == sb[0] and np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]          # same position in the stream


def _batch(B, R, seed, near=5):
    rng = np.random.default_rng(seed)
    bbs, idss, lens = [], [], []
    for i in range(B):
        bb, ids, l = synth.label_batch((i + seed) % 8, 1)
        bbs.append(bb); idss.append(ids); lens += l
    gt, ids = torch.cat(bbs), torch.cat(idss)
    x1, y1 = rng.uniform(0, 900, (B, R)), rng.uniform(0, 300, (B, R))
    w, h = rng.uniform(10, 150, (B, R)), rng.uniform(10, 80, (B, R))
    rois = np.round(np.stack([x1, y1, np.minimum(x1 + w, 1023), np.minimum(y1 + h, 374)], -1)).astype(np.float32)
    g, k = gt.numpy(), 0
    for b, l in enumerate(lens):                      # some proposals close to the boxes: foreground candidates, exact ties
        for j in range(l):
            for t in range(min(near, R // 4)):
                rois[b, j * near + t] = np.round(g[k + j] + rng.uniform(-6, 6, 4))
        k += l
    return torch.from_numpy(rois), gt, ids, lens


def _pre_from_numpy(layer, rois, gt, lens, cap):
    """What SetCriterion.precompute_proposal_iou hands over, computed with the reference-order NumPy IoU."""
    gt_pad, batched = layer.pad_gt(gt.numpy().astype(np.float32), lens)
    assert batched
    B, R, G = rois.shape[0], rois.shape[1], gt_pad.shape[1]
    rois_h = np.full((B, cap, 4), 7.5, np.float32)   # rows beyond the RoI count hold anything
    rois_h[:, :R] = rois.numpy()
    mx, asg = np.zeros((B, cap + G), np.float32), np.zeros((B, cap + G), np.int32)
    for b in range(B):
        ov = targets.box_iou_incl(np.concatenate([rois_h[b], gt_pad[b]]), gt_pad[b][:lens[b]])
        mx[b], asg[b] = ov.max(1), ov.argmax(1)
    return rois_h, mx, asg, cap


def test_precomputed_iou_path_equals_host_path():
    layer = targets.ProposalTargetLayer(default_args(device='cpu'))
    for seed, (B, R, cap) in enumerate([(16, 1000, 1024), (8, 40, 64), (8, 17, 32), (4, 300, 512)]):
        rois, gt, ids, lens = _batch(B, R, seed)
        np.random.seed(seed)
        ref, s_ref = layer(rois, gt, ids, lens), np.random.get_state()
        np.random.seed(seed)
        got, s_got = layer(rois, gt, ids, lens, pre=_pre_from_numpy(layer, rois, gt, lens, cap)), np.random.get_state()
        for a, b in zip(ref, got):
            assert (a is None and b is None) or torch.equal(a, b)
        assert np.array_equal(s_ref[1], s_got[1]) and s_ref[2] == s_got[2]


def _anchor_pre_from_numpy(layer, gt, lens, args, poison=()):
    """What SetCriterion.start_anchor_targets hands over, computed with the reference-order NumPy arithmetic."""
    n_in, idx = len(layer.inds_inside), np.cumsum([0] + list(lens))
    lab, amx, flag = np.zeros((len(lens), n_in), np.int8), np.zeros((len(lens), n_in), np.int16), np.zeros(len(lens), np.int32)
    for b in range(len(lens)):
        lb, a = layer._host_labels(targets.box_iou_incl(layer.anchors_np, gt[idx[b]:idx[b + 1]]), args)
        lab[b], amx[b] = lb, a
        if b in poison:                                  # the device raised the image's flag: whatever it wrote must be ignored
            lab[b], amx[b], flag[b] = 1, 0, 1
    return lab, amx, flag


def test_anchor_targets_with_device_half_equal_the_all_host_layer():
    """AnchorTargetLayer (reference layers.py:102-216) with its IoU / arg-max / threshold half handed over (`pre`, what
    nbm_anchor_targets computes): same labels (i.e. same NumPy draws, same stream position) and same regression targets as the all-host
    layer; an image whose flag is raised (degenerate box on the device) is recomputed on the host."""
    args = default_args(device='cpu')
    layer = targets.AnchorTargetLayer(args)
    for seed, B in enumerate([8, 3, 16]):
        bbs, lens = [], []
        for i in range(B):
            bb, _, l = synth.label_batch((i + seed) % 8, 1)
            bbs.append(bb); lens += l
        gt = torch.cat(bbs)
        np.random.seed(seed)
        l0, t0 = layer(gt, lens, device='cpu')
        s0 = np.random.get_state()
        pre = _anchor_pre_from_numpy(layer, gt.numpy().astype(np.float32), lens, args, poison=(1,) if seed == 2 else ())
        np.random.seed(seed)
        l1, t1 = layer(gt, lens, device='cpu', pre=pre)
        s1 = np.random.get_state()
        assert torch.equal(l0, l1) and torch.equal(t0, t1) and int((l0 == 1).sum()) > 0
        assert s0[0] == s1[0] and np.array_equal(s0[1], s1[1]) and s0[2:] == s1[2:]
    # a degenerate box (x2 < x1 - 1): the masked targets are NaN where the reference's are -- both paths take the full encode
    gt_bad = gt.clone()
    gt_bad[0] = torch.tensor([50., 40., 20., 90.])
    np.random.seed(9)
    l0, t0 = layer(gt_bad, lens, device='cpu')
    pre = _anchor_pre_from_numpy(layer, gt_bad.numpy().astype(np.float32), lens, args)
    np.random.seed(9)
    l1, t1 = layer(gt_bad, lens, device='cpu', pre=pre)
    assert torch.equal(l0, l1) and torch.equal(torch.isnan(t0), torch.isnan(t1)) and torch.equal(torch.nan_to_num(t0), torch.nan_to_num(t1))
