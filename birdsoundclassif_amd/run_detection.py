"""Per-file detection driver (reference nbm_model/run_detection.py): `load_model`, `run_detection`, `merge_images`.

`run_detection` keeps the reference's signature and output dict; the spectrogram windows stay on the GPU between the
front end and the detector, and the cross-window merge runs the device NMS."""
import json
import os

import numpy as np
import torch

from . import ops
from .nbm_datasets.prepare_dataset import File_Processor
from .nets.backbone import build_backbone
from .nets.fpn import build_fpn
from .nets.head import build_head
from .nets.nbm_model import NbmModel, initialize_model
from .nets.self_attention import build_sa_layers
from .nets.util.nets_utils import setattr_others

device = 'cuda'


def merge_images(fp, outputs, num_classes, nms_thresh=0.3):
    """Cross-window merge (reference run_detection.py:163-249): drop narrow border boxes (first / last / inner window
    rules, if/elif chain kept: Appendix C-5), shift by HOP_SPECTRO*i, drop boxes past the end of the file, then one
    class-agnostic greedy NMS over the file IN THE COLLECTED ORDER (class-major, not re-sorted by score)."""
    min_border_size = 0.9 * (fp.W_PIX - fp.HOP_SPECTRO)
    out = []
    for b_outputs in outputs:
        out.extend(b_outputs)
    boxes, scores, species = [], [], []
    for j in range(1, num_classes + 1):
        for i, img_out in enumerate(out):
            bb = img_out[str(j)]['bbox_coord']
            if len(bb) == 0:
                continue
            bb = bb.detach().float().cpu().clone()
            sc = img_out[str(j)]['scores'].detach().float().cpu().reshape(-1)
            widths = bb[:, 2] - bb[:, 0]
            if i == 0:
                drop = (bb[:, 2] >= fp.W_PIX - 5) & (widths < min_border_size)
            elif i == len(out) - 1:
                drop = (bb[:, 0] <= 4) & (widths < min_border_size)
            else:
                drop = ((bb[:, 0] <= 4) | (bb[:, 2] >= fp.W_PIX - 5)) & (widths < min_border_size)
            bb, sc = bb[~drop], sc[~drop]
            bb[:, 0] += fp.HOP_SPECTRO * i
            bb[:, 2] += fp.HOP_SPECTRO * i
            ok = ~(bb[:, 2] >= fp.spectrogram_length)
            bb, sc = bb[ok], sc[ok]
            if len(bb) == 0:
                continue
            boxes.append(bb), scores.append(sc), species.extend([j] * len(bb))
    class_bbox = {str(j): {'bbox_coord': torch.tensor([]), 'scores': torch.tensor([])} for j in range(1, num_classes + 1)}
    if not boxes:
        return class_bbox
    boxes, scores, species = torch.cat(boxes), torch.cat(scores), np.array(species)
    n = len(boxes)
    cap = max(64, (n + 63) // 64 * 64)
    if cap > 4096:
        raise NotImplementedError('more than 4096 candidate boxes in one file')
    bx = torch.zeros((1, cap, 4), device=device)
    sx = torch.zeros((1, cap), device=device)
    bx[0, :n], sx[0, :n] = boxes.to(device), scores.to(device)
    # index payload rides in the score slot so that the kept order can be recovered exactly
    idx_payload = torch.arange(cap, device=device, dtype=torch.float32)[None]
    n_in = torch.full((1,), n, device=device, dtype=torch.int32)
    _, kept_idx, n_out = ops.nms_batched(bx, idx_payload.contiguous(), n_in, nms_thresh, cap)
    keep = kept_idx[0, :int(n_out.item())].long().cpu()
    boxes, scores, species = boxes[keep], scores[keep], species[keep.numpy()]
    for j in range(1, num_classes + 1):
        m = torch.from_numpy(species == j)
        if m.any():
            class_bbox[str(j)] = {'bbox_coord': boxes[m], 'scores': scores[m]}
    return class_bbox


def run_detection(model, config, wav_path, bird_dicts_path, min_score=0.5, bs=10, visualise_outputs=False, show_sp_name=True):
    """reference run_detection.py:28-84 -> {species_name: {'bbox_coord': [[x1,y1,x2,y2]...], 'scores': [...]}}."""
    if visualise_outputs:
        raise NotImplementedError('visualisation is outside the hot-path scope')
    fp = File_Processor(wav_path)
    img_db, _ = fp.process_file(device=device)
    if img_db is None:
        return {}
    if len(img_db) > 0 and isinstance(img_db[0], list):
        # recordings longer than 56 min: process_file returns one image list per split (reference process_long_file);
        # the reference's own run_detection indexes that as a flat image list and fails (run_detection.py:44-55)
        raise NotImplementedError('recordings longer than 1.5e8 samples come back as nested per-split image lists, which '
                                  'the reference detection loop cannot consume either; split the recording first')
    imgs = fp.images_device                                   # [n_img, 375, 1024] on the GPU
    outputs = []
    for s in range(0, imgs.shape[0], bs):
        with torch.no_grad():
            outputs.append(model(imgs[s:s + bs][:, None].contiguous(), min_score=min_score))
    with open(bird_dicts_path, 'r') as f:
        birds_dict = json.load(f)
    birds_dict.update({'Non bird sound': 0})
    reverse_dict = {idx: name for name, idx in birds_dict.items()}
    class_bbox = merge_images(fp, outputs, config.num_classes)
    return {reverse_dict[idx]: {k: v.cpu().numpy().tolist() for k, v in class_bbox[str(idx)].items()}
            for idx in range(1, len(class_bbox) + 1) if len(class_bbox[str(idx)]['bbox_coord']) > 0}


def load_model(mod_p):
    """reference run_detection.py:87-122: directory with JSON `args` + `model_chkpt.pt`."""
    class Args:
        def __init__(self, **kwargs):
            for (k, v) in kwargs.items():
                setattr(self, k, v)

    with open(os.path.join(mod_p, 'args'), 'rb') as f:
        args = Args(**json.load(f))
    args.device = device          # placement follows this process, not the string stored in the file (Appendix C-13)
    setattr_others(args)
    backbone = build_backbone(args)
    attn = build_sa_layers(args, backbone.num_channels)
    fpn = build_fpn(args, backbone.num_channels)
    head = build_head(args)
    model = NbmModel(args, backbone, attn, fpn, head).to(device)
    model = initialize_model(model, path=os.path.join(mod_p, 'model_chkpt.pt'), train=False)
    return model, args
