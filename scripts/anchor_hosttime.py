"""Host time of the AnchorTargetLayer inside a training step (B = 128 by default), with its IoU / arg-max / threshold half on the
device (nbm_anchor_targets, default) and all on the host (train.ANCHOR_TARGETS_ON_DEVICE = False): mean ms per step inside
`AnchorTargetLayer.forward` + `SetCriterion.start_anchor_targets`, and the wall time of the step.
usage: python scripts/anchor_hosttime.py [B] [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from birdsoundclassif_amd import synth, train as T
from birdsoundclassif_amd.nets import build_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
args = T.default_args(device='cuda')
model, crit = build_model(args)
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().train()
crit.train()
opt, _ = T.build_optimizer(model, args)
img = torch.from_numpy(np.tile(synth.image_batch(0, 8), (-(-B // 8), 1, 1))[:B].copy()).cuda()
bbs, idss, lens = [], [], []
for i in range(B):
    bb, ids, l = synth.label_batch(i % 8, 1)
    bbs.append(bb), idss.append(ids)
    lens += l
batch = [img, img, torch.cat(bbs), torch.cat(idss), lens]
acc = {'layer': 0.0, 'start': 0.0}
layer = crit.anchor_target_layer
fwd, start = layer.forward, crit.start_anchor_targets


def timed_fwd(*a, **k):
    t = time.perf_counter()
    out = fwd(*a, **k)
    acc['layer'] += time.perf_counter() - t
    return out


def timed_start(*a, **k):
    t = time.perf_counter()
    out = start(*a, **k)
    acc['start'] += time.perf_counter() - t
    return out


layer.forward = timed_fwd
crit.start_anchor_targets = timed_start
np.random.seed(0)
for _ in range(4):
    T.train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=False)
for on in (True, False, True):
    T.ANCHOR_TARGETS_ON_DEVICE = on
    T.train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=False)
    acc['layer'] = acc['start'] = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        T.train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'B={B} anchor IoU on the {"device" if on else "host"}: AnchorTargetLayer.forward {1e3 * acc["layer"] / steps:.1f} ms / step on the host'
          f' (+ start_anchor_targets {1e3 * acc["start"] / steps:.2f} ms), step {1e3 * dt:.1f} ms', flush=True)
