"""The one-pixel RoI flip of the `--add_posenc` variant against the reference fixture: which proposal,
which coordinate, how far from x.5 BEFORE the reference's `.round()` (nets_utils.py:186), and which reassociation of the product
crosses x.5.  Prints, for image 1 / the proposal that decodes to x2 = 226|227:

    * the oracle's pre-round x2 (torch CPU fp32, the reference's arithmetic; pinned against the reference by tests/golden/*),
    * the product's pre-round x2 with the FPN 3x3 convolutions through Winograd F(2x2,3x3) (default) and through the direct
      implicit-GEMM kernel (Fn.WINOGRAD = False), with the streaming 1x1 kernel on / off, dense maps and on-demand maps.

Run on the GPU box: python scripts/posenc_flip.py > gpurun_out/posenc_flip.txt"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from birdsoundclassif_amd import synth
from birdsoundclassif_amd.nets import build_model, functional as Fn
from birdsoundclassif_amd.train import default_args
from helpers import filler_state_dict, load_golden, preround_corners
from oracle import nets_ref as O

tag, kw = 'posenc', dict(add_posenc=True)
g = load_golden('variants_b2.npz')
args = default_args(device='cuda', **kw)
m, _ = build_model(args)
sd = filler_state_dict(**kw)
m.load_state_dict(sd)
m = m.cuda().eval()
x = torch.from_numpy(synth.image_batch(0, 2))[:, None]
ref = torch.from_numpy(g[f'{tag}.rois.full'].reshape(g[f'{tag}.rois.shape']))
lim = torch.tensor([args.img_width - 1, args.img_height - 1, args.img_width - 1, args.img_height - 1], dtype=torch.float32)


def run(lazy):
    with torch.no_grad():
        o = m.forward_first_stage(x.cuda(), lazy=lazy)
    return o['rois'].cpu(), o['rpn_bbox_reg'].float().cpu()


def report(name, rois, reg, target=None):
    pre = preround_corners(reg, args)
    out = []
    for b in range(rois.shape[0]):
        for i in range(rois.shape[1]):
            if not torch.equal(rois[b, i], ref[b, i]):
                out.append((b, i))
    print(f'{name}: {len(out)} RoI rows differ from the reference fixture: {[(b, i, rois[b, i].tolist(), ref[b, i].tolist()) for b, i in out]}')
    if target is not None:
        b, anchor, j = target
        v = float(pre[b, anchor, j])
        print(f'   pre-round coordinate {j} of image {b}, anchor {anchor}: {v!r}  (distance to x.5: {abs(v - np.floor(v) - 0.5):.3e})')
    return pre, out


# the oracle (reference arithmetic on the CPU)
with torch.no_grad():
    r = O.forward_first_stage(sd, O.make_cfg(args), x)
print('oracle RoIs == fixture:', torch.equal(r['rois'], ref))
pre_o = preround_corners(r['rpn_bbox_reg'], args)

rois, reg = run(False)
pre, diff = report('product, dense maps, Winograd F(2x2,3x3) forward (default)', rois, reg)
target = None
if diff:
    b, i = diff[0]
    j = int((rois[b, i] != ref[b, i]).nonzero()[0])
    boxes = torch.minimum(pre[b].round().clamp(min=0), lim)
    cand = (boxes == rois[b, i]).all(-1).nonzero().flatten()
    half = 0.5 * float(rois[b, i, j] + ref[b, i, j])
    anchor = int(cand[(pre[b][cand, j] - half).abs().argmin()])
    target = (b, anchor, j)
else:
    # no flip in this run: look at the proposal of the fixture closest to x.5 among the fixture's RoIs of image 1
    b = 1
    boxes = torch.minimum(pre_o[b].round().clamp(min=0), lim)
    best = (1.0, None)
    for i in range(ref.shape[1]):
        cand = (boxes == ref[b, i]).all(-1).nonzero().flatten()
        for a in cand.tolist():
            fr = (pre_o[b, a] - pre_o[b, a].floor() - 0.5).abs()
            jj = int(fr.argmin())
            if float(fr[jj]) < best[0]:
                best = (float(fr[jj]), (b, a, jj))
    target = best[1]
b, anchor, j = target
vo = float(pre_o[b, anchor, j])
print(f'oracle (torch CPU fp32): pre-round coordinate {j} of image {b}, anchor {anchor}: {vo!r}  (distance to x.5: {abs(vo - np.floor(vo) - 0.5):.3e})')
report('product, dense maps, Winograd (default)', rois, reg, target)
for name, setup in (('product, dense maps, DIRECT 3x3 convolutions (Fn.WINOGRAD = False)', dict(wino=False)),
                    ('product, dense maps, Winograd, tiled 1x1 kernels (NBM_STREAM1X1=0)', dict(stream='0')),
                    ('product, dense maps, DIRECT 3x3 + tiled 1x1', dict(wino=False, stream='0')),
                    ('product, ON-DEMAND maps (cell transforms), Winograd', dict(lazy=True))):
    Fn.WINOGRAD = setup.get('wino', True)
    if 'stream' in setup:
        os.environ['NBM_STREAM1X1'] = setup['stream']
    try:
        rr, gg = run(setup.get('lazy', False))
    finally:
        Fn.WINOGRAD = True
        os.environ.pop('NBM_STREAM1X1', None)
    report(name, rr, gg, target)
d = reg.permute(0, 2, 3, 1).reshape(2, -1, 4)[b, anchor]
do = r['rpn_bbox_reg'].permute(0, 2, 3, 1).reshape(2, -1, 4)[b, anchor]
print(f'regression deltas of that anchor: product {d.tolist()}  oracle {do.tolist()}  |diff| {(d - do).abs().tolist()}')
