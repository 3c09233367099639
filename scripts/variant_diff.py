"""Which RoIs of a composition-flag variant differ from the reference golden (tests/golden/variants_b2.npz), and by how much."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from birdsoundclassif_amd import synth
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args
from helpers import filler_state_dict, load_golden

tag = sys.argv[1] if len(sys.argv) > 1 else 'posenc'
kw = dict(fpn_first=dict(fpn_first=True), sandwich=dict(sandwich_attn=True), posenc=dict(add_posenc=True),
          bifpn=dict(fpn='bifpn', n_bifpn_layers=2), attn5=dict(pyramid_top_n_attn=5))[tag]
g = load_golden('variants_b2.npz')
m, _ = build_model(default_args(device='cuda', **kw))
m.load_state_dict(filler_state_dict(**kw))
m = m.cuda().eval()
x = torch.from_numpy(synth.image_batch(0, 2))[:, None].cuda()
with torch.no_grad():
    fpn = m._fpn_nhwc(x)
    rois, scores, n, cls, reg, _ = m.head.forward_first_stage_device(fpn)
ref = g[f'{tag}.rois.full'].reshape(g[f'{tag}.rois.shape'])
rs = g[f'{tag}.roi_scores.full'].reshape(g[f'{tag}.roi_scores.shape'])
rois, scores = rois.cpu().numpy(), scores.cpu().numpy()
for b in range(2):
    for i in range(50):
        if not np.array_equal(rois[b, i], ref[b, i]):
            print(f'img {b} rank {i}: got {rois[b, i].tolist()} score {scores[b, i]:.8f} | ref {ref[b, i].tolist()} score {rs[b, i]:.8f}')
    gs, rset = set(map(tuple, rois[b].tolist())), set(map(tuple, ref[b].tolist()))
    print(f'img {b}: only in got {sorted(gs - rset)}  only in ref {sorted(rset - gs)}')
    print('   max |score diff| at equal rank', np.abs(scores[b] - rs[b]).max())
