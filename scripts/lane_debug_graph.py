"""Graph version of lane_debug.py: captured steps in given lanes, solo replays vs eager, then concurrent replays.
usage: lane_debug_graph.py B LANES(comma list, e.g. 0,1 or 1 or 0,0) INDEPENDENT(0/1) [concurrent(0/1)]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from birdsoundclassif_amd import bulk, ops, synth
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args
from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd
from helpers import filler_state_dict

B = int(sys.argv[1])
lanes = [int(v) for v in sys.argv[2].split(',')]
indep = bool(int(sys.argv[3]))
conc = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
tag = f'[B={B} lanes={lanes} independent={indep}]'
model, _ = build_model(default_args(device='cuda'))
model.load_state_dict(filler_state_dict())
model = model.cuda().eval()
pcm = [torch.from_numpy(synth.clip_batch_pcm16(300 + B * k, B)) for k in range(len(lanes))]
# eager references first (lane of its own: 7), before any graph exists
fe = SpectrogramFrontEnd('cuda')
eager = []
with torch.no_grad(), ops.lane(7):
    for k in range(len(lanes)):
        imgs, _ = fe(pcm[k].cuda(), 22050)
        d, n = model.detect(imgs[:, 0][:, None].contiguous(), 0.3, 0.05, independent=indep)
        torch.cuda.synchronize()
        eager.append((d.clone(), n.clone()))
if os.environ.get('NBM_DBG_PREWARM') == '1':          # allocate every lane's persistent scratch BEFORE the first capture
    with torch.no_grad():
        for l in sorted(set(lanes)):
            with ops.lane(l):
                imgs, _ = fe(pcm[0].cuda(), 22050)
                model.detect(imgs[:, 0][:, None].contiguous(), 0.3, 0.05, independent=indep)
    torch.cuda.synchronize()
mode = os.environ.get('NBM_DBG_COPY', 'after_pageable')
dets = []
pcm_dev = [p_.cuda() for p_ in pcm]
torch.cuda.synchronize()
for k, l in enumerate(lanes):
    dets.append(bulk.GraphedDetector(model, B, 66150, 22050, min_score=0.05, lane=l, independent=indep))
    if mode == 'each_d2d':                      # what bench.py does: device -> device, right after each capture
        dets[-1].pcm.copy_(pcm_dev[k])
    elif mode == 'each_pageable':
        dets[-1].pcm.copy_(pcm[k])
if mode == 'after_pageable':
    for k in range(len(lanes)):
        dets[k].pcm.copy_(pcm[k])
elif mode == 'after_d2d':
    for k in range(len(lanes)):
        dets[k].pcm.copy_(pcm_dev[k])
torch.cuda.synchronize()
tag += f'[{mode}]'
print(tag, 'captured', flush=True)
from birdsoundclassif_amd import ondemand
if os.environ.get('NBM_DBG_SNAPSHOT') == '1':
    for key, buf in ops._WINO_SCRATCH.items():
        print(tag, f'scratch lane {key[1]}: {buf.data_ptr():#x} .. {buf.data_ptr() + buf.numel() * 4:#x}', flush=True)
    for key, (tl, nb) in ondemand._ROI_TILE_BUF.items():
        print(tag, f'tile buffer {key[1:]}: {tl.data_ptr():#x} .. {tl.data_ptr() + tl.numel() * 4:#x}, counter {nb.data_ptr():#x}', flush=True)
    for k, d in enumerate(dets):
        print(tag, f'graph {k}: pcm {d.pcm.data_ptr():#x} det {d.det.data_ptr():#x} n_det {d.n_det.data_ptr():#x} basis {d.fe.basis.data_ptr():#x}', flush=True)
    for seg in sorted(torch.cuda.memory_snapshot(), key=lambda s: s['address']):
        used = sum(b['size'] for b in seg['blocks'] if b['state'] != 'inactive')
        print(tag, f"segment {seg['address']:#x} .. {seg['address'] + seg['total_size']:#x} pool {seg.get('segment_pool_id')} stream {seg.get('stream')} used {used}", flush=True)


def same(a, b):
    return torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


for k in range(len(lanes)):
    with torch.cuda.stream(dets[k].stream):
        dets[k].replay()
    torch.cuda.synchronize()
    print(tag, f'graph {k} (lane {lanes[k]}): solo replay == eager: {same((dets[k].det, dets[k].n_det), eager[k])}; detections {int(dets[k].n_det.sum())}', flush=True)
if conc and len(lanes) > 1:
    for trial in range(2):
        for _ in range(4):
            for k in range(len(lanes)):
                with torch.cuda.stream(dets[k].stream):
                    dets[k].replay()
        torch.cuda.synchronize()
        print(tag, f'trial {trial}: concurrent == eager:', [same((dets[k].det, dets[k].n_det), eager[k]) for k in range(len(lanes))], flush=True)
print(tag, 'done', flush=True)
