"""Which output of the detect step changes when two lanes run at the same time?  Eager steps on two streams (the host queues far ahead
of the GPU, so the two steps overlap on the device), stage outputs compared with a solo run of the same lane."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from birdsoundclassif_amd import ops, ondemand, synth
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args
from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd
from helpers import filler_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model, _ = build_model(default_args(device='cuda'))
model.load_state_dict(filler_state_dict())
model = model.cuda().eval()
fe = SpectrogramFrontEnd('cuda')
pcm = [torch.from_numpy(synth.clip_batch_pcm16(300 + B * k, B)).cuda() for k in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(k):
    out = {}
    imgs, _ = fe(pcm[k], 22050)
    out['img'] = imgs
    fpn = model._fpn_nhwc(imgs[:, 0][:, None].contiguous(), lazy=True)
    rois, sc, n_roi, cls, reg, _ = model.head.forward_first_stage_device(fpn)
    out.update(rois=rois, roi_scores=sc, n_roi=n_roi, cls=cls, reg=reg)
    det, n_det = model.head.fast_rcnn.detect_device(fpn, rois, n_roi, 0.3, 0.05)
    out.update(det=det, n_det=n_det)
    for i in (2, 3, 4):
        out[f'fpn{i}'] = fpn[i]
    return out


with torch.no_grad():
    for k in range(2):                 # warm-up + solo references
        with torch.cuda.stream(streams[k]), ops.lane(k):
            run(k); run(k)
            ref = run(k)
        torch.cuda.synchronize()
        if k == 0:
            refs = [ {n: t.clone() for n, t in ref.items()} ]
        else:
            refs.append({n: t.clone() for n, t in ref.items()})
    for trial in range(3):
        outs = []
        for rep in range(3):
            for k in range(2):
                with torch.cuda.stream(streams[k]), ops.lane(k):
                    o = run(k)
                if rep == 2:
                    outs.append(o)
        torch.cuda.synchronize()
        for k in range(2):
            bad = [n for n in refs[k] if not torch.equal(refs[k][n], outs[k][n])]
            print(f'trial {trial} lane {k}: differing outputs: {bad}')
            for n in bad[:4]:
                a, b = refs[k][n].float(), outs[k][n].float()
                d = (a - b).abs()
                print(f'    {n}: max |diff| {float(d.max()):.3e}, {int((d > 0).sum())} of {d.numel()} elements')
    # control: the same two-stream schedule, both streams in lane 0 (a shared scratch MUST corrupt) -- shows the probe can see a race
    outs = []
    for rep in range(3):
        for k in range(2):
            with torch.cuda.stream(streams[k]), ops.lane(0):
                o = run(k)
            if rep == 2:
                outs.append(o)
    torch.cuda.synchronize()
    for k in range(2):
        bad = [n for n in refs[k] if not torch.equal(refs[k][n], outs[k][n])]
        print(f'control (both streams in lane 0) lane {k}: differing outputs: {bad}')
