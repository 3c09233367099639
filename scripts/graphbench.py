"""B = 64 detect step: eager launches vs one hipGraph replay (bulk.GraphedDetector), same work, wall time per step."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import synth
from birdsoundclassif_amd.bulk import GraphedDetector
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args
B = 64
model, _ = build_model(default_args(device='cuda'))
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().eval()
pcm = torch.from_numpy(np.tile(synth.clip_batch_pcm16(0, 8), (8, 1))).cuda()
g = GraphedDetector(model, B, pcm.shape[1], 22050, min_score=0.2)
g.pcm.copy_(pcm)
for name, fn in (('graph replay', g.replay), ('eager', lambda: g._run())):
    with torch.no_grad():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        print(f'{name}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms / step of {B} clips')
