"""Safe probe (no replay): after capturing two GraphedDetectors in a fresh process, list every live CUDA tensor whose storage lies in
a graph-private pool -- anything besides the graphs' static outputs was created DURING a capture and cached somewhere."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from birdsoundclassif_amd import bulk, ops, synth
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args
from helpers import filler_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
model, _ = build_model(default_args(device='cuda'))
model.load_state_dict(filler_state_dict())
model = model.cuda().eval()
dets = []
for k in range(2):
    dets.append(bulk.GraphedDetector(model, B, 66150, 22050, min_score=0.05, lane=0, independent=True))
    torch.cuda.synchronize()
    gc.collect()
    segs = [(s['address'], s['address'] + s['total_size'], s.get('segment_pool_id')) for s in torch.cuda.memory_snapshot()]
    priv = [s for s in segs if s[2] not in ((0, 0), None)]
    print(f'after capture {k}: {len(priv)} private-pool segments, pools {sorted({s[2] for s in priv})}', flush=True)
    seen = set()
    for o in gc.get_objects():
        try:
            if isinstance(o, torch.Tensor) and o.is_cuda and o.numel() > 0:
                p = o.data_ptr()
                if p in seen:
                    continue
                seen.add(p)
                for a, b, pool in priv:
                    if a <= p < b:
                        owners = [type(r).__name__ for r in gc.get_referrers(o)][:6]
                        own = 'static output' if any(p == d.det.data_ptr() or p == d.n_det.data_ptr() for d in dets) else 'OTHER'
                        print(f'   live tensor in pool {pool}: {tuple(o.shape)} {o.dtype} at {p:#x} [{own}] referrers {owners}', flush=True)
        except Exception:
            pass
print('done', flush=True)
