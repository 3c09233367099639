import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import synth
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args
from oracle import nets_ref as O
model, _ = build_model(default_args(device='cpu'))
sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
cfg = O.make_cfg()
x = torch.from_numpy(synth.image_batch(0, 4))[:, None]
print('cpus', os.cpu_count())
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    with torch.no_grad():
        O.forward(sd, cfg, x[:2], min_score=0.2)
        t = time.perf_counter(); O.forward(sd, cfg, x, min_score=0.2); dt = time.perf_counter() - t
    print(nt, 'threads', f'{dt:.2f} s for 4 clips -> {4/dt:.2f} clips/s', flush=True)
