"""Soak of the inference path: N eager detect steps (B = 64, front end included) + N // 4 bulk-style graph replays; device and host
memory must stay flat."""
import sys, os, time, resource, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import synth, bulk
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args
from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
model, _ = build_model(default_args(device='cuda'))
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().eval()
fe = SpectrogramFrontEnd('cuda')
pcms = [torch.from_numpy(np.tile(synth.clip_batch_pcm16(s, 8), (8, 1))).cuda() for s in range(3)]
rss = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20
log = []
with torch.no_grad():
    for it in range(N):
        imgs, _ = fe(pcms[it % 3], 22050)
        out = model(imgs, min_score=0.2)                      # the reference's forward: dicts on the host
        if it % 50 == 0 or it == N - 1:
            torch.cuda.synchronize()
            log.append((it, torch.cuda.memory_allocated() / 2 ** 30, rss()))
            print(f'step {it}: allocated {log[-1][1]:.2f} GiB, host max RSS {log[-1][2]:.2f} GiB, detections in clip 0: {sum(len(v["scores"]) for v in out[0].values())}', flush=True)
assert log[-1][1] < log[1][1] + 0.2, 'device memory grows'
assert log[-1][2] < log[1][2] + 0.3, 'host memory grows'
print('flat')
