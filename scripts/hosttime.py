"""Host time to queue one detect step (B = 64) vs GPU time of the step: is the launch loop ahead of the GPU?"""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops, synth
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args
from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd
B = 64
model, _ = build_model(default_args(device='cuda'))
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().eval()
fe = SpectrogramFrontEnd('cuda')
pcm = torch.from_numpy(np.tile(synth.clip_batch_pcm16(0, 8), (8, 1))).cuda()
with torch.no_grad():
    for _ in range(3):
        imgs, _ = fe(pcm, 22050); model.detect(imgs)
    torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter()
        imgs, _ = fe(pcm, 22050); det = model.detect(imgs)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f'host queueing {1e3 * (t1 - t0):.1f} ms, until the GPU is done {1e3 * (t2 - t0):.1f} ms')
    for n in (5, 20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            imgs, _ = fe(pcm, 22050); det = model.detect(imgs)
        torch.cuda.synchronize()
        print(f'{n} steps back to back, no D2H / host post-processing: {1e3 * (time.perf_counter() - t0) / n:.2f} ms / step')
    fr = model.head.fast_rcnn
    for n in (5, 20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            imgs, _ = fe(pcm, 22050); det, nd = model.detect(imgs)
            out = fr.dets_to_dicts(det, nd, model.args.num_classes)       # synchronous D2H + dict building
        torch.cuda.synchronize()
        print(f'{n} steps, synchronous D2H + dicts each step: {1e3 * (time.perf_counter() - t0) / n:.2f} ms / step')
