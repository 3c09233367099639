"""Wall time of the B = 64 detect step (front end included), 3 blocks of 20 steps: python detectloop.py"""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import synth
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args
from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd
model, _ = build_model(default_args(device='cuda'))
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().eval()
fe = SpectrogramFrontEnd('cuda')
pcm = torch.from_numpy(np.tile(synth.clip_batch_pcm16(0, 8), (8, 1))).cuda()
with torch.no_grad():
    for _ in range(5):
        imgs, _ = fe(pcm, 22050); model.detect(imgs)
    for blk in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            imgs, _ = fe(pcm, 22050); det = model.detect(imgs)
        torch.cuda.synchronize()
        print(f'block {blk}: {1e3 * (time.perf_counter() - t0) / 20:.2f} ms / step', flush=True)
