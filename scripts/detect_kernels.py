"""13 detect steps (B = 64, front end included) for `rocprofv3 --kernel-trace --stats`: per-kernel time of the inference path alone."""
import sys, os, torch, numpy as np
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
    for _ in range(13):
        imgs, _ = fe(pcm, 22050); model.detect(imgs)
torch.cuda.synchronize()
