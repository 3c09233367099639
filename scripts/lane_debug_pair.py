"""ONE captured graph with two parallel branches (fork / join on two streams inside the capture), each branch a full detect step in
its own lane: correct right after the capture?  concurrent (faster than two sequential steps)?  usage: lane_debug_pair.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from birdsoundclassif_amd import ops, synth
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args
from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd
from helpers import filler_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model, _ = build_model(default_args(device='cuda'))
model.load_state_dict(filler_state_dict())
model = model.cuda().eval()
pcm_host = [torch.from_numpy(synth.clip_batch_pcm16(300 + B * k, B)) for k in range(2)]
fe = [SpectrogramFrontEnd('cuda'), SpectrogramFrontEnd('cuda')]
pcm = [torch.zeros((B, 66150), dtype=torch.int16, device='cuda') for _ in range(2)]
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()


def run(k):
    imgs, _ = fe[k](pcm[k], 22050)
    return model.detect(imgs[:, 0][:, None].contiguous(), 0.3, 0.05)


def both():
    s1.wait_stream(s0)                       # fork
    with torch.cuda.stream(s1), ops.lane(1):
        o1 = run(1)
    with ops.lane(0):
        o0 = run(0)
    s0.wait_stream(s1)                       # join
    return o0, o1


with torch.no_grad(), torch.cuda.stream(s0):
    for _ in range(2):
        both()
    s0.synchronize(); s1.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s0):
        outs = both()
for k in range(2):
    pcm[k].copy_(pcm_host[k])
torch.cuda.synchronize()
with torch.cuda.stream(s0):
    g.replay()
torch.cuda.synchronize()
got = [(o[0].clone(), o[1].clone()) for o in outs]
with torch.no_grad(), ops.lane(5):
    for k in range(2):
        d, n = run(k)
        torch.cuda.synchronize()
        print(f'[pair B={B}] branch {k}: first replay == eager: {torch.equal(d, got[k][0]) and torch.equal(n, got[k][1])}; detections {int(n.sum())}', flush=True)
# timing: paired replays vs a single-branch graph
with torch.no_grad(), torch.cuda.stream(s0):
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=s0), ops.lane(0):
        o_single = run(0)
time.sleep(1.5)
for name, gr, per in (('single', g1, 1), ('pair', g, 2)):
    with torch.cuda.stream(s0):
        for _ in range(6):
            gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s0):
        for _ in range(20):
            gr.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (20 * per)
    print(f'[pair B={B}] {name}: {dt * 1e3:.2f} ms per batch = {B / dt:.1f} clips/s', flush=True)
