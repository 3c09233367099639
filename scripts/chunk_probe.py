"""Does running the HBM-bound front of the backbone (stem, max-pool, layer1, layer2) in sub-batches of k images keep the
producer -> consumer traffic of the 256-channel maps inside the 256 MB Infinity Cache?  GPU time of a hipGraph replay per k."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import synth
from birdsoundclassif_amd.nets import build_model, functional as Fn
from birdsoundclassif_amd.train import default_args

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
upto = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # run up to layer `upto`
model, _ = build_model(default_args(device='cuda'))
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().eval()
body, init_conv = model.backbone[0].body, model.backbone[0].init_conv
x = torch.rand(B, 375, 1024, 1, device='cuda')


def run(k):
    outs = []
    for c in x.split(k):
        s, b = body.bn1.affine()
        y = Fn.Stem.apply(c, init_conv.weight, init_conv.bias, body.conv1.weight, s, b)
        t = Fn.MaxPool.apply(y)
        for li in range(1, upto + 1):
            for blk in getattr(body, f'layer{li}'):
                t = blk(t)
        outs.append(t)
    return outs


with torch.no_grad():
    ref = torch.cat(run(B))
    for k in (B, 32, 16, 8, 4, 2, 1):
        if k > B:
            continue
        got = torch.cat(run(k))
        assert torch.equal(got, ref), k
        del got
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            run(k); st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                o = run(k)
            g.replay(); st.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for s0, e0 in ev:
                s0.record(st); g.replay(); e0.record(st)
            st.synchronize()
        ms = sorted(s0.elapsed_time(e0) for s0, e0 in ev)[2]
        print(f'B={B} sub-batch {k:3d}: {ms:7.2f} ms (stem .. layer{upto}), identical outputs', flush=True)
        del g, o
        torch.cuda.empty_cache()
