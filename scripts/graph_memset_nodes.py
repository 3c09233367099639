"""Which HIP memory operations (memset / memcpy nodes) does the captured detect step hold besides its kernel nodes?  Memset nodes are the
node kind that the HIP runtime torch bundles (7.0.51831) loses in replayed execs (scripts/graph_pair_repro.hip).
usage: AMD_LOG_LEVEL=3 python graph_memset_nodes.py [B] 2> api.log ; scripts/graph_memset_nodes.py --summarize api.log"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from birdsoundclassif_amd import ops
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args
from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd
from helpers import filler_state_dict

if len(sys.argv) > 2 and sys.argv[1] == '--summarize':
    inside, calls = False, {}
    for line in open(sys.argv[2], errors='replace'):
        if '@@@ capture begins' in line:
            inside = True
        elif '@@@ capture ends' in line:
            inside = False
        elif inside:
            m = re.search(r'(hip[A-Z][A-Za-z0-9_]*)\s*\(', line)
            if m and 'returned' not in line.split(m.group(1))[0]:
                calls[m.group(1)] = calls.get(m.group(1), 0) + 1
                if 'Memset' in m.group(1) or 'Memcpy' in m.group(1):
                    print('   ', line.strip()[:260])
    print('HIP API calls inside the capture:', dict(sorted(calls.items(), key=lambda kv: -kv[1])))
    sys.exit(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model, _ = build_model(default_args(device='cuda'))
model.load_state_dict(filler_state_dict())
model = model.cuda().eval()
fe = SpectrogramFrontEnd('cuda')
pcm = torch.zeros((B, 66150), dtype=torch.int16, device='cuda')
st = torch.cuda.Stream()


def run():
    imgs, _ = fe(pcm, 22050)
    return model.detect(imgs[:, 0][:, None].contiguous(), 0.3, 0.05, independent=True)


with torch.no_grad(), torch.cuda.stream(st):
    for _ in range(2):
        run()
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    print('@@@ capture begins', file=sys.stderr, flush=True)
    with torch.cuda.graph(g, stream=st):
        out = run()
    print('@@@ capture ends', file=sys.stderr, flush=True)
