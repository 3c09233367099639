"""Where the small torch launches of a training step come from (round 4: 174 fills + 243 adds + ~236 copies per step at B = 128, 4.8 ms):
one step under torch.profiler with stacks, aten ops that launch fill / add / copy kernels grouped by the package's own call site.
`python scripts/small_launches.py [B]` on the GPU box."""
import os
import sys
import collections
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import synth
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args, build_optimizer, train_one_step
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
args = default_args(device='cuda')
model, crit = build_model(args)
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().train()
crit.train()
opt, _ = build_optimizer(model, args)
img = torch.from_numpy(np.tile(synth.image_batch(0, 8), (-(-B // 8), 1, 1))[:B].copy())
bbs, idss, lens = [], [], []
for i in range(B):
    bb, ids, l = synth.label_batch(i % 8, 1)
    bbs.append(bb), idss.append(ids)
    lens += l
batch = [img.cuda(), img.cuda(), torch.cat(bbs), torch.cat(idss), lens]
np.random.seed(0)
for _ in range(3):
    train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=False)
torch.cuda.synchronize()
import traceback
from torch.overrides import TorchFunctionMode, resolve_name
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WANT = ('zeros', 'zeros_like', 'zero_', 'fill_', 'full', 'add', 'add_', '__add__', '__iadd__', '__radd__', 'mul', 'mul_', '__mul__', '__rmul__',
        '__imul__', 'sub', '__sub__', '__rsub__', 'div', '__truediv__', 'copy_', 'clone', 'contiguous', 'cat', 'sum', 'rsqrt', 'sqrt', 'to', 'float')
cnt = collections.Counter()


class Spy(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = getattr(func, '__name__', str(func))
        out = func(*args, **(kwargs or {}))
        if name in WANT:
            t = out if isinstance(out, torch.Tensor) else (args[0] if args and isinstance(args[0], torch.Tensor) else None)
            if t is not None and t.is_cuda:
                site = 'other'
                for fr in reversed(traceback.extract_stack(limit=12)[:-1]):
                    if 'birdsoundclassif_amd' in fr.filename:
                        site = f"{fr.filename.split('birdsoundclassif_amd/')[-1]}:{fr.lineno} {fr.name}"
                        break
                cnt[(name, site, 'big' if t.numel() > (1 << 20) else 'small')] += 1
        return out


with Spy():
    train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=False)
    torch.cuda.synchronize()
tot = collections.Counter()
for (name, site, size), n in cnt.items():
    tot[name] += n
print('python-level torch calls on CUDA tensors in one step (launch-producing ones):', dict(tot))
for (name, site, size), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:90]:
    print(f'{n:5d}  {name:12s} {size:5s}  {site}')
