import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import synth
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args, build_optimizer, train_one_step
B = int(sys.argv[1]); steps = int(sys.argv[2])
args = default_args(device='cuda')
model, crit = build_model(args)
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().train(); crit.train()
opt, _ = build_optimizer(model, args)
base = synth.image_batch(0, 8); 
img = torch.from_numpy(np.tile(base, (-(-B//8),1,1))[:B].copy())
bbs, idss, lens = [], [], []
for i in range(B):
    bb, ids, l = synth.label_batch(i % 8, 1); bbs.append(bb); idss.append(ids); lens += l
batch = [img.cuda(), img.cuda(), torch.cat(bbs), torch.cat(idss), lens]
np.random.seed(0)
for neg in (False,):
    for it in range(steps + 1):
        torch.cuda.synchronize(); t = time.perf_counter()
        loss = train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=neg)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f'B={B} neg={neg} it={it} {dt*1e3:.1f} ms  {B/dt:.1f} clips/s  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB reserved {torch.cuda.memory_reserved()/2**30:.1f} retries {torch.cuda.memory_stats().get("num_alloc_retries", 0)}', {k: round(float(v),4) for k,v in loss.items()}, flush=True)
