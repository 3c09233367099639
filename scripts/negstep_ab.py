"""Negative training step (all 1000 proposals per image go through the second stage) with the FPN levels on demand vs dense."""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import synth, train as T
from birdsoundclassif_amd.nets import build_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
args = T.default_args(device='cuda')
model, crit = build_model(args)
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().train(); crit.train()
opt, _ = T.build_optimizer(model, args)
img = torch.from_numpy(np.tile(synth.image_batch(0, 8), (-(-B // 8), 1, 1))[:B].copy()).cuda()
neg = torch.from_numpy(np.tile(synth.image_batch(100, 8), (-(-B // 8), 1, 1))[:B].copy()).cuda()
bbs, idss, lens = [], [], []
for i in range(B):
    bb, ids, l = synth.label_batch(i % 8, 1); bbs.append(bb); idss.append(ids); lens += l
batch = [img, neg, torch.cat(bbs), torch.cat(idss), lens]
np.random.seed(0)
for _ in range(3):
    T.train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=False)
for mode in ('lazy', 'dense', 'lazy', 'dense'):
    from birdsoundclassif_amd import ondemand
    ondemand.LAZY_FINEST = mode != 'dense'            # dense: every pixel of the FPN levels is computed (the switch of the dense_finest_map leg)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        T.train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=True)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    print(mode, [round(x, 1) for x in ts], flush=True)
