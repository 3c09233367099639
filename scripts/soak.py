"""Soak: N training steps in the reference's 9 positive : 1 negative schedule at B = 128; prints losses every 10 steps, memory at the end.
Checks that nothing drifts (NaN), leaks (allocated / reserved memory) or trips the hand-over checks over many steps."""
import sys, os, time, resource, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import synth
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args, build_optimizer, train_one_step
B, N = int(sys.argv[1]), int(sys.argv[2])
args = default_args(device='cuda')
model, crit = build_model(args)
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().train(); crit.train()
opt, _ = build_optimizer(model, args)
batches = []
for s in range(4):
    img = torch.from_numpy(np.tile(synth.image_batch(s, 8), (-(-B // 8), 1, 1))[:B].copy()).cuda()
    bbs, idss, lens = [], [], []
    for i in range(B):
        bb, ids, l = synth.label_batch((i + s) % 8, 1); bbs.append(bb); idss.append(ids); lens += l
    batches.append([img, img, torch.cat(bbs), torch.cat(idss), lens])
np.random.seed(0)
mem = []
t0 = time.perf_counter()
for it in range(N):
    neg = it % 10 == 9
    loss = train_one_step(model, crit, opt, batches[it % 4], args.clip_max_norm, 'cuda', negative_sample=neg)
    if it % 10 in (0, 9) or it == N - 1:
        vals = {k: round(float(v), 4) for k, v in loss.items()}
        assert all(np.isfinite(v) for v in vals.values()), (it, vals)
        mem.append((torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20, neg))
        print(f'step {it:4d} neg={int(neg)} {vals}  allocated {mem[-1][0]:.1f} GiB reserved {mem[-1][1]:.1f} GiB host max RSS {mem[-1][2]:.2f} GiB', flush=True)
torch.cuda.synchronize()
print(f'{N} steps, {1e3 * (time.perf_counter() - t0) / N:.1f} ms / step incl. the printing syncs; max allocated {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB')
for kind in (False, True):           # like with like: a negative step leaves 2.7 GiB more allocated than a positive one
    m = [e for e in mem if e[3] == kind]
    if len(m) >= 4:
        assert m[-1][0] < m[len(m) // 2][0] + 1.0, 'allocated memory keeps growing'
        assert m[-1][2] < m[len(m) // 2][2] + 0.3, 'host memory keeps growing'
