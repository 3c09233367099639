"""Per-layer achieved TF/s of the data- and weight-gradient GEMMs in one training step (HIP events per launch)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops, synth, train as T
from birdsoundclassif_amd.nets import build_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
args = T.default_args(device='cuda')
model, crit = build_model(args)
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().train(); crit.train()
opt, _ = T.build_optimizer(model, args)
img = torch.from_numpy(np.tile(synth.image_batch(0, 8), (-(-B // 8), 1, 1))[:B].copy()).cuda()
bbs, idss, lens = [], [], []
for i in range(B):
    bb, ids, l = synth.label_batch(i % 8, 1); bbs.append(bb); idss.append(ids); lens += l
batch = [img, img, torch.cat(bbs), torch.cat(idss), lens]
np.random.seed(0)
for it in range(2):
    T.train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', False)
torch.cuda.synchronize()
ops.PROFILE_BWD = []
T.train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', False)
torch.cuda.synchronize()
prof, ops.PROFILE_BWD = ops.PROFILE_BWD, None
agg = {}
for tag, s, e in prof:
    agg.setdefault(tag, []).append(s.elapsed_time(e))
rows = []
for (kind, b, H, W, Cin, N, k, stride, groups), ts in agg.items():
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1            # 'same' padding everywhere on this path
    fl = 2.0 * b * Ho * Wo * N * Cin * k * k * groups / 1e9
    t = sum(ts)
    rows.append((t - len(ts) * fl / 140e3 * 1e3 / 1e3 * 1e0 if False else t - len(ts) * fl / 140.0, t, len(ts), kind, b, H, W, Cin, N, k, stride, groups, fl))
rows.sort(reverse=True)
for kind in ('dgrad', 'wgrad'):
    tt = sum(r[1] for r in rows if r[3] == kind); ff = sum(r[12] * r[2] for r in rows if r[3] == kind)
    print(f'{kind}: {tt:.1f} ms, {ff / 1e3:.1f} TFLOP -> {ff / tt:.1f} TF/s')
print('   lost(ms vs 140TF)  time   n  kind   B  HxW  Cin->N k s g   TF/s')
for lost, t, n, kind, b, H, W, Cin, N, k, stride, groups, fl in rows[:45]:
    print(f'{lost:8.2f} {t:8.2f} x{n:2d} {kind} B={b} {H}x{W} {Cin}->{N} k{k} s{stride} g{groups}  {fl * n / t:6.1f}')
