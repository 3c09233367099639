"""Host-side timeline of one training step (no extra syncs): where the Python thread spends its time."""
import os, sys, time, functools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import synth, train as T
from birdsoundclassif_amd.nets import build_model, targets, criterion as C, nbm_model, head

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
LOG = []
def wrap(obj, name, tag=None):
    f = getattr(obj, name)
    @functools.wraps(f)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); LOG.append((tag or name, t0, time.perf_counter())); return r
    setattr(obj, name, g)
wrap(targets.AnchorTargetLayer, 'forward', 'anchor_targets')
wrap(targets.ProposalTargetLayer, 'forward', 'proposal_targets')
wrap(C.SetCriterion, 'first_stage_loss'); wrap(C.SetCriterion, 'second_stage_loss'); wrap(C.SetCriterion, 'loss_cardinality')
wrap(nbm_model.NbmModel, '_fpn_nhwc', 'launch backbone+attn+fpn'); wrap(nbm_model.NbmModel, 'forward_second_stage')
wrap(head.Faster_RCNN, 'forward_first_stage_device', 'launch rpn+proposals')
wrap(nbm_model.NbmModel, 'forward_first_stage', 'forward_first_stage(total incl. sync)')
wrap(torch.Tensor, 'backward', 'backward launch'); wrap(T.FusedAdamW, 'step', 'optimizer')
wrap(T, 'allreduce_grads')
args = T.default_args(device='cuda')
model, crit = build_model(args)
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().train(); crit.train()
opt, _ = T.build_optimizer(model, args)
base = synth.image_batch(0, 8)
img = torch.from_numpy(np.tile(base, (-(-B // 8), 1, 1))[:B].copy()).cuda()
bbs, idss, lens = [], [], []
for i in range(B):
    bb, ids, l = synth.label_batch(i % 8, 1); bbs.append(bb); idss.append(ids); lens += l
batch = [img, img, torch.cat(bbs), torch.cat(idss), lens]
np.random.seed(0)
for it in range(4):
    LOG.clear()
    t0 = time.perf_counter()
    T.train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', False)
    t1 = time.perf_counter()
    if it == 3:
        print(f'host returns after {1e3 * (t1 - t0):.0f} ms')
        for n, a, b in sorted(LOG, key=lambda x: x[1]):
            print(f'  {1e3 * (a - t0):8.1f} -> {1e3 * (b - t0):8.1f}  ({1e3 * (b - a):7.1f} ms)  {n}')
torch.cuda.synchronize()
print(f'GPU drained {1e3 * (time.perf_counter() - t0):.0f} ms after step start')
