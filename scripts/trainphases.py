import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import synth
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args, build_optimizer
B = int(sys.argv[1])
args = default_args(device='cuda')
model, crit = build_model(args)
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().train(); crit.train()
opt, _ = build_optimizer(model, args)
base = synth.image_batch(0, 8)
img = torch.from_numpy(np.tile(base, (-(-B//8),1,1))[:B].copy()).cuda()
bbs, idss, lens = [], [], []
for i in range(B):
    bb, ids, l = synth.label_batch(i % 8, 1); bbs.append(bb); idss.append(ids); lens += l
bb, ids = torch.cat(bbs), torch.cat(idss)
np.random.seed(0)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(4):
    t0 = T()
    o = model.forward_first_stage(img[:, None]); t1 = T()
    l1 = crit.first_stage_loss(o['rpn_cls_scores'], o['rpn_bbox_reg'], bb, lens, False); t2 = T()
    pt = crit.generate_all_rois(o['rois'], bb, ids, lens); t3 = T()
    s = model.forward_second_stage(o['fpn_out'], pt['rois'], training=True)
    l2 = crit.second_stage_loss(s['bbox_reg'], s['bbox_classes'], pt['bbox_targets'], pt['labels'], False); t4 = T()
    loss = sum(l1.values()) + sum(l2.values())
    opt.zero_grad(); loss.backward(); t5 = T()
    opt.step(max_norm=0.1); t6 = T()
    print(f'it{it} fwd1 {1e3*(t1-t0):.0f}  loss1 {1e3*(t2-t1):.0f}  ptl {1e3*(t3-t2):.0f}  fwd2+loss2 {1e3*(t4-t3):.0f}  bwd {1e3*(t5-t4):.0f}  opt {1e3*(t6-t5):.1f}  total {1e3*(t6-t0):.0f} ms', flush=True)
