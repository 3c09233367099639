"""Peak device memory of a B = 128 training step against the number of RoIs per image that reach the second stage (VERDICT r4 weak #9:
headroom of the negative step, whose 1000 RoIs per image -- post_nms_topN, reference layers.py:233 -- go through the head unsampled).
usage: python scripts/negstep_memory.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from birdsoundclassif_amd import synth
from birdsoundclassif_amd.nets import build_model
from birdsoundclassif_amd.train import default_args, build_optimizer, train_one_step
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
img = torch.from_numpy(synth.image_batch(0, B)).cuda()
neg = torch.from_numpy(synth.image_batch(100000, B)).cuda()
bb, ids, lens = synth.label_batch(0, B)
for post in (500, 1000, 1500, 2000):
    args = default_args(device='cuda', post_nms_topN=post, pre_nms_topN=max(3000, post))
    model, crit = build_model(args)
    model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
    model = model.cuda().train(); crit.train()
    opt, _ = build_optimizer(model, args)
    np.random.seed(0)
    batch = [img, neg, bb, ids, list(lens)]
    train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=False)
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=False)
    torch.cuda.synchronize(); p_pos = torch.cuda.max_memory_allocated() / 2 ** 30
    torch.cuda.reset_peak_memory_stats()
    try:
        train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=True)
        torch.cuda.synchronize(); p_neg = torch.cuda.max_memory_allocated() / 2 ** 30
        note = ''
    except torch.cuda.OutOfMemoryError as exc:
        p_neg, note = float('nan'), ' OUT OF MEMORY'
    print(f'B = {B}, post_nms_topN = {post}: positive step peak {p_pos:6.1f} GiB, negative step ({post} RoIs per image through the head) peak {p_neg:6.1f} GiB{note}', flush=True)
    del model, crit, opt
    from birdsoundclassif_amd import ondemand
    ondemand.zero_pool_clear()
    torch.cuda.empty_cache()
