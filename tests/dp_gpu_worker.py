"""Worker of tests/test_gpu_rccl.py::test_two_ranks_share_the_gpu_overlapped_exchange: one data-parallel rank on cuda:0 (gloo: RCCL
refuses two ranks on one device).  Two optimisation steps of the real model (B = 2, different clips per rank) with the overlapped
exchange (NBM_DP_OVERLAP, default) or the serial one; prints one JSON line: exchange stats, a digest of the post-step parameters and
whether both replicas ended up bit-identical."""
import datetime
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np   # noqa: E402
import torch         # noqa: E402
import torch.distributed as dist   # noqa: E402


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', timeout=datetime.timedelta(minutes=5))
    from birdsoundclassif_amd import synth, train as T
    from birdsoundclassif_amd.nets import build_model
    from helpers import filler_state_dict
    T.init_control_group(dist)
    args = T.default_args(device='cuda')
    model, crit = build_model(args)
    model.load_state_dict(filler_state_dict())
    model = model.cuda().train()
    crit.train()
    opt, _ = T.build_optimizer(model, args)
    img = torch.from_numpy(synth.image_batch(rank * 2, 2)).cuda()
    bb, ids, lens = synth.label_batch(rank * 2, 2)
    data = [img, img, bb, ids, lens]
    np.random.seed(5 + rank)
    T.exchange_stats_reset()
    grads = []
    for it in range(2):
        T.train_one_step(model, crit, opt, data, args.clip_max_norm, 'cuda', negative_sample=False)
        grads.append(torch.cat([g.flatten() for g in opt.flat_grads()]).double().cpu())
    st = T.exchange_stats_summary()
    flat = torch.cat([f['p'].flatten() for f in opt._flat if f is not None])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    same = all(torch.equal(both[0], t) for t in both)
    gnorm = [float(g.norm()) for g in grads]
    # sampled averaged gradients (compared between the overlapped and the serial run by the test)
    idx = torch.arange(4096, dtype=torch.int64) * ((grads[0].numel() - 1) // 4095)
    out = {'rank': rank, 'stats': st, 'replicas_identical': bool(same), 'grad_norms': gnorm,
           'grad_samples': grads[0][idx].tolist(), 'param_digest': hashlib.sha1(flat.cpu().numpy().tobytes()).hexdigest(),
           'overlap': T.DP_OVERLAP}
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
