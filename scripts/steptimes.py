"""Per-step wall time (synchronised) of the first N training steps of a fresh process, 9 : 1 schedule: how long until the allocator settles."""
import sys, os, time, torch, numpy as np
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
img = torch.from_numpy(np.tile(synth.image_batch(0, 8), (-(-B // 8), 1, 1))[:B].copy()).cuda()
bbs, idss, lens = [], [], []
for i in range(B):
    bb, ids, l = synth.label_batch(i % 8, 1); bbs.append(bb); idss.append(ids); lens += l
batch = [img, img, torch.cat(bbs), torch.cat(idss), lens]
np.random.seed(0)
for it in range(N):
    neg = it % 10 == 9
    torch.cuda.synchronize(); t = time.perf_counter()
    train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=neg)
    torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    print(f'step {it:2d} neg={int(neg)} {1e3 * (time.perf_counter() - t):7.1f} ms  reserved {torch.cuda.memory_reserved() / 2 ** 30:6.1f} GiB  cudaMalloc calls {st.get("num_device_alloc", 0)}  retries {st.get("num_alloc_retries", 0)}', flush=True)
