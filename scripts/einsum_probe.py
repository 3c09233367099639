import sys, time, torch
sys.path.insert(0, '/root/repo')
from birdsoundclassif_amd.nets import _prep
w = torch.randn(256, 384, 3, 3, device='cuda')
dU = torch.randn(25, 256, 384, device='cuda')
for name, fn in (('cell_weight fwd', lambda: _prep.cell_weight(w, forward=True)), ('cell_weight bwd', lambda: _prep.cell_weight(w)),
                 ('cell_weight_grad', lambda: _prep.cell_weight_grad(dU))):
    for it in range(3):
        _prep.bump(); torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
        print(name, it, f'{(time.perf_counter() - t) * 1e3:.2f} ms')
