"""Time nbm_stem7x7 at B = 64 (HIP events, 20 launches)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
from birdsoundclassif_amd.nets import _prep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(B, 375, 1024, 1, device='cuda')
w1 = torch.randn(64, 3, 7, 7, device='cuda') * 0.1
wi, bi = torch.randn(3, 1, 1, 1, device='cuda'), torch.randn(3, device='cuda')
sc, sh = torch.rand(64, device='cuda') + 0.5, torch.randn(64, device='cuda')
f = _prep.stem_fold(w1, wi, bi)
for _ in range(3):
    ops.stem7x7(x, *f, sc, sh)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    y = ops.stem7x7(x, *f, sc, sh)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print(f'NBM_STEM_DBG={os.environ.get("NBM_STEM_DBG", "0")}: {ms:.3f} ms  write {y.numel() * 4 / ms / 1e6:.0f} GB/s  {2 * y.numel() * 56 / ms / 1e9:.1f} TF/s')
