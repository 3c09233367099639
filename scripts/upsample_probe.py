"""nbm_upsample_bilinear_bwd on the level-0 geometry (188x512 -> 94x256, 384 channels)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.randn(B, 188, 512, 384, device='cuda')
ops.upsample_bilinear_bwd(g, 94, 256); torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
for s, e in ev:
    s.record(); ops.upsample_bilinear_bwd(g, 94, 256); e.record()
torch.cuda.synchronize()
print(f'B = {B}: {sorted(s.elapsed_time(e) for s, e in ev)[2]:.2f} ms')
