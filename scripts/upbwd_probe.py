"""The bilinear backward of the finest top-down merge alone, at the training geometry: python upbwd_probe.py [B] [reps].
Prints ms per launch (HIP events) of the full gather and of the pattern-pruned one; under `rocprofv3 --pmc ...` the counters of
`upsample_bwd_kernel` tell what bounds it."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
Ho, Wo, Hi, Wi, C_, S = 188, 512, 94, 256, 384, 8
g = torch.zeros((B, Ho, Wo, C_), device='cuda')
def band(n):
    i = torch.arange(n, device='cuda')
    return ((i + 2) % S < 5) & ((i + 2) // S < (n - 1) // S + 1)
m = band(Ho)[:, None] & band(Wo)[None, :]
g[:, m] = 1.0
for name, kw in (('full', {}), ('pruned', {'pattern_stride': S})):
    ops.upsample_bilinear_bwd(g, Hi, Wi, **kw)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        out = ops.upsample_bilinear_bwd(g, Hi, Wi, **kw)
    ev[1].record()
    torch.cuda.synchronize()
    gb = (g.numel() * (1.0 if not kw else float(m.float().mean())) + out.numel()) * 4 / 1e9
    ms = ev[0].elapsed_time(ev[1]) / reps
    print(f'{name}: {ms:.3f} ms / launch, {gb:.1f} GB algorithmic -> {gb / ms:.2f} TB/s', flush=True)
