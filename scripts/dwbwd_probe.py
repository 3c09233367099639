"""Time the depthwise 3x3 data gradient at the RPN's level-0 / level-1 geometries (B = 128)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
for (H, W, st) in ((188, 512, 8), (94, 256, 4), (47, 128, 2)):
    B, C, mult = 128, 256, 2
    Ho, Wo = (H + 2 - 3) // st + 1, (W + 2 - 3) // st + 1
    g = torch.randn(B, Ho, Wo, C * mult, device='cuda')
    w = torch.randn(C * mult, 1, 3, 3, device='cuda')
    x = torch.empty(B, H, W, C, device='cuda')
    f = lambda: ops.dwconv3x3_bwd(x, g, w, mult, st, need_gx=True, need_gw=False)
    f(); f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): f()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f'{H}x{W} stride {st}: {ms:.2f} ms, {B * H * W * C * 4 / ms / 1e6:.0f} GB/s written')
