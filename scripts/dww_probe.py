"""Weight gradient of the RPN's depthwise 3x3 taps at the training geometry (B = 128): ms per launch per level."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for (H, W, S) in ((188, 512, 8), (94, 256, 4), (47, 128, 2), (24, 64, 1)):
    x = torch.randn((B, H, W, 256), device='cuda')
    Ho, Wo = (H - 1) // S + 1, (W - 1) // S + 1
    g = torch.randn((B, Ho, Wo, 512), device='cuda')
    w = torch.randn((512, 1, 3, 3), device='cuda')
    ops.dwconv3x3_bwd(x, g, w, 2, S, need_gx=False)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        _, gw, gb = ops.dwconv3x3_bwd(x, g, w, 2, S, need_gx=False)
    ev[1].record(); torch.cuda.synchronize()
    print(f'{H}x{W} s{S}: {ev[0].elapsed_time(ev[1]) / 5:.3f} ms  crc {float(gw.double().sum()):.6e} {float(gb.double().sum()):.6e}', flush=True)
    del x, g
