import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
from birdsoundclassif_amd.nets import _prep
def t(f, n=3):
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = f(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e), r
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for (H, W, C, N) in ((188, 512, 384, 256), (94, 256, 384, 256), (24, 64, 384, 256)):
    x = torch.relu(torch.randn(B, H, W, C, device='cuda'))
    w = torch.nn.Parameter(torch.randn(N, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5)
    b = torch.randn(N, device='cuda')
    ms_d, yd = t(lambda: ops.conv2d(x, _prep.krsc(w), 3, 3, 1, 1, shift=b))
    U = _prep.wino23(w)
    ms_w, yw = t(lambda: ops.conv3x3_winograd(x, U, b))
    ref = torch.nn.functional.conv2d(x[:1].permute(0, 3, 1, 2).double().cpu(), w.detach().double().cpu(), b.double().cpu(), padding=1)
    ed = (yd[:1].permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item()
    ew = (yw[:1].permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item()
    print(f'{H}x{W} B={B}: direct {ms_d:.2f} ms, winograd {ms_w:.2f} ms ({ms_d / ms_w:.2f}x); max err vs fp64: direct {ed:.2e}, winograd {ew:.2e} (scale {ref.abs().max().item():.2f})', flush=True)
