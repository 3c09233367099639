import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
B = 32
for (H, W, Cin, N, k) in ((188, 512, 384, 256, 3), (94, 256, 384, 256, 3), (94, 256, 64, 64, 3), (24, 64, 1024, 256, 1)):
    x = torch.randn(B, H, W, Cin, device='cuda'); g = torch.randn(B * H * W, N, device='cuda')
    out = torch.zeros(N, k * k * Cin, device='cuda')
    fl = 2.0 * B * H * W * N * Cin * k * k / 1e12
    for _ in range(3):
        out.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.conv_wgrad(g, x, out, B=B, H=H, W=W, Cin=Cin, N=N, kh=k, kw=k, stride=1, pad=k // 2)
        e.record(); torch.cuda.synchronize()
    print(f'wgrad {H}x{W} {Cin}->{N} k{k}: {s.elapsed_time(e):7.2f} ms  {fl / s.elapsed_time(e) * 1e3:6.1f} TF/s', flush=True)
