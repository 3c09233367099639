import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
B = 16
x = torch.randn(B, 188, 512, 384, device='cuda')
w = torch.randn(256, 3456, device='cuda') * 0.02
y = torch.empty(B, 188, 512, 256, device='cuda')
fl = 2.0 * B * 188 * 512 * 256 * 3456 / 1e12
for name, act in (('normal', 0), ('no loads/stores (stale LDS)', 0x100), ('no loads + no barrier', 0x300), ('no barrier only', 0x200)):
    for _ in range(2):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.gemm_conv(x, w, y, B=B, H=188, W=512, Cin=384, N=256, kh=3, kw=3, stride=1, pad=1, act=act)
        e.record(); torch.cuda.synchronize()
    print(f'{name:32s} {s.elapsed_time(e):7.2f} ms  {fl / s.elapsed_time(e) * 1e3:6.1f} TF/s')
