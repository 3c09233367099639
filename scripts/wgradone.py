import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
B, H, W, Cin, N, k = 32, 188, 512, 384, 256, 3
mode = sys.argv[1] if len(sys.argv) > 1 else 'randn'
x = torch.randn(B, H, W, Cin, device='cuda'); g = torch.randn(B * H * W, N, device='cuda')
if mode == 'relu':
    x = torch.relu(x); g = g * 1e-3
if mode == 'zero':
    x.zero_(); g.zero_()
out = torch.zeros(N, k * k * Cin, device='cuda')
w = torch.randn(N, k * k * Cin, device='cuda') * 0.02
y = torch.empty(B, H, W, N, device='cuda')
fl = 2.0 * B * H * W * N * Cin * k * k / 1e12
for _ in range(3):
    out.zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.conv_wgrad(g, x, out, B=B, H=H, W=W, Cin=Cin, N=N, kh=k, kw=k, stride=1, pad=1)
    e.record(); torch.cuda.synchronize()
print(f'{mode} wgrad: {s.elapsed_time(e):7.2f} ms  {fl / s.elapsed_time(e) * 1e3:6.1f} TF/s', flush=True)
for _ in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.gemm_conv(x, w, y, B=B, H=H, W=W, Cin=Cin, N=N, kh=k, kw=k, stride=1, pad=1)
    e.record(); torch.cuda.synchronize()
print(f'{mode} fwd  : {s.elapsed_time(e):7.2f} ms  {fl / s.elapsed_time(e) * 1e3:6.1f} TF/s', flush=True)
