import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
B = 32
T, C, N, G = B * 94 * 256, 384, 256, 16
V = torch.randn(G, T, C, device='cuda') * 0.5
U = torch.randn(G, N, C, device='cuda') * 0.05
M = torch.empty(G, T, N, device='cuda')
fl = 2.0 * G * T * C * N / 1e12
for _ in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.gemm_conv(V, U, M, B=1, H=T, W=1, Cin=C, N=N, groups=G, x_gs=T * C, w_gs=N * C, y_gs=T * N)
    e.record(); torch.cuda.synchronize()
print(f'16 x [{T} x {C}] x [{C} x {N}]: {s.elapsed_time(e):.2f} ms  {fl / s.elapsed_time(e) * 1e3:.1f} TF/s  (direct conv equivalent {2.25 * fl / s.elapsed_time(e) * 1e3:.1f} TF/s)')
x = torch.randn(B, 188, 512, C, device='cuda'); y = torch.empty(4 * x.numel() // 1, device='cuda') if False else None
# memory-pass proxies: write 4x|X| and read 4x|Y| + write |Y|
big = torch.empty(G, T, C, device='cuda')
for _ in range(2):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); big.copy_(V); e.record(); torch.cuda.synchronize()
print(f'copy {V.numel() * 4 / 1e9:.1f} GB -> {s.elapsed_time(e):.2f} ms ({2 * V.numel() * 4 / 1e9 / s.elapsed_time(e):.2f} TB/s r+w)')
