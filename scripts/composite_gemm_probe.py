import os, sys
sys.path.insert(0, os.getcwd())
import torch
from birdsoundclassif_amd import ops
from birdsoundclassif_amd.ops import gemm_conv
T, K, N = 46 * 1536, 448, 256
V = torch.randn(25 * T * K, device='cuda') * 0.1
U = torch.randn(25, N, K, device='cuda') * 0.05
Wf = torch.randn(N, 25 * K, device='cuda') * 0.05
M = torch.empty(25, T, N, device='cuda')
p5 = torch.empty(5, T, N, device='cuda')
f = torch.empty(T, N, device='cuda')
def bench(fn, name, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f'{name:<70} {ms:7.3f} ms  {2.0 * T * N * 25 * K / ms / 1e9:6.1f} TF/s', flush=True)
bench(lambda: gemm_conv(V, U, M, B=1, H=T, W=1, Cin=K, N=N, groups=25, x_gs=T * K, w_gs=N * K, y_gs=T * N), '25 plane GEMMs (groups = 25) -> [25][T][N]')
bench(lambda: gemm_conv(V, Wf, p5, B=1, H=5, W=T, Cin=K, N=N, kh=5, kw=1, Ho=1, Wo=T, x_ld=K, w_ld=25 * K, groups=5, x_gs=5 * T * K, w_gs=5 * K, y_gs=T * N), '5 groups x (kh = 5 planes) -> [5][T][N]')
bench(lambda: gemm_conv(V, Wf, M, B=1, H=1, W=T, Cin=K, N=N, kh=1, kw=1, Ho=1, Wo=T, x_ld=K, w_ld=25 * K, groups=25, x_gs=T * K, w_gs=K, y_gs=T * N), '25 groups, weights read from the tap-major [N][25K] layout')
def chain():
    for p0 in range(0, 25, 5):
        gemm_conv(V[p0 * T * K:], Wf[:, p0 * K:], f, B=1, H=5, W=T, Cin=K, N=N, kh=5, kw=1, Ho=1, Wo=T, x_ld=K, w_ld=25 * K, residual=f if p0 else None, res_ld=N if p0 else None)
bench(chain, 'chain of 5 launches (kh = 5), residual-linked')
bench(lambda: torch.sum(p5, 0, out=f), 'sum of the 5 partials')
Tn = T
g = torch.randn(Tn, N, device='cuda')
Wb = torch.randn(25, K, N, device='cuda') * 0.05
Mk = torch.empty(25 * Tn * K, device='cuda')
bench(lambda: gemm_conv(g, Wb, Mk, B=1, H=Tn, W=1, Cin=N, N=K, groups=25, x_gs=0, w_gs=K * N, y_gs=Tn * K), 'dgrad: [T][256] x 25 x [256 -> 448] (shared A)')
vg = torch.randn(25, Tn, N, device='cuda')
bench(lambda: gemm_conv(vg, Wb, Mk, B=1, H=Tn, W=1, Cin=N, N=K, groups=25, x_gs=Tn * N, w_gs=K * N, y_gs=Tn * K), 'dgrad old: [25][T][256] x [256 -> 448]')
dW = torch.zeros(N, 25 * K, device='cuda')
bench(lambda: ops.conv_wgrad(g, V, dW, B=1, H=Tn, W=1, Cin=K, N=N, groups=25, g_gs=0, x_gs=Tn * K, out_gs=K, out_ld=25 * K), 'wgrad: g^T V (shared g) -> [N][25K]')
dU = torch.zeros(25, N, K, device='cuda')
bench(lambda: ops.conv_wgrad(vg, V, dU, B=1, H=Tn, W=1, Cin=K, N=N, groups=25, g_gs=Tn * N, x_gs=Tn * K, out_gs=N * K), 'wgrad old: Vg^T V -> [25][N][K]')
