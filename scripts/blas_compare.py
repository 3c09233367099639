"""fp32 GEMM C = A W^T through torch.matmul (rocBLAS / hipBLASLt) against nbm_gemm_conv on the deep-K shapes of the detect step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
torch.backends.cuda.matmul.allow_tf32 = False
def timeit(f, n=7):
    f(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in ev:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in ev)[n // 2]
for M, N, K in [(24576, 3072, 2048), (98304, 1536, 1024), (98304, 256, 1024), (98304, 1024, 512), (393216, 128, 512), (24576, 512, 2048)]:
    a = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * K ** -0.5
    t_lib = timeit(lambda: torch.matmul(a, w.t()))
    x = a.view(1, M, 1, K)
    t_own = timeit(lambda: ops.conv2d(x, w))
    err = float((ops.conv2d(x, w).view(M, N) - torch.matmul(a, w.t())).abs().max())
    gf = 2.0 * M * N * K * 1e-9
    print(f'M={M:6d} N={N:4d} K={K:4d}: torch.matmul {t_lib:.3f} ms = {gf / t_lib:.1f} TF/s   nbm_gemm_conv {t_own:.3f} ms = {gf / t_own:.1f} TF/s   max |diff| {err:.2e}', flush=True)
