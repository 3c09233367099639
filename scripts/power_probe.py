"""Clock / power of the GPU while (a) an HBM stream, (b) a deep-K GEMM, (c) the HBM-bound 256 -> 64 1x1 layer run in a loop:
does the part sustain the matrix pipe and HBM together?  (rocm-smi is sampled from a thread.)"""
import os, sys, subprocess, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--csv'], capture_output=True, text=True, timeout=5).stdout
            out.append(r)
        except Exception as e:
            out.append(str(e))
        time.sleep(0.2)
B = 64
x256 = torch.randn(B, 94, 256, 256, device='cuda')
w_a = torch.randn(64, 256, device='cuda') * 0.06
big_a = torch.randn(1, 98304, 1, 1024, device='cuda'); big_w = torch.randn(1536, 1024, device='cuda') * 0.03
res = torch.randn(B, 94, 256, 64, device='cuda')
cases = {'hbm stream (relu_bwd 3 x 1.6 GB)': lambda: ops.relu_bwd(x256, x256),
         'deep-K GEMM 98304 x 1536 x 1024': lambda: ops.conv2d(big_a, big_w),
         '1x1 256 -> 64 + residual (HBM-bound GEMM)': lambda: ops.conv2d(x256, w_a, residual=res)}
for name, f in cases.items():
    f(); torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 3.0:
        for _ in range(50): f()
        torch.cuda.synchronize(); n += 50
    dt = (time.perf_counter() - t0) / n
    stop.set(); th.join()
    print(f'== {name}: {1e3 * dt:.3f} ms / call')
    for r in out[2:5]:
        print('   ', ' | '.join(l for l in r.strip().splitlines()[:3]))
