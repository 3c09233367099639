"""Split-bf16 weight-gradient kernel (csrc/igemm_split_tn.hip) against the fp32 kernel on the largest plain TN launches of a B = 128
training step: ms per launch and fp32-equivalent TF/s.  usage: python scripts/split_tn_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from birdsoundclassif_amd import ops
SHAPES = [  # label, M, N, K, groups
    ('cell-domain planes 448->256, 46 images', 46 * 1536, 256, 448, 25),
    ('F(4x4) planes 384->256 @47x128, 12288 tiles', 12288, 256, 384, 36),
    ('1x1 256->1024 @24x64 B=128', 196608, 1024, 256, 1),
    ('1x1 1024->256 @24x64 B=128', 196608, 256, 1024, 1),
    ('1x1 128->512 @47x128', 770048, 512, 128, 1),
    ('lateral 256->384 @94x256', 3080192, 384, 256, 1),
    ('attention qkv 1024->1408', 196608, 1408, 1024, 1),
    ('1x1 512->2048 @12x32', 49152, 2048, 512, 1),
]


def bench(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for label, M, N, K, G in SHAPES:
    g = torch.randn(G, M, N, device='cuda') * 0.1
    x = torch.randn(G, M, K, device='cuda') * 0.1
    out = torch.zeros(G, N, K, device='cuda')
    res = []
    for mode in ('0', '1'):
        os.environ['NBM_SPLIT_BF16'] = mode
        ms = bench(lambda: ops.conv_wgrad(g, x, out, B=1, H=M, W=1, Cin=K, N=N, groups=G, g_gs=M * N, x_gs=M * K, out_gs=N * K))
        res.append(ms)
    gf = 2.0 * G * M * N * K / 1e9
    print(f'{label:<50} fp32 kernel {res[0]:7.3f} ms ({gf / res[0]:6.1f} TF/s)   split bf16 {res[1]:7.3f} ms ({gf / res[1]:6.1f} TF/s fp32-equivalent)   x {res[0] / res[1]:.2f}', flush=True)
    del g, x, out
