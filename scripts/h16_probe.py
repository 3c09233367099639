"""Half-step kernel (csrc/igemm_h16.hip, NBM_H16=3: three workgroups per CU) against the two-stage kernel of igemm.hip on the deep-K launches
of the detect step (B = 64) -- same bits asserted.  usage: python scripts/h16_probe.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
torch.manual_seed(0)
# (label, H, W, Cin, N, k, stride, residual, relu, batch override)
SHAPES = [('1x1 512->128 @47x128', 47, 128, 512, 128, 1, 1, False, True, None), ('1x1 1024->256 @24x64', 24, 64, 1024, 256, 1, 1, False, True, None),
          ('1x1 1024->512 @24x64', 24, 64, 1024, 512, 1, 1, False, True, None), ('1x1 2048->512 @12x32', 12, 32, 2048, 512, 1, 1, False, True, None),
          ('1x1 512->2048 @12x32 + res', 12, 32, 512, 2048, 1, 1, True, True, None), ('3x3 s2 256->256 @47x128', 47, 128, 256, 256, 3, 2, False, True, None),
          ('3x3 s2 512->512 @24x64', 24, 64, 512, 512, 3, 2, False, True, None), ('1x1 512->384 @47x128', 47, 128, 512, 384, 1, 1, False, False, None),
          ('attention 1024->1408 @98304 rows', 1536 * B, 1, 1024, 1408, 1, 1, False, False, 1), ('attention 2048->2432 @24576 rows', 384 * B, 1, 2048, 2432, 1, 1, False, False, 1)]
print(f'{"launch (B = %d)" % B:<40}{"two-stage ms (TF/s)":>24}{"half-step, 3 per CU":>24}')
tot = [0.0, 0.0]
for label, H, W, Cin, N, k, st, res, relu, b1 in SHAPES:
    Bn = b1 or B
    x = torch.relu(torch.randn(Bn, H, W, Cin, device='cuda'))
    w = torch.randn(N, k * k * Cin, device='cuda') * 0.03
    sc, sh = torch.rand(N, device='cuda') + 0.5, torch.randn(N, device='cuda')
    Ho, Wo = (H + 2 * (k // 2) - k) // st + 1, (W + 2 * (k // 2) - k) // st + 1
    r = torch.randn(Bn, Ho, Wo, N, device='cuda') if res else None
    gf = 2.0 * Bn * Ho * Wo * N * k * k * Cin / 1e9
    outs, cells = [], []
    for mode in ('0', os.environ.get('H16_PROBE_MODE', '3')):
        os.environ['NBM_H16'] = mode
        f = lambda: ops.conv2d(x, w, k, k, st, k // 2, scale=sc, shift=sh, residual=r, act=ops.ACT_RELU if relu else ops.ACT_NONE)
        outs.append(f())
        ms = t(f)
        tot[int(mode != '0')] += ms
        cells.append(f'{ms:8.3f} ({gf / ms:5.1f})')
    assert torch.equal(outs[0], outs[1]), label
    print(f'{label:<40}' + ''.join(f'{c:>24}' for c in cells), flush=True)
    del x, w, r, outs
print(f'{"sum":<40}{tot[0]:>24.3f}{tot[1]:>24.3f}')
