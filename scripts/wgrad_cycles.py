"""Weight-gradient kernel (igemm_tn_kernel), the larger launches of a B = 128 training step: ms, TF/s and -- with the timing-only build
(`make -C birdsoundclassif_amd/csrc ablate_nn`, NBM_LIB=birdsoundclassif_amd/libnbm_hip_ablate_nn.so) -- cycles per workgroup in prologue /
K loop / epilogue (the atomics), thread 0 of every workgroup, clock64.  usage: [NBM_LIB=...] python scripts/wgrad_cycles.py [B]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from birdsoundclassif_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
instr = 'ablate' in os.environ.get('NBM_LIB', '')
L = _lib.load()
buf = (C.c_ulonglong * 24)()
# (label, H, W, Cin, N, k, stride, groups-as-planes or 0)
LAUNCHES = [('layer1 3x3 64->64 @94x256', 94, 256, 64, 64, 3, 1), ('layer1 1x1 64->256 @94x256', 94, 256, 64, 256, 1, 1),
            ('layer1 1x1 256->64 @94x256', 94, 256, 256, 64, 1, 1), ('FPN lateral 1x1 256->384 @94x256', 94, 256, 256, 384, 1, 1),
            ('layer2.0 3x3 s2 128->128 @94x256', 94, 256, 128, 128, 3, 2), ('layer2 1x1 512->128 @47x128', 47, 128, 512, 128, 1, 1),
            ('layer3 1x1 256->1024 @24x64', 24, 64, 256, 1024, 1, 1), ('layer4 1x1 2048->512 @12x32', 12, 32, 2048, 512, 1, 1)]
for label, H, W, Cin, N, k, st in LAUNCHES:
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    x = torch.randn(B, H, W, Cin, device='cuda')
    g = torch.randn(B * Ho * Wo, N, device='cuda') * 0.1
    out = torch.zeros(N, k * k * Cin, device='cuda')
    f = lambda: ops.conv_wgrad(g, x, out, B=B, H=H, W=W, Cin=Cin, N=N, kh=k, kw=k, stride=st, pad=pad)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    gf = 2.0 * B * Ho * Wo * N * Cin * k * k / 1e9
    line = f'{label:<36} {ms:7.3f} ms {gf / ms:6.1f} TF/s'
    if instr:
        L.nbm_nn_dbg_read(buf); f(); torch.cuda.synchronize(); L.nbm_nn_dbg_read(buf)
        n = int(buf[23])
        if n:
            line += f'   {n:6d} workgroups, cycles each: prologue {int(buf[20]) / n:8.0f}  K loop {int(buf[21]) / n:9.0f}  epilogue {int(buf[22]) / n:8.0f}'
    print(line, flush=True)
    del x, g, out
