"""Attribution of the data-gradient kernel's time: the largest `igemm_nn_kernel` launches of a B = 128 training step
(profiles/r04_train_layers.txt), each timed with ONE component of the kernel removed at a time (timing-only build: `make -C
birdsoundclassif_amd/csrc ablate_nn`, selected with NBM_LIB; NBM_NN_ABLATE bits are read per call).  Results of the ablated launches are
wrong by design; the row `shipped build` is the product library.

usage: NBM_LIB=birdsoundclassif_amd/libnbm_hip_ablate_nn.so python scripts/dgrad_ablate.py [B] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from birdsoundclassif_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
PEAK = 157.3
# (label, H, W, Cin = channels of dX, N = channels of G = K per tap, k, stride, a_scale, residual, mask, batch override)
LAUNCHES = [
    ('layer1 3x3 64->64 (G 64 ch, 9 taps -> dX 64)', 94, 256, 64, 64, 3, 1, True, False, True, None),
    ('layer1 1x1 256->64 (G 64 -> dX 256) + shortcut + mask', 94, 256, 256, 64, 1, 1, True, True, True, None),
    ('layer1 1x1 64->256 (G 256 -> dX 64) + mask', 94, 256, 64, 256, 1, 1, True, False, True, None),
    ('layer2 1x1 512->128 (G 128 -> dX 512) + shortcut + mask', 47, 128, 512, 128, 1, 1, True, True, True, None),
    ('layer3 1x1 1024->256 (G 256 -> dX 1024) + shortcut + mask', 24, 64, 1024, 256, 1, 1, True, True, True, None),
    ('layer3 1x1 256->1024 (G 1024 -> dX 256) + mask', 24, 64, 256, 1024, 1, 1, True, False, True, None),
    ('FPN lateral 1x1 256->384 @94x256 (G 384 -> dX 256) + residual', 94, 256, 256, 384, 1, 1, False, True, False, None),
    ('attention [q|k|v] 1024->1408 (G 1408 -> dX 1024), plain GEMM', 1536 * B, 1, 1024, 1408, 1, 1, False, False, False, 1),
    ('layer2.0 3x3 s2 128->128 @94x256 (by parity class) + mask', 94, 256, 128, 128, 3, 2, True, False, True, None),
    ('layer3.0 3x3 s2 256->256 @47x128 (by parity class) + mask', 47, 128, 256, 256, 3, 2, True, False, True, None),
]
if len(sys.argv) > 3:
    LAUNCHES = [l for l in LAUNCHES if sys.argv[3] in l[0]]
BITS = [(0, 'nothing removed'), (1, '- mask'), (2, '- residual(s)'), (4, '- a_scale multiply'), (16, '- global stores'),
        (32, '- whole epilogue'), (64, '- global loads (main loop)'), (128, '- LDS writes (main loop)'), (192, '- loads - LDS writes'),
        (224, 'MFMA + LDS reads + barriers only')]
ablate_build = 'ablate' in os.environ.get('NBM_LIB', '')


def bench(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


print(f'data-gradient kernel attribution, B = {B}, {reps} launches per cell, ms per launch (TF/s of the executed 2 M N K);',
      'timing-only ablation build' if ablate_build else 'SHIPPED build (only the first column means anything)')
hdr = f'{"launch":<66}' + ''.join(f'{name[:22]:>24}' for _, name in (BITS if ablate_build else BITS[:1]))
print(hdr)
for label, H, W, Cin, N, k, stride, sc, res, msk, b1 in LAUNCHES:
    Bn = b1 or B
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    g = torch.randn((Bn, Ho, Wo, N), device='cuda') * 0.1
    w = torch.randn((N, k * k * Cin), device='cuda') * 0.05
    out = torch.empty((Bn, H, W, Cin), device='cuda')
    a_scale = (torch.rand(N, device='cuda') + 0.5) if sc else None
    residual = torch.randn((Bn, H, W, Cin), device='cuda') if res else None
    mask = torch.randn((Bn, H, W, Cin), device='cuda') if msk else None
    gflop = 2.0 * Bn * Ho * Wo * N * k * k * Cin / 1e9
    cells = []
    for bits, _ in (BITS if ablate_build else BITS[:1]):
        os.environ['NBM_NN_ABLATE'] = str(bits)
        ms = bench(lambda: ops.conv_dgrad(g.view(-1, N), w, out, B=Bn, H=H, W=W, Cin=Cin, N=N, kh=k, kw=k, stride=stride, pad=k // 2, g_ld=N,
                                          w_ld=w.shape[1], a_scale=a_scale, residual=residual, mask=mask))
        cells.append(f'{ms:8.3f} ({gflop / ms:5.1f} TF/s)')
    os.environ['NBM_NN_ABLATE'] = '0'
    if ablate_build and os.environ.get('NBM_NN_CYCLES'):
        # cycle counters of the instrumented build (thread 0 of every workgroup, clock64): per parity class (class 0 when not phased)
        import ctypes as C
        from birdsoundclassif_amd import _lib
        L = _lib.load()
        buf = (C.c_ulonglong * 20)()
        L.nbm_nn_dbg_read(buf)                                       # clear
        for bits in (0, 224):
            os.environ['NBM_NN_ABLATE'] = str(bits)
            ops.conv_dgrad(g.view(-1, N), w, out, B=Bn, H=H, W=W, Cin=Cin, N=N, kh=k, kw=k, stride=stride, pad=k // 2, g_ld=N,
                           w_ld=w.shape[1], a_scale=a_scale, residual=residual, mask=mask)
            torch.cuda.synchronize()
            L.nbm_nn_dbg_read(buf)
            for c in range(4):
                adr, ld, loop, epi = (int(buf[4 * c + i]) for i in range(4))
                n = int(buf[16 + c])
                if n:
                    print(f'      ablate={bits:<3d} class {c}: {n:6d} tiles, cycles per tile: address arithmetic {adr / n:8.0f}  first loads + LDS write {ld / n:8.0f}  '
                          f'K loop {loop / n:8.0f}  epilogue {epi / n:8.0f}')
        os.environ['NBM_NN_ABLATE'] = '0'
    gb = (g.numel() + out.numel() + (residual.numel() if res else 0) + (mask.numel() if msk else 0)) * 4 / 1e9
    print(f'{label:<66}' + ''.join(f'{c:>24}' for c in cells) + f'   [{gb:.2f} GB mandatory -> {gb / 5.0:.2f} ms at 5 TB/s; {gflop / PEAK:.2f} ms at the MFMA peak]')
    del g, w, out, residual, mask
