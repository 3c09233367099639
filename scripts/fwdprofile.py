"""Per-launch table of one detect() forward at batch B (HIP events around every implicit-GEMM / fused-Winograd launch,
in launch order = layer order): time, executed TFLOP/s, algorithmic GB/s (input + weights + output once).
`python fwdprofile.py [B] [direct]` -- `direct` switches the Winograd path off (the direct 3x3 kernel everywhere)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops, synth
from birdsoundclassif_amd.nets import build_model, functional as Fn
from birdsoundclassif_amd.train import default_args
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
if 'direct' in sys.argv:
    Fn.WINOGRAD = False
model, _ = build_model(default_args(device='cuda'))
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().eval()
x = torch.from_numpy(np.tile(synth.image_batch(0, 8), (B // 8, 1, 1))).cuda()[:, None]
with torch.no_grad():
    model.detect(x); model.detect(x); torch.cuda.synchronize()
    runs = []
    for _ in range(3):
        ops.PROFILE = []
        model.detect(x); torch.cuda.synchronize()
        runs.append(ops.PROFILE)
        ops.PROFILE = None
rows = []
for i, (tag, s, e) in enumerate(runs[0]):
    ms = min(r[i][1].elapsed_time(r[i][2]) for r in runs)                 # same launch sequence in every run
    rows.append((tag, ms))
print(f'detect forward, B = {B}, {"direct 3x3 everywhere" if not Fn.WINOGRAD else "Winograd F(2x2,3x3) for 3x3/s1 with >= 128 channels"}; '
      f'best of 3 runs per launch')
print('   ms    TF/s(exec)  GB/s(alg)  launch')
tot = 0.0
for tag, ms in rows:
    if not isinstance(tag[0], str) and len(tag) != 9:
        continue
    if tag[0] in ('wino23', 'wino23-rois'):
        _, C, N, H, W, b = tag
        print(f'{ms:8.3f}      -         -      = whole Winograd layer above: 3x3 {C}->{N} @{H}x{W} B={b} (row transform + fused kernel)'
              + (' -- tiles under the RoIs only' if tag[0] == 'wino23-rois' else ''))
        continue
    Cin, N, k, H, W, Bb, G, st, label = tag
    Ho, Wo = -(-H // st), -(-W // st)
    fl = 2.0 * G * Bb * Ho * Wo * N * Cin * k * k / 1e9
    by = 4.0 * G * (Bb * H * W * Cin + Bb * Ho * Wo * N + N * Cin * k * k) / 1e9
    what = f'{k}x{k} s{st} {Cin}->{N} @{H}x{W} B={Bb}' + (f' groups={G}' if G > 1 else '')
    if label is not None and label[0] == 'rows-rois':
        print(f'{ms:8.3f}      -         -      {k}x{k} {Cin}->{N} @{label[1]}x{label[2]}, input patches of the tiles under the RoIs (device-side list)')
        tot += ms
        continue
    if label is not None and label[0] == 'rows':
        what = f'{k}x{k} {Cin}->{N} @{label[1]}x{label[2]}, {H} listed pixels of {Bb}x... (pattern input patches)'
        fl = 2.0 * H * N * Cin / 1e9
        by = 4.0 * H * (Cin + N) / 1e9
        tot += ms
        print(f'{ms:8.3f}  {fl / ms:8.1f}  {by / ms * 1e3:9.0f}   {what}')
        continue
    if label is not None and label[0] == 'wino23-rois':
        print(f'{ms:8.3f}      -         -      fused Winograd kernel of 3x3 {Cin}->{N} @{label[1]}x{label[2]}, tiles under the RoIs (device-side list)')
        tot += ms
        continue
    if label is not None and label[0] == 'rpn-composite':
        what = (f'{k} of the 25 taps of the RPN reader composed with the 3x3 in front of it, 5x5 {Cin}->{N} @{label[1]}x{label[2]} over {W} cells '
                '(evaluation mode: no pattern pixels; nbm_cell_patches is a separate HBM-bound kernel, not in this table)')
        fl = 2.0 * W * N * Cin * k / 1e9
        by = 4.0 * (k * W * Cin + 2 * W * N + N * Cin * k) / 1e9
        tot += ms
        print(f'{ms:8.3f}  {fl / ms:8.1f}  {by / ms * 1e3:9.0f}   {what}')
        continue
    if label is not None and label[0] == 'rpn-composite-map':
        what = (f'{Cin} of the channels of the RPN reader composed with the 3x3 in front of it, 5x5 s{st} {Cin}->{N} @{H}x{W} B={Bb}, gathered from '
                'the dense map (evaluation mode: no pattern pixels)')
        tot += ms
        print(f'{ms:8.3f}  {fl / ms:8.1f}  {by / ms * 1e3:9.0f}   {what}')
        continue
    if label is not None and label[0] == 'cell-fwd':
        what = (f'25 plane GEMMs of the cell transforms, 3x3 {Cin}->{N} @{label[1]}x{label[2]} on demand (pattern pixels: {H} cells; '
                'nbm_cell_input / nbm_cell_output are separate HBM-bound kernels, not in this table)')
        by = 4.0 * G * (H * Cin + H * N + N * Cin) / 1e9
    elif label is not None:
        what = f'fused Winograd kernel of 3x3 {Cin}->{N} @{label[1]}x{label[2]} ({G} planes x {H} tiles)'
        by = 4.0 * (H * 4 * Cin * 0.5 + H * 4 * N + G * N * Cin) / 1e9          # R (2x input) + y + U
    tot += ms
    print(f'{ms:8.3f}  {fl / ms:8.1f}  {by / ms * 1e3:9.0f}   {what}')
print(f'total of the GEMM-type launches: {tot:.1f} ms')
