"""Front-end timing on the GPU: 64 clips, per kernel (HIP events), image error vs nothing (timing only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from birdsoundclassif_amd import synth, ops
from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd

fe = SpectrogramFrontEnd('cuda')
pcm = torch.from_numpy(synth.clip_batch_pcm16(0, 8)).cuda().repeat(8, 1)
for _ in range(3):
    imgs, L = fe(pcm, 22050)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
N = 20
tt = np.zeros(4)
for _ in range(N):
    ev[0].record()
    wave_f = ops.pcm16_to_wave(pcm, 133632, 662, True, fe.hq)
    ev[1].record()
    db, mm = ops.stft_db(wave_f, 1003, 132, 1324, fe.basis, 375, fe.floor_amp)
    ev[2].record()
    n_img, cols = fe.last_window_columns([1003])
    img = ops.spec_windows(db, mm, 1003, n_img, 1024, 819, cols)
    ev[3].record()
    torch.cuda.synchronize()
    tt += [ev[i].elapsed_time(ev[i + 1]) for i in range(3)] + [ev[0].elapsed_time(ev[3])]
tt /= N
flop = 2 * 2 * 384 * 664 * 1024 * 64
print(f'pcm16_to_wave {tt[0]:.3f} ms  stft_db {tt[1]:.3f} ms ({flop / tt[1] / 1e9:.1f} TFLOP/s fp64 on executed flops)  '
      f'spec_windows {tt[2]:.3f} ms  total {tt[3]:.3f} ms / 64 clips')
