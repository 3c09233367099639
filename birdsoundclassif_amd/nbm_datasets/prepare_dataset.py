"""Spectrogram front end (reference nbm_model/nbm_datasets/prepare_dataset.py `File_Processor`), on HIP.

    PCM16 -> [2x half-band up-sampling to 44.1 kHz] -> centre-padded fp32 waveform            (nbm_pcm16_to_wave)
          -> STFT (n_fft = win = 1324, hop 132, periodic Hann) as a DFT-GEMM on the fp32 MFMA,
             |.| -> 20 log10(max(floor, .)) for rows [16:391], running min/max per file            (nbm_stft_db)
          -> (x - min)/(max - min), 1024-column windows with hop 819, reflect-padded last window   (nbm_spec_windows)

`SpectrogramFrontEnd` is the device-resident batch API (used by bulk inference and the benchmark);
`File_Processor` keeps the reference's per-file interface and attributes (`W_PIX`, `HOP_SPECTRO`,
`spectrogram_length`, ...) that `merge_images` reads later.
"""
import os
import wave

import numpy as np
import torch

from .. import ops

UP_TAPS = 16


def upsample2x_coeffs():
    """Q15 odd-phase taps of the 2x half-band interpolator (documented in DESIGN.md; the reference shells out
    to ffmpeg for this step, prepare_dataset.py:175-178, which is third-party and absent)."""
    k = np.arange(UP_TAPS, dtype=np.float64) + 0.5
    h = np.sinc(k) * np.i0(8.0 * np.sqrt(1 - (k / UP_TAPS) ** 2)) / np.i0(8.0)
    h = h / (2 * h.sum())
    q = np.round(h * 32768).astype(np.int64)
    q[0] += 16384 - q.sum()
    return q.astype(np.int32)


def dft_basis(n_fft, low_bin, n_bins):
    """fp32 [rows, ld] DFT basis with the periodic Hann window folded in; rows come in blocks of 64 =
    32 cosine rows + 32 sine rows of the same 32 bins (the kernel pairs them in registers)."""
    n_blk = -(-n_bins // 32)
    rows = -(-(n_blk * 64) // 128) * 128
    ld = -(-n_fft // 32) * 32
    n = np.arange(n_fft, dtype=np.float64)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft)
    basis = np.zeros((rows, ld), dtype=np.float64)
    for j in range(n_blk):
        nb = min(32, n_bins - 32 * j)
        f = (low_bin + 32 * j + np.arange(nb))[:, None]
        ang = 2 * np.pi * ((f * n[None, :]) % n_fft) / n_fft
        basis[64 * j:64 * j + nb, :n_fft] = win * np.cos(ang)
        basis[64 * j + 32:64 * j + 32 + nb, :n_fft] = -win * np.sin(ang)
    return torch.from_numpy(basis.astype(np.float32))


class SpectrogramFrontEnd:
    """Device-resident front end for batches of equal-length PCM16 clips."""

    H_PIX, LOW_FREQ, FREQ = 375, 500, 44100            # prepare_dataset.py:96-98

    def __init__(self, device='cuda', freq_accuracy=33.3, dt=0.003, overlap_spectro=0.2, w_pix=1024):
        self.device = torch.device(device)
        self.W_PIX = w_pix
        self.HOP_SPECTRO = int((1 - overlap_spectro) * w_pix)                       # :115
        self.WIN_LENGTH = int(self.FREQ / freq_accuracy)                             # :125
        self.HOP_LENGTH = int(self.FREQ * dt)                                        # :126
        overlap_fft = np.round(1 - self.HOP_LENGTH / self.WIN_LENGTH, 3)
        self.FREQ_ACCURACY = self.FREQ / self.WIN_LENGTH
        self.DT = int((1 - overlap_fft) * self.WIN_LENGTH) / self.FREQ
        self.LOW_IDX = 1 + int(self.LOW_FREQ / self.FREQ_ACCURACY)                   # :134
        self.HIGH_IDX = self.LOW_IDX + self.H_PIX
        self.floor_amp = float(np.exp(-100 / 20 * np.log(10)))                       # amp_to_db :228-230
        if self.HOP_LENGTH % 4:
            raise NotImplementedError('hop length must be a multiple of 4 samples (16-byte aligned frame rows)')
        self.basis = dft_basis(self.WIN_LENGTH, self.LOW_IDX, self.H_PIX).to(self.device)
        self.hq = torch.from_numpy(upsample2x_coeffs()).to(self.device)

    def n_frames(self, n_samples_44k):
        return 1 + n_samples_44k // self.HOP_LENGTH                                  # librosa.stft, center=True

    def n_images(self, n_frames):
        return max(1, int(1 + np.ceil((n_frames - self.W_PIX) / self.HOP_SPECTRO)))  # :267

    MAX_CHUNK = int(5e7)           # STFT chunk length in 44.1 kHz samples (reference prepare_dataset.py:234)
    MAX_FILE = int(15e7)           # beyond this the reference goes through process_long_file (:187-225)

    def _rate(self, sr):
        if sr == self.FREQ:
            return False
        if sr * 2 == self.FREQ:
            return True
        raise NotImplementedError(f'sample rate {sr}: only 44100 and 22050 Hz inputs are supported')

    def _stft_chunk(self, pcm, up, out=None, col0=0):
        n44 = pcm.shape[1] * (2 if up else 1)
        L = self.n_frames(n44)
        lead = self.WIN_LENGTH // 2
        ld = max(lead + n44 + lead, (L - 1) * self.HOP_LENGTH + self.basis.shape[1])
        ld = -(-ld // 4) * 4
        wave_f = ops.pcm16_to_wave(pcm.contiguous(), ld, lead, up, self.hq)
        db, mm = ops.stft_db(wave_f, L, self.HOP_LENGTH, self.basis, self.H_PIX, self.floor_amp, out=out, col0=col0)
        return db, mm, L

    def spectrogram_db(self, pcm, sr):
        """pcm int16 [batch, n] on the device -> (db [batch,375,L], minmax, L).  Rows longer than 5e7 samples (19 min)
        are transformed chunk by chunk like the reference (every chunk is centre-padded on its own, the min/max runs
        over the whole row)."""
        if pcm.dtype != torch.int16 or pcm.dim() != 2:
            raise TypeError('pcm must be int16 [batch, n]')
        up = self._rate(sr)
        n44 = pcm.shape[1] * (2 if up else 1)
        if n44 > self.MAX_FILE - self.MAX_FILE % self.FREQ:
            raise NotImplementedError('files longer than 1.5e8 samples: the reference re-enters process_file per 56-minute '
                                      'split and returns nested lists that its own run_detection cannot batch '
                                      '(prepare_dataset.py:187-225); split such recordings before detection')
        if n44 <= self.MAX_CHUNK:
            return self._stft_chunk(pcm, up)
        step_in = self.MAX_CHUNK // (2 if up else 1)
        pieces = [pcm[:, k * step_in:(k + 1) * step_in] for k in range(int(n44 / self.MAX_CHUNK) + 1)]
        pieces = [p for p in pieces if p.shape[1] > 0]
        Ls = [self.n_frames(p.shape[1] * (2 if up else 1)) for p in pieces]
        db = torch.empty((pcm.shape[0], self.H_PIX, sum(Ls)), device=pcm.device, dtype=torch.float32)
        mm = torch.empty((pcm.shape[0], 2), device=pcm.device, dtype=torch.int32)
        ops.check(ops.lib().nbm_minmax_init(ops._ptr(mm), pcm.shape[0], ops._stream()), 'nbm_minmax_init')
        col = 0
        for p, L in zip(pieces, Ls):
            self._stft_chunk(p, up, out=(db, mm), col0=col)
            col += L
        return db, mm, sum(Ls)

    def __call__(self, pcm, sr):
        """pcm int16 [batch, n] (device) -> images f32 [batch, n_img, 375, w_pix] in [0,1], spectrogram length."""
        db, mm, L = self.spectrogram_db(pcm, sr)
        return ops.spec_windows(db, mm, L, self.n_images(L), self.W_PIX, self.HOP_SPECTRO), L


def read_wav_pcm16(path):
    with wave.open(path, 'rb') as f:
        if f.getsampwidth() != 2:
            raise NotImplementedError('only 16-bit PCM wav files are supported')
        sr, nch, n = f.getframerate(), f.getnchannels(), f.getnframes()
        x = np.frombuffer(f.readframes(n), dtype='<i2').reshape(-1, nch)
    if nch > 1:
        x = np.round(x.astype(np.float64).mean(1)).astype(np.int16)[:, None]
    return np.array(x[:, 0], dtype=np.int16), sr          # own, writable copy


_FE = {}


class File_Processor:
    """Per-file interface of the reference (prepare_dataset.py:92-157)."""

    H_PIX, LOW_FREQ, FREQ = 375, 500, 44100

    def __init__(self, filepath, extra_str_label='', labels=None):
        if labels is not None:
            raise NotImplementedError('label merging (dataset preparation) is outside the hot-path scope')
        self.labels = labels
        self.ext = os.path.basename(filepath).split('.')[-1]
        self.filename = os.path.basename(filepath).replace('.' + self.ext, '').replace(extra_str_label, '')
        self.filepath = filepath

    def load(self):
        try:
            return read_wav_pcm16(self.filepath)
        except Exception:
            print('File loading failed')
            return None

    def process_file(self, freq_accuracy=33.3, dt=0.003, overlap_spectro=0.2, w_pix=1024, device='cuda'):
        """-> (list of np.float32 [375, w_pix], None); (None, None) when the file cannot be read."""
        key = (freq_accuracy, dt, overlap_spectro, w_pix, str(device))
        if key not in _FE:
            _FE[key] = SpectrogramFrontEnd(device, freq_accuracy, dt, overlap_spectro, w_pix)
        fe = _FE[key]
        for k in ('W_PIX', 'HOP_SPECTRO', 'WIN_LENGTH', 'HOP_LENGTH', 'FREQ_ACCURACY', 'DT', 'LOW_IDX', 'HIGH_IDX'):
            setattr(self, k, getattr(fe, k))
        data = self.load()
        if data is None:
            return None, None
        pcm, sr = data
        imgs, L = fe(torch.from_numpy(pcm)[None].to(fe.device), sr)
        self.spectrogram_length = L
        self.images_device = imgs[0]
        return [im for im in imgs[0].cpu().numpy()], None
