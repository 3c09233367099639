"""Spectrogram front end (reference nbm_model/nbm_datasets/prepare_dataset.py `File_Processor`), on HIP.

    PCM16 -> [2x half-band up-sampling to 44.1 kHz] -> centre-padded fp32 waveform            (nbm_pcm16_to_wave)
          -> STFT (n_fft = win = 1324, hop 132, periodic Hann) as two real DFT-GEMMs on the fp64 MFMA,
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


def dft_basis_f64(n_fft, low_bin, n_bins):
    """float64 [bin tiles, k steps, 64, 2] DFT basis of `nbm_stft_db` in MFMA fragment order: entry (bt, ks, l) holds
    {w[k] cos(2 pi f k / N), w[k] sin(2 pi f k / N)} for f = low_bin + 16 bt + (l & 15), k = 4 ks + (l >> 4), with the
    periodic Hann window w (librosa's default) folded in, the k = N/2 column halved (the kernel feeds x[k] + x[N-k] =
    2 x[N/2] there) and zeros for k > N/2 and for bins beyond n_bins.  Angles are reduced in integers (f k mod N)."""
    if n_fft % 2:
        raise NotImplementedError('odd n_fft')
    half = n_fft // 2
    n_ks = -(-(half + 1) // 4)
    n_bt = -(-n_bins // 128) * 8
    k = np.arange(4 * n_ks, dtype=np.int64)
    win = np.where(k <= half, 0.5 - 0.5 * np.cos(2 * np.pi * k / n_fft), 0.0)       # w[0] = 0 exactly
    win[half] *= 0.5
    f = low_bin + np.arange(16 * n_bt, dtype=np.int64)
    ang = 2 * np.pi * ((f[:, None] * k[None, :]) % n_fft) / n_fft
    valid = (np.arange(16 * n_bt) < n_bins)[:, None]
    cos = np.where(valid, win * np.cos(ang), 0.0).reshape(n_bt, 16, n_ks, 4)        # [bt, l & 15, ks, l >> 4]
    sin = np.where(valid, win * np.sin(ang), 0.0).reshape(n_bt, 16, n_ks, 4)
    frag = np.stack([cos, sin], axis=-1).transpose(0, 2, 3, 1, 4)                    # [bt, ks, l >> 4, l & 15, 2]
    return torch.from_numpy(np.ascontiguousarray(frag).reshape(n_bt, n_ks, 64, 2))


def window_columns(chunk_lengths, w_pix, hop_img, label_end_col=None):
    """Source columns of every window of `File_Processor.split_power_spec` (reference prepare_dataset.py:255-294) in the
    chunk-concatenated spectrogram: window k = columns [k hop, k hop + W); a window that runs past the end of the file
    stops at the end of the CHUNK its first column is in (:270-278 drop the later chunks there; for the usual one-chunk
    file that is the end of the file); the last window is then grown to W columns by np.pad(mode='reflect') in steps of
    min(empty_width, missing), `empty_width` starting at W (no labels) or at L - label_end_col (:283-287) and growing by
    every step.  Returns (number of windows, int32 column list of the LAST window)."""
    cum = np.cumsum([0] + [int(n) for n in chunk_lengths])
    L = int(cum[-1])
    n_img = max(1, int(1 + np.ceil((L - w_pix) / hop_img)))
    start = (n_img - 1) * hop_img
    end = start + w_pix
    if end > L:
        end = int(cum[np.searchsorted(cum, start, side='right')])
    cols = np.arange(start, end, dtype=np.int32)
    empty = w_pix if label_end_col is None else L - int(label_end_col)
    while len(cols) < w_pix:
        pad = max(1, min(empty, w_pix - len(cols)))
        cols = np.pad(cols, (0, pad), mode='reflect')
        empty += pad
    return n_img, cols


class SpectrogramFrontEnd:
    """Device-resident front end for batches of equal-length PCM16 clips."""

    H_PIX, LOW_FREQ, FREQ = 375, 500, 44100            # prepare_dataset.py:96-98

    def __init__(self, device='cuda', freq_accuracy=33.3, dt=0.003, overlap_spectro=0.2, w_pix=1024,
                 pad_mode='constant'):
        """`pad_mode`: how `librosa.stft(center=True)` pads each chunk -- 'constant' (zeros, librosa >= 0.10, default)
        or 'reflect' (librosa <= 0.9); the reference does not pin its librosa version (prepare_dataset.py:237)."""
        if pad_mode not in ('constant', 'reflect'):
            raise ValueError(f'pad_mode {pad_mode!r}')
        self.device = torch.device(device)
        self.pad_mode = pad_mode
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
        self.basis = dft_basis_f64(self.WIN_LENGTH, self.LOW_IDX, self.H_PIX).to(self.device)
        self.hq = torch.from_numpy(upsample2x_coeffs()).to(self.device)
        self._cols = {}
        self._taps = {}

    def n_frames(self, n_samples_44k):
        return 1 + n_samples_44k // self.HOP_LENGTH                                  # librosa.stft, center=True

    def n_images(self, n_frames):
        return max(1, int(1 + np.ceil((n_frames - self.W_PIX) / self.HOP_SPECTRO)))  # :267

    MAX_CHUNK = int(5e7)           # STFT chunk length in 44.1 kHz samples (reference prepare_dataset.py:234)
    MAX_FILE = int(15e7)           # beyond this the reference goes through process_long_file (:187-225)

    def _source(self, dtype, n, sr):
        """How the 44.1 kHz signal is obtained from the input rows `x` (int16 PCM or float32 samples in [-1, 1)):
        -> (kind, n44, extra).  int16 at 44.1 / 22.05 kHz keeps the exact integer path (`nbm_pcm16_to_wave`); everything
        else goes through `nbm_resample_to_wave`: float32 rows at 44.1 kHz are used as they are (what librosa.load hands
        the reference for 24-bit / 32-bit / float files), any other rate is resampled by the rational polyphase filter of
        `resample_taps` and rounded to the 16-bit grid, like the reference's ffmpeg step (prepare_dataset.py:175-178)."""
        if dtype == torch.int16 and sr == self.FREQ:
            return 'pcm16', n, False
        if dtype == torch.int16 and sr * 2 == self.FREQ:
            return 'pcm16', 2 * n, True
        if dtype not in (torch.int16, torch.float32):
            raise TypeError('input rows must be int16 PCM or float32 samples')
        if sr == self.FREQ:
            return 'f32', n, (1, 1, None)
        if sr not in self._taps:
            L, M, taps = resample_taps(sr, self.FREQ)
            self._taps[sr] = (L, M, torch.from_numpy(taps).to(self.device))
        L, M, taps = self._taps[sr]
        return 'f32', -(-n * L // M), (L, M, taps)

    def _stft_chunk(self, x, src, first, count, out=None, col0=0):
        """STFT of the piece [first, first + count) of the 44.1 kHz signal of every row."""
        kind, _, extra = src
        L = self.n_frames(count)
        lead = self.WIN_LENGTH // 2
        ld = max(lead + count + lead, (L - 1) * self.HOP_LENGTH + self.WIN_LENGTH)
        ld = -(-ld // 4) * 4
        reflect = self.pad_mode == 'reflect'
        if kind == 'pcm16':
            wave_f = ops.pcm16_to_wave(x, ld, lead, extra, self.hq, reflect, first, count)
        else:
            wave_f = ops.resample_to_wave(x, ld, lead, extra[0], extra[1], extra[2], reflect, first, count,
                                          quant16=extra[2] is not None)
        db, mm = ops.stft_db(wave_f, L, self.HOP_LENGTH, self.WIN_LENGTH, self.basis, self.H_PIX, self.floor_amp,
                             out=out, col0=col0)
        return db, mm, L

    def spectrogram_db(self, x, sr):
        """x int16 / float32 [batch, n] on the device -> (db [batch,375,L], minmax, [L_0, L_1, ...] frames per chunk).  Rows
        longer than 5e7 samples (19 min) are transformed chunk by chunk like the reference (the WHOLE row is resampled
        first, every chunk is centre-padded on its own, the min/max runs over the whole row)."""
        if x.dim() != 2:
            raise TypeError('input must be [batch, n]')
        x = x.contiguous()
        src = self._source(x.dtype, x.shape[1], sr)
        if src[0] == 'f32' and x.dtype == torch.int16:
            x = x.to(torch.float32) * (1.0 / 32768.0)             # exact
        n44 = src[1]
        if n44 > self.MAX_FILE - self.MAX_FILE % self.FREQ:
            raise ValueError('rows longer than 1.5e8 samples are split by File_Processor.process_long_file first '
                             '(prepare_dataset.py:187-225)')
        if n44 < self.MAX_CHUNK:
            db, mm, L = self._stft_chunk(x, src, 0, n44)
            return db, mm, [L]
        bounds = [(k * self.MAX_CHUNK, min(n44, (k + 1) * self.MAX_CHUNK)) for k in range(int(n44 / self.MAX_CHUNK) + 1)]
        if bounds[-1][0] == bounds[-1][1]:
            raise ValueError('file length is an exact multiple of the STFT chunk length: the reference calls librosa.stft '
                             'on an empty chunk there and fails (prepare_dataset.py:236-237)')
        Ls = [self.n_frames(b - a) for a, b in bounds]
        db = torch.empty((x.shape[0], self.H_PIX, sum(Ls)), device=x.device, dtype=torch.float32)
        mm = torch.empty((x.shape[0], 2), device=x.device, dtype=torch.int32)
        ops.check(ops.lib().nbm_minmax_init(ops._ptr(mm), x.shape[0], ops._stream()), 'nbm_minmax_init')
        col = 0
        for (a, b), L in zip(bounds, Ls):
            self._stft_chunk(x, src, a, b - a, out=(db, mm), col0=col)
            col += L
        return db, mm, Ls

    def last_window_columns(self, chunk_lengths, label_end_col=None):
        key = (tuple(chunk_lengths), label_end_col)
        if key not in self._cols:
            if len(self._cols) > 64:
                self._cols.clear()
            n_img, cols = window_columns(chunk_lengths, self.W_PIX, self.HOP_SPECTRO, label_end_col)
            self._cols[key] = (n_img, torch.from_numpy(cols).to(self.device))
        return self._cols[key]

    def __call__(self, pcm, sr, label_end_col=None):
        """pcm int16 (or float32 samples) [batch, n] (device) -> images f32 [batch, n_img, 375, w_pix] in [0,1],
        spectrogram length.
        `label_end_col`: int(t_end.max() / DT) of the file's annotations when it has any (it changes the reference's
        padding of the last window, prepare_dataset.py:283-287)."""
        db, mm, Ls = self.spectrogram_db(pcm, sr)
        L = int(sum(Ls))
        n_img, cols = self.last_window_columns(Ls, label_end_col)
        return ops.spec_windows(db, mm, L, n_img, self.W_PIX, self.HOP_SPECTRO, cols), L


RESAMPLE_ZEROS = 16          # zero crossings of the windowed sinc on each side
RESAMPLE_ROLLOFF = 0.95


def resample_taps(sr, target=44100):
    """Polyphase taps of the rational resampler sr -> target (reference: `ffmpeg -ar 44100`, prepare_dataset.py:175-178,
    third-party and absent; this build owns the filter): target / sr = L / M reduced, y[m] = sum_n x[n] g(m M / L - n) with
    g(tau) = rho s sinc(rho s tau) kaiser_8(tau s / 16), s = min(1, L / M), rho = 0.95, |tau| < 16 / s.  Returns
    (L, M, float64 [L, T]): row `ph` holds the taps of output phase ph for the inputs floor(m M / L) - T/2 + 1 + k,
    k = 0 .. T-1, each row normalised to unit DC gain."""
    from math import gcd
    g = gcd(int(sr), int(target))
    L, M = int(target) // g, int(sr) // g
    sc = min(1.0, L / M)
    half = int(np.ceil(RESAMPLE_ZEROS / sc))
    T = 2 * half
    k = np.arange(T, dtype=np.float64)
    ph = np.arange(L, dtype=np.float64)[:, None] / L
    tau = ph + (T // 2 - 1) - k[None, :]
    u = tau * sc / RESAMPLE_ZEROS
    win = np.where(np.abs(u) < 1.0, np.i0(8.0 * np.sqrt(np.clip(1.0 - u * u, 0.0, None))) / np.i0(8.0), 0.0)
    h = RESAMPLE_ROLLOFF * sc * np.sinc(RESAMPLE_ROLLOFF * sc * tau) * win
    return L, M, np.ascontiguousarray(h / h.sum(1, keepdims=True))


def read_wav(path):
    """RIFF/WAVE file -> (samples, sample rate): mono 16-bit PCM as int16 [n] (the exact-integer fast path), everything
    else as float32 [n] in [-1, 1) the way librosa.load(sr=None) returns it (prepare_dataset.py:162): 8-bit (unsigned),
    16-, 24-, 32-bit PCM scaled by 2^-(bits-1), IEEE float 32 / 64, WAVE_FORMAT_EXTENSIBLE of those; several channels are
    averaged in float32 (librosa.to_mono)."""
    import struct
    with open(path, 'rb') as f:
        raw = f.read()
    if raw[:4] != b'RIFF' or raw[8:12] != b'WAVE':
        raise ValueError('not a RIFF/WAVE file')
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(raw):
        cid, size = raw[pos:pos + 4], struct.unpack('<I', raw[pos + 4:pos + 8])[0]
        body = raw[pos + 8:pos + 8 + size]
        if cid == b'fmt ':
            fmt = body
        elif cid == b'data':
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError('wav file without fmt / data chunk')
    tag, nch, sr, _, _, bits = struct.unpack('<HHIIHH', fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:                                  # extensible: the sub-format GUID starts with the tag
        tag = struct.unpack('<H', fmt[24:26])[0]
    n = len(data) // (nch * (bits // 8))
    data = data[:n * nch * (bits // 8)]
    if tag == 1 and bits == 16:
        x = np.frombuffer(data, dtype='<i2').reshape(n, nch)
        if nch == 1:
            return np.array(x[:, 0], dtype=np.int16), sr                  # own, writable copy
        y = x.astype(np.float32) / np.float32(32768.0)
    elif tag == 1 and bits == 8:
        y = (np.frombuffer(data, dtype=np.uint8).reshape(n, nch).astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    elif tag == 1 and bits == 24:
        b = np.frombuffer(data, dtype=np.uint8).reshape(n, nch, 3).astype(np.int32)
        v = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        y = v.astype(np.float32) / np.float32(8388608.0)
    elif tag == 1 and bits == 32:
        y = (np.frombuffer(data, dtype='<i4').reshape(n, nch).astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == 3 and bits in (32, 64):
        y = np.frombuffer(data, dtype='<f4' if bits == 32 else '<f8').reshape(n, nch).astype(np.float32)
    else:
        raise NotImplementedError(f'wav format tag {tag} with {bits} bits per sample')
    return (y[:, 0].copy() if nch == 1 else y.mean(1, dtype=np.float32)), sr


def read_wav_pcm16(path):
    """Mono 16-bit PCM only (the bulk-inference shard format)."""
    x, sr = read_wav(path)
    if x.dtype != np.int16:
        raise NotImplementedError('bulk inference expects mono 16-bit PCM wav files')
    return x, sr


def soundfile_pcm16_round_trip(x):
    """What `soundfile.write(path, data, sr)` + `librosa.load` do to a float signal on a wav file (reference
    process_long_file, prepare_dataset.py:197-199, 217): libsndfile stores float data as 16-bit PCM with
    lrint(x * 32767) (its default normalisation) and the samples come back as k / 32768.  int16 input k means x = k / 32768."""
    if x.dtype == np.int16:
        k = x.astype(np.int32)
        return np.rint(k.astype(np.float64) * (32767.0 / 32768.0)).astype(np.int16)
    return np.clip(np.rint(x.astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16)


_FE = {}


class File_Processor:
    """Per-file interface of the reference (prepare_dataset.py:92-157)."""

    H_PIX, LOW_FREQ, FREQ = 375, 500, 44100

    def __init__(self, filepath, extra_str_label='', labels=None):
        self.labels = labels
        self.ext = os.path.basename(filepath).split('.')[-1]
        self.filename = os.path.basename(filepath).replace('.' + self.ext, '').replace(extra_str_label, '')
        self.filepath = filepath

    def load(self):
        try:
            return read_wav(self.filepath)
        except Exception:
            print('File loading failed')
            return None

    def _front_end(self, freq_accuracy, dt, overlap_spectro, w_pix, device, pad_mode):
        key = (freq_accuracy, dt, overlap_spectro, w_pix, str(device), pad_mode)
        if key not in _FE:
            _FE[key] = SpectrogramFrontEnd(device, freq_accuracy, dt, overlap_spectro, w_pix, pad_mode)
        return _FE[key]

    def process_file(self, freq_accuracy=33.3, dt=0.003, overlap_spectro=0.2, w_pix=1024, device='cuda',
                     pad_mode='constant', data=None):
        """-> (list of np.float32 [375, w_pix], annotations or None); (None, None) when the file cannot be read.  Files
        longer than 1.5e8 samples at 44.1 kHz (56 min) return what the reference's `process_long_file` returns: a list of
        such lists, one per split, and the list of the splits' annotation frames."""
        fe = self._front_end(freq_accuracy, dt, overlap_spectro, w_pix, device, pad_mode)
        self.W_PIX, self.HOP_SPECTRO = fe.W_PIX, fe.HOP_SPECTRO                 # prepare_dataset.py:114-115
        if data is None:
            data = self.load()
        if data is None:
            return None, None
        samples, sr = data
        long_out = self.process_long_file(samples, sr, fe, (freq_accuracy, dt, overlap_spectro, w_pix, device, pad_mode))
        if long_out is not None:
            return long_out
        for k in ('WIN_LENGTH', 'HOP_LENGTH', 'FREQ_ACCURACY', 'DT', 'LOW_IDX', 'HIGH_IDX'):
            setattr(self, k, getattr(fe, k))
        label_end_col = None
        if self.labels is not None:
            own = self.labels.loc[self.labels['filename'] == self.filename]
            if len(own) > 0:
                label_end_col = int(own['t_end'].max() / fe.DT)                   # prepare_dataset.py:283-284
        imgs, L = fe(torch.from_numpy(samples)[None].to(fe.device), sr, label_end_col)
        self.spectrogram_length = L
        self.images_device = imgs[0]
        img_db = [im for im in imgs[0].cpu().numpy()]
        if self.labels is None:
            return img_db, None
        self.LOW_FREQ = (self.LOW_IDX - 1) * self.FREQ_ACCURACY                  # prepare_dataset.py:137-138
        self.HIGH_FREQ = (self.HIGH_IDX - 1) * self.FREQ_ACCURACY
        labels_ = self.merge_and_filter_labels(img_db)
        if labels_ is None:
            print('Something went wrong with the annotation file, skipping~~')
            return None, None
        return img_db, labels_

    def process_long_file(self, samples, sr, fe, settings):
        """reference prepare_dataset.py:187-225: a file longer than max_l = 15e7 - 15e7 % 44100 samples (at 44.1 kHz) is cut
        into pieces of max_l samples; every piece goes through soundfile.write -> a fresh File_Processor.process_file
        (its own min / max normalisation, its own windows), with the annotations shifted into the piece's time frame
        (kept when they START inside it, their end clipped to it).  Returns (list of image lists, list of annotation
        frames) like the reference, or None for ordinary files.  The temporary wav round trip of the reference quantises
        the float samples to 16 bits with libsndfile's x * 32767 rule: reproduced by `soundfile_pcm16_round_trip`."""
        max_l = fe.MAX_FILE - fe.MAX_FILE % self.FREQ
        src = fe._source(torch.from_numpy(samples[:1]).dtype, len(samples), sr)
        if src[1] <= max_l:
            return None
        if not (sr == self.FREQ):
            # the reference resamples the whole file first (load), then cuts; do the same on the host grid: the resampled
            # signal is int16-valued, so fetch it from the device path once
            x = torch.from_numpy(samples)[None].to(fe.device)
            if src[0] == 'f32' and x.dtype == torch.int16:
                x = x.to(torch.float32) * (1.0 / 32768.0)
            lead = 0
            if src[0] == 'pcm16':
                w = ops.pcm16_to_wave(x.contiguous(), -(-src[1] // 4) * 4, lead, src[2], fe.hq)
            else:
                w = ops.resample_to_wave(x.contiguous(), -(-src[1] // 4) * 4, lead, src[2][0], src[2][1], src[2][2])
            samples = torch.round(w[0, :src[1]] * 32768.0).to(torch.int16).cpu().numpy()
            sr = self.FREQ
        print('Long file, processing in several steps...')
        time_increment = max_l / self.FREQ
        img_db, annotations = [], []
        for k in range(int(len(samples) / max_l) + 1):
            piece = soundfile_pcm16_round_trip(samples[k * max_l:(k + 1) * max_l])
            print(f'~~ Processing split # {k} ~~')
            labels = None
            if self.labels is not None:
                labels = self.labels.loc[self.labels['filename'] == self.filename].copy()
                for col in ('t_start', 't_end'):
                    labels[col] = labels[col] - k * time_increment
                labels = labels.loc[labels['t_start'].between(0, time_increment)].copy()
                labels['t_end'] = labels['t_end'].clip(upper=time_increment)
                labels['filename'] = f'temp{k}'
                if len(labels) == 0:
                    labels = None
            fp = File_Processor(f'temp{k}.{self.ext}', '', labels)
            img_inc, annot_inc = fp.process_file(*settings[:4], device=settings[4], pad_mode=settings[5], data=(piece, sr))
            img_db.append(img_inc)
            if annot_inc is not None:
                annotations.append(annot_inc)
        return img_db, annotations

    def merge_and_filter_labels(self, img_db):
        """Annotations of this file (seconds / Hz) -> one row per window that holds at least one box:
        DataFrame {index, coord: [(x1,y1,x2,y2), ...] in window pixels, bird_id: [...]} (reference
        prepare_dataset.py:297-375).  Rules kept: time -> column by truncation of t / DT, frequency clipped to the image
        band and truncated to rows, degenerate boxes dropped, a box belongs to every window it intersects unless the
        visible part is < 50 % of its width and < 20 px, or < 10 % and < 45 px; coordinates clipped to the window;
        'noise' labels (bird_id -1) only survive in windows without a real label.  None when the file has no label
        (the reference raises and skips the file)."""
        import pandas as pd
        lab = self.labels.loc[self.labels['filename'] == self.filename].copy()
        if self.ext == 'mp3':                                            # Audacity offset of mp3 decoding, :309-311
            for c in ('t_start', 't_end'):
                lab[c] = lab[c] - 0.03
        if len(lab) == 0:
            return None
        x1 = (lab['t_start'].astype(float) / self.DT).astype(int).to_numpy()
        x2 = (lab['t_end'].astype(float) / self.DT).astype(int).to_numpy()
        band = lambda c: ((lab[c].clip(lower=self.LOW_FREQ, upper=self.HIGH_FREQ) - self.LOW_FREQ)
                          / self.FREQ_ACCURACY).astype(int).to_numpy()
        y1, y2 = band('f_start'), band('f_end')
        bird = lab['bird_id'].to_numpy()
        keep = (y1 != y2) & (x2 - x1 + 1 > 0) & (y2 - y1 + 1 > 0)
        x1, x2, y1, y2, bird = x1[keep], x2[keep], y1[keep], y2[keep], bird[keep]
        w = x2 - x1 + 1
        rows = []                                                        # (window, x1, y1, x2, y2, bird), label-major order
        n_win = len(img_db)
        starts = np.arange(n_win) * self.HOP_SPECTRO
        ends = starts + self.W_PIX - 1
        for j in range(len(x1)):
            hit = ((x1[j] >= starts) & (x1[j] <= ends)) | ((x2[j] >= starts) & (x2[j] <= ends)) | \
                  ((x1[j] < starts) & (x2[j] > ends))
            for k in np.flatnonzero(hit):
                inside = min(x2[j], ends[k]) - max(x1[j], starts[k]) + 1
                if (inside < 0.5 * w[j] and inside < 20) or (inside < 0.1 * w[j] and inside < 45):
                    continue
                rows.append((int(k), max(int(x1[j] - starts[k]), 0), max(int(y1[j]), 0),
                             min(int(x2[j] - starts[k]), self.W_PIX - 1), min(int(y2[j]), self.H_PIX - 1), int(bird[j])))
        real = {}
        for r in rows:
            real[r[0]] = real.get(r[0], 0) + (r[5] != -1)
        # the reference's inner merge drops windows that hold no real label at all, and then noise rows never survive
        rows = [r for r in rows if real.get(r[0], 0) > 0 and r[5] != -1]
        out = {}
        for r in rows:
            out.setdefault(r[0], ([], []))
            out[r[0]][0].append(r[1:5])
            out[r[0]][1].append(r[5])
        idx = sorted(out)
        return pd.DataFrame({'index': idx, 'coord': [out[k][0] for k in idx], 'bird_id': [out[k][1] for k in idx]})


# =========================================================================== dataset preparation
def read_txt_file(file, extra_str_label=''):
    """Audacity spectral-label file -> DataFrame [t_start, t_end, f_start, f_end, species, filename], one row per
    annotation (reference utils.py:59-92): a time line `t0<TAB>t1<TAB>species` followed by its frequency line
    `\\<TAB>f0<TAB>f1`; a repeated time or frequency line inside one record is ignored, records missing either line are
    dropped."""
    import pandas as pd
    rows, cur = [], None
    with open(file, 'r') as f:
        for line in f:
            parts = line.rstrip('\n').split('\t')
            if parts[0] == '\\':
                if cur is not None and 'f' not in cur and len(parts) >= 3:
                    cur['f'] = (parts[1], parts[2])
            elif len(parts) >= 3:
                cur = {'t': (parts[0], parts[1]), 'species': parts[2]}
                rows.append(cur)
    name = os.path.basename(file).split('.')[0].replace(extra_str_label, '')
    recs = [(float(r['t'][0]), float(r['t'][1]), float(r['f'][0]), float(r['f'][1]), r['species'], name)
            for r in rows if 'f' in r and r['species'] != '']
    return pd.DataFrame(recs, columns=['t_start', 't_end', 'f_start', 'f_end', 'species', 'filename'])


def create_label_dataset(directory, birds_dict, noise_labels=(), not_bird_labels=(), suppress_others=True,
                         suppress_noise=True, other_id=None):
    """All `*.txt` annotation files of a directory -> one label DataFrame with `bird_id` (reference utils.py:95-173).
    Same steps: negative f_start clipped to 0, negative f_end -> 20 kHz, duplicates of (file, t_start, species) resolved in
    favour of the widest frequency range, species -> id through `birds_dict`; `noise_labels` -> -1, `not_bird_labels` (and
    any species containing 'autre') -> 0, unknown species -> `other_id` (default `birds_dict['Other']`).  The reference's
    hand-curated spelling-correction table and label lists are data of its authors' corpus: pass your own."""
    import pandas as pd
    files = sorted(f for f in os.listdir(directory) if os.path.splitext(f)[-1] == '.txt')
    labels = pd.concat([read_txt_file(os.path.join(directory, f)) for f in files])
    labels['f_start'] = labels['f_start'].clip(lower=0)
    labels.loc[labels['f_end'] < 0, 'f_end'] = 20000
    labels['f_delta'] = labels['f_end'] - labels['f_start']
    labels = labels.sort_values('f_delta', ascending=False, kind='stable').drop_duplicates(['filename', 't_start', 'species'])
    labels = labels.sort_values(['filename', 't_start'], kind='stable').drop(columns='f_delta')
    labels['bird_id'] = labels['species'].map(lambda x: birds_dict.get(x, np.nan))
    labels.loc[labels['species'].isin(list(noise_labels)), 'bird_id'] = -1
    others = labels['species'].map(lambda x: 'autre' in x.lower()) | labels['species'].isin(list(not_bird_labels))
    labels.loc[others, 'bird_id'] = 0
    labels['bird_id'] = labels['bird_id'].fillna(birds_dict['Other'] if other_id is None else other_id).astype(int)
    if suppress_noise:
        labels = labels.loc[labels['bird_id'] != -1]
    if suppress_others:
        labels = labels.loc[labels['bird_id'] != 0]
    labels.index = range(len(labels))
    return labels


def encode_png_gray8(img_u8):
    """uint8 [H,W] -> bytes of an 8-bit greyscale PNG (filter type 0 on every scanline)."""
    import struct
    import zlib
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    H, W = img.shape
    raw = np.zeros((H, W + 1), dtype=np.uint8)
    raw[:, 1:] = img
    def chunk(kind, data):
        return struct.pack('>I', len(data)) + kind + data + struct.pack('>I', zlib.crc32(kind + data) & 0xFFFFFFFF)
    return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', W, H, 8, 0, 0, 0, 0)) +
            chunk(b'IDAT', zlib.compress(raw.tobytes(), 6)) + chunk(b'IEND', b''))


def write_png_gray8(path, img_u8):
    """The counterpart of `imageio.imwrite` (reference prepare_dataset.py:85-87)."""
    with open(path, 'wb') as f:
        f.write(encode_png_gray8(img_u8))


def prepare_dataset(directory, out_directory, freq_accuracy=33.3, dt=0.003, overlap_spectro=0.2, w_pix=1024,
                    annotations=True, labels=None, audio_format='wav', keep_files=None, device='cuda'):
    """wav recordings (+ label DataFrame) -> the `Img_dataset` directory layout (reference prepare_dataset.py:12-89):
    `positive_files/<dir>__<rec>/<dir>__<rec>__<window:05d>.png` + `annotations.csv` (sep ';', columns index / coord /
    bird_id) for windows that hold a label, `negative_files/...` for the others (the first 1000 windows of a file only).
    Spectrogram windows come from the device front end; `labels` is what `create_label_dataset` returns (required when
    `annotations`).  Returns the number of (positive, negative) images written."""
    import glob
    top_dir = os.path.basename(os.path.normpath(directory))
    if annotations and labels is None:
        raise ValueError('annotations=True needs the label DataFrame (create_label_dataset)')
    n_pos = n_neg = 0
    for file in sorted(glob.glob(os.path.join(directory, f'*.{audio_format}'))):
        fp = File_Processor(file, '', labels if annotations else None)
        if keep_files is not None and fp.filename not in keep_files:
            print(f'** File {fp.filename} not included, going to next file **')
            continue
        rec = top_dir + '__' + fp.filename.replace('#', '__')
        out_pos_dir = os.path.join(out_directory, 'positive_files', rec)
        out_neg_dir = os.path.join(out_directory, 'negative_files', rec)
        if os.path.exists(out_pos_dir) or os.path.exists(out_neg_dir):
            continue
        print(f'~~~ Processing file {fp.filename} ~~~')
        img_db, annots = fp.process_file(freq_accuracy=freq_accuracy, dt=dt, overlap_spectro=overlap_spectro, w_pix=w_pix,
                                         device=device)
        if img_db is None:
            continue
        pos_idx = set() if annots is None else set(int(i) for i in annots['index'].values)
        if pos_idx:
            os.makedirs(out_pos_dir, exist_ok=True)
            annots.to_csv(os.path.join(out_pos_dir, 'annotations.csv'), sep=';', index=False)
        if len(pos_idx) < len(img_db):
            os.makedirs(out_neg_dir, exist_ok=True)
        for i, img in enumerate(img_db):
            name = '__'.join([top_dir, fp.filename.replace('#', '__'), format(i, '05d')]) + '.png'
            u8 = np.round(img * 255).astype(np.uint8)
            if i in pos_idx:
                write_png_gray8(os.path.join(out_pos_dir, name), u8)
                n_pos += 1
            elif i <= 999:
                write_png_gray8(os.path.join(out_neg_dir, name), u8)
                n_neg += 1
    return n_pos, n_neg
