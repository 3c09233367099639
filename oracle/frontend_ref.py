"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy, float64) of the spectrogram front end.

Follows reference nbm_model/nbm_datasets/prepare_dataset.py `File_Processor`
(:92-294): load -> [resample to 44.1 kHz] -> STFT(n_fft=win=1324, hop=132, periodic Hann,
center=True) -> |.| -> 20 log10(max(floor, .)) -> rows [16:391] -> per-file min/max normalise
-> 1024-column windows, hop 819, last one reflect-padded.

PARITY UNPINNED for two third-party steps that are absent from /root/reference and from this
image:
  * `librosa.stft` (prepare_dataset.py:237; version unpinned; pad_mode default 'constant' in
    librosa >= 0.10, 'reflect' before).  Restated from its documented semantics; pinned only by
    analytic known-answer tests (pure tone -> peak row, Parseval, silence -> NaN) in
    tests/test_frontend_oracle.py.  `pad_mode` is a switch; default 'constant'.
  * the `ffmpeg -ar 44100 -acodec pcm_s16le` resample (prepare_dataset.py:175-178).  The build
    owns a documented 2x half-band polyphase up-sampler in exact integer arithmetic
    (`upsample2x_pcm16`), bit-reproducible on any device.
Only tests, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module.
"""
import wave

import numpy as np
import scipy.fft

FREQ = 44100          # prepare_dataset.py:98
H_PIX = 375           # :96
LOW_FREQ = 500        # :97


def constants(freq_accuracy=33.3, dt=0.003, overlap_spectro=0.2, w_pix=1024):
    """process_file constants, prepare_dataset.py:114-138."""
    c = {}
    c['W_PIX'] = w_pix
    c['HOP_SPECTRO'] = int((1 - overlap_spectro) * w_pix)
    c['WIN_LENGTH'] = int(FREQ / freq_accuracy)
    c['HOP_LENGTH'] = int(FREQ * dt)
    overlap_fft = np.round(1 - c['HOP_LENGTH'] / c['WIN_LENGTH'], 3)
    c['FREQ_ACCURACY'] = FREQ / c['WIN_LENGTH']
    c['DT'] = int((1 - overlap_fft) * c['WIN_LENGTH']) / FREQ
    c['LOW_IDX'] = 1 + int(LOW_FREQ / c['FREQ_ACCURACY'])
    c['HIGH_IDX'] = c['LOW_IDX'] + H_PIX
    c['LOW_FREQ'] = (c['LOW_IDX'] - 1) * c['FREQ_ACCURACY']
    c['HIGH_FREQ'] = (c['HIGH_IDX'] - 1) * c['FREQ_ACCURACY']
    return c


# --------------------------------------------------------------------------- F1: load (+ resample)
def read_wav_pcm16(path):
    """mono/stereo 16-bit PCM wav -> (int16 [n] (channel mean, as librosa.load mono=True), sr)."""
    with wave.open(path, 'rb') as f:
        assert f.getsampwidth() == 2, 'only 16-bit PCM is supported'
        sr, nch, n = f.getframerate(), f.getnchannels(), f.getnframes()
        x = np.frombuffer(f.readframes(n), dtype='<i2').reshape(-1, nch)
    return (x[:, 0] if nch == 1 else np.round(x.astype(np.float64).mean(1)).astype(np.int16)), sr


UP_TAPS = 16


def upsample2x_coeffs():
    """Q15 odd-phase coefficients of the half-band windowed-sinc interpolator: h[k] =
    sinc(k+1/2) * kaiser(beta=8), k = 0..15 (symmetric), scaled so that 2*sum(h) = 32768."""
    k = np.arange(UP_TAPS, dtype=np.float64) + 0.5
    h = np.sinc(k) * np.i0(8.0 * np.sqrt(1 - (k / UP_TAPS) ** 2)) / np.i0(8.0)
    h = h / (2 * h.sum())
    q = np.round(h * 32768).astype(np.int64)
    q[0] += 16384 - q.sum()                                   # exact unity DC gain
    return q


def upsample2x_pcm16(x):
    """22.05 kHz -> 44.1 kHz.  Even output samples are the input samples; odd ones are the
    half-band interpolation in exact integer arithmetic: (sum_k hq[k]*(x[n-k]+x[n+1+k]) + 2^14) >> 15,
    saturated to int16.  Samples outside the clip are zero."""
    x = np.asarray(x, dtype=np.int64)
    n = len(x)
    hq = upsample2x_coeffs()
    xp = np.concatenate([np.zeros(UP_TAPS, np.int64), x, np.zeros(UP_TAPS + 1, np.int64)])
    acc = np.zeros(n, dtype=np.int64)
    for k in range(UP_TAPS):
        acc += hq[k] * (xp[UP_TAPS - k:UP_TAPS - k + n] + xp[UP_TAPS + 1 + k:UP_TAPS + 1 + k + n])
    odd = np.clip((acc + 16384) >> 15, -32768, 32767)
    out = np.empty(2 * n, dtype=np.int16)
    out[0::2] = x
    out[1::2] = odd
    return out


def read_wav(path):
    """What `librosa.core.load(path, sr=None)` returns for a wav file (prepare_dataset.py:162): float32 mono in [-1, 1) --
    PCM scaled by 2^-(bits-1) (8-bit is unsigned), IEEE float as stored, channels averaged -- and the sample rate.
    Restated from the documented libsndfile / librosa behaviour (both absent here)."""
    import struct
    raw = open(path, 'rb').read()
    assert raw[:4] == b'RIFF' and raw[8:12] == b'WAVE'
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(raw):
        cid, size = raw[pos:pos + 4], struct.unpack('<I', raw[pos + 4:pos + 8])[0]
        if cid == b'fmt ':
            fmt = raw[pos + 8:pos + 8 + size]
        if cid == b'data':
            data = raw[pos + 8:pos + 8 + size]
        pos += 8 + size + (size & 1)
    tag, nch, sr, _, _, bits = struct.unpack('<HHIIHH', fmt[:16])
    if tag == 0xFFFE:
        tag = struct.unpack('<H', fmt[24:26])[0]
    if tag == 3:
        y = np.frombuffer(data, dtype={32: '<f4', 64: '<f8'}[bits]).astype(np.float32)
    elif bits == 8:
        y = ((np.frombuffer(data, dtype=np.uint8).astype(np.float64) - 128) / 128).astype(np.float32)
    elif bits == 24:
        b = np.frombuffer(data, dtype=np.uint8).reshape(-1, 3).astype(np.int64)
        v = b[:, 0] + (b[:, 1] << 8) + (b[:, 2] << 16)
        y = ((v - ((v >> 23) << 24)) / 2.0 ** 23).astype(np.float32)
    else:
        y = (np.frombuffer(data, dtype={16: '<i2', 32: '<i4'}[bits]).astype(np.float64) / 2.0 ** (bits - 1)).astype(np.float32)
    y = y.reshape(-1, nch)
    return (y[:, 0] if nch == 1 else np.mean(y, axis=1, dtype=np.float32)), sr


def resample_taps(sr, target=FREQ, zeros=16, rolloff=0.95, beta=8.0):
    """The build's rational resampler (the reference shells out to `ffmpeg -ar 44100`, third-party and absent):
    windowed-sinc interpolation y(t) = sum_n x[n] g(t - n), g(tau) = rho s sinc(rho s tau) kaiser_beta(tau s / zeros) for
    |tau| < zeros / s, s = min(1, target / sr); returned as polyphase rows (L, M, taps [L, T]) with unit DC gain."""
    from math import gcd
    g = gcd(sr, target)
    L, M = target // g, sr // g
    s = min(1.0, L / M)
    T = 2 * int(np.ceil(zeros / s))
    taps = np.zeros((L, T))
    for ph in range(L):
        tau = ph / L + (T // 2 - 1) - np.arange(T)
        u = tau * s / zeros
        w = np.where(np.abs(u) < 1, np.i0(beta * np.sqrt(np.maximum(0.0, 1 - u * u))) / np.i0(beta), 0.0)
        h = rolloff * s * np.sinc(rolloff * s * tau) * w
        taps[ph] = h / h.sum()
    return L, M, taps


def resample_to_pcm16(x, sr, target=FREQ):
    """float samples at `sr` -> int16 at 44.1 kHz: ceil(n L / M) outputs, y[m] = sum_k taps[(m M) % L][k] x[m M // L - T/2 +
    1 + k] in float64, rounded to 16 bits (the `-acodec pcm_s16le` of the reference's ffmpeg call)."""
    L, M, taps = resample_taps(sr, target)
    T = taps.shape[1]
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    n_out = -(-n * L // M)
    m = np.arange(n_out, dtype=np.int64)
    n0, ph = (m * M) // L, (m * M) % L
    xp = np.concatenate([np.zeros(T, np.float64), x, np.zeros(T + 1, np.float64)])
    acc = np.zeros(n_out)
    for k in range(T):
        acc += taps[ph, k] * xp[n0 - T // 2 + 1 + k + T]
    return np.clip(np.rint(acc * 32768.0), -32768, 32767).astype(np.int16)


def load(path):
    """File_Processor.load prepare_dataset.py:160-184 -> float32 in [-1,1) at 44.1 kHz."""
    y, sr = read_wav(path)
    if sr == FREQ:
        return y
    if sr * 2 == FREQ and np.array_equal(y * np.float32(32768), np.rint(y * np.float32(32768))):
        # 16-bit material at 22.05 kHz: the exact-integer half-band interpolator
        return upsample2x_pcm16((y * np.float32(32768)).astype(np.int16)).astype(np.float32) / np.float32(32768.0)
    return resample_to_pcm16(y, sr).astype(np.float32) / np.float32(32768.0)


def soundfile_round_trip(y):
    """soundfile.write(..., float data) + librosa.load on a wav file (prepare_dataset.py:197-199, 217): libsndfile stores
    float samples as 16-bit PCM = lrint(x * 32767); they come back as k / 32768."""
    k = np.clip(np.rint(np.asarray(y, dtype=np.float64) * 32767.0), -32768, 32767)
    return (k / 32768.0).astype(np.float32)


def process_long_waveform(y, max_l, labels=None, filename=None, pad_mode='constant', chunk_l=int(5e7), **kw):
    """File_Processor.process_long_file prepare_dataset.py:187-225 on a loaded 44.1 kHz waveform: None when len(y) <= max_l,
    else (list of per-split image lists, list of (split index, label rows shifted into the split)).  `labels`: list of
    (t_start, t_end, ...) tuples of this file."""
    if len(y) <= max_l:
        return None
    inc = max_l / FREQ
    out, kept = [], []
    for k in range(int(len(y) / max_l) + 1):
        piece = soundfile_round_trip(y[k * max_l:(k + 1) * max_l])
        rows = None
        if labels is not None:
            rows = [(r[0] - k * inc, min(r[1] - k * inc, inc)) + tuple(r[2:]) for r in labels if 0 <= r[0] - k * inc <= inc]
            rows = rows or None
        t_end_max = max(r[1] for r in rows) if rows else None
        imgs, _ = process_waveform(piece, pad_mode, chunk_l, t_end_max, **kw)
        out.append(imgs)
        kept.append(rows)
    return out, kept


# --------------------------------------------------------------------------- F3: spectrogram
def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)


def stft(y, n_fft, hop, pad_mode='constant'):
    """librosa.stft(y, n_fft, hop_length=hop) with librosa defaults (win_length=n_fft, periodic Hann,
    center=True): complex128 [1+n_fft//2, 1+len(y)//hop].  Frames are transformed in blocks so that a
    19-minute chunk does not materialise a 4 GB frame matrix."""
    y = np.asarray(y, dtype=np.float64)
    yp = np.pad(y, n_fft // 2, mode=pad_mode)
    n_frames = 1 + (len(yp) - n_fft) // hop
    win = hann_periodic(n_fft)[None, :]
    out = np.empty((1 + n_fft // 2, n_frames), dtype=np.complex128)
    for t0 in range(0, n_frames, 4096):
        t1 = min(n_frames, t0 + 4096)
        idx = np.arange(n_fft)[None, :] + hop * np.arange(t0, t1)[:, None]
        out[:, t0:t1] = scipy.fft.rfft(yp[idx] * win, axis=1).T
    return out


def stft_mag(y, n_fft, hop, pad_mode='constant'):
    """|stft(...)|: [1+n_fft//2, 1+len(y)//hop] float64."""
    return np.abs(stft(y, n_fft, hop, pad_mode))


def amp_to_db(x, min_level_db=-100):
    """prepare_dataset.py:228-230 (floor = exp(-100/20*ln10) = 9.99999999999998e-06, Appendix C-7)."""
    min_level = np.exp(min_level_db / 20 * np.log(10))
    return 20 * np.log10(np.maximum(min_level, x))


def spectrogram(y, c, pad_mode='constant', max_l=int(5e7)):
    """File_Processor.spectrogram prepare_dataset.py:233-252: STFT in chunks of 5e7 samples (every chunk is
    centre-padded on its own by librosa), dB, crop, min/max over the WHOLE file.  Returns list of float64
    [375, L_k] in [0,1].  `max_l` is the reference's hard-coded chunk length (tests scale it down)."""
    parts = []
    for k in range(int(len(y) / max_l) + 1):
        m = stft_mag(y[k * max_l:(k + 1) * max_l], c['WIN_LENGTH'], c['HOP_LENGTH'], pad_mode)
        parts.append(amp_to_db(m)[c['LOW_IDX']:c['HIGH_IDX'], :])
    s_max = max(p.max() for p in parts)
    s_min = min(p.min() for p in parts)
    with np.errstate(invalid='ignore', divide='ignore'):
        return [(p - s_min) / (s_max - s_min) for p in parts]


# --------------------------------------------------------------------------- F4: windows
def window_columns(chunk_lengths, c, label_t_end_max=None):
    """Column bookkeeping of File_Processor.split_power_spec (prepare_dataset.py:255-294) as index vectors:
    for every window the list of source columns of the chunk-concatenated spectrogram.

    * window k covers columns [k*hop, k*hop + W) (:264-266);
    * a window that runs past the end of the file is cut at the end of the CHUNK its first column lies in
      (:270-278: `e_bin_idx` is None and `next_bin` False there, so the following chunks are dropped) -- for a
      single-chunk file that is simply the end of the file;
    * the last window is then grown to W columns by repeated `np.pad(mode='reflect')` in steps of
      `min(empty_width, missing)` columns, `empty_width` starting at W when the file has no label and at
      `L - int(t_end.max() / DT)` when it has (:280-292), and growing by every step."""
    W, hop = c['W_PIX'], c['HOP_SPECTRO']
    cum = np.cumsum([0] + [int(n) for n in chunk_lengths])
    L = int(cum[-1])
    cols = []
    for k in range(max(1, int(1 + np.ceil((L - W) / hop)))):
        start, end = k * hop, k * hop + W
        if end > L:
            end = int(cum[np.searchsorted(cum, start, side='right')])      # end of the chunk holding `start`
        cols.append(np.arange(start, end, dtype=np.int64))
    if len(cols[-1]) < W:
        empty = W if label_t_end_max is None else L - int(label_t_end_max / c['DT'])
        while len(cols[-1]) < W:
            pad = max(1, min(empty, W - len(cols[-1])))
            cols[-1] = np.pad(cols[-1], (0, pad), mode='reflect')
            empty += pad
    return cols


def split_power_spec(parts, c, label_t_end_max=None):
    """File_Processor.split_power_spec prepare_dataset.py:255-294."""
    spec = np.concatenate(parts, axis=1)
    return [spec[:, j] for j in window_columns([p.shape[1] for p in parts], c, label_t_end_max)]


def process_waveform(y, pad_mode='constant', max_l=int(5e7), label_t_end_max=None, **kw):
    """process_file prepare_dataset.py:108-157 from an already loaded 44.1 kHz float waveform.
    Returns (list of float32 [375,1024], meta dict with W_PIX, HOP_SPECTRO, spectrogram_length)."""
    c = constants(**kw)
    parts = spectrogram(y, c, pad_mode, max_l)
    c['spectrogram_length'] = int(sum(p.shape[1] for p in parts))
    return [im.astype(np.float32) for im in split_power_spec(parts, c, label_t_end_max)], c


def process_file(path, pad_mode='constant', **kw):
    return process_waveform(load(path), pad_mode, **kw)
