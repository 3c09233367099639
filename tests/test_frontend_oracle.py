"""CPU: known-answer tests pinning the front-end oracle (oracle/frontend_ref.py).  librosa / ffmpeg are
absent (parity unpinned for those third-party steps), so the restatement is pinned by analytic
properties of the STFT and by the reference's own constants (prepare_dataset.py:114-138)."""
import numpy as np

from birdsoundclassif_amd import synth
from oracle import frontend_ref as FR


def test_constants_match_reference_probe():
    c = FR.constants()
    assert (c['W_PIX'], c['HOP_SPECTRO'], c['WIN_LENGTH'], c['HOP_LENGTH']) == (1024, 819, 1324, 132)
    assert (c['LOW_IDX'], c['HIGH_IDX']) == (16, 391)
    assert abs(c['FREQ_ACCURACY'] - 33.3082) < 1e-3 and abs(c['DT'] - 132 / 44100) < 1e-12
    assert abs(c['LOW_FREQ'] - 499.62) < 0.01 and abs(c['HIGH_FREQ'] - 12990.18) < 0.01
    assert FR.amp_to_db(np.array([0.0]))[0] == 20 * np.log10(np.exp(-5 * np.log(10)))


def test_pure_tone_peaks_at_its_bin_and_parseval():
    c = FR.constants()
    k = 100                                                        # bin 100 -> image row 84
    t = np.arange(132300) / 44100
    y = 0.5 * np.sin(2 * np.pi * k * c['FREQ_ACCURACY'] * t)
    m = FR.stft_mag(y, 1324, 132)
    assert m.shape == (663, 1003)
    mid = m[:, 100:900]
    assert (mid.argmax(0) == k).all()
    assert np.allclose(mid[k], 0.5 * 1324 / 4, rtol=1e-3)          # A * sum(hann)/2 = A*N/4
    imgs, meta = FR.process_waveform(y.astype(np.float32))
    assert len(imgs) == 1 and imgs[0].shape == (375, 1024) and meta['spectrogram_length'] == 1003
    assert (imgs[0][:, 100:900].argmax(0) == k - 16).all()
    assert imgs[0].min() == 0.0 and imgs[0].max() == 1.0
    # reflect padding of the last (only) window: columns 1003.. mirror 1001, 1000, ...
    assert np.array_equal(imgs[0][:, 1003:1024], imgs[0][:, 1001:980:-1])


def test_window_split_counts():
    c = FR.constants()
    for L, n in ((1003, 1), (1024, 1), (1025, 2), (1843, 2), (1844, 3), (3157, 4)):
        spec = [np.random.RandomState(0).rand(375, L)]
        imgs = FR.split_power_spec(spec, c)
        assert len(imgs) == n and all(i.shape == (375, 1024) for i in imgs)
        assert np.array_equal(imgs[0][:, :min(L, 1024)], spec[0][:, :min(L, 1024)])


def test_silence_is_nan():
    imgs, _ = FR.process_waveform(np.zeros(132300, np.float32))
    assert np.isnan(imgs[0]).all()                                  # 0/0, reference Appendix C-8


def test_upsampler_properties():
    hq = FR.upsample2x_coeffs()
    assert 2 * hq.sum() == 32768          # unity DC gain in Q15 (the kernels accumulate in int64)
    x = synth.clip_pcm16(3)
    y = FR.upsample2x_pcm16(x)
    assert y.dtype == np.int16 and len(y) == 2 * len(x) and np.array_equal(y[0::2], x)
    # a 3 kHz tone sampled at 22.05 kHz is reconstructed at 44.1 kHz to < 0.1 % of full scale
    n = np.arange(8000)
    tone = np.round(12000 * np.sin(2 * np.pi * 3000 * n / 22050)).astype(np.int16)
    up = FR.upsample2x_pcm16(tone).astype(np.float64)
    ideal = 12000 * np.sin(2 * np.pi * 3000 * np.arange(16000) / 44100)
    assert np.abs(up[200:-200] - ideal[200:-200]).max() < 33
    # DC gain exactly 1
    assert np.array_equal(FR.upsample2x_pcm16(np.full(200, 1000, np.int16))[40:-40], np.full(320, 1000, np.int16))


# --------------------------------------------------------------------------- wav formats and the generic resampler
def _write_wav(path, tag, bits, sr, channels):
    """channels: list of equally long sample lists (already in the file's native number format)."""
    import struct
    nch, n = len(channels), len(channels[0])
    fmtc = {(1, 8): 'B', (1, 16): 'h', (1, 32): 'i', (3, 32): 'f', (3, 64): 'd'}.get((tag, bits))
    body = bytearray()
    for i in range(n):
        for ch in channels:
            if bits == 24:
                body += int(ch[i] & 0xFFFFFF).to_bytes(3, 'little')
            else:
                body += struct.pack('<' + fmtc, ch[i])
    fmt = struct.pack('<HHIIHH', tag, nch, sr, sr * nch * bits // 8, nch * bits // 8, bits)
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<I', 4 + 8 + len(fmt) + 8 + len(body)) + b'WAVE' + b'fmt ' + struct.pack('<I', len(fmt))
                + fmt + b'data' + struct.pack('<I', len(body)) + bytes(body))


def test_wav_formats_known_answers(tmp_path):
    """Both readers (oracle and product) on hand-built files: librosa.load conventions."""
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import read_wav
    cases = [((1, 8), [[0, 128, 255]], [-1.0, 0.0, 127 / 128]),
             ((1, 16), [[-32768, 0, 32767]], [-1.0, 0.0, 32767 / 32768]),
             ((1, 24), [[-8388608, 1, 8388607]], [-1.0, 2.0 ** -23, 8388607 / 8388608]),
             ((1, 32), [[-2147483648, 65536, 2147483647]], [-1.0, 2.0 ** -15, 1.0]),
             ((3, 32), [[-0.5, 0.25, 0.999]], [-0.5, 0.25, np.float32(0.999)]),
             ((3, 64), [[-0.5, 0.25, 0.3]], [-0.5, 0.25, np.float32(0.3)]),
             ((1, 16), [[1000, -2000, 3], [3000, 1000, 4]], [2000 / 32768, -500 / 32768, 3.5 / 32768])]
    for (tag, bits), chans, want in cases:
        p = str(tmp_path / f'f{tag}_{bits}_{len(chans)}.wav')
        _write_wav(p, tag, bits, 48000, chans)
        for reader in (FR.read_wav, read_wav):
            y, sr = reader(p)
            y = y.astype(np.float32) / np.float32(32768) if y.dtype == np.int16 else y
            assert sr == 48000 and y.dtype == np.float32 and np.array_equal(y, np.asarray(want, dtype=np.float32)), (tag, bits, y)
    y, _ = read_wav(str(tmp_path / 'f1_16_1.wav'))
    assert y.dtype == np.int16                                       # mono 16-bit PCM keeps the exact integer path


def test_resampler_known_answers():
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import resample_taps
    for sr, (L, M) in ((48000, (147, 160)), (16000, (441, 160)), (32000, (441, 320)), (96000, (147, 320)), (8000, (441, 80))):
        l, m, taps = FR.resample_taps(sr)
        l2, m2, taps2 = resample_taps(sr)
        assert (l, m) == (l2, m2) == (L, M) and taps.shape == taps2.shape and np.abs(taps - taps2).max() < 1e-15
        assert np.allclose(taps.sum(1), 1.0, atol=1e-14)
    # a 3 kHz tone at 48 kHz comes out as the same tone at 44.1 kHz (filter error + 16-bit rounding)
    n = np.arange(48000)
    y = FR.resample_to_pcm16(0.5 * np.sin(2 * np.pi * 3000 * n / 48000), 48000)
    assert len(y) == 44100
    ideal = 0.5 * 32768 * np.sin(2 * np.pi * 3000 * np.arange(44100) / 44100)
    assert np.abs(y[100:-100] - ideal[100:-100]).max() < 4
    # DC gain exactly one, 12 kHz survives from 32 kHz, 30 kHz of a 96 kHz file is removed by the anti-alias filter
    assert np.array_equal(FR.resample_to_pcm16(np.full(4000, 1000 / 32768), 16000)[200:-200], np.full(11025 - 400, 1000))
    t12 = FR.resample_to_pcm16(0.5 * np.sin(2 * np.pi * 12000 * np.arange(32000) / 32000), 32000)
    assert 0.45 * 32768 < np.abs(t12[1000:-1000]).max() < 0.51 * 32768
    t30 = FR.resample_to_pcm16(0.5 * np.sin(2 * np.pi * 30000 * np.arange(96000) / 96000), 96000)
    assert np.abs(t30[1000:-1000]).max() < 0.001 * 32768
    assert FR.soundfile_round_trip(np.array([16384 / 32768, 16385 / 32768, -1.0, 1 / 32768], np.float32)).tolist() == \
        [16384 / 32768, 16384 / 32768, -32767 / 32768, 1 / 32768]
