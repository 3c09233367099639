"""CPU: host logic of the bulk-inference route (birdsoundclassif_amd/bulk.py): wav header probing, the file grouping the CLI
routes by, and the rows -> per-file output dictionary conversion against the oracle's `merge_images` (reference
run_detection.py:163-249) on a single-window file."""
import os
import struct

import numpy as np
import torch

from birdsoundclassif_amd import bulk, synth
from birdsoundclassif_amd.nets.layers import FastRCNN
from oracle import nets_ref as O


def test_wav_header_and_groups(tmp_path):
    p = lambda n: str(tmp_path / n)
    synth.write_wav(p('a.wav'), np.zeros(66150, np.int16), 22050)
    synth.write_wav(p('b.wav'), np.ones(66150, np.int16), 22050)
    synth.write_wav(p('c.wav'), np.zeros(132300, np.int16), 44100)          # 3 s at 44.1 kHz: its own group
    synth.write_wav(p('long.wav'), np.zeros(3 * 66150, np.int16), 22050)    # several windows
    synth.write_wav(p('odd.wav'), np.zeros(48000, np.int16), 16000)         # another rate: resampler path
    # exactly 1024 frames is the last single-window length: 1 + n44 // 132 <= 1024
    synth.write_wav(p('edge_in.wav'), np.zeros((1023 * 132 + 131) // 2, np.int16), 22050)
    synth.write_wav(p('edge_out.wav'), np.zeros(1024 * 132 // 2, np.int16), 22050)
    # stereo + a LIST chunk in front of the data
    body = np.zeros((1000, 2), '<i2').tobytes()
    raw = b'RIFF' + struct.pack('<I', 36 + 12 + len(body)) + b'WAVE' + b'fmt ' + struct.pack('<IHHIIHH', 16, 1, 2, 22050, 88200, 4, 16) + \
        b'LIST' + struct.pack('<I', 4) + b'abcd' + b'data' + struct.pack('<I', len(body)) + body
    open(p('stereo.wav'), 'wb').write(raw)
    open(p('junk.wav'), 'wb').write(b'not a wav file at all')
    assert bulk.wav_header(p('a.wav')) == (1, 1, 22050, 16, 66150, 44)
    assert bulk.wav_header(p('stereo.wav'))[:5] == (1, 2, 22050, 16, 1000) and bulk.wav_header(p('stereo.wav'))[5] == 56
    files = sorted(str(f) for f in tmp_path.glob('*.wav'))
    groups, rest = bulk.bulk_groups(files)
    names = {k: sorted(os.path.basename(f) for f in v) for k, v in groups.items()}
    assert names == {(22050, 66150): ['a.wav', 'b.wav'], (44100, 132300): ['c.wav'], (22050, (1023 * 132 + 131) // 2): ['edge_in.wav']}
    assert sorted(os.path.basename(f) for f in rest) == ['edge_out.wav', 'junk.wav', 'long.wav', 'odd.wav', 'stereo.wav']


def _rows(seed, n, spectrogram_length):
    """n non-overlapping boxes on a grid (so that no NMS can remove one), some at the right border, sorted by (class, score desc)."""
    u = synth.uniform(('bulkrows', seed), 8 * n).reshape(n, 8)
    rows = np.zeros((n, 6), np.float32)
    for i in range(n):
        gx, gy = i % 10, i // 10
        x1 = 100.0 * gx + np.floor(u[i, 0] * 20)
        w = np.floor(20 + u[i, 1] * 60)
        y1 = 60.0 * gy + np.floor(u[i, 2] * 10)
        h = np.floor(10 + u[i, 3] * 40)
        if gx == 9:                                  # reaches the right border: narrow ones are dropped, wide ones stay
            x1, w = (1023 - w, w) if i % 2 else (800.0, 223.0)
        rows[i] = (1 + int(u[i, 4] * 6), x1, y1, min(x1 + w, 1023), min(y1 + h, 374), np.float32(0.05 + 0.9 * u[i, 5]))
    order = np.lexsort((-rows[:, 5], rows[:, 0]))
    return rows[order]


def test_rows_to_result_equals_merge_images_for_a_single_window():
    names = {i: f'Species {i}' for i in range(0, 151)}
    for seed, n, L in ((0, 37, 1003), (1, 50, 1003), (2, 12, 700), (3, 0, 1003)):
        rows = _rows(seed, n, L) if n else np.zeros((0, 6), np.float32)
        det = torch.zeros((1, 50, 6))
        det[0, :n] = torch.from_numpy(rows)
        d = FastRCNN.dets_to_dicts(det, torch.tensor([n], dtype=torch.int32), 150)[0]
        ref = O.merge_images(1024, 819, L, [d], 150)
        ref = {names[int(k)]: {'bbox_coord': v['bbox_coord'].numpy().tolist(), 'scores': v['scores'].reshape(-1).numpy().tolist()}
               for k, v in ref.items() if len(v['bbox_coord'])}
        got = bulk.rows_to_result(det[0].numpy(), n, 1024, 819, L, names)
        assert got == ref and list(got) == list(ref), (seed, got, ref)
        assert str(got) == str(ref)
        assert bulk.single_window_merge(d, 1024, 819, L, names) == ref
        if n:
            dropped = n - sum(len(v['scores']) for v in got.values())
            assert dropped > 0 or L == 1003 and seed == 3


def test_txt_path_and_result_order():
    assert bulk.txt_path('/a/b/clip.wav') == '/a/b/clip.txt'
    rows = np.array([[3, 10, 10, 40, 40, 0.9], [3, 100, 10, 140, 40, 0.5], [7, 10, 100, 60, 160, 0.8]], dtype=np.float32)
    got = bulk.rows_to_result(rows, 3, 1024, 819, 1003, None)
    assert list(got) == ['3', '7'] and got['3']['scores'] == [float(np.float32(0.9)), 0.5]
    assert bulk.rows_to_result(rows, 0, 1024, 819, 1003, None) == {}
