"""TEST INFRASTRUCTURE ONLY (oracle): 8-bit greyscale PNG codec in numpy / pure Python.

The reference reads its training images with `imageio.imread` (image_dataset.py:44,62,77); imageio is absent from this
image, so the container format is restated here from the PNG specification (ISO/IEC 15948: signature, IHDR/IDAT/IEND
chunks, zlib stream, per-scanline filter types 0-4).  The product's decoder (zlib inflate on the host, wavefront
unfilter kernel on the device) is checked against `decode_png_gray8`; `encode_png_gray8` writes the synthetic datasets
the tests and `oracle/make_golden.py` use.
"""
import struct
import zlib

import numpy as np

SIGNATURE = b'\x89PNG\r\n\x1a\n'


def _chunk(kind, data):
    return struct.pack('>I', len(data)) + kind + data + struct.pack('>I', zlib.crc32(kind + data) & 0xFFFFFFFF)


def _paeth_predict(a, b, c):
    """a = left, b = up, c = up-left (int arrays) -> predictor, PNG spec 9.4."""
    p = a + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))


def encode_png_gray8(img, filters=None, level=6):
    """img uint8 [H,W] -> PNG bytes; `filters[r]` in 0..4 per scanline (default: cycles through all five types so that
    decoders are exercised on every predictor)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    if filters is None:
        filters = np.arange(H) % 5
    x = img.astype(np.int32)
    up = np.vstack([np.zeros((1, W), np.int32), x[:-1]])
    left = np.hstack([np.zeros((H, 1), np.int32), x[:, :-1]])
    upleft = np.hstack([np.zeros((H, 1), np.int32), up[:, :-1]])
    pred = [np.zeros_like(x), left, up, (left + up) // 2, _paeth_predict(left, up, upleft)]
    raw = np.empty((H, W + 1), np.uint8)
    for r in range(H):
        f = int(filters[r])
        raw[r, 0] = f
        raw[r, 1:] = ((x[r] - pred[f][r]) & 0xFF).astype(np.uint8)
    ihdr = struct.pack('>IIBBBBB', W, H, 8, 0, 0, 0, 0)
    return SIGNATURE + _chunk(b'IHDR', ihdr) + _chunk(b'IDAT', zlib.compress(raw.tobytes(), level)) + _chunk(b'IEND', b'')


def parse_png(data):
    """-> (width, height, concatenated IDAT payload); raises ValueError for anything but 8-bit grey, non-interlaced."""
    if data[:8] != SIGNATURE:
        raise ValueError('not a PNG file')
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, kind = struct.unpack('>I4s', data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', body)
        elif kind == b'IDAT':
            idat.append(body)
        elif kind == b'IEND':
            break
    if hdr is None or hdr[2:] != (8, 0, 0, 0, 0):
        raise ValueError(f'unsupported PNG flavour {hdr}: 8-bit greyscale, non-interlaced only')
    return hdr[0], hdr[1], b''.join(idat)


def decode_png_gray8(data):
    """PNG bytes -> uint8 [H,W] (scanline reconstruction, PNG spec 9.2)."""
    W, H, payload = parse_png(data)
    raw = np.frombuffer(zlib.decompress(payload), dtype=np.uint8).reshape(H, W + 1)
    out = np.zeros((H, W), np.uint8)
    prev = np.zeros(W, np.int32)
    for r in range(H):
        f, line = int(raw[r, 0]), raw[r, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 1:
            cur = np.cumsum(line) & 0xFF
        elif f == 2:
            cur = (line + prev) & 0xFF
        elif f in (3, 4):
            cur = np.empty(W, np.int32)
            a = c = 0
            pl = prev.tolist()
            ll = line.tolist()
            res = [0] * W
            for i in range(W):
                b = pl[i]
                if f == 3:
                    p = (a + b) >> 1
                else:
                    q = a + b - c
                    pa, pb, pc = abs(q - a), abs(q - b), abs(q - c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                a = (ll[i] + p) & 0xFF
                c = b
                res[i] = a
            cur = np.array(res, np.int32)
        else:
            raise ValueError(f'bad filter type {f} on scanline {r}')
        out[r] = cur.astype(np.uint8)
        prev = cur
    return out
