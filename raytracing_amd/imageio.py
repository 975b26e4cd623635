"""The output step of the reference's ``main()`` (reference main.py:32,
``plt.imsave(path, image)``): 8-bit quantisation and PNG encoding.

matplotlib turns a float RGB image into bytes as ``(clip(v, 0, 1) * 255).astype(uint8)``
(truncation, not rounding) and writes RGBA with an opaque alpha channel.  The
quantisation runs on the GPU (``rt_read_image_rgb8``); :func:`quantise` is the same
arithmetic in NumPy for host arrays.  The PNG encoder needs only zlib.
"""
import struct
import zlib

import numpy as np


def quantise(image):
    """``(..., 3)`` float colours -> uint8 the way ``plt.imsave`` stores them."""
    return (np.clip(np.asarray(image, dtype=np.float64), 0.0, 1.0) * 255).astype(np.uint8)


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_png(rgb8):
    """PNG bytes (8-bit RGBA, alpha 255, no interlace) of an ``(H, W, 3)`` uint8 image."""
    rgb8 = np.asarray(rgb8)
    if rgb8.dtype != np.uint8 or rgb8.ndim != 3 or rgb8.shape[2] != 3:
        raise ValueError("expected an (H, W, 3) uint8 image")
    h, w, _ = rgb8.shape
    rows = np.empty((h, 1 + 4 * w), dtype=np.uint8)
    rows[:, 0] = 0                                   # filter type 0 (none) for every scanline
    px = rows[:, 1:].reshape(h, w, 4)
    px[..., :3] = rgb8
    px[..., 3] = 255
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)
    return (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(rows.tobytes(), 6))
            + _chunk(b"IEND", b""))


def decode_png(data):
    """Inverse of :func:`encode_png` for 8-bit non-interlaced RGB/RGBA PNGs (all five filter
    types); returns ``(H, W, channels)`` uint8.  Used by the tests to read images back."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, w = 8, [], None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if depth != 8 or ctype not in (2, 6) or interlace:
                raise ValueError("only 8-bit non-interlaced RGB/RGBA PNGs are supported")
            ch = 3 if ctype == 2 else 4
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), dtype=np.uint8)
    prev = np.zeros(w * ch, dtype=np.int32)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(w * ch, dtype=np.int32)
            for x in range(w * ch):
                a = cur[x - ch] if x >= ch else 0
                b = prev[x]
                c = prev[x - ch] if x >= ch else 0
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + p) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, ch)


def save_png(path, rgb8):
    with open(path, "wb") as f:
        f.write(encode_png(rgb8))
