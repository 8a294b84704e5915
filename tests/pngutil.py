"""Minimal PNG reader / writer on zlib for the command-line tool tests (8 bit, colour types 0/2/4/6,
all five filters, optional Adam7).  Independent of tools/cli_common.hpp so that it can check it."""
from __future__ import annotations

import struct
import zlib

import numpy as np

_CTYPE = {1: 0, 2: 4, 3: 2, 4: 6}
_CHANNELS = {0: 1, 4: 2, 2: 3, 6: 4}
_A7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]  # x0 y0 dx dy


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _filter_rows(img: np.ndarray, filters) -> bytes:
    h, w, ch = img.shape
    out = bytearray()
    prev = np.zeros(w * ch, dtype=np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        f = filters[y % len(filters)]
        left = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        upleft = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        if f == 0:
            pred = np.zeros_like(cur)
        elif f == 1:
            pred = left
        elif f == 2:
            pred = prev
        elif f == 3:
            pred = (left + prev) >> 1
        else:
            pred = np.array([_paeth(int(a), int(b), int(c)) for a, b, c in zip(left, prev, upleft)], dtype=np.int32)
        out.append(f)
        out += ((cur - pred) & 255).astype(np.uint8).tobytes()
        prev = cur
    return bytes(out)


def write_png(img: np.ndarray, filters=(0, 1, 2, 3, 4), interlace: bool = False, idat_split: int = 0) -> bytes:
    """img: uint8 [h][w][channels]; filters: filter type per row, cycled."""
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, ch = img.shape
    if not interlace:
        raw = _filter_rows(img, filters)
    else:
        raw = b""
        for (x0, y0, dx, dy) in _A7:
            sub = img[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += _filter_rows(np.ascontiguousarray(sub), filters)
    packed = zlib.compress(raw, 6)

    def chunk(kind: bytes, body: bytes) -> bytes:
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, _CTYPE[ch], 0, 0, 1 if interlace else 0))
    out += chunk(b"tEXt", b"Comment\x00ako test image")  # an ancillary chunk readers must skip
    if idat_split:
        for at in range(0, len(packed), idat_split):
            out += chunk(b"IDAT", packed[at:at + idat_split])
    else:
        out += chunk(b"IDAT", packed)
    return out + chunk(b"IEND", b"")


def read_png(data: bytes) -> np.ndarray:
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, head = 8, b"", None
    while at < len(data):
        n, kind = struct.unpack(">I4s", data[at:at + 8])
        body = data[at + 8:at + 8 + n]
        assert struct.unpack(">I", data[at + 8 + n:at + 12 + n])[0] == zlib.crc32(kind + body) & 0xFFFFFFFF
        if kind == b"IHDR":
            head = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat += body
        at += n + 12
    w, h, depth, ctype, _, _, interlace = head
    assert depth == 8 and interlace == 0
    ch = _CHANNELS[ctype]
    raw = zlib.decompress(idat)
    line = w * ch
    img = np.zeros((h, line), dtype=np.uint8)
    prev = np.zeros(line, dtype=np.int32)
    for y in range(h):
        f = raw[y * (line + 1)]
        cur = np.frombuffer(raw, dtype=np.uint8, count=line, offset=y * (line + 1) + 1).astype(np.int32)
        rec = np.zeros(line, dtype=np.int32)
        if f == 0:
            rec = cur
        elif f == 2:
            rec = (cur + prev) & 255
        else:
            for i in range(line):
                a = rec[i - ch] if i >= ch else 0
                b = prev[i]
                c = prev[i - ch] if i >= ch else 0
                pred = a if f == 1 else ((a + b) >> 1 if f == 3 else _paeth(int(a), int(b), int(c)))
                rec[i] = (cur[i] + pred) & 255
        img[y] = rec.astype(np.uint8)
        prev = rec
    return img.reshape(h, w, ch)
