"""ctypes front-end to the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It loads

* ``oracle/libako_oracle.so``  -- our scalar restatement (oracle/ako_oracle.c), always present
  after ``make -C oracle`` (it travels to the GPU box as a built file), and
* ``oracle/_ref/libako_ref.so`` -- the reference itself compiled from /root/reference by
  oracle/Makefile, when it has been built (``have_ref()``).

Nothing here reads /root/reference at run time.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# enums of include/ako.h (reference: library/ako.h:43-73)
DD137, CDF53, HAAR, WAVELET_NONE = 0, 1, 2, 3
YCOCG, SUBTRACT_G, COLOR_NONE, YCOCG_Q = 0, 1, 2, 3
CLAMP, MIRROR, REPEAT, ZERO = 0, 1, 2, 3
KAGARI, MANBAVARAN, COMPRESSION_NONE = 0, 1, 2


class Settings(C.Structure):
    """struct akoSettings (include/ako.h; reference library/ako.h:86-99)."""

    _fields_ = [
        ("wavelet", C.c_int),
        ("color", C.c_int),
        ("wrap", C.c_int),
        ("compression", C.c_int),
        ("tiles_dimension", C.c_size_t),
        ("quantization", C.c_int),
        ("gate", C.c_int),
        ("chroma_loss", C.c_int),
        ("discard_non_visible", C.c_int),
    ]


def settings(wavelet=DD137, color=YCOCG, wrap=CLAMP, compression=COMPRESSION_NONE, tiles=0, q=16, g=0,
             chroma_loss=1, discard=0) -> Settings:
    return Settings(wavelet, color, wrap, compression, tiles, q, g, chroma_loss, discard)


def effective_color(s: Settings) -> int:
    """The encoder's YCOCG <-> YCOCG_Q fix-up (reference: library/encode.c:59-64)."""
    if s.color == YCOCG and (s.quantization > 0 or s.gate > 0):
        return YCOCG_Q
    if s.color == YCOCG_Q and s.quantization <= 0 and s.gate <= 0:
        return YCOCG
    return s.color


def build(force: bool = False) -> None:
    """make -C oracle (restatement, and the reference when /root/reference exists)."""
    if force or not os.path.exists(os.path.join(HERE, "libako_oracle.so")):
        subprocess.run(["make", "-C", HERE, "-s"], check=True)


_orc = None
_ref = None

_u8p = C.POINTER(C.c_uint8)
_i16p = C.POINTER(C.c_int16)


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


def lib() -> C.CDLL:
    global _orc
    if _orc is None:
        build()
        L = C.CDLL(os.path.join(HERE, "libako_oracle.so"))
        L.orcTileStreamBytes.restype = C.c_size_t
        L.orcTileStreamBytes.argtypes = [C.c_size_t, C.c_size_t]
        L.orcLevels.restype = C.c_size_t
        L.orcLevels.argtypes = [C.c_size_t, C.c_size_t]
        L.orcTilesNo.restype = C.c_size_t
        L.orcTilesNo.argtypes = [C.c_size_t] * 3
        L.orcQuantStep.restype = C.c_int16
        L.orcQuantStep.argtypes = [C.c_int, C.c_int] + [C.c_size_t] * 4
        L.orcGateStep.restype = C.c_int16
        L.orcGateStep.argtypes = [C.c_int, C.c_int] + [C.c_size_t] * 4
        L.orcLift1d.restype = None
        L.orcLift1d.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_int, _i16p, C.c_ssize_t, _i16p, C.c_ssize_t, _i16p,
                                C.c_ssize_t]
        L.orcUnlift1d.restype = None
        L.orcUnlift1d.argtypes = [C.c_int, C.c_int, C.c_size_t, _i16p, C.c_ssize_t, _i16p, C.c_ssize_t, _i16p,
                                  C.c_ssize_t, _i16p, C.c_ssize_t]
        L.orcLiftPlane.restype = C.c_int
        L.orcLiftPlane.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, _i16p, _i16p]
        L.orcUnliftPlane.restype = C.c_int
        L.orcUnliftPlane.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, _i16p, _i16p]
        L.orcEncodeTile.restype = C.c_int
        L.orcEncodeTile.argtypes = [C.POINTER(Settings), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _u8p, _i16p]
        L.orcDecodeTile.restype = C.c_int
        L.orcDecodeTile.argtypes = [C.POINTER(Settings), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _i16p, _u8p]
        L.orcEncodeImage.restype = C.c_size_t
        L.orcEncodeImage.argtypes = [C.POINTER(Settings), C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        L.orcDecodeImage.restype = C.c_void_p
        L.orcDecodeImage.argtypes = [C.c_size_t, C.c_void_p, C.POINTER(Settings), C.POINTER(C.c_size_t),
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.orcLastTransformSeconds.restype = C.c_double
        L.orcKagariEncode.restype = C.c_size_t
        L.orcKagariEncode.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orcKagariDecode.restype = C.c_size_t
        L.orcKagariDecode.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orcHeadWrite.restype = C.c_int
        L.orcHeadWrite.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(Settings), C.c_void_p]
        L.orcGenImage.restype = None
        L.orcGenImage.argtypes = [C.c_int, C.c_uint32, C.c_size_t, C.c_size_t, _u8p]
        L.orcGenPlane.restype = None
        L.orcGenPlane.argtypes = [C.c_uint32, C.c_size_t, _i16p]
        L.orcAdler32.restype = C.c_uint32
        L.orcAdler32.argtypes = [_u8p, C.c_size_t]
        _orc = L
    return _orc


_libc_handle = None


def _libc():
    global _libc_handle
    if _libc_handle is None:
        _libc_handle = C.CDLL(None)
        _libc_handle.free.argtypes = [C.c_void_p]
        _libc_handle.free.restype = None
    return _libc_handle


def have_ref() -> bool:
    return os.path.exists(os.path.join(HERE, "_ref", "libako_ref.so"))


def ref() -> C.CDLL:
    """The compiled reference (oracle/_ref/libako_ref.so)."""
    global _ref
    if _ref is None:
        R = C.CDLL(os.path.join(HERE, "_ref", "libako_ref.so"))
        R.akoEncodeExt.restype = C.c_size_t
        R.akoEncodeExt.argtypes = [C.c_void_p, C.POINTER(Settings), C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        R.akoDecodeExt.restype = C.c_void_p
        R.akoDecodeExt.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Settings), C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        R.akoDefaultFree.argtypes = [C.c_void_p]
        R.akoTileDataSize.restype = C.c_size_t
        R.akoTileDataSize.argtypes = [C.c_size_t, C.c_size_t]
        for name in ("akoQuantization", "akoGate"):
            f = getattr(R, name)
            f.restype = C.c_int16
            f.argtypes = [C.c_int, C.c_int] + [C.c_size_t] * 4
        for name in ("akoDd137LiftH", "akoCdf53LiftH"):
            f = getattr(R, name)
            f.restype = None
            f.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _i16p, _i16p]
        for name in ("akoDd137LiftV", "akoCdf53LiftV"):
            f = getattr(R, name)
            f.restype = None
            f.argtypes = [C.c_int, C.c_size_t, C.c_size_t, _i16p, _i16p]
        for name in ("akoDd137UnliftH", "akoCdf53UnliftH"):
            f = getattr(R, name)
            f.restype = None
            f.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _i16p, _i16p, _i16p]
        R.akoKagariEncode.restype = C.c_size_t
        R.akoKagariEncode.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        R.akoKagariDecode.restype = C.c_size_t
        R.akoKagariDecode.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        _ref = R
    return _ref


# ---------------------------------------------------------------------------------------------
# numpy-level helpers
# ---------------------------------------------------------------------------------------------

def adler32(buf) -> int:
    """Adler-32 as printed by the reference tools' -ch flag (tools/misc.hpp:59-82) == zlib's."""
    return zlib.adler32(memoryview(np.ascontiguousarray(buf)).cast("B")) & 0xFFFFFFFF


def gen_image(generator: int, w: int, h: int, seed: int = 0x9E3779B9) -> np.ndarray:
    """Synthetic RGBA image, SURVEY 8d: generator 0 = G0 smooth, 1 = G1 noise."""
    img = np.empty((h, w, 4), dtype=np.uint8)
    lib().orcGenImage(generator, seed & 0xFFFFFFFF, w, h, _ptr(img, _u8p))
    return img


def gen_plane(n: int, seed: int = 0x9E3779B9) -> np.ndarray:
    """Synthetic int16 plane G2 (values in [-512, 511])."""
    p = np.empty(n, dtype=np.int16)
    lib().orcGenPlane(seed & 0xFFFFFFFF, n, _ptr(p, _i16p))
    return p


def tile_stream_values(w: int, h: int) -> int:
    return lib().orcTileStreamBytes(w, h) // 2


def _encode_with(fn, first_null: bool, s: Settings, img: np.ndarray):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    out = C.c_void_p()
    st = C.c_int(-1)
    args = [C.byref(s), ch, w, h, img.ctypes.data_as(C.c_void_p), C.byref(out), C.byref(st)]
    if first_null:
        args = [None] + args
    size = fn(*args)
    if size == 0:
        return None, st.value
    blob = np.ctypeslib.as_array(C.cast(out, _u8p), shape=(size,)).copy()
    _libc().free(out)
    return blob, st.value


def _decode_with(fn, first_null: bool, blob: np.ndarray):
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    s = Settings()
    ch, w, h = C.c_size_t(), C.c_size_t(), C.c_size_t()
    st = C.c_int(-1)
    args = [blob.size, blob.ctypes.data_as(C.c_void_p), C.byref(s), C.byref(ch), C.byref(w), C.byref(h),
            C.byref(st)]
    if first_null:
        args = [None] + args
    p = fn(*args)
    if not p:
        return None, s, st.value
    img = np.ctypeslib.as_array(C.cast(p, _u8p), shape=(h.value, w.value, ch.value)).copy()
    _libc().free(p)
    return img, s, st.value


def encode_image(s: Settings, img: np.ndarray):
    """Our restatement: (blob | None, status)."""
    return _encode_with(lib().orcEncodeImage, False, s, img)


def decode_image(blob: np.ndarray):
    return _decode_with(lib().orcDecodeImage, False, blob)


def ref_encode_image(s: Settings, img: np.ndarray):
    """The compiled reference's akoEncodeExt with default callbacks."""
    return _encode_with(ref().akoEncodeExt, True, s, img)


def ref_decode_image(blob: np.ndarray):
    return _decode_with(ref().akoDecodeExt, True, blob)


def lift_plane(wavelet: int, wrap: int, plane: np.ndarray) -> np.ndarray:
    plane = np.ascontiguousarray(plane, dtype=np.int16)
    h, w = plane.shape
    out = np.empty(tile_stream_values(w, h), dtype=np.int16)
    lib().orcLiftPlane(wavelet, wrap, w, h, _ptr(plane, _i16p), _ptr(out, _i16p))
    return out


def unlift_plane(wavelet: int, wrap: int, w: int, h: int, stream: np.ndarray) -> np.ndarray:
    stream = np.ascontiguousarray(stream, dtype=np.int16)
    out = np.empty((h, w), dtype=np.int16)
    lib().orcUnliftPlane(wavelet, wrap, w, h, _ptr(stream, _i16p), _ptr(out, _i16p))
    return out


def ref_lift_plane(wavelet: int, wrap: int, plane: np.ndarray, timing: dict | None = None) -> np.ndarray:
    """The compiled reference's private akoLift (library/lifting.c:171; prototype library/ako-private.h:81-82) on ONE
    int16 plane taken as a one-channel tile, buffers laid out as library/encode.c:84-148 lays them out (plane at the start
    of work area A, akoPlanesSpacing() values of spacing behind it, stream written to the start of work area B).
    timing["lift_s"] receives the call's wall time."""
    import time

    plane = np.ascontiguousarray(plane, dtype=np.int16)
    h, w = plane.shape
    R = ref()
    n = R.akoTileDataSize(w, h) // 2
    spacing = 2 * w + 2 * h  # akoPlanesSpacing, library/misc.c:104-107
    a = np.zeros(n + spacing, dtype=np.int16)
    b = np.zeros(n + spacing, dtype=np.int16)
    a[: w * h] = plane.reshape(-1)
    st = settings(wavelet=wavelet, wrap=wrap, color=2, q=0, g=0)
    R.akoLift.restype = None
    R.akoLift.argtypes = [C.c_size_t, C.POINTER(Settings), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _i16p, _i16p]
    t0 = time.perf_counter()
    R.akoLift(0, C.byref(st), 1, w, h, spacing, _ptr(a, _i16p), _ptr(b, _i16p))
    if timing is not None:
        timing["lift_s"] = time.perf_counter() - t0
    return b[:n].copy()


def ref_unlift_plane(wavelet: int, wrap: int, w: int, h: int, stream: np.ndarray, timing: dict | None = None) -> np.ndarray:
    """The compiled reference's private akoUnlift (library/lifting.c:295) on a one-channel tile stream
    (call as in library/decode.c:183-187)."""
    import time

    R = ref()
    n = R.akoTileDataSize(w, h) // 2
    spacing = 2 * w + 2 * h
    a = np.zeros(n + spacing, dtype=np.int16)
    b = np.zeros(n + spacing, dtype=np.int16)
    a[:n] = np.ascontiguousarray(stream, dtype=np.int16).reshape(-1)[:n]
    st = settings(wavelet=wavelet, wrap=wrap, color=2, q=0, g=0)
    R.akoUnlift.restype = None
    R.akoUnlift.argtypes = [C.POINTER(Settings), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _i16p, _i16p]
    t0 = time.perf_counter()
    R.akoUnlift(C.byref(st), 1, 0, w, h, spacing, _ptr(a, _i16p), _ptr(b, _i16p))
    if timing is not None:
        timing["unlift_s"] = time.perf_counter() - t0
    return b[: w * h].reshape(h, w).copy()


def encode_tile(s: Settings, img: np.ndarray) -> np.ndarray:
    """u8 image (h, w, ch) taken as ONE tile -> int16 coefficient stream.  s.color must be effective."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    n = tile_stream_values(w, h) * ch if s.wavelet != WAVELET_NONE else w * h * ch
    out = np.empty(n, dtype=np.int16)
    rc = lib().orcEncodeTile(C.byref(s), ch, w, h, w, _ptr(img, _u8p), _ptr(out, _i16p))
    assert rc == 0
    return out


def decode_tile(s: Settings, ch: int, w: int, h: int, stream: np.ndarray) -> np.ndarray:
    stream = np.ascontiguousarray(stream, dtype=np.int16)
    out = np.empty((h, w, ch), dtype=np.uint8)
    rc = lib().orcDecodeTile(C.byref(s), ch, w, h, w, _ptr(stream, _i16p), _ptr(out, _u8p))
    assert rc == 0
    return out


def quant_table(s: Settings, tile_w: int, tile_h: int):
    """[(cur_w, cur_h, tgt_w, tgt_h, qY, gY, qC, gC)] from the largest level down (lifting.c:182-211)."""
    rows = []
    w, h = tile_w, tile_h
    while w > 2 and h > 2:
        cw, ch = w, h
        w, h = (w + 1) // 2, (h + 1) // 2
        L = lib()
        rows.append((cw, ch, w, h,
                     L.orcQuantStep(s.quantization, 1, tile_w, tile_h, cw, ch),
                     L.orcGateStep(s.gate, 1, tile_w, tile_h, cw, ch),
                     L.orcQuantStep(s.quantization, s.chroma_loss + 1, tile_w, tile_h, cw, ch),
                     L.orcGateStep(s.gate, s.chroma_loss + 1, tile_w, tile_h, cw, ch)))
    return rows
