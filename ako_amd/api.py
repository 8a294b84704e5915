"""Python mirror of the reference's ``ako.h`` interface, bound to ``libako.so`` over its C-ABI.

Two levels, both thin:

* :func:`encode` / :func:`decode` call ``akoEncodeExt`` / ``akoDecodeExt`` (include/ako.h;
  reference library/encode.c:38, library/decode.c:38) on host buffers -- what ``akoenc`` /
  ``akodec`` do (tools/akoenc.cpp:112-217, tools/akodec.cpp:100-154).
* :class:`Plan` drives the device-resident C-ABI of include/ako_hip.h with ``torch`` CUDA tensors
  (PyTorch is only used for device memory and streams).

There is no CPU fallback anywhere: if ``libako.so`` is missing, or no HIP device is usable, the
calls raise / return the library's error status.
"""
from __future__ import annotations

import ctypes as C
import time
import os
import weakref
from typing import Callable, Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AKO_LIB_OVERRIDE") or os.path.join(HERE, "libako.so")  # override: timing experiments

# enum akoWavelet / akoColor / akoWrap / akoCompression / akoStatus (include/ako.h)
DD137, CDF53, HAAR, WAVELET_NONE = 0, 1, 2, 3
YCOCG, SUBTRACT_G, COLOR_NONE, YCOCG_Q = 0, 1, 2, 3
CLAMP, MIRROR, REPEAT, ZERO = 0, 1, 2, 3
KAGARI, MANBAVARAN, COMPRESSION_NONE = 0, 1, 2
AKO_OK, AKO_ERROR = 0, 1
PLAN_PLANES_I16 = 1

EVENT_NAMES = {1: "FORMAT_START", 2: "FORMAT_END", 3: "WAVELET_START", 4: "WAVELET_END", 5: "COMPRESSION_START",
               6: "COMPRESSION_END"}


class AkoError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        msg = f"{where}: status {status} ({status_string(status)})"
        if detail:
            msg += f": {detail}"
        super().__init__(msg)


class Settings(C.Structure):
    """struct akoSettings (include/ako.h)."""

    _fields_ = [("wavelet", C.c_int), ("color", C.c_int), ("wrap", C.c_int), ("compression", C.c_int),
                ("tiles_dimension", C.c_size_t), ("quantization", C.c_int), ("gate", C.c_int),
                ("chroma_loss", C.c_int), ("discard_non_visible", C.c_int)]

    def copy(self) -> "Settings":
        out = Settings()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(Settings))
        return out


_EVENTS_FN = C.CFUNCTYPE(None, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p)


class Callbacks(C.Structure):
    """struct akoCallbacks (include/ako.h)."""

    _fields_ = [("malloc", C.c_void_p), ("realloc", C.c_void_p), ("free", C.c_void_p), ("events", _EVENTS_FN),
                ("events_data", C.c_void_p)]


class KagariRun(C.Structure):
    _fields_ = [("out_start", C.c_uint32), ("count", C.c_uint32), ("after", C.c_uint32), ("pad", C.c_uint32)]


class KagariTokens(C.Structure):
    _fields_ = [("literals", C.POINTER(C.c_int16)), ("n_literals", C.c_size_t), ("cap_literals", C.c_size_t),
                ("runs", C.POINTER(KagariRun)), ("n_runs", C.c_size_t), ("cap_runs", C.c_size_t)]


class KernelRecord(C.Structure):
    """struct akoHipKernelRecord (include/ako_hip.h)."""

    _fields_ = [("name", C.c_char * 48), ("ms", C.c_float), ("level", C.c_uint32), ("group", C.c_uint32),
                ("units", C.c_uint64), ("bytes_rd", C.c_uint64), ("bytes_wr", C.c_uint64)]


_lib = None


def lib() -> C.CDLL:
    """Load libako.so; loud failure when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m ako_amd.build` "
                          "(there is no CPU fallback for the transform path)")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1.  Whichever HIP
    # runtime is loaded FIRST serves the whole process (same sonames); if the system runtime that
    # libako.so links against comes first, torch afterwards reports "No HIP GPUs are available".
    # So make sure torch's copy is in before libako.so is opened.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz = C.c_void_p, C.c_size_t
    L.akoEncodeExt.restype = sz
    L.akoEncodeExt.argtypes = [C.POINTER(Callbacks), C.POINTER(Settings), sz, sz, sz, vp, C.POINTER(vp),
                               C.POINTER(C.c_int)]
    L.akoDecodeExt.restype = vp
    L.akoDecodeExt.argtypes = [C.POINTER(Callbacks), sz, vp, C.POINTER(Settings), C.POINTER(sz), C.POINTER(sz),
                               C.POINTER(sz), C.POINTER(C.c_int)]
    L.akoDefaultSettings.restype = Settings
    L.akoDefaultCallbacks.restype = Callbacks
    L.akoDefaultFree.argtypes = [vp]
    L.akoDefaultFree.restype = None
    L.akoStatusString.restype = C.c_char_p
    L.akoStatusString.argtypes = [C.c_int]
    for name in ("akoVersionMajor", "akoVersionMinor", "akoVersionPatch", "akoFormatVersion", "akoHipDeviceCount"):
        getattr(L, name).restype = C.c_int
    L.akoHipLastError.restype = C.c_char_p
    L.akoHipEffectiveColor.restype = C.c_int
    L.akoHipEffectiveColor.argtypes = [C.POINTER(Settings)]
    L.akoHipPlanCreate.restype = vp
    L.akoHipPlanCreate.argtypes = [C.c_int, C.POINTER(Settings), sz, sz, sz, sz, vp, C.c_uint, C.POINTER(C.c_int)]
    L.akoHipPlanDestroy.argtypes = [vp]
    L.akoHipPlanDestroy.restype = None
    for name in ("akoHipPlanImageBytes", "akoHipPlanStreamBytes", "akoHipPlanTiles", "akoHipPlanBatch"):
        f = getattr(L, name)
        f.restype = sz
        f.argtypes = [vp]
    L.akoHipPlanTileInfo.restype = C.c_int
    L.akoHipPlanTileInfo.argtypes = [vp, sz] + [C.POINTER(sz)] * 6
    L.akoHipPlanLevels.restype = C.c_int
    L.akoHipPlanLevels.argtypes = [vp, sz]
    L.akoHipPlanQuant.restype = C.c_int
    L.akoHipPlanQuant.argtypes = [vp, sz, sz, sz, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for name in ("akoHipEncode", "akoHipDecode", "akoHipEncodeHost", "akoHipDecodeHost"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [vp, vp, vp]
    L.akoHipSynchronize.restype = C.c_int
    L.akoHipSynchronize.argtypes = [vp]
    L.akoHipPlanSetProfiling.restype = C.c_int
    L.akoHipPlanSetProfiling.argtypes = [vp, C.c_int]
    L.akoHipPlanKernelRecords.restype = sz
    L.akoHipPlanKernelRecords.argtypes = [vp, C.c_int, C.POINTER(KernelRecord), sz]
    # host-side helpers (exported for the host-logic tests)
    L.akoHipEncodeUpload.restype = C.c_int
    L.akoHipEncodeUpload.argtypes = [vp, vp]
    L.akoHipKagariEncode.restype = C.c_int
    L.akoHipKagariEncode.argtypes = [vp, vp, sz, C.POINTER(sz), C.POINTER(sz)]
    L.akoHipKagariFetch.restype = C.c_int
    L.akoHipKagariFetch.argtypes = [vp, vp]
    L.akoHipKagariBody.restype = vp
    L.akoHipKagariBody.argtypes = [vp]
    L.akoHipKagariExpand.restype = C.c_int
    L.akoHipKagariExpand.argtypes = [vp, C.POINTER(C.c_int16), sz, C.POINTER(KagariRun), sz, vp, sz]
    L.akoHipDecodeDownload.restype = C.c_int
    L.akoHipDecodeDownload.argtypes = [vp, vp]
    L.akoHostKagariTokenize.restype = sz
    L.akoHostKagariTokenize.argtypes = [sz, sz, vp, C.c_uint64, C.POINTER(KagariTokens)]
    L.akoHostKagariTokensFree.restype = None
    L.akoHostKagariTokensFree.argtypes = [C.POINTER(KagariTokens)]
    L.akoHostQuantStep.restype = C.c_int16
    L.akoHostQuantStep.argtypes = [C.c_int, C.c_int, sz, sz, sz, sz]
    L.akoHostGateStep.restype = C.c_int16
    L.akoHostGateStep.argtypes = [C.c_int, C.c_int, sz, sz, sz, sz]
    L.akoHostHeadWrite.restype = C.c_int
    L.akoHostHeadWrite.argtypes = [sz, sz, sz, C.POINTER(Settings), vp]
    L.akoHostHeadRead.restype = C.c_int
    L.akoHostHeadRead.argtypes = [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(Settings)]
    L.akoHostKagariEncode.restype = sz
    L.akoHostKagariEncode.argtypes = [sz, sz, vp, vp]
    L.akoHostKagariDecode.restype = sz
    L.akoHostKagariDecode.argtypes = [sz, sz, sz, vp, vp]
    if not hasattr(L, "akoHipBatchCreate"):  # an older build of the library (timing comparisons): the common subset only
        _lib = L
        return L
    L.akoHipThreadRelease.restype = None
    L.akoHipThreadRelease.argtypes = []
    L.akoHipTuningSignature.restype = C.c_uint64
    L.akoHipBatchCreate.restype = vp
    L.akoHipBatchCreate.argtypes = [C.POINTER(C.c_int), sz, sz, C.POINTER(Settings), sz, sz, sz, C.POINTER(C.c_int)]
    L.akoHipBatchDestroy.restype = None
    L.akoHipBatchDestroy.argtypes = [vp]
    L.akoHipBatchLanes.restype = sz
    L.akoHipBatchLanes.argtypes = [vp]
    L.akoHipEncodeBatch.restype = C.c_int
    L.akoHipEncodeBatch.argtypes = [vp, sz, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz), C.POINTER(C.c_int)]
    L.akoHipDecodeBatch.restype = C.c_int
    L.akoHipDecodeBatch.argtypes = [vp, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(C.c_int)]
    L.akoEncodeRatioExt.restype = sz
    L.akoEncodeRatioExt.argtypes = [C.POINTER(Callbacks), C.POINTER(Settings), sz, sz, sz, vp, C.c_int, C.POINTER(vp),
                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.akoHipRequantize.restype = C.c_int
    L.akoHipRequantize.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.akoHipHostAlloc.restype = vp
    L.akoHipHostAlloc.argtypes = [sz]
    if hasattr(L, "akoHipHostIsPinned"):
        L.akoHipHostIsPinned.restype = C.c_int
        L.akoHipHostIsPinned.argtypes = [vp]
    L.akoHipHostFree.restype = None
    L.akoHipHostFree.argtypes = [vp]
    L.akoHostSynthImage.restype = None
    L.akoHostSynthImage.argtypes = [C.c_int, C.c_uint32, sz, sz, vp]
    L.akoHostSynthPlane.restype = None
    L.akoHostSynthPlane.argtypes = [C.c_uint32, sz, vp]
    _lib = L
    return L


def status_string(status: int) -> str:
    try:
        return lib().akoStatusString(int(status)).decode()
    except Exception:  # library missing: still give the number
        return "?"


def default_settings() -> Settings:
    return lib().akoDefaultSettings()


def settings(wavelet=DD137, color=YCOCG, wrap=CLAMP, compression=KAGARI, tiles=0, q=16, g=0, chroma_loss=1,
             discard=0) -> Settings:
    return Settings(wavelet, color, wrap, compression, tiles, q, g, chroma_loss, discard)


def device_count() -> int:
    return lib().akoHipDeviceCount()


def last_error() -> str:
    return lib().akoHipLastError().decode()


# ---------------------------------------------------------------------------------------------
# synthetic inputs of the benchmark configurations, generated by the library's host side
# ---------------------------------------------------------------------------------------------

def synth_image(generator: int, w: int, h: int, seed: int = 0x9E3779B9) -> np.ndarray:
    """RGBA image (h, w, 4) uint8: generator 0 = G0 "smooth", 1 = G1 "noise"."""
    img = np.empty((h, w, 4), dtype=np.uint8)
    lib().akoHostSynthImage(generator, seed & 0xFFFFFFFF, w, h, img.ctypes.data_as(C.c_void_p))
    return img


def synth_plane(n: int, seed: int = 0x9E3779B9) -> np.ndarray:
    """G2: n int16 samples in [-512, 511]."""
    p = np.empty(n, dtype=np.int16)
    lib().akoHostSynthPlane(seed & 0xFFFFFFFF, n, p.ctypes.data_as(C.c_void_p))
    return p


# ---------------------------------------------------------------------------------------------
# ako.h level: host buffers in, host buffers out
# ---------------------------------------------------------------------------------------------

def _callbacks(events: Optional[Callable[[int, int, int], None]]):
    cb = lib().akoDefaultCallbacks()
    keep = None
    if events is not None:
        keep = _EVENTS_FN(lambda tile, total, ev, _data: events(int(tile), int(total), int(ev)))
        cb.events = keep
    return cb, keep


# wall-clock of the library call alone inside the most recent encode() / decode() of this module (the wrappers around
# it copy the blob, and dropping a previous 268 MB result costs more than decoding the next one)
last_call_seconds = {}


def encode(image: np.ndarray, s: Optional[Settings] = None, events=None) -> np.ndarray:
    """akoEncodeExt on an (h, w, channels) or (h, w) uint8 array -> blob (uint8 array)."""
    image = np.ascontiguousarray(image, dtype=np.uint8)
    h, w = image.shape[:2]
    ch = 1 if image.ndim == 2 else image.shape[2]
    cb, keep = _callbacks(events)
    out = C.c_void_p()
    st = C.c_int(-1)
    t0 = time.perf_counter()
    n = lib().akoEncodeExt(C.byref(cb), C.byref(s) if s is not None else None, ch, w, h,
                           image.ctypes.data_as(C.c_void_p), C.byref(out), C.byref(st))
    last_call_seconds["akoEncodeExt"] = time.perf_counter() - t0
    del keep
    if n == 0:
        raise AkoError(st.value, "akoEncodeExt", last_error())
    blob = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(n,)).copy()
    lib().akoDefaultFree(out)
    return blob


def encode_ratio(image: np.ndarray, ratio: int, s: Optional[Settings] = None):
    """akoEncodeRatioExt: the quantization search of tools/akoenc.cpp:112-217 in one call ->
    (blob, quantization it settled on, candidate encodes it replaced, forward transforms it ran)."""
    image = np.ascontiguousarray(image, dtype=np.uint8)
    h, w = image.shape[:2]
    ch = 1 if image.ndim == 2 else image.shape[2]
    out = C.c_void_p()
    st, q, enc, tr = C.c_int(-1), C.c_int(0), C.c_int(0), C.c_int(0)
    n = lib().akoEncodeRatioExt(None, C.byref(s) if s is not None else None, ch, w, h, image.ctypes.data_as(C.c_void_p),
                                ratio, C.byref(out), C.byref(q), C.byref(enc), C.byref(tr), C.byref(st))
    if n == 0:
        raise AkoError(st.value, "akoEncodeRatioExt", last_error())
    blob = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(n,)).copy()
    lib().akoDefaultFree(out)
    return blob, q.value, enc.value, tr.value


def decode(blob, events=None):
    """akoDecodeExt -> (image (h, w, channels) uint8, Settings)."""
    blob = np.ascontiguousarray(np.frombuffer(blob, dtype=np.uint8) if isinstance(blob, (bytes, bytearray)) else blob,
                                dtype=np.uint8)
    cb, keep = _callbacks(events)
    s = Settings()
    ch, w, h = C.c_size_t(), C.c_size_t(), C.c_size_t()
    st = C.c_int(-1)
    t0 = time.perf_counter()
    p = lib().akoDecodeExt(C.byref(cb), blob.size, blob.ctypes.data_as(C.c_void_p), C.byref(s), C.byref(ch),
                           C.byref(w), C.byref(h), C.byref(st))
    last_call_seconds["akoDecodeExt"] = time.perf_counter() - t0
    del keep
    if not p:
        raise AkoError(st.value, "akoDecodeExt", last_error())
    # the library's buffer IS the result (no copy: for a large image that copy costs more than the decode); it is
    # released through akoDefaultFree when the array -- and every view of it -- is gone
    n = h.value * w.value * ch.value
    return _owned_array(p, n, np.uint8, (h.value, w.value, ch.value), lib().akoDefaultFree), s


def _owned_array(p: int, nbytes: int, dtype, shape, free) -> np.ndarray:
    """ndarray over `nbytes` at address `p` that the library allocated.  The finalizer hangs on the ctypes buffer at the
    ROOT of every view chain (NumPy collapses the .base of a plain view -- np.asarray(a), a.view(np.ndarray), a slice --
    to that buffer, not to the array it was taken from), so the memory lives exactly as long as any array over it."""
    root = (C.c_uint8 * nbytes).from_address(p)
    weakref.finalize(root, free, C.c_void_p(p))
    return np.frombuffer(root, dtype=dtype).reshape(shape).view(_Owned)


def pinned_empty(shape, dtype=np.uint8):
    """An uninitialised array in page-locked host memory (akoHipHostAlloc): device copies to / from it run at link
    rate, and the batched API uses it in place.  Freed (akoHipHostFree) when the array and its views are gone."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = lib().akoHipHostAlloc(max(n, 1))
    if not p:
        raise MemoryError(f"akoHipHostAlloc({n})")
    return _owned_array(p, n, dt, shape, lib().akoHipHostFree)


class _Owned(np.ndarray):
    """An ndarray over memory the library allocated (see _owned_array for when it is freed)."""


# ---------------------------------------------------------------------------------------------
# ako_hip.h level: device resident batches
# ---------------------------------------------------------------------------------------------

class Plan:
    """akoHipPlan wrapper.  Tensors are torch CUDA tensors on the plan's device.

    images : uint8 [batch, h, w, channels]            (int16 [batch, channels, h, w] with planes_i16)
    streams: int16 [batch, stream_bytes // 2]
    """

    def __init__(self, s: Settings, channels: int, w: int, h: int, batch: int = 1, device: int = 0,
                 stream: Optional[int] = None, planes_i16: bool = False, effective_color: bool = True):
        import torch  # device memory / stream plumbing only

        self._torch = torch
        self.settings = s.copy()
        if effective_color:
            self.settings.color = lib().akoHipEffectiveColor(C.byref(self.settings))
        self.channels, self.w, self.h, self.batch, self.device = channels, w, h, batch, device
        self.planes_i16 = planes_i16
        if stream is None:
            stream = torch.cuda.current_stream(device).cuda_stream
        st = C.c_int(-1)
        self._p = lib().akoHipPlanCreate(device, C.byref(self.settings), channels, w, h, batch, C.c_void_p(stream),
                                         PLAN_PLANES_I16 if planes_i16 else 0, C.byref(st))
        if not self._p:
            raise AkoError(st.value, "akoHipPlanCreate", last_error())
        self.image_bytes = lib().akoHipPlanImageBytes(self._p)
        self.stream_bytes = lib().akoHipPlanStreamBytes(self._p)
        self.tiles = lib().akoHipPlanTiles(self._p)

    def close(self):
        if getattr(self, "_p", None):
            lib().akoHipPlanDestroy(self._p)
            self._p = None

    def __del__(self):
        if lib is None:  # interpreter shutdown: module globals are gone, the process takes the plan with it
            return
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int, where: str):
        if rc != 0:
            raise AkoError(rc, where, last_error())

    def new_streams(self):
        t = self._torch
        return t.empty((self.batch, self.stream_bytes // 2), dtype=t.int16, device=f"cuda:{self.device}")

    def new_images(self):
        t = self._torch
        if self.planes_i16:
            return t.empty((self.batch, self.channels, self.h, self.w), dtype=t.int16, device=f"cuda:{self.device}")
        return t.empty((self.batch, self.h, self.w, self.channels), dtype=t.uint8, device=f"cuda:{self.device}")

    def encode(self, images, streams=None):
        """akoHipEncode: asynchronous on the plan's stream."""
        assert images.is_cuda and images.is_contiguous()
        assert images.numel() * images.element_size() == self.image_bytes * self.batch
        if streams is None:
            streams = self.new_streams()
        assert streams.is_cuda and streams.is_contiguous()
        assert streams.numel() * streams.element_size() == self.stream_bytes * self.batch
        self._check(lib().akoHipEncode(self._p, C.c_void_p(images.data_ptr()), C.c_void_p(streams.data_ptr())),
                    "akoHipEncode")
        return streams

    def decode(self, streams, images=None):
        """akoHipDecode: asynchronous on the plan's stream."""
        assert streams.is_cuda and streams.is_contiguous()
        assert streams.numel() * streams.element_size() == self.stream_bytes * self.batch
        if images is None:
            images = self.new_images()
        assert images.is_cuda and images.is_contiguous()
        assert images.numel() * images.element_size() == self.image_bytes * self.batch
        self._check(lib().akoHipDecode(self._p, C.c_void_p(streams.data_ptr()), C.c_void_p(images.data_ptr())),
                    "akoHipDecode")
        return images

    def synchronize(self):
        self._check(lib().akoHipSynchronize(self._p), "akoHipSynchronize")

    def tile_info(self, t: int):
        v = [C.c_size_t() for _ in range(6)]
        self._check(lib().akoHipPlanTileInfo(self._p, t, *[C.byref(x) for x in v]), "akoHipPlanTileInfo")
        return dict(zip(("x", "y", "w", "h", "stream_offset", "stream_bytes"), (x.value for x in v)))

    def levels(self, tile: int = 0) -> int:
        return lib().akoHipPlanLevels(self._p, tile)

    def quant(self, tile: int, level: int, channel: int):
        q, g = C.c_int(), C.c_int()
        self._check(lib().akoHipPlanQuant(self._p, tile, level, channel, C.byref(q), C.byref(g)), "akoHipPlanQuant")
        return q.value, g.value

    def kagari_encode(self, streams=None, image: int = 0, fetch: bool = True):
        """akoHipKagariEncode (+ akoHipKagariFetch): the blob body (uint32 size + payload per tile) of one
        image's coefficient streams, entropy-coded on the GPU.  Raises AkoError when a tile does not shrink."""
        n, bad = C.c_size_t(0), C.c_size_t(0)
        ptr = C.c_void_p(streams.data_ptr()) if streams is not None else None
        self._check(lib().akoHipKagariEncode(self._p, ptr, image, C.byref(n), C.byref(bad)), "akoHipKagariEncode")
        if not fetch:
            return n.value
        out = np.empty(n.value, dtype=np.uint8)
        self._check(lib().akoHipKagariFetch(self._p, out.ctypes.data_as(C.c_void_p)), "akoHipKagariFetch")
        return out

    def kagari_decode_body(self, body: np.ndarray, streams=None, image: int = 0):
        """The decoder's entropy stage, device route: parse a blob body ([uint32 size][payload] per tile) on the
        host (akoHostKagariTokenize), expand the runs on the GPU (akoHipKagariExpand) into `streams`."""
        body = np.ascontiguousarray(body, dtype=np.uint8)
        if streams is None:
            streams = self.new_streams()
        tok = KagariTokens()
        at = 0
        try:
            for t in range(self.tiles):
                ti = self.tile_info(t)
                if body.size - at < 4:
                    raise AkoError(15, "kagari_decode_body", "truncated body")
                block = int(body[at:at + 4].view("<u4")[0])
                if body.size - at - 4 < block:
                    raise AkoError(15, "kagari_decode_body", "truncated body")
                used = lib().akoHostKagariTokenize(ti["stream_bytes"] // 2, block, C.c_void_p(body.ctypes.data + at + 4),
                                                   ti["stream_offset"] // 2, C.byref(tok))
                if used == 0 or used != block:
                    raise AkoError(15, "kagari_decode_body", f"tile {t}: broken bit-stream")
                at += block + 4
            self._check(lib().akoHipKagariExpand(self._p, tok.literals, tok.n_literals, tok.runs, tok.n_runs,
                                                 C.c_void_p(streams.data_ptr()), image), "akoHipKagariExpand")
        finally:
            lib().akoHostKagariTokensFree(C.byref(tok))
        return streams

    def set_profiling(self, on: bool):
        lib().akoHipPlanSetProfiling(self._p, 1 if on else 0)

    def kernel_records(self, decode: bool):
        buf = (KernelRecord * 8192)()
        n = lib().akoHipPlanKernelRecords(self._p, 1 if decode else 0, buf, 8192)
        return [dict(name=buf[i].name.decode(), ms=buf[i].ms, level=buf[i].level, group=buf[i].group,
                     units=buf[i].units, bytes_rd=buf[i].bytes_rd, bytes_wr=buf[i].bytes_wr) for i in range(n)]


# ---------------------------------------------------------------------------------------------
# batched host API (include/ako_hip.h: akoHipBatch*): arrays of equally shaped images, all devices
# ---------------------------------------------------------------------------------------------

class Batch:
    """akoHipBatch wrapper: host images -> .ako blobs -> host images, many at a time, over the lanes of one or more
    devices (the caller of tools/akoenc.cpp:112-217 for a whole array; BASELINE configs[3])."""

    def __init__(self, s: Settings, channels: int, w: int, h: int, devices=None, lanes_per_device: int = 0):
        self.channels, self.w, self.h = channels, w, h
        devs = list(devices) if devices else [0]
        arr = (C.c_int * len(devs))(*devs)
        st = C.c_int(-1)
        self._b = lib().akoHipBatchCreate(arr, len(devs), lanes_per_device, C.byref(s), channels, w, h, C.byref(st))
        if not self._b:
            raise AkoError(st.value, "akoHipBatchCreate", last_error())
        self.lanes = lib().akoHipBatchLanes(self._b)

    def close(self):
        if getattr(self, "_b", None):
            lib().akoHipBatchDestroy(self._b)
            self._b = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        if lib is None:
            return
        self.close()

    def encode(self, images):
        """images: sequence of (h, w, channels) uint8 arrays -> (list of blobs | None, list of status)."""
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        for im in imgs:
            assert im.size == self.w * self.h * self.channels
        n = len(imgs)
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        out = (C.c_void_p * n)()
        sizes = (C.c_size_t * n)()
        st = (C.c_int * n)()
        lib().akoHipEncodeBatch(self._b, n, ptrs, out, sizes, st)
        blobs = []
        for i in range(n):
            if st[i] == 0 and out[i]:
                blobs.append(np.ctypeslib.as_array(C.cast(out[i], C.POINTER(C.c_uint8)), shape=(sizes[i],)).copy())
                lib().akoDefaultFree(out[i])
            else:
                blobs.append(None)
        return blobs, list(st)

    def decode(self, blobs, outs=None):
        """blobs of images of this batch's shape -> (list of (h, w, channels) arrays | None, list of status).
        outs: arrays to decode into (reused buffers, or pinned ones from pinned_empty(): those are filled by the
        device copy itself)."""
        bl = [np.ascontiguousarray(np.frombuffer(b, dtype=np.uint8) if isinstance(b, (bytes, bytearray)) else b, dtype=np.uint8)
              for b in blobs]
        n = len(bl)
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bl])
        sizes = (C.c_size_t * n)(*[b.size for b in bl])
        if outs is None:
            outs = [np.empty((self.h, self.w, self.channels), dtype=np.uint8) for _ in range(n)]
        assert len(outs) == n and all(o.dtype == np.uint8 and o.flags.c_contiguous and o.size == self.h * self.w * self.channels
                                      for o in outs)
        optr = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        st = (C.c_int * n)()
        lib().akoHipDecodeBatch(self._b, n, ptrs, sizes, optr, st)
        return [o if st[i] == 0 else None for i, o in enumerate(outs)], list(st)
