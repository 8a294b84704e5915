"""CPU, build container only: the oracle against the compiled reference (oracle/_ref).

Skipped when oracle/_ref has not been built (it cannot be built without /root/reference);
tests/test_oracle_golden.py then carries the pinning through the committed vectors.
"""
import ctypes as C
import random

import numpy as np
import pytest

from oracle import pyoracle as pyo

pytestmark = pytest.mark.skipif(not pyo.have_ref(), reason="oracle/_ref not built (needs /root/reference)")

I16 = C.POINTER(C.c_int16)


def _degenerate(w, h, tiles, wavelet):
    if wavelet == 3:
        return False
    ws = [w] if tiles == 0 else ([tiles] * (w // tiles) + ([w % tiles] if w % tiles else []))
    hs = [h] if tiles == 0 else ([tiles] * (h // tiles) + ([h % tiles] if h % tiles else []))
    return min(ws) <= 2 or min(hs) <= 2


def test_random_settings_sweep(po):
    rng = random.Random(1234)
    nrng = np.random.default_rng(7)
    sizes = [(3, 3), (4, 4), (5, 7), (8, 8), (9, 9), (15, 16), (16, 16), (17, 23), (31, 33), (32, 32), (33, 31),
             (63, 65), (64, 64), (100, 75), (127, 129), (130, 70), (3, 50), (50, 3), (200, 17)]
    compared = 0
    for _ in range(500):
        w, h = rng.choice(sizes)
        ch = rng.choice([1, 2, 3, 4, 5])
        wavelet = rng.choice([0, 0, 1, 2, 3])
        tiles = rng.choice([0, 0, 8, 16, 32, 64])
        comp = rng.choice([2, 2, 0])
        s = po.settings(wavelet=wavelet, color=rng.choice([0, 1, 2]), wrap=rng.randrange(4), compression=comp,
                        tiles=tiles, q=rng.choice([0, 0, 1, 16, 100, 2000]), g=rng.choice([0, 0, 16, 300]),
                        chroma_loss=rng.choice([0, 1, 3]), discard=rng.choice([0, 1]))
        kind = rng.choice(["noise", "smooth", "extreme"])
        if kind == "noise":
            img = nrng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        elif kind == "smooth":
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([((xx * 3 + yy * 2 + c * 40) % 256) for c in range(ch)], -1).astype(np.uint8)
        else:
            img = nrng.choice(np.array([0, 255], dtype=np.uint8), (h, w, ch))
        if s.discard_non_visible and ch in (2, 4):
            img[..., -1] = np.where(nrng.random((h, w)) < 0.3, 0, img[..., -1])

        ob, ost = po.encode_image(s, img)
        if _degenerate(w, h, tiles, wavelet):
            assert ob is None  # refused on purpose, see oracle/ako_oracle.c
            continue
        if wavelet == 3 and comp != 2:
            continue  # the reference entropy-codes past the planes here (uninitialised bytes)
        rb, rst = po.ref_encode_image(s, img)
        if rb is None or ob is None:
            assert (rb is None) == (ob is None) and rst == ost
            continue
        assert np.array_equal(rb, ob)
        rd, _, _ = po.ref_decode_image(rb)
        od, _, _ = po.decode_image(rb)
        assert np.array_equal(rd, od)
        compared += 1
    assert compared > 300


def test_adversarial_streams_decode_alike(po):
    """Full-range int16 coefficient streams: every int16 wrap-around must agree."""
    rng = random.Random(99)
    nrng = np.random.default_rng(3)
    for _ in range(300):
        w = rng.choice([3, 4, 8, 9, 16, 17, 33, 64, 100])
        h = rng.choice([3, 5, 8, 16, 23, 64, 75])
        ch = rng.choice([1, 3, 4])
        s = po.settings(wavelet=rng.choice([0, 1, 2]), color=rng.choice([0, 1, 2, 3]), wrap=rng.randrange(4),
                        compression=2, q=0, g=0)
        head = np.zeros(16, np.uint8)
        assert po.lib().orcHeadWrite(ch, w, h, C.byref(s), head.ctypes.data_as(C.c_void_p)) == 0
        body = nrng.integers(-32768, 32768, po.tile_stream_values(w, h) * ch, dtype=np.int16)
        if rng.random() < 0.5:
            body = (body // 64).astype(np.int16)
        blob = np.concatenate([head, body.view(np.uint8)])
        rd, _, _ = po.ref_decode_image(blob)
        od, _, _ = po.decode_image(blob)
        assert rd is not None and np.array_equal(rd, od)


def test_1d_kernels_full_range(po):
    R, L = po.ref(), po.lib()
    rng = random.Random(5)
    nrng = np.random.default_rng(8)
    for _ in range(2000):
        wv = rng.choice([0, 1])
        wrap = rng.randrange(4)
        T = rng.randrange(5 if wv == 0 else 2, 40)  # the reference's DD137 inverse needs T >= 5 (library: >= 8)
        fake = rng.randrange(2)
        src = nrng.integers(-32768, 32768, 2 * T + 2, dtype=np.int16)
        ref_out = np.zeros(2 * T, np.int16)
        lp, hp = np.zeros(T, np.int16), np.zeros(T, np.int16)
        (R.akoDd137LiftH if wv == 0 else R.akoCdf53LiftH)(wrap, 1, T, fake, 0, src.ctypes.data_as(I16),
                                                           ref_out.ctypes.data_as(I16))
        L.orcLift1d(wv, wrap, T, fake, src.ctypes.data_as(I16), 1, lp.ctypes.data_as(I16), 1,
                    hp.ctypes.data_as(I16), 1)
        assert np.array_equal(ref_out[:T], lp) and np.array_equal(ref_out[T:], hp)

        lp2 = nrng.integers(-32768, 32768, T, dtype=np.int16)
        hp2 = nrng.integers(-32768, 32768, T, dtype=np.int16)
        rout = np.zeros(2 * T + 2, np.int16)
        (R.akoDd137UnliftH if wv == 0 else R.akoCdf53UnliftH)(wrap, T, 1, 0, fake, lp2.ctypes.data_as(I16),
                                                               hp2.ctypes.data_as(I16), rout.ctypes.data_as(I16))
        ev, od = np.zeros(T, np.int16), np.zeros(T, np.int16)
        L.orcUnlift1d(wv, wrap, T, lp2.ctypes.data_as(I16), 1, hp2.ctypes.data_as(I16), 1, ev.ctypes.data_as(I16), 1,
                      od.ctypes.data_as(I16), 1)
        mine = np.zeros(2 * T, np.int16)
        mine[0::2], mine[1::2] = ev, od
        n = 2 * T - fake
        assert np.array_equal(rout[:n], mine[:n])


def test_quant_tables_random(po):
    R, L = po.ref(), po.lib()
    rng = random.Random(17)
    for _ in range(2000):
        tw, th = rng.randrange(3, 20000), rng.randrange(3, 20000)
        f, m = rng.choice([0, 1, 2, 5, 16, 31, 100, 1000, 8192]), rng.choice([1, 2, 3, 9])
        w, h = tw, th
        while w > 2 and h > 2:
            assert R.akoQuantization(f, m, tw, th, w, h) == L.orcQuantStep(f, m, tw, th, w, h)
            assert R.akoGate(f, m, tw, th, w, h) == L.orcGateStep(f, m, tw, th, w, h)
            w, h = (w + 1) // 2, (h + 1) // 2


def test_kagari_bitstreams(po):
    R, L = po.ref(), po.lib()
    rng = random.Random(23)
    nrng = np.random.default_rng(4)
    V = C.c_void_p
    for _ in range(200):
        n = rng.randrange(1, 5000)
        kind = rng.randrange(3)
        if kind == 0:
            v = nrng.integers(-5, 6, n, dtype=np.int16)
        elif kind == 1:
            v = np.repeat(nrng.integers(-300, 300, n // 7 + 1, dtype=np.int16), 7)[:n].copy()
        else:
            v = np.zeros(n, np.int16)
            v[::max(1, n // 5)] = 77
        cap = n * 2 + 64
        o1, o2 = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        a = R.akoKagariEncode(n * 2, cap, v.ctypes.data_as(V), o1.ctypes.data_as(V))
        b = L.orcKagariEncode(n * 2, cap, v.ctypes.data_as(V), o2.ctypes.data_as(V))
        assert a == b and np.array_equal(o1[:a], o2[:b])
        if a:
            d = np.zeros(n, np.int16)
            assert L.orcKagariDecode(n, a, n * 2, o1.ctypes.data_as(V), d.ctypes.data_as(V)) == a
            assert np.array_equal(d, v)


def test_kagari_decoder_on_damaged_payloads(po):
    """The oracle's decoder against the reference's on bit flips, noise, truncation, trailing garbage and a
    wrong expected count: same 'bytes consumed' (the figure compression.c:69 compares with the block size)
    and the same values when it is non-zero."""
    R, L = po.ref(), po.lib()
    V = C.c_void_p
    for lib, name in ((R, "akoKagariDecode"), (L, "orcKagariDecode")):
        getattr(lib, name).restype = C.c_size_t
        getattr(lib, name).argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, V, V]
    R.akoKagariEncode.restype = C.c_size_t
    R.akoKagariEncode.argtypes = [C.c_size_t, C.c_size_t, V, V]
    rng = np.random.default_rng(77)
    accepted = 0
    for trial in range(400):
        n = int(rng.integers(2, 1500))
        kind = trial % 3
        if kind == 0:
            v = np.where(rng.random(n) < 0.85, 0, rng.integers(-900, 900, n))
        elif kind == 1:
            v = np.repeat(rng.integers(-20, 20, (n + 5) // 6), 6)[:n]
        else:
            v = rng.integers(-4, 5, n)
        v = np.ascontiguousarray(v.astype(np.int16))
        packed = np.zeros(8 * n + 64, dtype=np.uint8)
        size = R.akoKagariEncode(n * 2, packed.size, v.ctypes.data_as(V), packed.ctypes.data_as(V))
        assert size > 0
        for damage in range(6):
            bad = packed[:size].copy()
            length = size
            if damage < 3:
                bad[int(rng.integers(0, size))] ^= 1 << int(rng.integers(0, 8))
            elif damage == 3:
                at = int(rng.integers(0, size))
                bad[at:at + 3] = rng.integers(0, 256, bad[at:at + 3].size, dtype=np.uint8)
            elif damage == 4:
                length = int(rng.integers(1, size + 1))
            else:
                bad = np.concatenate([bad, rng.integers(0, 256, 5, dtype=np.uint8)])
                length = bad.size
            bad = np.ascontiguousarray(np.concatenate([bad, np.zeros(16, np.uint8)]))  # slack behind the payload
            for values in (n, n + 1, max(1, n - 1)):
                cap = 2 * values + 2 * 64
                o1 = np.zeros(cap // 2 + 8, dtype=np.int16)
                o2 = np.zeros(cap // 2 + 8, dtype=np.int16)
                a = R.akoKagariDecode(values, length, cap, bad.ctypes.data_as(V), o1.ctypes.data_as(V))
                b = L.orcKagariDecode(values, length, cap, bad.ctypes.data_as(V), o2.ctypes.data_as(V))
                assert a == b, (trial, damage, values, a, b)
                if a:
                    accepted += 1
                    assert np.array_equal(o1[:values], o2[:values])
    assert accepted > 50


def test_private_lift_and_unlift_of_one_plane(po):
    """The reference's private akoLift / akoUnlift (library/lifting.c:171,295) called on ONE int16 plane the way
    library/encode.c:132-148 and library/decode.c:183-187 call them: the restatement's lift_plane must give the same
    stream, and akoUnlift must give the plane back.  (bench.py times exactly these two calls as the CPU baseline of the
    lifting-only workload, BASELINE.json configs[1].)"""
    for (w, h) in [(64, 64), (100, 75), (257, 131), (640, 360)]:
        for wavelet in (0, 1, 2):
            for wrap in (0, 1, 2, 3):
                plane = po.gen_plane(w * h, seed=w * 131 + h + wavelet).reshape(h, w)
                ours = po.lift_plane(wavelet, wrap, plane)
                theirs = po.ref_lift_plane(wavelet, wrap, plane)
                assert np.array_equal(ours, theirs), (w, h, wavelet, wrap)
                assert np.array_equal(po.ref_unlift_plane(wavelet, wrap, w, h, theirs), plane), (w, h, wavelet, wrap)
                assert np.array_equal(po.unlift_plane(wavelet, wrap, w, h, theirs), plane), (w, h, wavelet, wrap)
