"""Device entropy stage: the GPU Kagari encoder must produce, bit for bit, what the ORACLE's
restatement of library/kagari.c produces (oracle/ako_oracle.c: orcKagariEncode, pinned against the compiled
reference in tests/test_oracle_vs_ref.py) -- and so must the product's own host coder, checked alongside."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

from ako_amd import api


def host_body(plan, streams_i16: np.ndarray):
    """[uint32 size][payload] per tile with the ORACLE's encoder (the checker), or the index of the first tile that
    fails; the product's host coder (akoHostKagariEncode, same .so as the device coder) must agree with it."""
    from oracle import pyoracle as po

    po.build()
    L, O = api.lib(), po.lib()
    out = bytearray()
    raw = streams_i16.view(np.uint8)
    for t in range(plan.tiles):
        ti = plan.tile_info(t)
        off, n = ti["stream_offset"], ti["stream_bytes"]
        src = np.ascontiguousarray(raw[off:off + n])
        dst = np.zeros(n + 16, dtype=np.uint8)
        size = O.orcKagariEncode(n, n - 4, src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p))
        mine = np.zeros(n + 16, dtype=np.uint8)
        size2 = L.akoHostKagariEncode(n, n - 4, src.ctypes.data_as(C.c_void_p), mine.ctypes.data_as(C.c_void_p))
        assert size2 == size and np.array_equal(mine[:size], dst[:size]), f"host coder differs from the oracle on tile {t}"
        if size == 0:
            return None, t
        out += struct.pack("<I", size) + dst[:size].tobytes()
    return bytes(out), None


def crafted_stream(kind: str, n: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    if kind == "sparse":  # what a quantized high band looks like
        v = np.zeros(n, dtype=np.int16)
        idx = rng.integers(0, n, n // 6)
        v[idx] = rng.integers(-40, 41, idx.size)
        return v
    if kind == "runs":  # runs of every length around 1, 2, 3 and the 65534 restart
        out, left = [], n
        lengths = [1, 2, 3, 4, 65533, 65534, 65535, 65536, 65537, 2 * 65534 + 1, 2 * 65534 + 2, 3 * 65534 + 3, 7, 1, 1, 2]
        k = 0
        while left > 0:
            ln = min(left, lengths[k % len(lengths)] if rng.random() < 0.7 else int(rng.integers(1, 400)))
            out.append(np.full(ln, int(rng.integers(-3, 4)), dtype=np.int16))
            left -= ln
            k += 1
        return np.concatenate(out)
    if kind == "extremes":  # the full code range incl. -32768 (code wraps to 0) between long zero runs
        v = np.zeros(n, dtype=np.int16)
        idx = rng.integers(0, n, max(n // 400, 8))
        v[idx] = rng.choice(np.array([-32768, 32767, -32767, 1, -1, 16384, -16384, 255, -256], dtype=np.int16), idx.size)
        return v
    if kind == "zeros":
        return np.zeros(n, dtype=np.int16)
    if kind == "noise":  # does not shrink
        return rng.integers(-3000, 3000, n).astype(np.int16)
    raise KeyError(kind)


PLANS = [
    # w, h, channels, tiles
    (64, 64, 1, 0),
    (300, 200, 4, 64),
    (1000, 777, 3, 256),
    (2048, 1024, 4, 0),
    (131, 67, 2, 32),
]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,ch,td", PLANS)
@pytest.mark.parametrize("kind", ["sparse", "runs", "extremes", "zeros"])
def test_device_kagari_matches_host_encoder(w, h, ch, td, kind):
    import torch

    s = api.settings(wavelet=api.CDF53, compression=api.KAGARI, q=0, g=0, tiles=td)
    with api.Plan(s, ch, w, h) as plan:
        n = plan.stream_bytes // 2
        v = crafted_stream(kind, n, seed=w * 7 + ch)
        d = torch.from_numpy(v.view(np.uint8)).cuda().reshape(1, -1)
        want, bad = host_body(plan, v)
        assert want is not None, f"host encoder failed on tile {bad}"
        got = plan.kagari_encode(d)
        assert got.size == len(want)
        assert got.tobytes() == want
        # a second call on the same plan (buffers reused) and on different data
        v2 = crafted_stream("sparse", n, seed=99)
        want2, _ = host_body(plan, v2)
        got2 = plan.kagari_encode(torch.from_numpy(v2.view(np.uint8)).cuda().reshape(1, -1))
        assert got2.tobytes() == want2


@pytest.mark.gpu
def test_device_kagari_reports_the_tile_that_does_not_shrink():
    import torch

    s = api.settings(wavelet=api.CDF53, compression=api.KAGARI, q=0, g=0, tiles=64)
    with api.Plan(s, 4, 300, 200) as plan:
        n = plan.stream_bytes // 2
        v = crafted_stream("sparse", n, seed=5)
        t_bad = 7
        ti = plan.tile_info(t_bad)
        a, b = ti["stream_offset"] // 2, (ti["stream_offset"] + ti["stream_bytes"]) // 2
        v[a:b] = crafted_stream("noise", b - a, seed=6)
        want, bad = host_body(plan, v)
        assert want is None and bad == t_bad
        with pytest.raises(api.AkoError):
            plan.kagari_encode(torch.from_numpy(v.view(np.uint8)).cuda().reshape(1, -1))
        assert "did not shrink" in api.last_error()
        # and the plan still works afterwards
        v[a:b] = 0
        want, _ = host_body(plan, v)
        assert plan.kagari_encode(torch.from_numpy(v.view(np.uint8)).cuda().reshape(1, -1)).tobytes() == want


@pytest.mark.gpu
def test_device_kagari_borderline_payloads_follow_the_reference_rule():
    """Payload sizes right at 'tile bytes - 5' (the largest the reference accepts, kagari.c:64-112)."""
    import torch

    s = api.settings(wavelet=api.WAVELET_NONE, color=api.COLOR_NONE, compression=api.KAGARI, q=0, g=0)
    for w in (20, 21, 24, 37):
        with api.Plan(s, 1, w, 8) as plan:
            n = plan.stream_bytes // 2
            rng = np.random.default_rng(w)
            hits = {True: 0, False: 0}
            for trial in range(300):
                # values of 15 bits cost 31 bits each (does not shrink), small ones few bits: mix to land near the limit
                v = np.where(rng.random(n) < 0.5 + 0.02 * (trial % 11 - 5), rng.integers(-2, 3, n), rng.integers(8192, 16000, n)).astype(np.int16)
                want, bad = host_body(plan, v)
                d = torch.from_numpy(v.view(np.uint8)).cuda().reshape(1, -1)
                if want is None:
                    with pytest.raises(api.AkoError):
                        plan.kagari_encode(d)
                else:
                    assert plan.kagari_encode(d).tobytes() == want
                hits[want is not None] += 1
            assert hits[True] and hits[False], hits  # both sides of the rule were exercised


@pytest.mark.gpu
def test_device_kagari_on_real_streams_4096(po):
    import torch

    img = po.gen_image(0, 4096, 4096)
    s = api.settings(wavelet=api.DD137, compression=api.KAGARI, q=16, g=16)
    with api.Plan(s, 4, 4096, 4096) as plan:
        d_img = torch.from_numpy(img).cuda().reshape(1, 4096, 4096, 4)
        streams = plan.encode(d_img)
        plan.synchronize()
        host = streams.cpu().numpy().reshape(-1).view(np.int16)
        want, bad = host_body(plan, host)
        assert want is not None
        got = plan.kagari_encode(streams)
        assert got.tobytes() == want
        # SURVEY 8c anchor: the reference's blob for this input is 559 811 bytes, Adler-32 beeeebc6
        head = bytes([65, 107, 111, 2]) + struct.pack("<III", 4096, 4096, 3 | (0 << 4) | (0 << 6) | (3 << 8) | (0 << 10))
        assert 16 + got.size == 559811
        assert f"{zlib.adler32(got.tobytes(), zlib.adler32(head)) & 0xFFFFFFFF:08x}" == "beeeebc6"


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,ch,td", PLANS)
@pytest.mark.parametrize("kind", ["sparse", "runs", "zeros"])
def test_device_run_expansion_matches_host_decoder(w, h, ch, td, kind):
    """Decoder side: host tokenizer + akoHipKagariExpand rebuilds exactly the stream that was encoded."""
    import torch

    s = api.settings(wavelet=api.CDF53, compression=api.KAGARI, q=0, g=0, tiles=td)
    with api.Plan(s, ch, w, h) as plan:
        n = plan.stream_bytes // 2
        v = crafted_stream(kind, n, seed=w + 3 * ch)
        body, bad = host_body(plan, v)
        assert body is not None
        streams = plan.kagari_decode_body(np.frombuffer(body, dtype=np.uint8))
        plan.synchronize()
        assert np.array_equal(streams.cpu().numpy().reshape(-1), v)
        # and the pair encoder -> decoder on the device
        streams2 = plan.kagari_decode_body(plan.kagari_encode(torch.from_numpy(v.view(np.uint8)).cuda().reshape(1, -1)))
        assert np.array_equal(streams2.cpu().numpy().reshape(-1), v)


@pytest.mark.gpu
def test_device_run_expansion_rejects_forged_records():
    import ctypes as C

    s = api.settings(wavelet=api.CDF53, compression=api.KAGARI, q=0, g=0)
    with api.Plan(s, 1, 64, 64) as plan:
        n = plan.stream_bytes // 2
        L = api.lib()
        lit = (C.c_int16 * 8)(*range(8))
        streams = plan.new_streams()

        def expand(n_lit, runs):
            arr = (api.KagariRun * max(len(runs), 1))(*[api.KagariRun(*r, 0) for r in runs])
            return L.akoHipKagariExpand(plan._p, lit, n_lit, arr, len(runs), C.c_void_p(streams.data_ptr()), 0)

        assert expand(8, [(8, n - 8, 8)]) == 0                      # 8 literals then one long run: valid
        assert expand(8, [(8, n - 7, 8)]) != 0                      # one value too many
        assert expand(8, [(8, n - 9, 8)]) != 0                      # one too few
        assert expand(8, [(4, n - 8, 8)]) != 0                      # run placed where literals are
        assert expand(8, [(8, n - 8, 9)]) != 0                      # refers to a literal that does not exist
        assert expand(8, [(8, n - 8, 0)]) != 0
        assert expand(8, [(2, 10, 2), (2, n - 18, 2)]) != 0         # overlapping runs
        assert expand(8, [(2, 10, 2), (18, n - 18, 8)]) == 0
