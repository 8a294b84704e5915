"""GPU: the batched host API (include/ako_hip.h akoHipBatch*) and the multi-device routes.

Every blob a batch call returns must be byte-identical to what akoEncodeExt returns for the same image, which in turn is
the oracle's blob; decoded images likewise.  The 1-GPU test box runs the multi-device logic with a device list that
names device 0 more than once (bands / lanes are then separate plans and streams on the same GPU)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from ako_amd import api  # noqa: E402


def test_encode_batch_equals_per_image_encode_and_the_oracle(po):
    w, h, n = 640, 360, 24
    imgs = [po.gen_image(0, w, h, seed=0x9E3779B9 + i) for i in range(n)]   # configs[3] seeding rule
    for (comp, q, g, td) in [(api.KAGARI, 16, 16, 0), (api.COMPRESSION_NONE, 16, 16, 0), (api.KAGARI, 0, 0, 128)]:
        s = api.settings(wavelet=api.DD137 if q else api.CDF53, compression=comp, q=q, g=g, tiles=td)
        with api.Batch(s, 4, w, h, devices=[0, 0], lanes_per_device=2) as b:
            assert b.lanes == 4
            blobs, st = b.encode(imgs)
            assert st == [0] * n
            for i in (0, 7, n - 1):
                want, ost = po.encode_image(po.settings(wavelet=s.wavelet, compression=comp, q=q, g=g, tiles=td), imgs[i])
                assert ost == 0 and np.array_equal(blobs[i], want), (comp, q, td, i)
            for i in range(n):
                assert np.array_equal(blobs[i], api.encode(imgs[i], s)), (comp, q, td, i)
            # and back
            decs, st = b.decode(blobs)
            assert st == [0] * n
            for i in range(n):
                want, _, _ = po.decode_image(blobs[i])
                assert np.array_equal(decs[i], want), (comp, q, td, i)


def test_batch_with_pinned_and_reused_buffers(po):
    """Images in page-locked memory (akoHipHostAlloc) are used in place by the lanes, pageable ones go through the
    lanes' staging: same blobs, same pixels either way, also when the output buffers of an earlier call are handed
    in again; mixed lists work."""
    w, h, n = 1024, 544, 12          # 2.2 MB images: the decode path also takes its huge-page advice branch
    imgs = [po.gen_image(0, w, h, seed=77 + i) for i in range(n)]
    s = api.settings(wavelet=api.DD137, compression=api.KAGARI, q=9, g=4)
    assert api.lib().akoHipHostIsPinned(imgs[0].ctypes.data) == 0
    with api.Batch(s, 4, w, h, devices=[0], lanes_per_device=3) as b:
        blobs, st = b.encode(imgs)
        assert st == [0] * n
        pin_in = [api.pinned_empty((h, w, 4)) for _ in range(n)]
        assert api.lib().akoHipHostIsPinned(pin_in[0].ctypes.data) == 1
        for a, im in zip(pin_in, imgs):
            a[...] = im
        mixed = [pin_in[i] if i % 2 else imgs[i] for i in range(n)]
        blobs_p, st = b.encode(mixed)
        assert st == [0] * n and all(np.array_equal(x, y) for x, y in zip(blobs_p, blobs))
        want = [po.decode_image(bl)[0] for bl in blobs]
        decs, st = b.decode(blobs)
        assert st == [0] * n and all(np.array_equal(x, y) for x, y in zip(decs, want))
        for d in decs:
            d[...] = 0
        again, st = b.decode(blobs, outs=decs)                       # the same pageable buffers
        assert st == [0] * n and all(x is y for x, y in zip(again, decs)) and all(np.array_equal(x, y) for x, y in zip(again, want))
        outs = [api.pinned_empty((h, w, 4)) if i % 2 else np.zeros((h, w, 4), np.uint8) for i in range(n)]
        got, st = b.decode(blobs, outs=outs)                         # pinned and pageable, mixed
        assert st == [0] * n and all(np.array_equal(x, y) for x, y in zip(got, want))


def test_batch_reports_per_image_failures(po):
    import ctypes as C

    w, h = 256, 128
    s = api.settings(wavelet=api.CDF53, compression=api.KAGARI, q=0, g=0)
    good = po.gen_image(0, w, h)
    want, ost = po.encode_image(po.settings(wavelet=1, compression=0, q=0, g=0), good)
    assert ost == 0
    with api.Batch(s, 4, w, h) as b:
        # encode: a missing image fails alone, its neighbours come through
        ptrs = (C.c_void_p * 3)(good.ctypes.data, None, good.ctypes.data)
        out, sizes, st = (C.c_void_p * 3)(), (C.c_size_t * 3)(), (C.c_int * 3)()
        rc = api.lib().akoHipEncodeBatch(b._b, 3, ptrs, out, sizes, st)
        assert rc != 0 and list(st) == [0, 9, 0] and not out[1] and sizes[1] == 0     # 9 = AKO_INVALID_INPUT
        for i in (0, 2):
            got = np.ctypeslib.as_array(C.cast(out[i], C.POINTER(C.c_uint8)), shape=(sizes[i],)).copy()
            api.lib().akoDefaultFree(out[i])
            assert np.array_equal(got, want)
        # decode: a truncated blob and a blob of another shape fail alone
        other = api.encode(po.gen_image(0, 64, 64), s)
        decs, st = b.decode([want, want[:200], other, want])
        assert st[0] == 0 and st[1] == 15 and st[2] != 0 and st[3] == 0
        assert np.array_equal(decs[0], good) and np.array_equal(decs[3], good) and decs[1] is None


def test_tiled_image_over_several_devices_matches_the_single_device_blob(po, golden_sums):
    """akoEncodeExt / akoDecodeExt of a TILED image split over the devices named by AKO_HIP_DEVICES
    (bands of whole tile rows, library/encode.c:115-205; the host joins the bodies in tile order).  Device 0 is named
    once per band here, so the band logic runs on a one-GPU box too; with more GPUs visible the second band goes to
    device 1."""
    ndev = api.device_count()
    second = 1 if ndev > 1 else 0
    old = os.environ.get("AKO_HIP_DEVICES")
    try:
        for (w, h, ch, td, wavelet, q, comp) in [(1000, 777, 4, 256, 0, 16, 2), (1000, 777, 4, 256, 0, 16, 0), (300, 200, 3, 64, 1, 0, 0),
                                                 (517, 1031, 4, 128, 0, 16, 0), (96, 64, 4, 64, 2, 0, 2)]:
            img = np.ascontiguousarray(po.gen_image(1 if comp == 2 else 0, w, h)[:, :, :ch])
            want, ost = po.encode_image(po.settings(wavelet=wavelet, compression=comp, q=q, g=q, tiles=td), img)
            assert ost == 0
            dec_want, _, _ = po.decode_image(want)
            for devs in (f"0,{second}", f"0,{second},0", "0"):
                os.environ["AKO_HIP_DEVICES"] = devs
                s = api.settings(wavelet=wavelet, compression=comp, q=q, g=q, tiles=td)
                events = []
                blob = api.encode(img, s, events=lambda t, n, e: events.append((t, n, e)))
                assert np.array_equal(blob, want), (w, h, td, devs)
                tiles = ((w + td - 1) // td) * ((h + td - 1) // td)
                assert [e[0] for e in events if e[2] == 1] == list(range(tiles))   # FORMAT_START of every tile, in order
                dec, _ = api.decode(blob)
                assert np.array_equal(dec, dec_want), (w, h, td, devs)
        # the anchor of SURVEY 8c: G1 1000x777 DD137 q16 g16 tiles 256, blob adler 370428a2
        os.environ["AKO_HIP_DEVICES"] = f"0,{second}"
        blob = api.encode(po.gen_image(1, 1000, 777), api.settings(wavelet=0, compression=2, q=16, g=16, tiles=256))
        assert blob.size == 6231936 and f"{po.adler32(blob):08x}" == "370428a2"
    finally:
        if old is None:
            os.environ.pop("AKO_HIP_DEVICES", None)
        else:
            os.environ["AKO_HIP_DEVICES"] = old


def _search_by_reencoding(img, ratio, base):
    """tools/akoenc.cpp:130-214 step for step, every candidate a full akoEncodeExt (what the tools did before N4)."""
    def run(q):
        s = base.copy()
        s.quantization = q
        try:
            return api.encode(img, s)
        except api.AkoError:
            return None
    size = lambda b: 0 if b is None else b.size
    target = img.size // ratio
    margin = target * 4 // 100
    runs = 1
    last = run(0)
    ceil_size = floor_size = size(last)
    ceil_q = floor_q = 0
    q = 1
    while True:
        q *= 4
        ceil_size, ceil_q = floor_size, floor_q
        last = run(q)
        runs += 1
        floor_size, floor_q = size(last), q
        if not floor_size > target:
            break
    last_size = floor_size
    while abs(floor_size - ceil_size) > margin and abs(floor_q - ceil_q) > 1:
        q = (ceil_q + floor_q) // 2
        last = run(q)
        runs += 1
        last_size = size(last)
        if last_size > target:
            ceil_size, ceil_q = last_size, q
        else:
            floor_size, floor_q = last_size, q
    take_floor = abs(floor_size - target) < abs(ceil_size - target)
    best_q, best_size = (floor_q, floor_size) if take_floor else (ceil_q, ceil_size)
    if last_size != best_size:
        last = run(best_q)
        runs += 1
    return last, runs


def test_ratio_search_on_one_transform_gives_the_blob_of_repeated_encodes(po, golden_sums):
    """akoEncodeRatioExt transforms once per colour transformation and re-quantizes per candidate; the
    blob (and therefore the quantization it settles on) must be the one the search by repeated akoEncodeExt calls
    ends with, and the golden `ratio20` file of the reference's own akoenc."""
    import zlib

    img512 = po.gen_image(0, 512, 512)
    blob, q, encodes, transforms = api.encode_ratio(img512, 20, api.default_settings())
    assert blob.size == 51597 and f"{zlib.adler32(blob.tobytes()) & 0xFFFFFFFF:08x}" == "a6c44283"   # tests/golden/cli.json ratio20
    assert transforms == 2 and encodes >= 4     # quantization 0 keeps plain YCoCg, every other candidate YCoCg_Q
    cases = [(img512, 20, api.default_settings()),
             (img512, 8, api.settings(wavelet=api.CDF53)),
             (img512, 12, api.settings(g=8)),                                    # gate > 0: one colour, ONE transform
             (img512, 30, api.settings(color=api.SUBTRACT_G, tiles=128)),        # tiled, colour independent of q
             (np.ascontiguousarray(po.gen_image(0, 300, 200)[:, :, :3]), 10, api.settings(wrap=api.MIRROR, chroma_loss=2)),
             (np.ascontiguousarray(po.gen_image(0, 97, 131)[:, :, :1]), 6, api.settings(wavelet=api.HAAR))]
    for (img, ratio, base) in cases:
        blob, q, encodes, transforms = api.encode_ratio(img, ratio, base)
        want, runs = _search_by_reencoding(img, ratio, base)
        assert want is not None and np.array_equal(blob, want), (img.shape, ratio)
        assert encodes == runs
        one_colour = base.color != api.YCOCG or base.gate > 0 or img.shape[2] < 3
        assert transforms == (1 if one_colour else 2), (img.shape, ratio, transforms)
        # and it is the oracle's blob for that quantization
        s = po.settings(wavelet=base.wavelet, color=base.color, wrap=base.wrap, compression=0, tiles=base.tiles_dimension,
                        q=q, g=base.gate, chroma_loss=base.chroma_loss)
        ob, ost = po.encode_image(s, img)
        assert ost == 0 and np.array_equal(blob, ob), (img.shape, ratio, q)
    # no search for ratio 0 / 1 (tools/akoenc.cpp:116-128)
    b0, q0, e0, _ = api.encode_ratio(img512, 0, api.default_settings())
    assert np.array_equal(b0, api.encode(img512, api.default_settings())) and e0 == 1 and q0 == 16
    b1, q1, _, _ = api.encode_ratio(img512, 1, api.default_settings())
    assert np.array_equal(b1, api.encode(img512, api.settings(q=0, g=0))) and q1 == 0


@pytest.mark.gpu
def test_bench_host_routes_rehearsed_on_one_gpu():
    """`bench.py --route lanes` (akoHipBatch lanes over devices, configs[3]) and `--route bands` (one tiled image cut over
    AKO_HIP_DEVICES, configs[4]) print a line with per-device busy time and the slowest device.  Rehearsed on the one GPU of
    the box with device 0 named twice: the devices then share one memory system and one link, so the two-"device" figure
    of the lanes route must agree with its one-device figure (not fall below 0.8 of it: the lanes of both halves compete
    for the same link), and every band / lane must have been used."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(*a, env=None):
        e = dict(os.environ, **(env or {}))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *a], capture_output=True, text=True, env=e, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])

    one = run("--workload", "batch4k", "--route", "lanes", "--route-devices", "0", "--steps", "2")
    two = run("--workload", "batch4k", "--route", "lanes", "--route-devices", "0,0", "--steps", "2")
    for line, n_lanes in ((one, 8), (two, 16)):
        assert line["config"]["route"] == "lanes" and line["lanes"] == n_lanes
        pd = line["per_device"]["0"]
        assert pd["images"] == 64 and pd["lanes"] == n_lanes and pd["encode_busy_s"] > 0 and pd["decode_busy_s"] > 0
        assert line["slowest_device"] == 0
    # (structure only: each side is a two-step run on a shared box, the routes are bound by the host link, and boxes differ by
    # +-6 %: rate comparisons belong to profiles/, not to an assertion)
    assert two["value"] > 0 and one["value"] > 0

    env = {"AKO_BENCH_TILES": "512"}
    one = run("--workload", "tiles16k", "--route", "bands", "--route-devices", "0", "--steps", "1", env=env)
    two = run("--workload", "tiles16k", "--route", "bands", "--route-devices", "0,0", "--steps", "1", env=env)
    assert one["bands"] == 0 and two["bands"] == 2  # one device: the call is not split
    assert two["per_device"]["0"]["rows"] == 16384 and two["per_device"]["0"]["bands"] == 2
    assert two["per_device"]["0"]["encode_busy_s"] > 0 and two["per_device"]["0"]["decode_busy_s"] > 0
    # the band route keeps its plans from call to call (the warm-up call created them): no timed call creates one
    assert two["band_plans_created_per_timed_call"] == {"encode": [0], "decode": [0]}, two["band_plans_created_per_timed_call"]
    # (no rate comparison here: a split call builds a plan per band and parses on fresh threads, which on ONE GPU costs more
    # than the split saves -- 148 against 459 Mpx/s)
    assert two["value"] > 0 and one["value"] > 0
