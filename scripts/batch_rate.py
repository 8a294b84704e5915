"""Rate of the batched host API (akoHipEncodeBatch / akoHipDecodeBatch): 64 images of 3840x2160 RGBA (BASELINE
configs[3]) from host memory to .ako blobs and back, one calling thread; against the PCIe bound (4 B/px at the
pinned copy rate the bench measures, ~55 GB/s) and against a loop of akoEncodeExt calls.  Checks 4 blobs against
akoEncodeExt's bytes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ako_amd import api

w, h, n = 3840, 2160, int(os.environ.get("N", "64"))
imgs = [api.synth_image(0, w, h, seed=0x9E3779B9 + i) for i in range(n)]
s = api.settings(wavelet=api.DD137, compression=api.KAGARI, q=16, g=16)
px = w * h * n
for lanes in (4, 6, 8, 12):
    with api.Batch(s, 4, w, h, devices=[0], lanes_per_device=lanes) as b:
        b.encode(imgs[:lanes])                      # plans warm
        t0 = time.perf_counter(); blobs, st = b.encode(imgs); t1 = time.perf_counter()
        assert st == [0] * n
        decs, st = b.decode(blobs[:lanes])
        t2 = time.perf_counter(); decs, st = b.decode(blobs); t3 = time.perf_counter()
        assert st == [0] * n
        t4 = time.perf_counter(); decs, st = b.decode(blobs, outs=decs); t5 = time.perf_counter()   # the caller's buffers again
        assert st == [0] * n
        print(f"lanes {lanes}: encode {px / (t1 - t0) / 1e9:.2f} Gpx/s ({4 * px / (t1 - t0) / 1e9:.1f} GB/s of pixels), "
              f"decode into fresh buffers {px / (t3 - t2) / 1e9:.2f} Gpx/s, into the same buffers again {px / (t5 - t4) / 1e9:.2f}", flush=True)
        if lanes == 6:
            assert all(np.array_equal(d, api.decode(blobs[i])[0]) for i, d in ((0, decs[0]), (n - 1, decs[n - 1])))
            pin_in = [api.pinned_empty((h, w, 4)) for _ in range(n)]
            for a, im in zip(pin_in, imgs):
                a[...] = im
            pin_out = [api.pinned_empty((h, w, 4)) for _ in range(n)]
            b.encode(pin_in[:lanes])
            t0 = time.perf_counter(); blobs_p, st = b.encode(pin_in); t1 = time.perf_counter()
            assert st == [0] * n and all(np.array_equal(x, y) for x, y in zip(blobs_p, blobs))
            t2 = time.perf_counter(); decs_p, st = b.decode(blobs, outs=pin_out); t3 = time.perf_counter()
            assert st == [0] * n and np.array_equal(decs_p[3], decs[3])
            print(f"lanes {lanes}, images in pinned memory (akoHipHostAlloc): encode {px / (t1 - t0) / 1e9:.2f} Gpx/s, "
                  f"decode {px / (t3 - t2) / 1e9:.2f} Gpx/s", flush=True)
            del pin_in, pin_out, decs_p, blobs_p
for i in (0, 1, n // 2, n - 1):
    assert np.array_equal(blobs[i], api.encode(imgs[i], s)), i
t0 = time.perf_counter()
for im in imgs[:16]:
    api.encode(im, s)
t1 = time.perf_counter()
print(f"loop of akoEncodeExt, one thread: {w * h * 16 / (t1 - t0) / 1e9:.2f} Gpx/s; blobs identical: yes; total blob bytes {sum(b.size for b in blobs)}")
