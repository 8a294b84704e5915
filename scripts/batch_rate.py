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
        print(f"lanes {lanes}: encode {px / (t1 - t0) / 1e9:.2f} Gpx/s ({4 * px / (t1 - t0) / 1e9:.1f} GB/s of pixels), "
              f"decode {px / (t3 - t2) / 1e9:.2f} Gpx/s", flush=True)
for i in (0, 1, n // 2, n - 1):
    assert np.array_equal(blobs[i], api.encode(imgs[i], s)), i
t0 = time.perf_counter()
for im in imgs[:16]:
    api.encode(im, s)
t1 = time.perf_counter()
print(f"loop of akoEncodeExt, one thread: {w * h * 16 / (t1 - t0) / 1e9:.2f} Gpx/s; blobs identical: yes; total blob bytes {sum(b.size for b in blobs)}")
