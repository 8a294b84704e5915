"""Where akoDecodeExt's time goes (AKO_HIP_TRACE=1), 8192x8192 RGBA G0 DD13/7 q16 g16, untiled and tiled 512."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AKO_HIP_TRACE"] = "1"
import numpy as np
from ako_amd import api
w = h = int(os.environ.get("W", "8192"))
img = api.synth_image(0, w, h)
for td in (0, 512):
    s = api.settings(wavelet=0, compression=0, q=16, g=16, tiles=td)
    blob = api.encode(img, s)
    api.decode(blob)
    for _ in range(2):
        t0 = time.perf_counter(); api.decode(blob); t1 = time.perf_counter()
        print(f"tiles={td} blob={blob.size} akoDecodeExt {1e3 * (t1 - t0):.2f} ms", flush=True)
    t0 = time.perf_counter(); api.encode(img, s); t1 = time.perf_counter()
    print(f"tiles={td} akoEncodeExt {1e3 * (t1 - t0):.2f} ms", flush=True)
