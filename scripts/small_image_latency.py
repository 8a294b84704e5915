"""akoEncodeExt / akoDecodeExt wall time per call for small images (plan set-up cost dominates)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ako_amd import api
from oracle import pyoracle as po
out = {}
for (w, h) in ((512, 512), (1920, 1080), (4096, 4096)):
    img = po.gen_image(0, w, h)
    s = api.settings(wavelet=api.DD137, compression=api.KAGARI, q=16, g=16)
    blob = api.encode(img, s)
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        blob = api.encode(img, s)
    t1 = time.perf_counter()
    for _ in range(n):
        back, _ = api.decode(blob)
    t2 = time.perf_counter()
    out[f"{w}x{h}"] = {"encode_ms": round((t1 - t0) / n * 1e3, 3), "decode_ms": round((t2 - t1) / n * 1e3, 3)}
print(json.dumps(out))
