"""End-to-end akoEncodeExt / akoDecodeExt wall time (host buffers in, .ako blob out) with the entropy stage on
the GPU (default) or on the host (AKO_HIP_KAGARI=host), plus the device Kagari encoder alone."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ako_amd import api
from oracle import pyoracle as po

w = h = int(os.environ.get("W", "8192"))
img = po.gen_image(0, w, h)
s = api.settings(wavelet=api.DD137, compression=api.KAGARI, q=16, g=16)
out = {"image": f"{w}x{h} RGBA G0, DD137 q16 g16"}
for mode in ("device", "host"):
    os.environ["AKO_HIP_KAGARI"] = mode
    api.encode(img[:64, :64].copy(), s)  # warm up the runtime
    t0 = time.perf_counter(); blob = api.encode(img, s); t1 = time.perf_counter()
    out[f"encode_{mode}_kagari_s"] = round(t1 - t0, 4)
    out["blob_bytes"] = int(blob.size)
ref = None
for mode in ("device", "host"):
    os.environ["AKO_HIP_KAGARI"] = mode
    t0 = time.perf_counter(); back, _ = api.decode(blob); t1 = time.perf_counter()
    out[f"decode_{mode}_kagari_s"] = round(t1 - t0, 4)
    assert ref is None or np.array_equal(ref, back)
    ref = back
os.environ["AKO_HIP_KAGARI"] = "device"
with api.Plan(s, 4, w, h) as plan:
    d = torch.from_numpy(img).cuda().reshape(1, h, w, 4)
    st = plan.encode(d); plan.synchronize()
    plan.kagari_encode(st, fetch=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        n = plan.kagari_encode(st, fetch=False)
    plan.synchronize(); t1 = time.perf_counter()
    out["device_kagari_ms"] = round((t1 - t0) / 5 * 1e3, 3)
    out["device_kagari_Mpx_s"] = round(w * h / ((t1 - t0) / 5) / 1e6, 1)
    out["body_bytes"] = int(n)
print(json.dumps(out))
