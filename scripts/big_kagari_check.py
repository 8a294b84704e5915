"""One-off: device Kagari encoder + host tokenizer + device run expansion on the largest configuration
(16384x16384 RGBA, CDF5/3 lossless, 256-px tiles = 4096 tiles, 2.1 GB of coefficients)."""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ako_amd import api
from oracle import pyoracle as po
w = h = 16384
img = po.gen_image(0, w, h)
s = api.settings(wavelet=api.CDF53, compression=api.KAGARI, q=0, g=0, tiles=256)
with api.Plan(s, 4, w, h) as plan:
    d = torch.from_numpy(img).cuda().reshape(1, h, w, 4)
    st = plan.encode(d); plan.synchronize()
    t0 = time.perf_counter(); body = plan.kagari_encode(st); t1 = time.perf_counter()
    print("tiles", plan.tiles, "body bytes", body.size, "device encode + fetch s", round(t1 - t0, 3))
    t0 = time.perf_counter(); st2 = plan.kagari_decode_body(body); plan.synchronize(); t1 = time.perf_counter()
    print("tokenize + expand s", round(t1 - t0, 3), "streams equal", bool(torch.equal(st, st2)))
    back = plan.decode(st2); plan.synchronize()
    print("pixels equal", bool(torch.equal(back, d)))
    head = bytes([65, 107, 111, 2]) + int(w).to_bytes(4, "little") + int(h).to_bytes(4, "little") + int(3 | (1 << 6) | (0 << 8) | (0 << 10) | ((8 - 2) << 12)).to_bytes(4, "little")
    print("blob bytes", 16 + body.size, "adler32 %08x" % (zlib.adler32(body.tobytes(), zlib.adler32(head)) & 0xFFFFFFFF))
blob = np.concatenate([np.frombuffer(head, np.uint8), body])
t0 = time.perf_counter(); back2, _ = api.decode(blob); t1 = time.perf_counter()
print("akoDecodeExt (tiles parsed on worker threads) s", round(t1 - t0, 3), "pixels equal", bool(np.array_equal(back2, img)))
