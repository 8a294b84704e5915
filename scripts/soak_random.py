"""One-off soak: random images / settings through akoEncodeExt + akoDecodeExt and through the device C-ABI,
against the oracle.  Usage: soak_random.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ako_amd import api
from oracle import pyoracle as po
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
nrng = np.random.default_rng(rng.randrange(1 << 30))
bad = 0
for case in range(n_cases):
    w = rng.choice([rng.randrange(3, 80), rng.randrange(80, 400), rng.randrange(400, 1500), rng.randrange(1500, 3000)])
    h = rng.choice([rng.randrange(3, 40), rng.randrange(40, 200), rng.randrange(200, 400)])
    ch = rng.choice([1, 2, 3, 4, 4, 4, 5])
    td = rng.choice([0, 0, 0, 8, 16, 32, 64, 128, 256])
    wavelet, wrap, color = rng.randrange(4), rng.randrange(4), rng.randrange(3)
    q, g = rng.choice([0, 0, 1, 7, 16, 60]), rng.choice([0, 0, 5, 16])
    comp = rng.choice([0, 2, 2])
    if wavelet == 3 and comp != 2:
        comp = 2  # reference bug: NONE + Kagari codes akoTileDataSize bytes, i.e. past the planes
    kind = rng.randrange(3)
    if kind == 0:
        img = nrng.integers(0, 256, (h, w, ch), dtype=np.uint8)
    elif kind == 1:
        img = (po.gen_image(0, w, h)[:, :, [0, 1, 2, 3, 0][:ch]]).copy()
    else:
        img = np.zeros((h, w, ch), np.uint8); img[h // 3:, w // 2:] = 200
    s = po.settings(wavelet=wavelet, wrap=wrap, color=color, compression=comp, q=q, g=g, tiles=td,
                    chroma_loss=rng.choice([0, 1, 3]), discard=rng.randrange(2))
    ob, st = po.encode_image(s, img)
    try:
        gb = api.encode(img, api.settings(wavelet=wavelet, wrap=wrap, color=color, compression=comp, q=q, g=g, tiles=td,
                                          chroma_loss=s.chroma_loss, discard=s.discard_non_visible))
        gst = 0
    except api.AkoError as e:
        gb, gst = None, e.status
    ok = ((ob is None) == (gb is None)) and (ob is None or np.array_equal(ob, gb)) and (ob is not None or st == gst)
    if ok and ob is not None:
        od, _, _ = po.decode_image(ob)
        gd, _ = api.decode(ob)
        ok = np.array_equal(od, gd)
    if not ok:
        bad += 1
        print("MISMATCH", dict(w=w, h=h, ch=ch, td=td, wavelet=wavelet, wrap=wrap, color=color, q=q, g=g, comp=comp, kind=kind, st=st, gst=gst))
        if bad > 5:
            break
    if case % 100 == 99:
        print("cases", case + 1, "mismatches", bad, flush=True)
print("done", n_cases, "mismatches", bad)
sys.exit(1 if bad else 0)
