"""Long random parity run on the GPU box: akoEncodeExt / akoDecodeExt against the oracle on shapes, settings and tuning
knobs drawn at random (a superset of what tests/test_hip_parity.py fixes).  usage: python scripts/fuzz_parity.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ako_amd import api
from oracle import pyoracle as po

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
print("seed", seed, flush=True)
t_end = time.time() + budget
done = 0
knobs = ["AKO_HIP_PATH", "AKO_HIP_TAIL", "AKO_HIP_OPT", "AKO_HIP_LOCKSTEP", "AKO_HIP_FUSE2", "AKO_HIP_INV_PAIRS", "AKO_HIP_FWD_PAIRS",
         "AKO_HIP_TAIL_MANY", "AKO_HIP_WIDE", "AKO_HIP_STAGED", "AKO_HIP_DEEP", "AKO_KAGARI_THREADS", "AKO_KAGARI_PAR_MIN",
         "AKO_HIP_GROUP", "AKO_HIP_GROUP_MIN", "AKO_HIP_SEG_ROWS", "AKO_HIP_LEAN", "AKO_HIP_PACK", "AKO_HIP_ROW_STRIPS"]
while time.time() < t_end:
    for k in knobs:
        os.environ.pop(k, None)
    if rng.random() < 0.6:
        os.environ["AKO_HIP_PATH"] = str(rng.choice(["auto", "stream", "generic"]))
        os.environ["AKO_HIP_LOCKSTEP"] = str(rng.integers(0, 4))
        os.environ["AKO_HIP_FUSE2"] = str(rng.integers(0, 4))
        os.environ["AKO_HIP_GROUP"] = str(rng.integers(0, 2))
        os.environ["AKO_HIP_GROUP_MIN"] = str(rng.choice([64, 128, 1024]))
        os.environ["AKO_HIP_STAGED"] = str(rng.choice([0, 1, 1, 2]))
        if rng.random() < 0.3:
            os.environ["AKO_HIP_SEG_ROWS"] = str(rng.choice([2, 6, 7, 12, 40]))
        os.environ["AKO_HIP_INV_PAIRS"] = str(rng.choice([1, 2, 4]))
        os.environ["AKO_HIP_FWD_PAIRS"] = str(rng.choice([1, 2, 4]))
        os.environ["AKO_HIP_TAIL_MANY"] = str(rng.choice([4, 8, 16, 32, 64]))
        os.environ["AKO_HIP_OPT"] = str(rng.integers(0, 2))
        os.environ["AKO_HIP_DEEP"] = str(rng.integers(0, 2))
        os.environ["AKO_HIP_LEAN"] = str(rng.choice([0, 1, 1]))
        os.environ["AKO_HIP_PACK"] = str(rng.choice([0, 1, 1]))
        os.environ["AKO_HIP_ROW_STRIPS"] = str(rng.choice([0, 1, 1]))
        os.environ["AKO_KAGARI_THREADS"] = str(rng.choice([1, 4, 16]))
        os.environ["AKO_KAGARI_PAR_MIN"] = str(rng.choice([256, 4096, 131072]))
        if rng.random() < 0.3:
            os.environ["AKO_HIP_TAIL"] = str(rng.integers(0, 2))
    big = rng.random() < 0.25
    w = int(rng.integers(3, 2600 if big else 400)); h = int(rng.integers(3, 1800 if big else 400))
    if rng.random() < 0.35:  # the column-group kernel wants multiples of 128, the native RGB kernels multiples of 4
        w = int(rng.choice([128, 256, 384, 512, 896, 1024, 1152, 2048, 2560])) if rng.random() < 0.6 else (w + 3) // 4 * 4
    ch = int(rng.choice([1, 2, 3, 4, 4, 4, 5]))
    wavelet = int(rng.choice([0, 0, 1, 2, 3])); wrap = int(rng.integers(0, 4)); color = int(rng.integers(0, 4))
    tiles = int(rng.choice([0, 0, 0, 8, 16, 32, 64, 128, 256, 512]))
    q = int(rng.choice([0, 0, 1, 3, 16, 40, 200])); g = int(rng.choice([0, 0, 2, 16, 100]))
    comp = int(rng.choice([0, 2]))
    if wavelet == 3:
        comp = 2  # wavelet NONE under Kagari: the reference codes past the planes and cannot decode its own output
    discard = int(rng.integers(0, 2)); chroma = int(rng.integers(0, 3))
    kind = rng.integers(0, 3)
    if kind == 0 and ch == 4:
        img = po.gen_image(int(rng.integers(0, 2)), w, h, int(rng.integers(1, 1 << 31)))
    elif kind == 1:
        img = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
    else:
        img = (rng.integers(0, 4, (h, w, ch)) * 85).astype(np.uint8)
    so = po.settings(wavelet=wavelet, color=color, wrap=wrap, compression=comp, tiles=tiles, q=q, g=g, chroma_loss=chroma, discard=discard)
    want, st = po.encode_image(so, img)
    sa = api.Settings(so.wavelet, so.color, so.wrap, so.compression, so.tiles_dimension, so.quantization, so.gate, so.chroma_loss, so.discard_non_visible)
    tag = (w, h, ch, wavelet, wrap, color, tiles, q, g, comp, discard, chroma, {k: os.environ.get(k) for k in knobs if k in os.environ})
    try:
        got = api.encode(img, sa)
        ok = (st == 0)
    except api.AkoError as e:
        got, ok = None, False
        if st == 0 and not (min(w, h) <= 2 or (tiles and min(w % tiles or tiles, h % tiles or tiles) <= 2)):
            # the oracle encoded it: the product must too, except the documented refusals (extent <= 2; a Kagari tile that does not shrink fails in both)
            print("ENCODE FAILED where the oracle succeeded", tag, e, flush=True); sys.exit(1)
    if ok:
        if not np.array_equal(got, want):
            print("BLOB MISMATCH", tag, flush=True); sys.exit(1)
        back, _ = api.decode(got)
        wantpx, _, _ = po.decode_image(want)
        if not np.array_equal(np.asarray(back).reshape(wantpx.shape), wantpx):
            print("PIXEL MISMATCH", tag, flush=True); sys.exit(1)
        del back
    done += 1
    if done % 200 == 0:
        print(done, "cases", flush=True)
print("ok:", done, "cases, seed", seed)
