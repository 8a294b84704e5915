"""16384 x 16384 RGBA in 256-pixel tiles, CDF5/3 lossless, wrap REPEAT: whole step with the small levels packed (AKO_HIP_PACK=1,
REPEAT included since round 4) and one tile per wave (AKO_HIP_PACK=0).  usage: python3 scripts/repeat_pack_rate.py [tiles]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, statistics
sys.path.insert(0, %r)
import numpy as np, torch
from ako_amd import api
w = h = 16384
tiles = int(sys.argv[1])
s = api.settings(wavelet=api.CDF53, wrap=int(sys.argv[2]), compression=2, q=0, g=0, tiles=tiles)
with api.Plan(s, 4, w, h) as plan:
    d = torch.randint(0, 256, (1, h, w, 4), dtype=torch.uint8, device="cuda")
    st, back = plan.new_streams(), plan.new_images()
    for _ in range(3):
        plan.encode(d, st); plan.decode(st, back)
    plan.synchronize()
    assert torch.equal(d, back)
    t = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            plan.encode(d, st); plan.decode(st, back)
        plan.synchronize()
        t.append(time.perf_counter() - t0)
    print("RESULT %%.1f" %% (w * h * 10 / statistics.median(t) / 1e9))
''' % ROOT
tiles = sys.argv[1] if len(sys.argv) > 1 else "256"
for rnd in range(2):
    for wrap, name in ((0, "CLAMP"), (2, "REPEAT")):
        for pack in ("1", "0"):
            r = subprocess.run([sys.executable, "-c", CHILD, tiles, str(wrap)], env=dict(os.environ, AKO_HIP_PACK=pack), capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            print(f"tiles {tiles} {name:6s} AKO_HIP_PACK={pack}: {line[-1][7:] if line else 'failed: ' + r.stderr[-300:]} Gpx/s (one step in flight)", flush=True)
