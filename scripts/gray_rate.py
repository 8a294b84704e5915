"""One- and two-channel 8192 x 8192 images, DD13/7 q16 g16: encode + decode rate on the native gray kernels (default) against the
staged route (AKO_HIP_STAGED=2), same box, alternating.  usage (GPU box): python3 scripts/gray_rate.py"""
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, statistics
sys.path.insert(0, %r)
import numpy as np, torch
from ako_amd import api
from oracle import pyoracle as po
ch = int(sys.argv[1]); w = h = 8192
img = np.ascontiguousarray(po.gen_image(0, w, h)[:, :, :ch])
s = api.settings(wavelet=0, compression=2, q=16, g=16)
with api.Plan(s, ch, w, h) as plan:
    d = torch.from_numpy(img).cuda().reshape(1, h, w, ch)
    st = plan.new_streams(); back = plan.new_images()
    for _ in range(3):
        plan.encode(d, st); plan.decode(st, back)
    plan.synchronize()
    t = []
    for rep in range(7):
        t0 = time.perf_counter()
        for i in range(10):
            plan.encode(d, st); plan.decode(st, back)
        plan.synchronize()
        t.append(time.perf_counter() - t0)
    plan.set_profiling(True)
    plan.encode(d, st); plan.decode(st, back); plan.synchronize()
    names = sorted({r["name"] for r in plan.kernel_records(False) + plan.kernel_records(True) if r["level"] == 0})
    print("RESULT %%.1f Gpx/s  level-0 kernels: %%s" %% (w * h * 10 / statistics.median(t) / 1e9, ", ".join(names)))
''' % ROOT
for rnd in range(2):
    for ch in (1, 2):
        for staged in ("1", "2"):
            env = dict(os.environ, AKO_HIP_STAGED=staged)
            r = subprocess.run([sys.executable, "-c", CHILD, str(ch)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            print(f"channels {ch}  {'native' if staged == '1' else 'staged'}: ", line[-1][7:] if line else ("failed: " + r.stderr[-300:]), flush=True)
