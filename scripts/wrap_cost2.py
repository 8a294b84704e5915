"""CLAMP against REPEAT in ONE process, alternating (the process-to-process state of DESIGN.md 5.0 cannot come between them): kernel
times per level of the default workload.  REPEAT strips run no left / right border code (they wrap their load addresses), so the
difference is what the border bodies cost a launch -- level 1 and 2 run as one round of waves, whose slowest wave ends the launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import defaultdict
from ako_amd import api
from oracle import pyoracle as po
w = h = 8192
img = po.gen_image(0, w, h)
d = torch.from_numpy(img).cuda().reshape(1, h, w, 4)
plans = {}
for wrap, name in ((0, "CLAMP"), (2, "REPEAT"), (3, "ZERO")):
    p = api.Plan(api.settings(wavelet=0, wrap=wrap, compression=2, q=16, g=16), 4, w, h)
    plans[name] = (p, p.new_streams(), p.new_images())
for rnd in range(3):
    for name, (p, st, back) in plans.items():
        for _ in range(3):
            p.encode(d, st); p.decode(st, back)
        p.synchronize(); p.set_profiling(True)
        for _ in range(10):
            p.encode(d, st); p.decode(st, back)
        p.synchronize()
        agg = defaultdict(list)
        for r in p.kernel_records(False) + p.kernel_records(True):
            agg[(r["name"], r["level"])].append(r["ms"])
        p.set_profiling(False)
        keys = [k for k in sorted(agg, key=lambda k: (k[1], k[0])) if k[1] <= 3 and "exact" not in k[0]]
        print(f"{name:7s}", " ".join(f"{k[0].replace('_stream_dd137', '')}:{k[1]} {sum(agg[k]) / len(agg[k]) * 1000:.1f}" for k in keys), flush=True)
