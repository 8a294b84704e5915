"""Kernel table of the default workload (8192 x 8192 RGBA, DD13/7 q16 g16) under each border rule: REPEAT strips run no left / right
border code at all (they wrap their load addresses), so the difference to CLAMP is what the border bodies cost a launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import defaultdict
from ako_amd import api
from oracle import pyoracle as po
w = h = 8192
img = po.gen_image(0, w, h)
for wrap, name in ((0, "CLAMP"), (2, "REPEAT"), (3, "ZERO"), (1, "MIRROR")):
    s = api.settings(wavelet=0, wrap=wrap, compression=2, q=16, g=16)
    with api.Plan(s, 4, w, h) as plan:
        d = torch.from_numpy(img).cuda().reshape(1, h, w, 4)
        st = plan.new_streams(); back = plan.new_images()
        for _ in range(3):
            plan.encode(d, st); plan.decode(st, back)
        plan.synchronize(); plan.set_profiling(True)
        for _ in range(10):
            plan.encode(d, st); plan.decode(st, back)
        plan.synchronize()
        agg = defaultdict(list)
        for r in plan.kernel_records(False) + plan.kernel_records(True):
            agg[(r["name"], r["level"])].append(r["ms"])
        tot = sum(sum(v) / len(v) for v in agg.values())
        print(f"{name:7s} sum {tot:.4f} ms ", [(k[0].replace('_stream_dd137', ''), k[1], round(sum(v) / len(v) * 1000, 1)) for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]], flush=True)
