"""bench.py without the checksum assertion, for timing experiments that produce wrong output on purpose."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ako_amd import api
from oracle import pyoracle as po
w = h = int(os.environ.get("W", "8192"))
img = po.gen_image(0, w, h)
s = api.settings(wavelet=int(os.environ.get("WAVELET", "0")), compression=2, q=int(os.environ.get("Q", "16")), g=int(os.environ.get("G", "16")))
with api.Plan(s, 4, w, h) as plan:
    d = torch.from_numpy(img).cuda().reshape(1, h, w, 4)
    st = plan.new_streams(); back = plan.new_images()
    for _ in range(3):
        plan.encode(d, st); plan.decode(st, back)
    plan.synchronize(); plan.set_profiling(True)
    for _ in range(10):
        plan.encode(d, st); plan.decode(st, back)
    plan.synchronize()
    from collections import defaultdict
    agg = defaultdict(list)
    for r in plan.kernel_records(False) + plan.kernel_records(True):
        agg[(r["name"], r["level"])].append(r["ms"])
    tot = sum(sum(v) / len(v) for v in agg.values())
    print("sum_kernel_ms %.4f" % tot, [(k[0], k[1], round(sum(v) / len(v), 4)) for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(os.environ.get("TOP", "16"))]])
