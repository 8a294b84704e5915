"""Which allocation decides the fast / slow placement (scripts/alloc_modes.py): the plan's own scratch or the caller's
buffers?  Phase A keeps one plan and re-allocates the buffers, phase B keeps the buffers and re-creates the plan."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ako_amd import api
from oracle import pyoracle as po
w = h = 8192
host = torch.from_numpy(po.gen_image(0, w, h))
s = api.settings(wavelet=0, compression=2, q=16, g=16)
def measure(plan, d, st, back):
    for _ in range(3):
        plan.encode(d, st); plan.decode(st, back)
    plan.synchronize(); plan.set_profiling(True)
    for _ in range(10):
        plan.encode(d, st); plan.decode(st, back)
    plan.synchronize()
    rec = {}
    for r in plan.kernel_records(False) + plan.kernel_records(True):
        rec.setdefault((r["name"], r["level"]), []).append(r["ms"])
    plan.set_profiling(False)
    return sum(rec[("fwd_stream_dd137_u8", 0)]) / 10, sum(rec[("inv_stream_dd137_u8", 0)]) / 10, sum(rec[("fwd_stream_dd137", 1)]) / 10
keep = []
plan = api.Plan(s, 4, w, h)
for rnd in range(5):
    d = host.cuda().reshape(1, h, w, 4); st = plan.new_streams(); back = plan.new_images()
    print("A (same plan, new buffers) %d: fwd0 %.4f inv0 %.4f fwd1 %.4f" % ((rnd,) + measure(plan, d, st, back)), flush=True)
    keep.append((d, st, back))
d, st, back = keep[-1]
for rnd in range(5):
    p2 = api.Plan(s, 4, w, h)
    print("B (new plan, same buffers) %d: fwd0 %.4f inv0 %.4f fwd1 %.4f" % ((rnd,) + measure(p2, d, st, back)), flush=True)
    keep.append(p2)
