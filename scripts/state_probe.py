"""Is the fast / slow level of the kernel times a property of the plan's HIP stream (hardware queue) or of its buffers?  Four plans
alive at once, each with buffers of its own, measured in turn twice; then the same four buffer sets on ONE shared stream."""
import os, sys
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
    g = lambda k: sum(rec[k]) / len(rec[k])
    return g(("fwd_stream_dd137_u8", 0)), g(("inv_stream_dd137_u8", 0)), g(("fwd_stream_dd137", 1)), g(("inv_stream_dd137", 1))
sets = []
for k in range(4):
    plan = api.Plan(s, 4, w, h)
    sets.append((plan, host.cuda().reshape(1, h, w, 4), plan.new_streams(), plan.new_images()))
for rep in range(2):
    for k, (plan, d, st, back) in enumerate(sets):
        print(f"own stream, plan {k} (pass {rep}): fwd0 %.4f inv0 %.4f fwd1 %.4f inv1 %.4f" % measure(plan, d, st, back), flush=True)
shared = torch.cuda.Stream()
plans2 = [api.Plan(s, 4, w, h, stream=shared.cuda_stream) for _ in range(4)]
for rep in range(2):
    for k, p2 in enumerate(plans2):
        _, d, st, back = sets[k]
        print(f"shared stream, plan {k} buffers {k} (pass {rep}): fwd0 %.4f inv0 %.4f fwd1 %.4f inv1 %.4f" % measure(p2, d, st, back), flush=True)
streams = [torch.cuda.Stream() for _ in range(4)]
plans3 = [api.Plan(s, 4, w, h, stream=streams[k].cuda_stream) for k in range(4)]
for k, p3 in enumerate(plans3):
    _, d, st, back = sets[0]
    print(f"torch stream {k}, buffers 0: fwd0 %.4f inv0 %.4f fwd1 %.4f inv1 %.4f" % measure(p3, d, st, back), flush=True)
