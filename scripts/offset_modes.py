"""Does the relative position of image and stream buffers matter?  One allocation, the three buffers of a step carved
out of it at varying offsets (profiles/r2_repeatability.txt shows two levels of kernel time between identical
processes; this separates "where in the address space" from "which physical pages")."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ako_amd import api
from oracle import pyoracle as po
w = h = 8192
img = po.gen_image(0, w, h)
s = api.settings(wavelet=0, compression=2, q=16, g=16)
plan = api.Plan(s, 4, w, h)
nbytes_img = w * h * 4
nbytes_str = plan.new_streams().numel() * 2
pool = torch.empty(nbytes_img * 2 + nbytes_str + (1200 << 20), dtype=torch.uint8, device="cuda")
host = torch.from_numpy(img).reshape(-1)
def carve(off, n, dtype):
    t = pool[off:off + n]
    return t.view(dtype)
deltas = [int(x) for x in os.environ["DELTAS_MB"].split(",")] if os.environ.get("DELTAS_MB") else None
for delta in ([d << 20 for d in deltas] if deltas else [0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096, 3 << 19, 1 << 21, 0]):
    o_img = 0
    o_str = nbytes_img + (8 << 20) + delta
    o_back = o_str + nbytes_str + (8 << 20) + delta
    d = carve(o_img, nbytes_img, torch.uint8).reshape(1, h, w, 4); d.copy_(host.cuda().reshape(1, h, w, 4))
    st = carve(o_str, nbytes_str, torch.int16).reshape(1, -1)
    back = carve(o_back, nbytes_img, torch.uint8).reshape(1, h, w, 4)
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
    f0 = sum(rec[("fwd_stream_dd137_u8", 0)]) / 10; i0 = sum(rec[("inv_stream_dd137_u8", 0)]) / 10
    tot = sum(sum(v) / len(v) for v in rec.values())
    print(f"delta {delta:8d}: fwd0 {f0:.4f} inv0 {i0:.4f} sum {tot:.4f}", flush=True)
