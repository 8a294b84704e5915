"""Do the level-0 kernel times depend on where the buffers landed?  Identical processes alternate between a fast and a
slow mode on the same box (profiles/r2_repeatability.txt); this re-allocates plan and buffers inside ONE process, keeping
the old ones alive so that every round gets new addresses, and prints times beside the device pointers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ako_amd import api
from oracle import pyoracle as po
w = h = 8192
img = po.gen_image(0, w, h)
s = api.settings(wavelet=0, compression=2, q=16, g=16)
keep = []
host = torch.from_numpy(img)
for rnd in range(int(os.environ.get("ROUNDS", "6"))):
    plan = api.Plan(s, 4, w, h)
    d = host.cuda().reshape(1, h, w, 4)
    st = plan.new_streams(); back = plan.new_images()
    for _ in range(3):
        plan.encode(d, st); plan.decode(st, back)
    plan.synchronize(); plan.set_profiling(True)
    for _ in range(10):
        plan.encode(d, st); plan.decode(st, back)
    plan.synchronize()
    rec = {}
    for r in plan.kernel_records(False) + plan.kernel_records(True):
        rec.setdefault((r["name"], r["level"]), []).append(r["ms"])
    f0 = sum(rec[("fwd_stream_dd137_u8", 0)]) / 10; i0 = sum(rec[("inv_stream_dd137_u8", 0)]) / 10
    tot = sum(sum(v) / len(v) for v in rec.values())
    print(f"round {rnd}: fwd0 {f0:.4f} inv0 {i0:.4f} sum {tot:.4f}  img {d.data_ptr():#x} stream {st.data_ptr():#x} back {back.data_ptr():#x}", flush=True)
    plan.set_profiling(False)
    keep.append((plan, d, st, back))
    if os.environ.get("PAD"):
        keep.append(torch.empty(int(os.environ["PAD"]) << 20, dtype=torch.uint8, device="cuda"))
