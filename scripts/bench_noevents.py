"""The default bench's timed loop (3 steps in flight) without the per-kernel HIP events: do the events cost
throughput?  (They do not measurably: this script exists to keep checking that.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ako_amd import api
from oracle import pyoracle as po
w = h = 8192
nfl = int(os.environ.get("INFLIGHT", "3"))
img = po.gen_image(0, w, h)
s = api.settings(wavelet=api.DD137, compression=api.COMPRESSION_NONE, q=16, g=16)
dev = torch.device("cuda", 0)
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(nfl - 1)]
plans = [api.Plan(s, 4, w, h, stream=st.cuda_stream) for st in streams]
d = torch.from_numpy(img).to(dev).reshape(1, h, w, 4)
strs = [p.new_streams() for p in plans]; backs = [p.new_images() for p in plans]
for events in (True, False, True, False):
    for p in plans:
        p.set_profiling(events)
    for i in range(6):
        plans[i % nfl].encode(d, strs[i % nfl]); plans[i % nfl].decode(strs[i % nfl], backs[i % nfl])
    torch.cuda.synchronize()
    for p in plans:
        p.set_profiling(events)
    K = 60
    t0 = time.perf_counter()
    for i in range(K):
        plans[i % nfl].encode(d, strs[i % nfl]); plans[i % nfl].decode(strs[i % nfl], backs[i % nfl])
    t_submit = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"events={events} ms/step {(t1 - t0) / K * 1e3:.4f}  host submit ms/step {(t_submit - t0) / K * 1e3:.4f}  Gpx/s {w * h * K / (t1 - t0) / 1e9:.1f}")
