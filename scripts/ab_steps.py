"""Same-box comparison of builds of the library on the bench's step (8192x8192 RGBA, DD13/7 q16 g16, encode + decode):
steps in flight 1 and 4, median of 7 regions of 20 steps.  usage: python scripts/ab_steps.py base r1 mid ..."""
import json, os, statistics, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, statistics
sys.path.insert(0, %r)
import numpy as np, torch
from ako_amd import api
from oracle import pyoracle as po
w = h = 8192
s = api.settings(wavelet=0, compression=2, q=16, g=16)
dev = torch.device("cuda", 0)
res = {}
for nfl in (1, 4):
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(nfl - 1)]
    plans = [api.Plan(s, 4, w, h, stream=st.cuda_stream) for st in streams]
    imgs = [torch.from_numpy(po.gen_image(0, w, h, seed=0x9E3779B9 + 7919 * k)).to(dev).reshape(1, h, w, 4) for k in range(nfl)]
    strs = [p.new_streams() for p in plans]; backs = [p.new_images() for p in plans]
    def step(i):
        k = i %% nfl
        plans[k].encode(imgs[k], strs[k]); plans[k].decode(strs[k], backs[k])
    for i in range(8): step(i)
    torch.cuda.synchronize()
    t = []
    for rep in range(7):
        t0 = time.perf_counter()
        for i in range(20): step(i)
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    res[nfl] = w * h * 20 / statistics.median(t) / 1e9
    del plans, imgs, strs, backs
print("RESULT %%.1f %%.1f" %% (res[1], res[4]))
''' % ROOT
for rnd in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ)
        name, _, extra = name.partition(":")
        for kv in filter(None, extra.split(",")):
            env[kv.split("=")[0]] = kv.split("=")[1]
        if name != "base":
            env["AKO_LIB_OVERRIDE"] = os.path.join(ROOT, "ako_amd", f"libako_{name}.so")
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print(name, extra, "Gpx/s inflight1 / inflight4:", line[-1][7:] if line else ("failed: " + r.stderr[-300:]), flush=True)
