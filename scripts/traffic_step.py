"""Exactly STEPS (default 4) encode + decode steps of the bench workload full8192 on one plan, nothing else: the program
scripts/collect_traffic.sh puts under rocprofv3 --pmc, so that 'bytes per step' is the counter total / STEPS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ako_amd import api
w = h = 8192
steps = int(os.environ.get("STEPS", "4"))
s = api.settings(wavelet=0, compression=2, q=16, g=16)
with api.Plan(s, 4, w, h) as plan:
    d = torch.from_numpy(api.synth_image(0, w, h)).cuda().reshape(1, h, w, 4)
    st, back = plan.new_streams(), plan.new_images()
    for _ in range(steps):
        plan.encode(d, st)
        plan.decode(st, back)
    plan.synchronize()
print("steps", steps)
