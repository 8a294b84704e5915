"""Exactly STEPS (default 4) encode + decode steps of one bench workload on one plan, nothing else: the program
scripts/collect_traffic.sh puts under rocprofv3 --pmc, so that 'bytes per step' is the counter total / STEPS.
usage: traffic_step.py [full8192 | full8192x4 | rgb8192 | batch4k | lift4096 | tiles16k_512 | tiles16k_256]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ako_amd import api
wl = sys.argv[1] if len(sys.argv) > 1 else "full8192"
steps = int(os.environ.get("STEPS", "4"))
planes = False
if wl == "full8192":
    w, h, ch, batch, s = 8192, 8192, 4, 1, api.settings(wavelet=0, compression=2, q=16, g=16)
elif wl == "full8192x4":  # four images per launch: dispatches of 0.8 ms (scripts/collect_clock.sh: GRBM_GUI_ACTIVE reads high on short ones)
    w, h, ch, batch, s = 8192, 8192, 4, 4, api.settings(wavelet=0, compression=2, q=16, g=16)
elif wl == "rgb8192":
    w, h, ch, batch, s = 8192, 8192, 3, 1, api.settings(wavelet=0, compression=2, q=16, g=16)
elif wl == "batch4k":
    w, h, ch, batch, s = 3840, 2160, 4, 8, api.settings(wavelet=0, compression=2, q=16, g=16)
elif wl == "lift4096":
    w, h, ch, batch, planes = 4096, 4096, 1, 1, True
    s = api.settings(wavelet=0, compression=2, q=0, g=0, color=api.COLOR_NONE)
elif wl.startswith("tiles16k"):
    w, h, ch, batch = 16384, 16384, 4, 1
    s = api.settings(wavelet=api.CDF53, compression=2, q=0, g=0, tiles=int(wl.split("_")[1]))
else:
    raise SystemExit("unknown workload " + wl)
with api.Plan(s, ch, w, h, batch=batch, planes_i16=planes) as plan:
    if planes:
        host = np.stack([api.synth_plane(w * h).reshape(1, h, w)])
    else:
        host = np.stack([api.synth_image(0, w, h, seed=0x9E3779B9 + k) for k in range(batch)])[..., :ch]
    d = torch.from_numpy(np.ascontiguousarray(host)).cuda()
    st, back = plan.new_streams(), plan.new_images()
    for _ in range(steps):
        plan.encode(d, st)
        plan.decode(st, back)
    plan.synchronize()
print("steps", steps)
