"""Micro-benchmark of the fused tail kernels: PLANES_I16 plans small enough to be tail-only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ako_amd import api

for (w, planes, batch) in [(128, 1, 1), (128, 4, 1), (64, 4, 1), (32, 4, 1), (16, 4, 1), (128, 4, 64), (128, 4, 256)]:
    s = api.settings(wavelet=0, wrap=0, compression=2, q=0, g=0, color=2)
    with api.Plan(s, planes, w, w, batch=batch, planes_i16=True) as plan:
        d = (torch.randint(-512, 512, (batch, planes, w, w), dtype=torch.int16, device="cuda"))
        st = plan.encode(d)
        back = plan.decode(st)
        plan.synchronize()
        assert torch.equal(back, d)
        plan.set_profiling(True)
        for _ in range(20):
            plan.encode(d, st)
            plan.decode(st, back)
        plan.synchronize()
        e = plan.kernel_records(False); dd = plan.kernel_records(True)
        fe = sorted(r["ms"] for r in e)[len(e)//2]; fd = sorted(r["ms"] for r in dd)[len(dd)//2]
        print(f"w={w} planes={planes} batch={batch}: launches/enc={len(e)//20} fwd_tail {fe*1e3:.1f} us  inv_tail {fd*1e3:.1f} us  names={e[0]['name']},{dd[0]['name']}")
