"""where do the column-group kernel's streams differ from the oracle's? (development aid)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
os.environ.setdefault("AKO_HIP_GROUP_MIN", "64")
from ako_amd import api
from oracle import pyoracle as po

def run(w, h, wavelet, wrap, tiles=0):
    img = np.random.default_rng(3).integers(0, 256, (h, w, 4), dtype=np.uint8)
    s = po.settings(wavelet=wavelet, wrap=wrap, color=0, compression=2, q=0, g=0, tiles=tiles)
    ob, st = po.encode_image(s, img)
    s.color = po.effective_color(s)
    a = api.Settings(s.wavelet, s.color, s.wrap, s.compression, s.tiles_dimension, s.quantization, s.gate, s.chroma_loss, s.discard_non_visible)
    with api.Plan(a, 4, w, h, batch=1) as plan:
        plan.set_profiling(True)
        d = plan.encode(torch.from_numpy(img[None]).cuda())
        plan.synchronize()
        names = [r["name"] for r in plan.kernel_records(False)]
        got = d.cpu().numpy().reshape(-1).view(np.int16)
    exp = ob[16:].view(np.int16)
    bad = np.nonzero(got != exp)[0]
    print(w, h, wavelet, wrap, names[:2], "mismatches", bad.size, "of", exp.size)
    if bad.size:
        # level 0 groups sit at the end of the stream: plane p at off0 + p * (1 + 3 N)
        Tc, Tr = w // 2, (h + 1) // 2
        N = Tc * Tr
        off0 = exp.size - 4 * (1 + 3 * N)
        for b in bad[:20]:
            if b < off0:
                print("  value", b, "below level 0")
                continue
            p, rem = divmod(b - off0, 1 + 3 * N)
            if rem == 0:
                print("  head of plane", p); continue
            sb, idx = divmod(rem - 1, N)
            print("  plane", p, "sub-band", "CBD"[sb], "row", idx // Tc, "col", idx % Tc, "got", got[b], "exp", exp[b])
        rows = sorted({((b - off0) % (1 + 3 * N) - 1) % N // Tc for b in bad if b >= off0})
        print("  rows:", rows[:40])

for a in sys.argv[1:]:
    run(*[int(x) for x in a.split(",")])
