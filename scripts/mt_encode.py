"""akoEncodeExt called from several host threads at once (each thread keeps its own plan and stream):
images per second from host memory, 1 / 2 / 4 / 8 threads."""
import os, sys, time, json, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ako_amd import api
from oracle import pyoracle as po
w = h = int(os.environ.get("W", "4096"))
img = po.gen_image(0, w, h)
s = api.settings(wavelet=api.DD137, compression=api.KAGARI, q=16, g=16)
ref = api.encode(img, s)
out = {"image": f"{w}x{h} RGBA"}
for n in (1, 2, 4, 8):
    imgs = [img.copy() for _ in range(n)]
    reps = 12
    ok = [True] * n
    def work(i):
        for _ in range(reps):
            b = api.encode(imgs[i], s)
            ok[i] = ok[i] and np.array_equal(b, ref)
    work(0)  # warm this thread's plan
    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    out[f"threads_{n}"] = {"images_per_s": round(n * reps / dt, 1), "Mpx_s": round(n * reps * w * h / dt / 1e6, 0), "all_equal": all(ok)}
print(json.dumps(out))
