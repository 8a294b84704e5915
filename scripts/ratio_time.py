"""akoEncodeRatioExt (one transform per colour, candidates re-quantized on the device) against the search by re-encoding
(tools/akoenc.cpp:130-214 step for step, every candidate a full akoEncodeExt): wall time for the 4096 x 4096 `ratio 20` case.
usage (GPU box): python scripts/ratio_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from ako_amd import api


def search_by_reencoding(img, ratio, base):
    def run(q):
        s = base.copy()
        s.quantization = q
        try:
            return api.encode(img, s)
        except api.AkoError:
            return None
    size = lambda b: 0 if b is None else b.size
    target = img.size // ratio
    margin = target * 4 // 100
    runs, last = 1, run(0)
    ceil_size = floor_size = size(last)
    ceil_q = floor_q = 0
    q = 1
    while True:
        q *= 4
        ceil_size, ceil_q = floor_size, floor_q
        last = run(q)
        runs += 1
        floor_size, floor_q = size(last), q
        if not floor_size > target:
            break
    while abs(floor_size - ceil_size) > margin and abs(floor_q - ceil_q) > 1:
        q = (ceil_q + floor_q) // 2
        last = run(q)
        runs += 1
        if size(last) > target:
            ceil_size, ceil_q = size(last), q
        else:
            floor_size, floor_q = size(last), q
    return runs


for side in (512, 4096):
    img = api.synth_image(0, side, side)
    base = api.default_settings()
    api.encode_ratio(img, 20, base)  # warm-up: plans
    search_by_reencoding(img, 20, base)
    t0 = time.perf_counter()
    blob, q, encodes, transforms = api.encode_ratio(img, 20, base)
    t1 = time.perf_counter()
    runs = search_by_reencoding(img, 20, base)
    t2 = time.perf_counter()
    print(f"{side} x {side} RGBA, ratio 20: akoEncodeRatioExt {1e3 * (t1 - t0):.1f} ms ({encodes} candidates, {transforms} transforms, quantization {q}, "
          f"{blob.size} bytes); search by re-encoding {1e3 * (t2 - t1):.1f} ms ({runs} akoEncodeExt calls)")
