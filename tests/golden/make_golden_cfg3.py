#!/usr/bin/env python3
"""Adds the reference's checksums of ALL 64 images of BASELINE configs[3] (3840x2160 RGBA, image i seeded 0x9E3779B9 + i,
DD13/7 q16 g16, compression NONE) to tests/golden/checksums.json: cfg3_4k_image0 .. cfg3_4k_image63.  make_golden.py keeps
the first four; this script generates the rest with the same compiled reference (oracle/_ref/libako_ref.so, built by
oracle/Makefile from /root/reference) without re-running the 16384 x 16384 cases.  Run in the build container:
    python tests/golden/make_golden_cfg3.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as po  # noqa: E402


def digest(a):
    return {"bytes": int(a.size), "adler32": f"{po.adler32(a):08x}",
            "sha256": hashlib.sha256(memoryview(np.ascontiguousarray(a)).cast("B")).hexdigest()}


def main():
    path = os.path.join(HERE, "checksums.json")
    sums = json.load(open(path))
    s = po.settings(wavelet=po.DD137, compression=po.COMPRESSION_NONE, q=16, g=16)
    for i in range(64):
        key = f"cfg3_4k_image{i}"
        img = po.gen_image(0, 3840, 2160, seed=0x9E3779B9 + i)
        blob, _ = po.ref_encode_image(s, img)
        dec, _, _ = po.ref_decode_image(blob)
        new = {"seed_offset": i, "input_adler32": f"{po.adler32(img):08x}", "blob": digest(blob), "decoded": digest(dec)}
        if key in sums["baseline"]:
            assert sums["baseline"][key] == new, key  # the first four: what make_golden.py wrote
        sums["baseline"][key] = new
        print(key, new["blob"]["adler32"], new["decoded"]["adler32"], flush=True)
    json.dump(sums, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
