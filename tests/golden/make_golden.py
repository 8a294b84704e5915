#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the COMPILED REFERENCE (oracle/_ref).

Run in the build container (where /root/reference exists) after ``make -C oracle``:

    python tests/golden/make_golden.py            # small + medium vectors (seconds)
    python tests/golden/make_golden.py --big      # also the 8192^2 / 16384^2 checksums (minutes, ~20 GiB RAM)

Outputs (data only -- inputs and expected outputs, never reference source text):

* ``blobs.npz``      full raw blobs (compression NONE: 16 byte header + coefficient streams) and
                     decoded images of small cases, keyed by case id
* ``checksums.json`` Adler-32 / length / SHA-256 of blob and decoded image for a wider grid and
                     for the BASELINE.json configurations (SURVEY 8c/8d anchors)
* ``kat_1d.json``    the Lp / Hp rows printed by the reference's own dd137-test / cdf53-test
* ``quant.json``     quantizer / gate tables (akoQuantization / akoGate) for the benchmark tiles

Inputs are the seeded synthetic generators G0 / G1 / G2 of SURVEY 8d (oracle/ako_oracle.c:orcGen*),
whose Adler-32 is stored next to every vector so a mis-stated generator is caught first.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle as po  # noqa: E402

WAVELETS = {"dd137": po.DD137, "cdf53": po.CDF53, "haar": po.HAAR}
WRAPS = {"clamp": po.CLAMP, "mirror": po.MIRROR, "repeat": po.REPEAT, "zero": po.ZERO}


def case_id(gen, w, h, ch, wavelet, wrap, q, g, tiles, color="ycocg", cl=1, disc=0):
    return f"G{gen}_{w}x{h}x{ch}_{wavelet}_{wrap}_q{q}_g{g}_t{tiles}_{color}_cl{cl}_d{disc}"


COLORS = {"ycocg": po.YCOCG, "subg": po.SUBTRACT_G, "none": po.COLOR_NONE}


def make_input(gen, w, h, ch):
    img = po.gen_image(gen, w, h)  # RGBA
    return np.ascontiguousarray(img[:, :, :ch])


def run_ref(gen, w, h, ch, wavelet, wrap, q, g, tiles, color="ycocg", cl=1, disc=0, compression=po.COMPRESSION_NONE):
    img = make_input(gen, w, h, ch)
    s = po.settings(wavelet=WAVELETS[wavelet], color=COLORS[color], wrap=WRAPS[wrap], compression=compression,
                    tiles=tiles, q=q, g=g, chroma_loss=cl, discard=disc)
    blob, st = po.ref_encode_image(s, img)
    assert blob is not None, (gen, w, h, ch, wavelet, wrap, q, g, tiles, st)
    dec, _, st = po.ref_decode_image(blob)
    assert dec is not None
    return img, blob, dec


def digest(a):
    return {"bytes": int(a.size), "adler32": f"{po.adler32(a):08x}",
            "sha256": hashlib.sha256(memoryview(np.ascontiguousarray(a)).cast("B")).hexdigest()}


def small_blob_cases():
    cases = []
    # every wavelet x wrap on an odd-sized RGBA noise image, lossy and lossless
    for wv in WAVELETS:
        for wr in WRAPS:
            cases.append((1, 37, 29, 4, wv, wr, 16, 16, 0))
            cases.append((1, 37, 29, 4, wv, wr, 0, 0, 0))
    # channel counts, tiling with ragged edge tiles, tiny and non-square tiles
    for ch in (1, 2, 3, 4):
        cases.append((1, 64, 64, ch, "dd137", "clamp", 16, 0, 0))
        cases.append((0, 51, 41, ch, "cdf53", "clamp", 0, 0, 16))  # 51 % 16 = 3: narrowest legal edge tile
    cases += [
        (0, 64, 64, 4, "dd137", "clamp", 16, 16, 0),   # SURVEY anchor
        (1, 64, 64, 4, "dd137", "clamp", 0, 0, 0),     # SURVEY anchor (lossless)
        (1, 8, 8, 4, "dd137", "clamp", 16, 16, 0),
        (1, 19, 23, 3, "dd137", "mirror", 16, 16, 8),
        (1, 3, 3, 4, "cdf53", "repeat", 0, 0, 0),
        (1, 5, 40, 4, "dd137", "zero", 7, 3, 0),
        (1, 40, 5, 1, "haar", "clamp", 3, 0, 0),
        (0, 100, 75, 4, "dd137", "clamp", 16, 16, 32),
    ]
    return cases


def grid_cases():
    cases = []
    for wv in WAVELETS:
        for wr in WRAPS:
            for (q, g) in ((0, 0), (16, 0), (16, 16)):
                for ch in (1, 3, 4):
                    for tiles in (0, 64):
                        cases.append((1, 100, 75, ch, wv, wr, q, g, tiles))
    for wr in WRAPS:
        cases.append((0, 257, 131, 4, "dd137", wr, 16, 16, 0))
        cases.append((1, 256, 256, 4, "dd137", wr, 16, 16, 0))
        cases.append((1, 300, 200, 4, "cdf53", wr, 0, 0, 128))
    return cases


def parse_1d_tests():
    """Run the reference's own 1-D test programs and turn the rows they print into vectors."""
    out = []
    for prog, name in (("dd137-test", "dd137"), ("cdf53-test", "cdf53")):
        exe = os.path.join(ROOT, "oracle", "_ref", prog)
        txt = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
        cur = None
        wrap = None
        for line in txt.splitlines():
            m = re.match(r"# \w+ (Horizontal|Vertical) \(len: (\d+)\):", line)
            if m:
                cur = {"wavelet": name, "direction": m.group(1).lower(), "len": int(m.group(2)), "input": None,
                       "wraps": {}}
                out.append(cur)
                continue
            if cur is None:
                continue
            m = re.match(r"\[(\w+)\]", line)
            if m:
                wrap = m.group(1)
                cur["wraps"][wrap] = {}
                continue
            fields = line.split()
            if line.startswith("Lp:"):
                cur["wraps"][wrap]["lp"] = [int(v) for v in fields[1:]]
            elif line.startswith("Hp:"):
                cur["wraps"][wrap]["hp"] = [int(v) for v in fields[1:]]
            elif fields and re.fullmatch(r"-?\d+", fields[0]) and cur["input"] is None:
                cur["input"] = [int(v) for v in fields]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    args = ap.parse_args()
    assert po.have_ref(), "build oracle/_ref first: make -C oracle"

    blobs = {}
    meta = {}
    for c in small_blob_cases():
        img, blob, dec = run_ref(*c)
        cid = case_id(*c)
        blobs[cid + "/blob"] = blob
        blobs[cid + "/dec"] = dec
        meta[cid] = {"input_adler32": f"{po.adler32(img):08x}"}
    # variety: colour modes, chroma loss, discard, kagari file
    for extra in (dict(color="subg"), dict(color="none"), dict(cl=0), dict(cl=3), dict(disc=1)):
        c = (1, 48, 40, 4, "dd137", "clamp", 16, 16, 0)
        img, blob, dec = run_ref(*c, **extra)
        cid = case_id(*c, **extra)
        blobs[cid + "/blob"] = blob
        blobs[cid + "/dec"] = dec
        meta[cid] = {"input_adler32": f"{po.adler32(img):08x}"}
    np.savez_compressed(os.path.join(HERE, "blobs.npz"), **blobs)

    sums = {"small": meta, "grid": {}, "baseline": {}, "kagari": {}}
    for c in grid_cases():
        img, blob, dec = run_ref(*c)
        sums["grid"][case_id(*c)] = {"input_adler32": f"{po.adler32(img):08x}", "blob": digest(blob),
                                     "decoded": digest(dec)}

    # BASELINE.json configurations (SURVEY 8c anchor table)
    base = [
        ("cfg0_512_cdf53_q16", (0, 512, 512, 4, "cdf53", "clamp", 16, 0, 0)),
        ("cfg2_4k_dd137_q16g16", (0, 3840, 2160, 4, "dd137", "clamp", 16, 16, 0)),
        ("g0_4096_dd137_q16g16", (0, 4096, 4096, 4, "dd137", "clamp", 16, 16, 0)),
        ("g0_4096_cdf53_lossless", (0, 4096, 4096, 4, "cdf53", "clamp", 0, 0, 0)),
        ("g0_4096_dd137_q16g16_t256", (0, 4096, 4096, 4, "dd137", "clamp", 16, 16, 256)),
        ("g1_4096_dd137_q16g16", (1, 4096, 4096, 4, "dd137", "clamp", 16, 16, 0)),
        ("g1_4096_dd137_lossless", (1, 4096, 4096, 4, "dd137", "clamp", 0, 0, 0)),
        ("g1_1000x777_dd137_q16g16_t256", (1, 1000, 777, 4, "dd137", "clamp", 16, 16, 256)),
    ]
    if args.big:
        base += [
            ("cfg2_8192_dd137_q16g16", (0, 8192, 8192, 4, "dd137", "clamp", 16, 16, 0)),
            ("cfg4_16384_cdf53_lossless_t256", (0, 16384, 16384, 4, "cdf53", "clamp", 0, 0, 256)),
            ("cfg4_16384_cdf53_lossless_t512", (0, 16384, 16384, 4, "cdf53", "clamp", 0, 0, 512)),
            ("cfg4_16384_cdf53_lossless", (0, 16384, 16384, 4, "cdf53", "clamp", 0, 0, 0)),
        ]
    prev = {}
    path = os.path.join(HERE, "checksums.json")
    if os.path.exists(path):
        prev = json.load(open(path)).get("baseline", {})
    for name, c in base:
        img, blob, dec = run_ref(*c)
        sums["baseline"][name] = {"case": case_id(*c), "input_adler32": f"{po.adler32(img):08x}",
                                  "blob": digest(blob), "decoded": digest(dec)}
        print(name, sums["baseline"][name]["blob"]["adler32"], sums["baseline"][name]["decoded"]["adler32"])
        del img, blob, dec
    for k, v in prev.items():  # keep big entries generated by an earlier --big run
        sums["baseline"].setdefault(k, v)

    # batch of 4K images: image i seeded 0x9E3779B9 + i (config 3); keep the first 4
    for i in range(4):
        img = po.gen_image(0, 3840, 2160, seed=0x9E3779B9 + i)
        s = po.settings(wavelet=po.DD137, compression=po.COMPRESSION_NONE, q=16, g=16)
        blob, _ = po.ref_encode_image(s, img)
        dec, _, _ = po.ref_decode_image(blob)
        sums["baseline"][f"cfg3_4k_image{i}"] = {"seed_offset": i, "input_adler32": f"{po.adler32(img):08x}",
                                                 "blob": digest(blob), "decoded": digest(dec)}

    # lifting-only plane (config 1): G2 plane through the public API as 1 channel, colour NONE
    # is impossible for values outside 0..255, so the plane oracle is the private akoLift path:
    # here we pin it through the 1-channel u8-range variant ...
    img = (po.gen_plane(4096 * 4096).astype(np.int32) & 255).astype(np.uint8).reshape(4096, 4096, 1)
    s = po.settings(wavelet=po.DD137, color=po.COLOR_NONE, compression=po.COMPRESSION_NONE, q=0, g=0)
    blob, _ = po.ref_encode_image(s, img)
    dec, _, _ = po.ref_decode_image(blob)
    sums["baseline"]["cfg1_plane4096_u8variant"] = {"input_adler32": f"{po.adler32(img):08x}", "blob": digest(blob),
                                                    "decoded": digest(dec)}

    # real .ako files (Kagari) for the host entropy stage
    for name, c in (("cfg0_512_cdf53_q16", (0, 512, 512, 4, "cdf53", "clamp", 16, 0, 0)),
                    ("g0_4096_dd137_q16g16", (0, 4096, 4096, 4, "dd137", "clamp", 16, 16, 0)),
                    ("g0_300x200_dd137_q16_t64", (0, 300, 200, 4, "dd137", "clamp", 16, 0, 64))):
        img, blob, dec = run_ref(*c, compression=po.KAGARI)
        sums["kagari"][name] = {"case": case_id(*c), "input_adler32": f"{po.adler32(img):08x}", "blob": digest(blob),
                                "decoded": digest(dec)}
    json.dump(sums, open(path, "w"), indent=1, sort_keys=True)

    json.dump(parse_1d_tests(), open(os.path.join(HERE, "kat_1d.json"), "w"), indent=None)

    # quantizer tables
    R = po.ref()
    q = {}
    for (tw, th) in ((512, 512), (4096, 4096), (8192, 8192), (16384, 16384), (3840, 2160), (256, 256), (1000, 777)):
        rows = []
        for factor in (16, 1, 100):
            w, h = tw, th
            while w > 2 and h > 2:
                rows.append([factor, w, h] + [int(R.akoQuantization(factor, m, tw, th, w, h)) for m in (1, 2)] +
                            [int(R.akoGate(factor, m, tw, th, w, h)) for m in (1, 2)])
                w, h = (w + 1) // 2, (h + 1) // 2
        q[f"{tw}x{th}"] = rows
    json.dump(q, open(os.path.join(HERE, "quant.json"), "w"))
    print("done")


if __name__ == "__main__":
    main()
