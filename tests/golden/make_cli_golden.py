#!/usr/bin/env python3
"""Golden vectors for the command-line tools: what the REFERENCE's own akoenc / akodec
(built from /root/reference by oracle/Makefile into oracle/_ref/) print and write for a set of seeded
PNG inputs.  Run in the build container only; commits tests/golden/cli.json (data, no reference text).

    python tests/golden/make_cli_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pngutil  # noqa: E402
from cli_cases import CASES, make_image  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, image, flags in CASES:
            png = os.path.join(tmp, "in.png")
            with open(png, "wb") as f:
                f.write(pngutil.write_png(make_image(image)))
            ako = os.path.join(tmp, "out.ako")
            enc = subprocess.run([os.path.join(REF, "akoenc-ref"), "-i", png, "-o", ako, "-ch"] + flags,
                                 capture_output=True, text=True)
            assert enc.returncode == 0, (name, enc.stdout, enc.stderr)
            blob = open(ako, "rb").read()
            back = os.path.join(tmp, "back.png")
            dec = subprocess.run([os.path.join(REF, "akodec-ref"), "-i", ako, "-o", back, "-ch"], capture_output=True, text=True)
            assert dec.returncode == 0, (name, dec.stdout, dec.stderr)
            # (the reference's PNG writer may pick a palette / lower bit depth: its '-ch' line is the
            # Adler-32 of the decoded pixels themselves, tools/akodec.cpp:193-195)
            out[name] = {
                "image": image, "flags": flags,
                "blob_bytes": len(blob), "blob_adler32": f"{zlib.adler32(blob) & 0xFFFFFFFF:08x}",
                "encoder_summary": enc.stdout.strip().splitlines()[-1],
                "decoder_summary": dec.stdout.strip().splitlines()[-1],
                "decoded_adler32": dec.stdout.strip().splitlines()[-1].split(")")[0].lstrip("("),
            }
            print(name, out[name]["blob_bytes"], out[name]["encoder_summary"])
    with open(os.path.join(HERE, "cli.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
