"""Inputs and flag sets shared by tests/golden/make_cli_golden.py (reference tools) and tests/test_cli.py."""
import numpy as np

from oracle import pyoracle as po

CASES = [
    # name, image, akoenc flags
    ("cfg0_cdf53_q16", "rgba512", ["-w", "CDF53", "-q", "16"]),  # BASELINE configs[0]
    ("defaults", "rgba512", []),
    ("lossless_rgb", "rgb100x75", ["-q", "0", "-w", "CDF53"]),
    ("ratio20", "rgba512", ["-dev-r", "20"]),
    ("ratio1", "rgba512", ["-dev-r", "1"]),
    ("ratio8_cdf53", "rgba512", ["-dev-r", "8", "-w", "CDF53"]),
    ("subg_mirror_gate", "rgba512", ["-c", "SUBTRACT-G", "-wr", "MIRROR", "-g", "8", "-chroma-loss", "2", "-d"]),
    ("haar_gray", "gray64", ["-w", "HAAR", "-q", "0", "-c", "NONE", "-wr", "REPEAT"]),
    ("none_ga", "ga51x41", ["-dev-compression", "NONE"]),
    ("zero_wrap", "rgb100x75", ["-wr", "ZERO", "-q", "40"]),
]


def make_image(kind: str) -> np.ndarray:
    if kind == "rgba512":
        return po.gen_image(0, 512, 512)
    if kind == "rgb100x75":
        return np.ascontiguousarray(po.gen_image(0, 100, 75)[:, :, :3])
    if kind == "gray64":
        return np.ascontiguousarray(po.gen_image(0, 64, 64)[:, :, :1])
    if kind == "ga51x41":
        return np.ascontiguousarray(po.gen_image(0, 51, 41)[:, :, [0, 3]])
    raise KeyError(kind)
