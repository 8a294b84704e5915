"""pytest configuration: registers the `gpu` marker and shared helpers.

`-m "not gpu"` : oracle vs golden vectors / compiled reference, host logic, C-ABI symbol checks (CPU only).
`-m gpu`       : parity tests proper -- the HIP path, called through the C-ABI, against the oracle.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-second CPU oracle runs at benchmark sizes")


@pytest.fixture(scope="session")
def po():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def golden_blobs():
    return np.load(os.path.join(GOLDEN, "blobs.npz"))


@pytest.fixture(scope="session")
def golden_sums():
    with open(os.path.join(GOLDEN, "checksums.json")) as f:
        return json.load(f)


def parse_case_id(cid: str):
    """Inverse of tests/golden/make_golden.py:case_id."""
    gen, dims, wavelet, wrap, q, g, t, color, cl, d = cid.split("_")
    w, h, ch = (int(v) for v in dims.split("x"))
    return dict(gen=int(gen[1:]), w=w, h=h, ch=ch, wavelet=wavelet, wrap=wrap, q=int(q[1:]), g=int(g[1:]),
                tiles=int(t[1:]), color=color, cl=int(cl[2:]), disc=int(d[1:]))


WAVELETS = {"dd137": 0, "cdf53": 1, "haar": 2}
WRAPS = {"clamp": 0, "mirror": 1, "repeat": 2, "zero": 3}
COLORS = {"ycocg": 0, "subg": 1, "none": 2}


def case_settings(po, c, compression=2):
    return po.settings(wavelet=WAVELETS[c["wavelet"]], color=COLORS[c["color"]], wrap=WRAPS[c["wrap"]],
                       compression=compression, tiles=c["tiles"], q=c["q"], g=c["g"], chroma_loss=c["cl"],
                       discard=c["disc"])


def case_input(po, c):
    return np.ascontiguousarray(po.gen_image(c["gen"], c["w"], c["h"])[:, :, :c["ch"]])
