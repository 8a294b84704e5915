"""Command-line counterparts of the reference tools: tools/akoenc.cpp, tools/akodec.cpp.

CPU tests cover the PNG codec (through tools/pngcheck) and the argument surface; the GPU tests run the
real tools and compare blobs, printed summaries and decoded pixels with what the REFERENCE's own tools
produced for the same PNGs (tests/golden/cli.json, made by tests/golden/make_cli_golden.py)."""
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import pngutil
from cli_cases import CASES, make_image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "cli.json")))


@pytest.fixture(scope="module")
def tools():
    from ako_amd import build

    return build.build_tools()


def run(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, text=True, timeout=300, **kw)


def pngcheck(tools, tmp_path, data: bytes, effort=None):
    src, raw, out = tmp_path / "in.png", tmp_path / "out.raw", tmp_path / "out.png"
    src.write_bytes(data)
    cmd = [os.path.join(tools, "pngcheck"), str(src), str(raw)] + ([str(out), str(effort)] if effort else [])
    r = run(cmd)
    if r.returncode != 0:
        return r, None, None
    blob = raw.read_bytes()
    head, _, pixels = blob.partition(b"\n")
    ch, w, h = (int(x) for x in head.split())
    img = np.frombuffer(pixels, dtype=np.uint8).reshape(h, w, ch)
    return r, img, (out.read_bytes() if effort else None)


# ---- CPU ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("ch", [1, 2, 3, 4])
@pytest.mark.parametrize("shape", [(1, 1), (7, 5), (33, 64), (75, 100)])
def test_png_reader_all_filters(tools, tmp_path, ch, shape):
    rng = np.random.default_rng(ch * 100 + shape[0])
    img = rng.integers(0, 256, (shape[0], shape[1], ch), dtype=np.uint8)
    img[: shape[0] // 2] //= 8  # smooth half: makes Paeth / Average predictions matter
    for filters in ((0,), (1,), (2,), (3,), (4,), (0, 1, 2, 3, 4)):
        r, got, _ = pngcheck(tools, tmp_path, pngutil.write_png(img, filters=filters))
        assert r.returncode == 0, r.stdout
        assert np.array_equal(got, img), (filters, ch, shape)


@pytest.mark.parametrize("ch", [1, 3, 4])
def test_png_reader_interlaced_and_split_idat(tools, tmp_path, ch):
    rng = np.random.default_rng(ch)
    for shape in ((1, 1), (3, 2), (9, 9), (37, 53)):
        img = rng.integers(0, 256, (shape[0], shape[1], ch), dtype=np.uint8)
        r, got, _ = pngcheck(tools, tmp_path, pngutil.write_png(img, interlace=True, idat_split=97))
        assert r.returncode == 0, r.stdout
        assert np.array_equal(got, img), (ch, shape)


@pytest.mark.parametrize("effort", [1, 2, 7, 10])
def test_png_writer_round_trip(tools, tmp_path, effort):
    for ch in (1, 2, 3, 4):
        img = make_image("rgba512")[:90, :130, :ch].copy()
        r, got, png = pngcheck(tools, tmp_path, pngutil.write_png(img), effort=effort)
        assert r.returncode == 0, r.stdout
        assert np.array_equal(pngutil.read_png(png), img)


def test_png_reader_rejects_what_the_reference_rejects(tools, tmp_path):
    img = np.zeros((4, 4, 3), dtype=np.uint8)
    good = pngutil.write_png(img)

    def patched_head(depth, ctype):
        body = struct.pack(">IIBBBBB", 4, 4, depth, ctype, 0, 0, 0)
        return good[:8] + struct.pack(">I", 13) + b"IHDR" + body + struct.pack(">I", zlib.crc32(b"IHDR" + body) & 0xFFFFFFFF) + good[33:]

    r, _, _ = pngcheck(tools, tmp_path, patched_head(16, 2))
    assert r.returncode == 1 and "Unsupported bits per pixel-component (16)" in r.stdout  # tools/akoenc.cpp:88-90
    r, _, _ = pngcheck(tools, tmp_path, patched_head(8, 3))
    assert r.returncode == 1 and "Unsupported channels number (3)" in r.stdout  # tools/akoenc.cpp:79-86
    broken = bytearray(good)
    broken[-20] ^= 0x55  # inside the IDAT body: checksum no longer matches
    r, _, _ = pngcheck(tools, tmp_path, bytes(broken))
    assert r.returncode == 1 and "PNG error" in r.stdout
    r, _, _ = pngcheck(tools, tmp_path, good[: len(good) // 2])
    assert r.returncode == 1 and "PNG error" in r.stdout
    r, _, _ = pngcheck(tools, tmp_path, b"not a png at all, just text" * 3)
    assert r.returncode == 1 and "not a PNG" in r.stdout


def test_cli_argument_surface(tools, tmp_path):
    enc, dec = os.path.join(tools, "akoenc"), os.path.join(tools, "akodec")
    for exe in (enc, dec):
        r = run([exe, "-h"])
        assert r.returncode == 0 and "USAGE" in r.stdout and "--input" in r.stdout
        r = run([exe, "--version"])
        assert r.returncode == 0 and "libako v" in r.stdout and "format 2" in r.stdout
        r = run([exe, "--no-such-flag"])
        assert r.returncode == 1 and "Unknown option" in r.stdout
        r = run([exe])
        assert r.returncode == 1 and "No input filename specified" in r.stdout
        r = run([exe, "-i", str(tmp_path / "missing.file")])
        assert r.returncode == 1 and "Error at opening file" in r.stdout
    # every reference flag is accepted by name (tools/akoenc.cpp:340-395), ranges are enforced
    help_text = run([enc, "--help"]).stdout
    for flag in ("--quantization", "--noise-gate", "--wavelet", "--color", "--wrap", "--chroma-loss",
                 "--discard-non-visible", "--benchmark", "--checksum", "--dev-ratio", "--dev-compression",
                 "--verbose", "--quiet", "--output"):
        assert flag in help_text, flag
    assert "--effort" in run([dec, "--help"]).stdout
    assert run([enc, "-q", "8193", "-i", "x"]).returncode == 1
    assert run([enc, "-w", "DCT", "-i", "x"]).returncode == 1
    assert run([dec, "-e", "0", "-i", "x"]).returncode == 1


def test_cli_golden_covers_every_case():
    assert sorted(GOLD) == sorted(name for name, _, _ in CASES)
    assert GOLD["cfg0_cdf53_q16"]["blob_bytes"] == 71825 and GOLD["cfg0_cdf53_q16"]["blob_adler32"] == "5e6a6736"


def test_cli_fails_loudly_without_a_gpu(tools, tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    png = tmp_path / "in.png"
    png.write_bytes(pngutil.write_png(make_image("gray64")))
    r = run([os.path.join(tools, "akoenc"), "-i", str(png)], env=dict(os.environ, AKO_HIP_QUIET="1"))
    assert r.returncode == 1 and "Ako error" in r.stdout


# ---- GPU ---------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("name,image,flags", CASES, ids=[c[0] for c in CASES])
def test_cli_matches_the_reference_tools(tools, tmp_path, name, image, flags):
    gold = GOLD[name]
    img = make_image(image)
    png, ako, back = tmp_path / "in.png", tmp_path / "out.ako", tmp_path / "back.png"
    png.write_bytes(pngutil.write_png(img))
    r = run([os.path.join(tools, "akoenc"), "-i", str(png), "-o", str(ako), "-ch"] + flags)
    assert r.returncode == 0, r.stdout + r.stderr
    blob = ako.read_bytes()
    assert len(blob) == gold["blob_bytes"]
    assert f"{zlib.adler32(blob) & 0xFFFFFFFF:08x}" == gold["blob_adler32"]
    assert r.stdout.strip().splitlines()[-1] == gold["encoder_summary"]

    r = run([os.path.join(tools, "akodec"), "-i", str(ako), "-o", str(back), "-ch"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines()[-1] == gold["decoder_summary"]
    pixels = pngutil.read_png(back.read_bytes())
    assert pixels.shape == img.shape
    assert f"{zlib.adler32(pixels.tobytes()) & 0xFFFFFFFF:08x}" == gold["decoded_adler32"]
    if "-q" in flags and flags[flags.index("-q") + 1] == "0" and "-g" not in flags:
        assert np.array_equal(pixels, img)  # lossless cases come back exactly


@pytest.mark.gpu
def test_cli_benchmark_and_verbose_output(tools, tmp_path):
    png, ako = tmp_path / "in.png", tmp_path / "out.ako"
    png.write_bytes(pngutil.write_png(make_image("rgba512")))
    r = run([os.path.join(tools, "akoenc"), "-i", str(png), "-o", str(ako), "-b", "-verbose"])
    assert r.returncode == 0, r.stdout + r.stderr
    for line in ("Benchmark:", " - Format:", " - Wavelet transformation:", " - Compression:", " - Total:",
                 "Input data: 4 channels, 512x512 px"):
        assert line in r.stdout, line
    r = run([os.path.join(tools, "akodec"), "-i", str(ako), "-b", "-verbose"])
    assert r.returncode == 0, r.stdout + r.stderr
    for line in ("Benchmark:", " - Compression:", " - Wavelet transformation:", " - Format:", " - Total:",
                 "Input data: 4 channels, 512x512 px, wavelet: 0, color: 3, wrap: 0, compression: 0"):
        assert line in r.stdout, line
    r = run([os.path.join(tools, "akoenc"), "-i", str(png), "-quiet"])
    assert r.returncode == 0 and r.stdout == ""


@pytest.mark.gpu
def test_cli_tiles_extra_round_trips(tools, tmp_path):
    img = make_image("rgba512")[:300, :421].copy()
    png, ako, back = tmp_path / "in.png", tmp_path / "out.ako", tmp_path / "back.png"
    png.write_bytes(pngutil.write_png(img))
    r = run([os.path.join(tools, "akoenc"), "-i", str(png), "-o", str(ako), "-q", "0", "-w", "CDF53", "--tiles", "64"])
    assert r.returncode == 0, r.stdout + r.stderr
    r = run([os.path.join(tools, "akodec"), "-i", str(ako), "-o", str(back), "-e", "3"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(pngutil.read_png(back.read_bytes()), img)


@pytest.mark.gpu
@pytest.mark.parametrize("name,image,flags", [c for c in CASES if c[0] in ("cfg0_cdf53_q16", "ratio20", "subg_mirror_gate",
                                                                             "haar_gray", "none_ga")],
                         ids=lambda v: v if isinstance(v, str) and "_" in v else None)
def test_reference_tools_run_unchanged_on_this_library(tmp_path, name, image, flags):
    """INTEGRATION.md, way 1: the REFERENCE'S OWN akoenc / akodec binaries (built into oracle/_ref/ by
    oracle/Makefile) with this library preloaded in place of theirs -- same files, same printed lines."""
    ref_enc = os.path.join(ROOT, "oracle", "_ref", "akoenc-ref")
    ref_dec = os.path.join(ROOT, "oracle", "_ref", "akodec-ref")
    lib = os.path.join(ROOT, "ako_amd", "libako.so")
    if not (os.path.exists(ref_enc) and os.path.exists(ref_dec)):
        pytest.skip("the reference tools were not built (oracle/_ref)")
    gold = GOLD[name]
    env = dict(os.environ, LD_PRELOAD=lib)
    png, ako, back = tmp_path / "in.png", tmp_path / "out.ako", tmp_path / "back.png"
    png.write_bytes(pngutil.write_png(make_image(image)))
    r = run([ref_enc, "-i", str(png), "-o", str(ako), "-ch"] + flags, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    blob = ako.read_bytes()
    assert len(blob) == gold["blob_bytes"] and f"{zlib.adler32(blob) & 0xFFFFFFFF:08x}" == gold["blob_adler32"]
    assert r.stdout.strip().splitlines()[-1] == gold["encoder_summary"]
    r = run([ref_dec, "-i", str(ako), "-o", str(back), "-ch"], env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines()[-1] == gold["decoder_summary"]
    # the preload really took effect: without a usable GPU path this library refuses, the reference's would not
    r = run([ref_enc, "-i", str(png)], env=dict(env, AKO_HIP_DEVICE="63", AKO_HIP_QUIET="1"))
    assert r.returncode == 1 and "Ako error" in r.stdout
