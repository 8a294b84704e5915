"""CPU: the C-ABI library loads, exports every symbol include/*.h declares, and its HOST logic
(header, quantizer curve, Kagari) agrees with the oracle and the golden vectors.  No GPU compute here.
"""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

from ako_amd import api


def _declared_symbols():
    names = set()
    for header in ("ako.h", "ako_hip.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b(ako[A-Z]\w*)\s*\(", text):
            names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol():
    L = api.lib()
    names = _declared_symbols()
    assert {"akoEncodeExt", "akoDecodeExt", "akoDefaultSettings", "akoDefaultCallbacks", "akoDefaultFree",
            "akoStatusString", "akoVersionMajor", "akoVersionMinor", "akoVersionPatch", "akoFormatVersion",
            "akoHipPlanCreate", "akoHipEncode", "akoHipDecode"} <= names
    for n in sorted(names):
        assert hasattr(L, n), f"{n} declared in include/ but not exported by libako.so"


def test_defaults_and_strings():
    s = api.default_settings()   # library/misc.c:30-47
    assert (s.wavelet, s.color, s.wrap, s.compression, s.tiles_dimension) == (0, 0, 0, 0, 0)
    assert (s.quantization, s.gate, s.chroma_loss, s.discard_non_visible) == (16, 0, 1, 0)
    L = api.lib()
    assert (L.akoVersionMajor(), L.akoVersionMinor(), L.akoVersionPatch(), L.akoFormatVersion()) == (0, 2, 0, 2)
    assert api.status_string(0) == "Everything Ok!"
    assert api.status_string(15) == "Broken input/premature end"
    assert api.status_string(99) == "Unknown status code"
    assert C.sizeof(api.Settings) == 40 and C.sizeof(api.Callbacks) == 40


def test_effective_color_rule():
    L = api.lib()
    for color in range(4):
        for q in (0, 16):
            for g in (0, 5):
                s = api.settings(color=color, q=q, g=g)
                exp = color
                if color == 0 and (q > 0 or g > 0):
                    exp = 3
                if color == 3 and q <= 0 and g <= 0:
                    exp = 0
                assert L.akoHipEffectiveColor(C.byref(s)) == exp


def test_quant_curve_matches_golden_and_oracle(po):
    L = api.lib()
    q = json.load(open(os.path.join(GOLDEN, "quant.json")))
    for key, rows in q.items():
        tw, th = (int(v) for v in key.split("x"))
        for factor, w, h, q1, q2, g1, g2 in rows:
            assert L.akoHostQuantStep(factor, 1, tw, th, w, h) == q1
            assert L.akoHostQuantStep(factor, 2, tw, th, w, h) == q2
            assert L.akoHostGateStep(factor, 1, tw, th, w, h) == g1
            assert L.akoHostGateStep(factor, 2, tw, th, w, h) == g2
    rng = np.random.default_rng(1)
    O = po.lib()
    for _ in range(500):
        tw, th = int(rng.integers(3, 20000)), int(rng.integers(3, 20000))
        f, m = int(rng.choice([0, 1, 7, 16, 100, 8192])), int(rng.choice([1, 2, 5]))
        w, h = tw, th
        while w > 2 and h > 2:
            assert L.akoHostQuantStep(f, m, tw, th, w, h) == O.orcQuantStep(f, m, tw, th, w, h)
            assert L.akoHostGateStep(f, m, tw, th, w, h) == O.orcGateStep(f, m, tw, th, w, h)
            w, h = (w + 1) // 2, (h + 1) // 2


def test_float_quantizer_is_exact_truncating_division():
    """The kernels quantize with trunc(float(x) * rq), rq = float((1/q)(1 + 1e-6)) (ako_kernels.hip.h:quantize).
    Sweep every q and, for each, every multiple of q and its neighbours (the only places where
    rounding could cross an integer) plus random values, in IEEE float32 as the GPU does."""
    qs = np.arange(1, 32766, dtype=np.int64)
    rng = np.random.default_rng(0)
    for q in qs[::1]:
        rq = np.float32((1.0 / float(q)) * (1.0 + 1e-6))
        k = np.arange(0, 32768 // q + 1, dtype=np.int64) * q
        x = np.unique(np.clip(np.concatenate([k - 1, k, k + 1, rng.integers(-32768, 32768, 16)]), -32768, 32767))
        x = np.concatenate([x, -x[x > -32768 + 1]])
        x = np.clip(x, -32768, 32767)
        got = np.trunc(x.astype(np.float32) * rq).astype(np.int64)
        exp = np.sign(x) * (np.abs(x) // q)
        assert np.array_equal(got, exp), int(q)


def test_head_bytes_and_validation(po):
    L = api.lib()
    O = po.lib()
    rng = np.random.default_rng(3)
    for _ in range(300):
        s = api.settings(wavelet=int(rng.integers(0, 5)), color=int(rng.integers(0, 5)), wrap=int(rng.integers(0, 5)),
                         compression=int(rng.integers(0, 4)),
                         tiles=int(rng.choice([0, 0, 4, 8, 12, 64, 512, 1024, 4096])))
        ch, w, h = int(rng.integers(1, 18)), int(rng.integers(0, 5000)), int(rng.integers(0, 5000))
        a, b = np.zeros(16, np.uint8), np.zeros(16, np.uint8)
        os_ = po.Settings(s.wavelet, s.color, s.wrap, s.compression, s.tiles_dimension, 16, 0, 1, 0)
        ra = L.akoHostHeadWrite(ch, w, h, C.byref(s), a.ctypes.data_as(C.c_void_p))
        rb = O.orcHeadWrite(ch, w, h, C.byref(os_), b.ctypes.data_as(C.c_void_p))
        assert ra == rb
        if ra == 0:
            assert np.array_equal(a, b)
            s2 = api.Settings()
            c2, w2, h2 = C.c_size_t(), C.c_size_t(), C.c_size_t()
            r = L.akoHostHeadRead(a.ctypes.data_as(C.c_void_p), C.byref(c2), C.byref(w2), C.byref(h2), C.byref(s2))
            if s.tiles_dimension >= 1024:
                assert r == 14   # AKO_INVALID_FLAGS: the reference's reader quirk (head.c:124)
            else:
                assert r == 0 and (c2.value, w2.value, h2.value) == (ch, w, h)
                assert (s2.wavelet, s2.color, s2.wrap, s2.compression, s2.tiles_dimension) == (
                    s.wavelet, s.color, s.wrap, s.compression, s.tiles_dimension)


def test_kagari_matches_oracle_bit_for_bit(po):
    L, O = api.lib(), po.lib()
    rng = np.random.default_rng(4)
    V = C.c_void_p
    cases = []
    for _ in range(200):
        n = int(rng.integers(1, 6000))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            v = rng.integers(-5, 6, n, dtype=np.int16)
        elif kind == 1:
            v = np.repeat(rng.integers(-300, 300, n // 7 + 1, dtype=np.int16), 7)[:n].copy()
        elif kind == 2:
            v = np.zeros(n, np.int16)
            v[::max(1, n // 5)] = 77
        else:
            v = rng.integers(-32768, 32768, n, dtype=np.int16)
        cases.append(v)
    long_run = np.zeros(200000, np.int16)
    long_run[150000:] = 3
    cases.append(long_run)
    cases.append(np.full(65534 * 2 + 5, -7, np.int16))
    for v in cases:
        n = v.size
        for cap in (n * 2 + 64, max(2, n), max(2, n // 2)):
            o1, o2 = np.zeros(cap + 8, np.uint8), np.zeros(cap + 8, np.uint8)
            a = L.akoHostKagariEncode(n * 2, cap, v.ctypes.data_as(V), o1.ctypes.data_as(V))
            b = O.orcKagariEncode(n * 2, cap, v.ctypes.data_as(V), o2.ctypes.data_as(V))
            assert a == b, (n, cap)
            if a:
                assert np.array_equal(o1[:a], o2[:a])
                d = np.zeros(n, np.int16)
                assert L.akoHostKagariDecode(n, a, n * 2, o1.ctypes.data_as(V), d.ctypes.data_as(V)) == a
                assert np.array_equal(d, v)
    # truncated / corrupt payloads must be refused, not crash
    v = rng.integers(-50, 50, 1000, dtype=np.int16)
    o = np.zeros(4096, np.uint8)
    a = L.akoHostKagariEncode(2000, 4096, v.ctypes.data_as(V), o.ctypes.data_as(V))
    d = np.zeros(1000, np.int16)
    assert L.akoHostKagariDecode(1000, a // 2, 2000, o.ctypes.data_as(V), d.ctypes.data_as(V)) == 0


def test_kagari_golden_file_payload(po, golden_sums):
    """The reference's .ako for BASELINE configs[0]: re-compress the oracle's raw stream with the
    product's host coder and compare with the golden file checksum."""
    exp = golden_sums["kagari"]["cfg0_512_cdf53_q16"]
    img = po.gen_image(0, 512, 512)
    s = po.settings(wavelet=1, q=16, g=0, compression=2)
    raw, _ = po.encode_image(s, img)
    body = raw[16:]
    L = api.lib()
    out = np.zeros(body.size, np.uint8)
    n = L.akoHostKagariEncode(body.size, body.size - 4, body.ctypes.data_as(C.c_void_p),
                              out[4:].ctypes.data_as(C.c_void_p))
    assert n + 4 + 16 == exp["blob"]["bytes"]
    head = raw[:16].copy()
    head[13] = int(head[13]) & 0xF3            # compression field (flag bits 10-11) := KAGARI (0)
    blob = np.concatenate([head, np.frombuffer(np.uint32(n).tobytes(), np.uint8), out[4:4 + n]])
    assert f"{po.adler32(blob):08x}" == exp["blob"]["adler32"]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present: covered by the gpu tests")
def test_no_gpu_fails_loudly_not_silently():
    """Without a HIP device the product path must refuse to work (no CPU fallback)."""
    os.environ["AKO_HIP_QUIET"] = "1"
    img = np.zeros((16, 16, 4), np.uint8)
    with pytest.raises(api.AkoError) as e:
        api.encode(img)
    assert e.value.status == api.AKO_ERROR and "no usable HIP device" in str(e.value)
    # validation still happens before the device is touched, in the reference's order
    with pytest.raises(api.AkoError) as e:
        api.encode(img, api.settings(tiles=12))
    assert e.value.status == 4
    with pytest.raises(api.AkoError) as e:
        api.decode(np.zeros(64, np.uint8))
    assert e.value.status == 11


def test_kagari_capacity_rule_closed_form():
    """The device encoder decides 'did the tile shrink' from the payload size alone: the host encoder
    (= library/kagari.c:64-112, pinned above) must succeed exactly when capacity >= payload + 1."""
    import ctypes as C

    from ako_amd import api

    L = api.lib()
    V = C.c_void_p
    rng = np.random.default_rng(11)
    for trial in range(200):
        n = int(rng.integers(1, 400))
        kind = trial % 4
        if kind == 0:
            v = rng.integers(-3, 4, n)
        elif kind == 1:
            v = rng.integers(-32768, 32768, n)
        elif kind == 2:
            v = np.repeat(rng.integers(-100, 100, (n + 7) // 8), 8)[:n]
        else:
            v = np.where(rng.random(n) < 0.8, 0, rng.integers(-5000, 5000, n))
        v = np.ascontiguousarray(v.astype(np.int16))
        big = np.zeros(8 * n + 64, dtype=np.uint8)
        size = L.akoHostKagariEncode(n * 2, big.size, v.ctypes.data_as(V), big.ctypes.data_as(V))
        assert size > 0
        for cap in range(max(size - 12, 1), size + 12):
            out = np.zeros(cap + 8, dtype=np.uint8)
            got = L.akoHostKagariEncode(n * 2, cap, v.ctypes.data_as(V), out.ctypes.data_as(V))
            assert (got == size) if cap >= size + 1 else (got == 0), (trial, n, size, cap, got)


def _expand_tokens(tok, n_values):
    """numpy model of akoHipKagariExpand: literals + run records -> the value sequence."""
    lit = np.ctypeslib.as_array(tok.literals, shape=(tok.n_literals,)).copy() if tok.n_literals else np.zeros(0, np.int16)
    out = np.zeros(n_values, dtype=np.int16)
    pos, used = 0, 0
    for k in range(tok.n_runs):
        r = tok.runs[k]
        n = r.after - used  # literals in front of this run
        out[pos:pos + n] = lit[used:r.after]
        pos += n
        used = r.after
        assert r.out_start == pos
        out[pos:pos + r.count] = lit[r.after - 1]
        pos += r.count
    out[pos:pos + (lit.size - used)] = lit[used:]
    assert pos + lit.size - used == n_values
    return out


def test_kagari_tokenizer_equals_decoder():
    """akoHostKagariTokenize + run expansion == akoHostKagariDecode, and both reject the same broken inputs."""
    import ctypes as C

    from ako_amd import api

    L = api.lib()
    V = C.c_void_p
    rng = np.random.default_rng(5)
    for trial in range(120):
        n = int(rng.integers(1, 3000))
        kind = trial % 4
        if kind == 0:
            v = rng.integers(-3, 4, n)
        elif kind == 1:
            v = np.repeat(rng.integers(-9, 9, (n + 15) // 16), 16)[:n]
        elif kind == 2:
            v = np.where(rng.random(n) < 0.9, 0, rng.integers(-5000, 5000, n))
        else:
            v = np.zeros(n)
            v[n // 2:] = 7
        v = np.ascontiguousarray(v.astype(np.int16))
        packed = np.zeros(8 * n + 64, dtype=np.uint8)
        size = L.akoHostKagariEncode(n * 2, packed.size, v.ctypes.data_as(V), packed.ctypes.data_as(V))
        assert size > 0
        tok = api.KagariTokens()
        used = L.akoHostKagariTokenize(n, size, packed.ctypes.data_as(V), 0, C.byref(tok))
        assert used == size
        assert np.array_equal(_expand_tokens(tok, n), v)
        assert tok.n_literals <= n and tok.n_runs <= n // 3 + 1
        L.akoHostKagariTokensFree(C.byref(tok))
        # damaged / truncated payloads: same verdict from both parsers
        for cut in (size - 1, size // 2, 1):
            if cut <= 0:
                continue
            bad = packed[:cut].copy()
            out = np.zeros(n, dtype=np.int16)
            a = L.akoHostKagariDecode(n, cut, n * 2, bad.ctypes.data_as(V), out.ctypes.data_as(V))
            tok = api.KagariTokens()
            b = L.akoHostKagariTokenize(n, cut, bad.ctypes.data_as(V), 0, C.byref(tok))
            assert a == b, (trial, cut, a, b)
            L.akoHostKagariTokensFree(C.byref(tok))
        flip = packed[:size].copy()
        flip[int(rng.integers(0, size))] ^= 1 << int(rng.integers(0, 8))
        out = np.zeros(n, dtype=np.int16)
        a = L.akoHostKagariDecode(n, size, n * 2, flip.ctypes.data_as(V), out.ctypes.data_as(V))
        tok = api.KagariTokens()
        b = L.akoHostKagariTokenize(n, size, flip.ctypes.data_as(V), 0, C.byref(tok))
        assert a == b
        if a:
            assert np.array_equal(_expand_tokens(tok, n), out)
        L.akoHostKagariTokensFree(C.byref(tok))


def _tokens_snapshot(tok):
    lit = np.ctypeslib.as_array(tok.literals, shape=(tok.n_literals,)).copy() if tok.n_literals else np.zeros(0, np.int16)
    runs = [(tok.runs[k].out_start, tok.runs[k].count, tok.runs[k].after) for k in range(tok.n_runs)]
    return lit, runs


def test_kagari_tokenizer_in_parallel_equals_the_sequential_parse(po):
    """Blocks of >= 128 KiB are parsed by several threads that start speculatively inside the bit-stream and join
    where they re-synchronise (ako_kagari.c).  With the threshold lowered so that small blocks take that route:
    same token lists, same return value as the sequential loop -- for clean blocks of every flavour (where the
    threads must actually have done the work), for blocks appended to existing lists at an offset, and for damaged,
    truncated, padded and mis-sized blocks (which the threads hand back to the sequential loop)."""
    L = api.lib()
    V = C.c_void_p
    L.akoHostKagariParallelStats.restype = None
    acc, back = C.c_size_t(0), C.c_size_t(0)

    def stats():
        L.akoHostKagariParallelStats(C.byref(acc), C.byref(back))
        return acc.value, back.value

    def tokenize(values, length, buf, base, threads, prefix=None):
        os.environ["AKO_KAGARI_THREADS"] = str(threads)
        tok = api.KagariTokens()
        if prefix is not None:  # lists that already hold another block's tokens
            pv = np.ascontiguousarray(prefix.astype(np.int16))
            pk = np.zeros(8 * pv.size + 64, dtype=np.uint8)
            ps = L.akoHostKagariEncode(pv.size * 2, pk.size, pv.ctypes.data_as(V), pk.ctypes.data_as(V))
            os.environ["AKO_KAGARI_THREADS"] = "1"
            assert L.akoHostKagariTokenize(pv.size, ps, pk.ctypes.data_as(V), 0, C.byref(tok)) == ps
            os.environ["AKO_KAGARI_THREADS"] = str(threads)
        used = L.akoHostKagariTokenize(values, length, buf.ctypes.data_as(V), base, C.byref(tok))
        snap = _tokens_snapshot(tok)
        L.akoHostKagariTokensFree(C.byref(tok))
        return used, snap

    old = {k: os.environ.get(k) for k in ("AKO_KAGARI_THREADS", "AKO_KAGARI_PAR_MIN")}
    rng = np.random.default_rng(77)
    try:
        os.environ["AKO_KAGARI_PAR_MIN"] = "256"
        clean_parallel = 0
        for trial in range(60):
            n = int(rng.integers(2000, 120000))
            kind = trial % 6
            if kind == 0:
                v = rng.integers(-3, 4, n)
            elif kind == 1:
                v = np.repeat(rng.integers(-9, 9, (n + 15) // 16), 16)[:n]
            elif kind == 2:
                v = np.where(rng.random(n) < 0.93, 0, rng.integers(-5000, 5000, n))
            elif kind == 3:
                v = rng.integers(-32767, 32768, n)           # long codes, no runs (-32768 has no code)
            elif kind == 4:
                v = np.zeros(n)                              # a handful of maximal runs: most ranges hold no code start
                v[rng.integers(0, n, 5)] = 9
            else:                                            # a real coefficient stream
                s = po.settings(wavelet=0, compression=2, q=int(rng.integers(0, 30)), g=int(rng.integers(0, 20)))
                ob, st = po.encode_image(s, po.gen_image(0, 160, 120, int(rng.integers(1, 1 << 30))))
                assert st == 0
                v = ob[16:].view(np.int16)
                n = v.size
            v = np.ascontiguousarray(np.asarray(v).astype(np.int16))
            packed = np.zeros(8 * n + 64, dtype=np.uint8)
            size = L.akoHostKagariEncode(n * 2, packed.size, v.ctypes.data_as(V), packed.ctypes.data_as(V))
            assert size > 0
            base = int(rng.integers(0, 1 << 20))
            prefix = rng.integers(-2, 3, int(rng.integers(1, 500))) if trial % 2 else None
            ref = tokenize(n, size, packed, base, 1, prefix)
            assert ref[0] == size
            for threads in (2, 3, 8, 16):
                before = stats()
                got = tokenize(n, size, packed, base, threads, prefix)
                after = stats()
                assert got[0] == ref[0], (trial, threads)
                assert np.array_equal(got[1][0], ref[1][0]) and got[1][1] == ref[1][1], (trial, threads)
                if size >= 256:
                    assert (after[0] - before[0]) + (after[1] - before[1]) == 1, (trial, threads, size)
                    clean_parallel += after[0] - before[0]
            # damaged / mis-sized blocks: the sequential verdict, whatever the threads made of them
            for damage in range(10):
                bad, length, values = packed[:size].copy(), size, n
                if damage == 8:
                    # the block ends in a third equal value whose run code is missing (a speculative trail that
                    # fails in the last byte must not be accepted): same values with an equal triple at the end, the last
                    # set bit -- the run code "1" -- cleared
                    v3 = v.copy()
                    v3[-3:] = 7
                    pk3 = np.zeros(8 * n + 64, dtype=np.uint8)
                    s3 = L.akoHostKagariEncode(n * 2, pk3.size, v3.ctypes.data_as(V), pk3.ctypes.data_as(V))
                    bad, length = pk3[:s3].copy(), s3
                    last = int(np.flatnonzero(bad)[-1])
                    bad[last] &= ~np.uint8(bad[last] & -bad[last])  # lowest set bit of the last non-zero byte
                elif damage == 9:
                    # a code of more than 31 bits (a run of zero bytes) somewhere behind the first range: whether it still
                    # "fits" depends on the reader's refill history, so the threads must leave the verdict to the sequential loop
                    at = int(rng.integers(size // 3, max(size // 3 + 1, size - 6)))
                    bad[at:at + int(rng.integers(3, 6))] = 0
                elif damage < 3:
                    bad[int(rng.integers(0, size))] ^= 1 << int(rng.integers(0, 8))
                elif damage == 3:
                    length = int(rng.integers(1, size + 1))
                elif damage == 4:
                    bad = np.concatenate([bad, rng.integers(0, 256, 5, dtype=np.uint8)])
                    length = bad.size
                elif damage == 5:
                    bad = np.concatenate([bad, np.zeros(3, dtype=np.uint8)])  # zero padding behind the last code
                    length = bad.size
                elif damage == 6:
                    values = n + 1
                else:
                    values = n - 1
                bad = np.ascontiguousarray(bad)
                want = tokenize(values, length, bad, base, 1)
                got = tokenize(values, length, bad, base, 8)
                assert got[0] == want[0], (trial, damage, got[0], want[0])
                if want[0]:
                    assert np.array_equal(got[1][0], want[1][0]) and got[1][1] == want[1][1], (trial, damage)
        assert clean_parallel > 150, clean_parallel  # the threads finished most clean blocks themselves
    finally:
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val


def test_kagari_decoder_verdicts_on_damaged_payloads_match_the_oracle(po):
    """compression.c:69 accepts a block only when the decoder reports exactly the block size as consumed, and
    kagari.c reports the bytes its eager reader FETCHED: on damaged payloads the verdict depends on that fetch
    pattern.  Host decoder, host tokenizer and oracle must agree on every input (the oracle is pinned against
    the compiled reference in tests/test_oracle_vs_ref.py)."""
    L, O = api.lib(), po.lib()
    V = C.c_void_p
    O.orcKagariDecode.restype = C.c_size_t
    O.orcKagariDecode.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, V, V]
    rng = np.random.default_rng(2024)
    agree_nonzero = 0
    for trial in range(400):
        n = int(rng.integers(2, 1500))
        kind = trial % 3
        if kind == 0:
            v = np.where(rng.random(n) < 0.85, 0, rng.integers(-900, 900, n))
        elif kind == 1:
            v = np.repeat(rng.integers(-20, 20, (n + 5) // 6), 6)[:n]
        else:
            v = rng.integers(-4, 5, n)
        v = np.ascontiguousarray(v.astype(np.int16))
        packed = np.zeros(8 * n + 64, dtype=np.uint8)
        size = L.akoHostKagariEncode(n * 2, packed.size, v.ctypes.data_as(V), packed.ctypes.data_as(V))
        assert size > 0
        for damage in range(6):
            bad = packed[:size].copy()
            length = size
            if damage < 3:
                bad[int(rng.integers(0, size))] ^= 1 << int(rng.integers(0, 8))
            elif damage == 3:
                at = int(rng.integers(0, size))
                bad[at:at + 3] = rng.integers(0, 256, bad[at:at + 3].size, dtype=np.uint8)
            elif damage == 4:
                length = int(rng.integers(1, size + 1))
            else:
                bad = np.concatenate([bad, rng.integers(0, 256, 5, dtype=np.uint8)])  # trailing garbage inside the block
                length = bad.size
            bad = np.ascontiguousarray(bad)
            for values in (n, n + 1, max(1, n - 1)):  # also a wrong expected count
                cap = 2 * values + 2 * 64  # what decode.c:152 hands over: plane bytes + spacing
                o1 = np.zeros(cap // 2 + 8, dtype=np.int16)
                o2 = np.zeros(cap // 2 + 8, dtype=np.int16)
                a = L.akoHostKagariDecode(values, length, 2 * values, bad.ctypes.data_as(V), o1.ctypes.data_as(V))
                b = O.orcKagariDecode(values, length, cap, bad.ctypes.data_as(V), o2.ctypes.data_as(V))
                tok = api.KagariTokens()
                c = L.akoHostKagariTokenize(values, length, bad.ctypes.data_as(V), 0, C.byref(tok))
                L.akoHostKagariTokensFree(C.byref(tok))
                assert a == b == c, (trial, damage, values, a, b, c)
                if a:
                    agree_nonzero += 1
                    assert np.array_equal(o1[:values], o2[:values])
    assert agree_nonzero > 50  # damaged-but-accepted streams were among them


def test_synthetic_generators_match_the_oracle_and_the_survey_anchors(po, golden_sums):
    """bench.py draws its inputs from the library's host side (akoHostSynthImage / akoHostSynthPlane); the oracle
    keeps its own restatement of SURVEY 8d's generators.  The two must agree, and G0 must hit the input Adler-32
    the compiled reference was fed for the anchor table (64x64 -> 15173ae6, 100x75 -> 189d0ecc)."""
    for gen in (0, 1):
        for (w, h, seed) in [(64, 64, 0x9E3779B9), (100, 75, 5), (517, 233, 0x9E3779B9 + 3), (1, 1, 1)]:
            assert np.array_equal(api.synth_image(gen, w, h, seed), po.gen_image(gen, w, h, seed=seed))
    assert np.array_equal(api.synth_plane(10000, 77), po.gen_plane(10000, seed=77))
    assert f"{po.adler32(api.synth_image(0, 64, 64)):08x}" == "15173ae6"
    assert f"{po.adler32(api.synth_image(0, 100, 75)):08x}" == "189d0ecc"
    assert f"{po.adler32(api.synth_image(1, 100, 75)):08x}" == "57cbae7c"


def test_forged_heads_are_rejected_before_anything_is_sized_from_them(po):
    """A 16 byte head is all an attacker needs to write: 2^32-1 x 2^32-1 pixels in 8 pixel tiles asks for
    ~10^17 tile records; w = h = 2^31 with 4 channels wraps w*h*channels to 0.  akoDecodeExt must answer with a
    status -- not abort, not allocate -- and must do so before it looks for a device (so this runs without one)."""
    def head(w, h, ch, tiles_log2_minus2, compression):
        flags = (ch - 1) | (0 << 4) | (0 << 6) | (0 << 8) | (compression << 10) | (tiles_log2_minus2 << 12)
        return np.frombuffer(bytes([65, 107, 111, 2]) + int(w).to_bytes(4, "little") + int(h).to_bytes(4, "little") +
                             int(flags).to_bytes(4, "little"), dtype=np.uint8)

    big = 0xFFFFFFFF
    for blob, allowed in [
        (head(big, big, 4, 1, 0), (13, 15)),              # Kagari, 8 px tiles: 2.9e17 tiles in a 16 byte blob
        (head(big, big, 16, 0, 2), (13, 15)),             # no compression, untiled: 2^68 bytes of stream
        (head(1 << 31, 1 << 31, 4, 0, 0), (13, 15)),      # w*h*ch*2 = 2^65 wraps
        (head(1 << 20, 1 << 20, 4, 1, 0), (13, 15)),      # 1.7e10 tiles, blob of 16 bytes
        (np.concatenate([head(4096, 4096, 4, 1, 0), np.zeros(1000, np.uint8)]), (15,)),  # 262144 tiles need >= 1.3 MB
        (np.concatenate([head(64, 64, 4, 0, 2), np.zeros(100, np.uint8)]), (15,)),       # raw stream shorter than w*h*ch*2
    ]:
        with pytest.raises(api.AkoError) as e:
            api.decode(blob)
        assert e.value.status in allowed, (blob[:16].tobytes().hex(), e.value.status)  # 13 no memory, 15 broken input


def test_library_owned_arrays_live_as_long_as_any_view():
    """api.decode() / api.pinned_empty() hand out arrays over memory the library allocated, without a copy.  NumPy
    collapses the .base of a plain view (np.asarray(a), a.view(np.ndarray), a slice) to the buffer at the root of the
    chain, so the memory must be freed when THAT goes, not when the first array does (use-after-free)."""
    import ctypes as C
    import gc

    from ako_amd import api

    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    freed = []

    def fake_free(ptr):
        freed.append(ptr.value)
        libc.free(ptr)

    p = libc.malloc(4096)
    a = api._owned_array(p, 4096, np.uint8, (32, 32, 4), fake_free)
    a[...] = 7
    plain, sl, vw = np.asarray(a), a[3:5], a.view(np.ndarray)
    del a
    gc.collect()
    assert freed == [] and int(plain.sum()) == 7 * 4096
    del plain, sl
    gc.collect()
    assert freed == []
    assert int(vw[0, 0, 0]) == 7
    del vw
    gc.collect()
    assert freed == [p]
