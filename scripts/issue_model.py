#!/usr/bin/env python3
"""Where a wave of the lean u8 level-0 kernels spends its cycles, measured and modelled.

    scripts/build_rgba_variant.sh stamps -DAKO_STAMPS          (here; the .so travels to the box)
    AKO_LIB_OVERRIDE=$PWD/ako_amd/libako_stamps.so python3 scripts/issue_model.py [out.txt]     (on the box)

Measured: the in-kernel s_memtime stamps of a measurement build (ako_u8_lean.hip.h, AKO_STAMPS): per phase of a row slot the
cycles a wave spends there, summed over all waves of the level-0 launches of the bench's default workload (8192 x 8192 RGBA,
DD13/7 q16 g16), divided by waves x full slots.  Modelled: the instruction mix of the marked loops of the SHIPPED build
(scripts/isa_lint.py --json over ako_amd/csrc/build/*.s; pass the JSON as ISA_JSON=...) times the issue costs measured in
rounds 1-2 (profiles/r2_valu_issue_rates*.txt: plain VALU 2 cycles of a SIMD's pipe per wave instruction at four waves per
SIMD, DPP / SDWA / conversions 4), i.e. what a slot costs when four resident waves share one VALU pipe and nothing else
holds them up.
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ako_amd import api  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

FWD = ["wait for the slot's pixels", "pixels -> samples (YCoCg)", "row pass", "column pass", "gate + quantizer + pack", "stores (issue)",
       "trip barrier (lockstep)", "first trip (pipeline fill, 6 slots)"]
INV = ["wait for the slot's coefficients", "unpack + de-quantize + column pass", "row pass", "LDS writes + barrier", "LDS reads",
       "colour + clamp + pack", "pixel store (issue)", "first trip (pipeline fill, 6 slots)"]


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    L = api.lib()
    if not hasattr(L, "akoHipLeanStamps"):
        raise SystemExit("this library has no stamps: build it with scripts/build_rgba_variant.sh stamps -DAKO_STAMPS and set AKO_LIB_OVERRIDE")
    L.akoHipLeanStamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    w = h = 8192
    img = po.gen_image(0, w, h)
    s = api.settings(wavelet=0, compression=2, q=16, g=16)
    NW = 16384
    buf = (C.c_ulonglong * (20 + 4 * NW))()
    with api.Plan(s, 4, w, h) as plan:
        d = torch.from_numpy(img).cuda().reshape(1, h, w, 4)
        st = plan.new_streams()
        back = plan.new_images()
        for _ in range(3):
            plan.encode(d, st)
            plan.decode(st, back)
        plan.synchronize()
        assert L.akoHipLeanStamps(buf, 1) == 0
        reps = 10
        plan.set_profiling(True)
        for _ in range(reps):
            plan.encode(d, st)
            plan.decode(st, back)
        plan.synchronize()
        assert L.akoHipLeanStamps(buf, 3) == 0
        ms = {}
        for r in plan.kernel_records(False) + plan.kernel_records(True):
            if r["level"] == 0 and "exact" not in r["name"]:
                ms.setdefault(r["name"], []).append(r["ms"])
    vals = list(buf)
    print("in-kernel stamps of the lean u8 level-0 kernels (measurement build: stamps cost ~10 % themselves), 8192 x 8192 RGBA, DD13/7 q16 g16", file=out)
    for d_, names, kname in ((0, FWD, "fwd_stream_dd137_u8"), (1, INV, "inv_stream_dd137_u8")):
        v = vals[10 * d_:10 * d_ + 10]
        waves, life = v[9], v[8]
        t = sum(ms.get(kname, [0])) / max(1, len(ms.get(kname, [0])))
        print(f"\n{kname}: {t * 1000:.1f} us per launch (this build), {waves // reps} waves per launch, mean wave lifetime {life / max(1, waves):.0f} cycles", file=out)
        # one slot in six is stamped (phases of a slot), the trip barrier and the first trip every time
        scale = [6.0] * 8
        scale[7] = 1.0
        if d_ == 0:
            scale[6] = 1.0
        est = [v[i] * scale[i] for i in range(8)]
        tot = sum(est)
        for i, n in enumerate(names):
            print(f"  {n:40s} {est[i] / max(1, waves):10.0f} cycles per wave  {100.0 * est[i] / max(1, tot):5.1f} %", file=out)
        print(f"  {'(sum of the phases; lifetime above)':40s} {tot / max(1, waves):10.0f}", file=out)
        if t > 0:
            print(f"  cycles of wave lifetime per microsecond of launch: {life / max(1, waves) / (t * 1e3):.0f} (two rounds of resident waves: x2 = the clock, if every wave lived half the launch)", file=out)
    # when were the waves of the LAST launch alive?  (s_memtime is one counter for the whole chip)
    import numpy as np
    raw = np.array(buf[20:], dtype=np.uint64).reshape(2, NW, 2).astype(np.int64)
    for d_, kname in ((0, "forward"), (1, "inverse")):
        b, e = raw[d_, :, 0], raw[d_, :, 1]
        ok = e > 0
        b, e = b[ok], e[ok]
        t0 = b.min()
        b, e = b - t0, e - t0
        span = e.max()
        print(f"\n{kname}: last launch, {ok.sum()} waves, first birth to last end {span} ticks; wave lifetime min / median / max {np.min(e - b)} / {int(np.median(e - b))} / {np.max(e - b)}", file=out)
        edges = np.linspace(0, span, 21)
        alive = [int(((b <= t) & (e > t)).sum()) for t in (edges[:-1] + edges[1:]) / 2]
        print("  waves alive at the middle of each twentieth of that span: " + " ".join(str(a) for a in alive), file=out)
        late = np.sort(e)[-8:]
        print("  last eight waves end at (fraction of the span): " + " ".join(f"{x / span:.3f}" for x in late), file=out)
        print(f"  births: 50 % of the waves by {np.sort(b)[len(b) // 2] / span:.3f}, 90 % by {np.sort(b)[int(len(b) * 0.9)] / span:.3f} of the span", file=out)
    isa = os.environ.get("ISA_JSON")
    if isa and os.path.exists(isa):
        recs = json.load(open(isa))
        print("\nissue model: VALU pipe cycles per six-slot trip of ONE wave (plain 2, DPP / SDWA / conversion 4), x 4 resident waves = what a trip costs each of them when only the pipe holds them up", file=out)
        for r in recs:
            c = r.get("counts")
            if not c or "lean" not in r["loop"] or not r["loop"].endswith("h0_v0"):
                continue
            slow = c.get("v_mov_dpp", 0) + c.get("valu_dpp", 0) + c.get("valu_sdwa", 0) + c.get("valu_cvt", 0) - c.get("valu_sdwa", 0) * 0
            # (SDWA conversions are counted in both valu_sdwa and valu_cvt)
            slow = c.get("v_mov_dpp", 0) + c.get("valu_dpp", 0) + c.get("valu_cvt", 0)
            plain = c["valu"] - slow
            pipe = 2 * plain + 4 * slow
            print(f"  {r['loop']:28s} VALU {c['valu']}: plain {plain}, half-rate {slow} -> {pipe} pipe cycles per trip, x4 = {4 * pipe} cycles of wave time per trip, {4 * pipe / 6:.0f} per slot", file=out)


if __name__ == "__main__":
    main()
