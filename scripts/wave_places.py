#!/usr/bin/env python3
"""Where and when the waves of the lean u8 level-0 kernels ran: birth, end and PLACE (XCC / SE / CU / SIMD / wave slot) of
every wave of one launch, from a -DAKO_STAMPS=2 build (two s_memtime and two s_getreg per wave, the slots as shipped).

    scripts/build_rgba_variant.sh places -DAKO_STAMPS=2                                   (here; the .so travels to the box)
    AKO_LIB_OVERRIDE=$PWD/ako_amd/libako_places.so python3 scripts/wave_places.py [out.txt] [raw.npz]    (on the box)

Questions it answers (profiles/r4_issue_model.txt ends with "the slowest wave lives 1.55 x the median"): what do the slow
waves have in common -- their role, their strip / segment (border code), their round (first / second), their XCC or CU, who
shared their SIMD -- and which CUs / XCCs end the launch.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ako_amd import api  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

NW = 16384


def pct(x, qs=(0, 5, 25, 50, 75, 95, 99, 100)):
    return " ".join(f"{np.percentile(x, q) / 1000:.0f}k" for q in qs)


def analyse(name, rows, out):
    b, e = rows[:, 0].astype(np.int64), rows[:, 1].astype(np.int64)
    ok = (e > 0) & (b > 0)
    rows, b, e = rows[ok], b[ok], e[ok]
    hw, xcc = rows[:, 2].astype(np.int64), rows[:, 3].astype(np.int64) & 15
    blk, wib, strip, seg, role, flags = (rows[:, k].astype(np.int64) for k in range(4, 10))
    t_dec, t_lc = seg >> 32, role >> 32  # (cycles from the kernel's entry to the end of decode_unit() / of lane_columns())
    seg, role = seg & 0xFFFFFFFF, role & 0xFFFFFFFF
    slot, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    top, drained = rows[:, 10].astype(np.int64), rows[:, 11].astype(np.int64)
    # s_memtime is a counter per CU (their values differ by milliseconds): one time base per CU, its first kernel entry
    cu_id = (((xcc * 8 + se) * 2 + sh) * 16 + cu)
    for k in np.unique(cu_id):
        m = cu_id == k
        t0 = int(top[m].min())
        b[m] -= t0
        e[m] -= t0
        top[m] -= t0
        drained[m] -= t0
    sane = (top > -10_000_000) & (e < 10_000_000_000)
    if not sane.all():
        print(f"  ({int((~sane).sum())} rows with stamps out of range dropped)", file=out)
        rows, b, e, top, drained, hw, xcc = rows[sane], b[sane], e[sane], top[sane], drained[sane], hw[sane], xcc[sane]
        blk, wib, strip, seg, role, flags = (rows[:, k].astype(np.int64) for k in range(4, 10))
        t_dec, t_lc = seg >> 32, role >> 32
        seg, role = seg & 0xFFFFFFFF, role & 0xFFFFFFFF
        slot, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    life = e - b
    span = e.max()
    print(f"\n== {name}: {len(b)} waves, first birth to last end {span} cycles; lifetimes (percentiles 0 5 25 50 75 95 99 100): {pct(life)}", file=out)
    edges = np.linspace(0, span, 21)
    mids = (edges[:-1] + edges[1:]) / 2
    print("  waves alive at the middle of each twentieth of the span: " + " ".join(str(int(((b <= t) & (e > t)).sum())) for t in mids), file=out)
    print("  ends by twentieth: " + " ".join(str(int(((e > lo) & (e <= hi)).sum())) for lo, hi in zip(edges[:-1], edges[1:])), file=out)
    first = b < np.percentile(b, 45)  # first round: born at launch
    print(f"  births: {int((b < span * 0.05).sum())} waves in the first 5 % of the span; first-round waves live {pct(life[first])}; later ones {pct(life[~first])}", file=out)
    print(f"  kernel entry -> decode_unit() done: {pct(t_dec)}; -> lane_columns() done: {pct(t_lc)}", file=out)
    print(f"  kernel entry -> first slot (unit decoding, tile descriptor, addresses): percentiles {pct(b - top)}; end -> last stores acknowledged: {pct(drained - e)}", file=out)
    # how long is a wave slot empty between two workgroups?
    key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd) * 16 + slot
    order = np.lexsort((top, key))
    same = key[order][1:] == key[order][:-1]
    gap = (top[order][1:] - drained[order][:-1])[same]
    gap_e = (b[order][1:] - e[order][:-1])[same]
    print(f"  wave slot empty between a wave's last acknowledged store and the next wave's kernel entry ({len(gap)} successions): percentiles {pct(gap)}; "
          f"between a wave's last slot and the next wave's first slot: {pct(gap_e)}", file=out)
    busy = (e - b).sum() / (len(np.unique(key)) * float(e.max() - b.min()))
    print(f"  wave slots used {len(np.unique(key))}; share of (slots x span) inside a wave's first..last slot: {busy:.3f}", file=out)
    # by role and border flags
    for r in (0, 1):
        for f in (0, 1, 2, 3):
            m = (role == r) & (flags == f)
            if m.sum():
                print(f"  role {r} border flags {f} (1 = left/right strip, 2 = top/bottom segment): {int(m.sum()):5d} waves, lifetimes {pct(life[m])}", file=out)
    # by place
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simd_key = cu_key * 4 + simd
    print(f"  places: {len(np.unique(xcc))} XCCs, {len(np.unique(cu_key))} CUs, {len(np.unique(simd_key))} SIMDs, wave slots used {sorted(np.unique(slot).tolist())}", file=out)
    per_x = [(x, int((xcc == x).sum()), e[xcc == x].max(), np.median(life[xcc == x])) for x in np.unique(xcc)]
    print("  per XCC (waves, last end on its CUs' own clocks as a fraction of the span, median lifetime): " + "  ".join(f"{x}: {n} {le / span:.3f} {ml / 1000:.0f}k" for x, n, le, ml in per_x), file=out)
    cu_end = np.array([e[cu_key == k].max() for k in np.unique(cu_key)])
    cu_n = np.array([(cu_key == k).sum() for k in np.unique(cu_key)])
    print(f"  per CU: waves min / median / max {cu_n.min()} / {int(np.median(cu_n))} / {cu_n.max()}; last end as a fraction of the span (percentiles 0 5 25 50 75 95 100): "
          + " ".join(f"{np.percentile(cu_end, q) / span:.3f}" for q in (0, 5, 25, 50, 75, 95, 100)), file=out)
    # who shared a SIMD with whom: time-weighted mean number of co-resident waves, and of role-0 co-residents, during each wave's life
    co = np.zeros(len(b))
    co0 = np.zeros(len(b))
    for k in np.unique(simd_key):
        idx = np.nonzero(simd_key == k)[0]
        bb, ee, rr = b[idx], e[idx], role[idx]
        ov = np.clip(np.minimum(ee[:, None], ee[None, :]) - np.maximum(bb[:, None], bb[None, :]), 0, None).astype(np.float64)
        np.fill_diagonal(ov, 0)
        co[idx] = ov.sum(1) / np.maximum(1, ee - bb)
        co0[idx] = (ov * (rr[None, :] == 0)).sum(1) / np.maximum(1, ee - bb)
    print(f"  co-resident waves on the wave's SIMD (time-weighted mean over its life): percentiles {' '.join(f'{np.percentile(co, q):.2f}' for q in (0, 5, 25, 50, 75, 95, 100))}", file=out)
    inner = flags == 0
    for lo, hi in ((0, 1.5), (1.5, 2.5), (2.5, 2.9), (2.9, 3.01)):
        m = inner & (co >= lo) & (co < hi)
        if m.sum() > 20:
            print(f"    interior waves with {lo}-{hi} co-residents: {int(m.sum()):5d}, lifetimes {pct(life[m])}", file=out)
    m3 = inner & (co > 2.9)
    for lo, hi in ((0, 0.75), (0.75, 1.25), (1.25, 1.75), (1.75, 2.25), (2.25, 3.01)):
        m = m3 & (co0 >= lo) & (co0 < hi)
        if m.sum() > 20:
            print(f"    ... of the fully shared ones, {lo}-{hi} role-0 co-residents: {int(m.sum()):5d} waves, lifetimes role 0 {pct(life[m & (role == 0)]) if (m & (role == 0)).sum() else '-'} | role 1 {pct(life[m & (role == 1)]) if (m & (role == 1)).sum() else '-'}", file=out)
    # the slowest 2 %: what are they?
    slow = life >= np.percentile(life, 98)
    print(f"  slowest 2 % ({int(slow.sum())} waves): role 0 share {np.mean(role[slow] == 0):.2f}, border flags {np.bincount(flags[slow], minlength=4).tolist()}, first-round share {np.mean(first[slow]):.2f}, "
          f"XCC histogram {np.bincount(xcc[slow], minlength=8).tolist()}, distinct CUs {len(np.unique(cu_key[slow]))}, strips (min / max) {strip[slow].min()} / {strip[slow].max()}, "
          f"co-residents median {np.median(co[slow]):.2f}", file=out)
    # workgroups: do the waves of a workgroup end together (lockstep)?  spread of ends inside a workgroup
    wg_spread = []
    for k in np.unique(blk)[:2000]:
        m = blk == k
        if m.sum() > 1:
            wg_spread.append(e[m].max() - e[m].min())
    if wg_spread:
        print(f"  spread of the ends inside a workgroup (first 2000 workgroups): median {np.median(wg_spread) / 1000:.1f}k, max {np.max(wg_spread) / 1000:.1f}k cycles", file=out)
    # does a workgroup's waves sit on distinct SIMDs of one CU?
    same_cu = np.mean([len(np.unique(cu_key[blk == k])) == 1 for k in np.unique(blk)[:500]])
    simds = np.mean([len(np.unique(simd[blk == k])) for k in np.unique(blk)[:500]])
    print(f"  workgroups on one CU: {same_cu:.2f}; distinct SIMDs per workgroup: {simds:.2f} of {np.mean([np.sum(blk == k) for k in np.unique(blk)[:500]]):.1f} waves", file=out)


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    L = api.lib()
    if not hasattr(L, "akoHipLeanStamps"):
        raise SystemExit("this library has no stamps: scripts/build_rgba_variant.sh places -DAKO_STAMPS=2 and set AKO_LIB_OVERRIDE")
    L.akoHipLeanStamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    w = h = 8192
    img = po.gen_image(0, w, h)
    s = api.settings(wavelet=0, compression=2, q=16, g=16)
    buf = (C.c_ulonglong * (20 + 2 * NW * 12))()
    with api.Plan(s, 4, w, h) as plan:
        d = torch.from_numpy(img).cuda().reshape(1, h, w, 4)
        st, back = plan.new_streams(), plan.new_images()
        for _ in range(5):
            plan.encode(d, st)
            plan.decode(st, back)
        plan.synchronize()
        assert L.akoHipLeanStamps(buf, 1) == 0
        plan.set_profiling(True)
        plan.encode(d, st)
        plan.decode(st, back)
        plan.synchronize()
        assert L.akoHipLeanStamps(buf, 5) == 0
        for r in plan.kernel_records(False) + plan.kernel_records(True):
            if r["level"] == 0 and "exact" not in r["name"]:
                print(f"{r['name']}: {r['ms'] * 1000:.1f} us (this launch, this build)", file=out)
    raw = np.array(buf[20:], dtype=np.uint64).reshape(2, NW, 12)
    if len(sys.argv) > 2:
        np.savez_compressed(sys.argv[2], raw=raw)
    analyse("forward (k_forward_u8_lean)", raw[0], out)
    analyse("inverse (k_inverse_u8_lean)", raw[1], out)


if __name__ == "__main__":
    main()
