#!/usr/bin/env python3
"""profiles/traffic.json from the raw PMC passes of scripts/collect_traffic.sh (gpurun_out/traffic_<workload>/traffic_raw.json):
HBM bytes per launch of the level kernels and per whole step, for every workload given.

hbm_bytes = (F * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests
as 64 bytes for wide coalesced reads (MI355X_MICROARCH.md, HBM section): F = 2 for kernels that read 16 bytes per lane.  For
the dword-per-lane reads of the inverse kernels F comes from scripts/fetch_calib.hip run under the same counter
(profiles/r3_fetch_calibration.txt); the factor used is recorded beside every entry, the uncorrected sum too.
usage: make_traffic_json.py <out.json> <commit> <read factor for dword-per-lane kernels> <workload>=<traffic_raw.json>:<steps> ..."""
import json, re, sys, time

out_path, commit, f_dword = sys.argv[1], sys.argv[2], float(sys.argv[3])
KINDS = ["dd137", "cdf53", "haar"]


def label(name, workload):
    m = re.search(r"k_(forward|inverse)_u8_(?:lean|gray|rows)<(\d)", name)  # round 4: the lean / gray level-0 kernels report under the same record names
    if m:
        return ("fwd" if m.group(1) == "forward" else "inv") + "_stream_" + KINDS[int(m.group(2))] + "_u8:0"
    m = re.search(r"k_(forward|inverse)_stream_u8<(\d)", name)
    if m:
        return ("fwd" if m.group(1) == "forward" else "inv") + "_stream_" + KINDS[int(m.group(2))] + "_u8:0"
    m = re.search(r"k_(forward|inverse)_stream<(\d)", name)
    if m:  # the largest dispatch of the int16 kernels: level 1 behind a u8 level 0, level 0 of a planes workload
        return ("fwd" if m.group(1) == "forward" else "inv") + "_stream_" + KINDS[int(m.group(2))] + (":0" if workload == "lift4096" else ":1")
    m = re.search(r"k_fused2_(forward|inverse)<(\d)", name)
    if m:
        return ("fwd" if m.group(1) == "forward" else "inv") + "_fused2_" + KINDS[int(m.group(2))] + "_u8:0"
    return None


out = {"_how": __doc__.strip().split("\n\n")[1],
       "_source": {"collected": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes (scripts/collect_traffic.sh), " + time.strftime("%Y-%m-%d"),
                   "commit": commit, "read_factor_16B_per_lane": 2.0, "read_factor_dword_per_lane": f_dword}}
for arg in sys.argv[4:]:
    workload, rest = arg.split("=")
    path, steps = rest.rsplit(":", 1)
    steps = int(steps)
    raw = json.load(open(path))
    fetch, write = raw["fetch_by_dispatch"], raw["write_by_dispatch"]
    res, tot_f16, tot_f4, total_w = {}, 0.0, 0.0, 0.0
    for name in set(fetch) | set(write):
        f, w = fetch.get(name, []), write.get(name, [])
        dword_reader = "k_inverse_stream" in name or "k_fused2_inverse" in name or "k_inverse_u8_" in name  # a dword per lane and sub-band row
        if dword_reader:
            tot_f4 += sum(f)
        else:
            tot_f16 += sum(f)
        total_w += sum(w)
        lab = label(name, workload)
        if lab is None or (", false" in name and "_u8" in lab and "inverse" in name):  # (the exact-if-flagged re-run returns at once)
            continue
        fk, wk = (max(f) if f else 0.0), (max(w) if w else 0.0)  # the largest dispatch of a kernel name = its largest level
        if lab in res and res[lab]["fetch_KiB_raw"] + res[lab]["write_KiB_raw"] > fk + wk:
            continue
        F = f_dword if dword_reader else 2.0
        res[lab] = {"kernel": name[:90], "fetch_KiB_raw": fk, "write_KiB_raw": wk, "read_factor": F,
                    "hbm_bytes_per_launch": int((F * fk + wk) * 1024), "hbm_bytes_per_launch_uncorrected": int((fk + wk) * 1024)}
    out[workload] = res
    out[workload + "_whole_step"] = {"steps_counted": steps, "fetch_KiB_raw_per_step": (tot_f16 + tot_f4) / steps,
                                     "write_KiB_raw_per_step": total_w / steps,
                                     "hbm_bytes_per_step": int((2.0 * tot_f16 + f_dword * tot_f4 + total_w) * 1024 / steps),
                                     "hbm_bytes_per_step_uncorrected": int((tot_f16 + tot_f4 + total_w) * 1024 / steps)}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.endswith("_whole_step")}, indent=1))
