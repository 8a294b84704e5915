#!/usr/bin/env python3
"""profiles/traffic.json from the raw PMC passes (gpurun_out/traffic_<workload>/traffic_raw.json, written by
scripts/collect_traffic.sh): HBM bytes per launch of the level kernels and per whole step.

hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports KiB, and on gfx950 FETCH_SIZE tallies 128-byte read
requests as 64 bytes for wide coalesced reads (MI355X_MICROARCH.md, HBM section) -- the uncorrected sum is kept beside it.
usage: make_traffic_json.py <traffic_raw.json> <workload> <steps counted in that run> <out.json>"""
import json, re, sys

raw = json.load(open(sys.argv[1]))
workload, steps, out_path = sys.argv[2], int(sys.argv[3]), sys.argv[4]
fetch, write = raw["fetch_by_dispatch"], raw["write_by_dispatch"]


def label(name):
    m = re.search(r"k_(forward|inverse)_stream_u8<(\d)", name)
    if m:
        return ("fwd" if m.group(1) == "forward" else "inv") + "_stream_" + ["dd137", "cdf53", "haar"][int(m.group(2))] + "_u8:0"
    m = re.search(r"k_(forward|inverse)_stream<(\d)", name)
    if m:
        return ("fwd" if m.group(1) == "forward" else "inv") + "_stream_" + ["dd137", "cdf53", "haar"][int(m.group(2))] + ":1"
    return None


res, total_f, total_w = {}, 0.0, 0.0
for name in set(fetch) | set(write):
    f, w = fetch.get(name, []), write.get(name, [])
    total_f += sum(f)
    total_w += sum(w)
    lab = label(name)
    if lab is None or ", false>" in name and "_u8" in lab:   # (the exact-if-flagged re-run returns at once)
        continue
    fk, wk = (max(f) if f else 0.0), (max(w) if w else 0.0)   # the largest dispatch of a kernel name = its largest level
    if lab in res and res[lab]["fetch_KiB_raw"] + res[lab]["write_KiB_raw"] > fk + wk:
        continue
    res[lab] = {"fetch_KiB_raw": fk, "write_KiB_raw": wk, "hbm_bytes_per_launch": int((2 * fk + wk) * 1024),
                "hbm_bytes_per_launch_uncorrected": int((fk + wk) * 1024)}
launches = max(len(v) for v in fetch.values())
out = {"_how": __doc__.strip().split("\n\n")[1],
       workload: res,
       workload + "_whole_step": {"steps_counted": steps, "fetch_KiB_raw_per_step": total_f / steps, "write_KiB_raw_per_step": total_w / steps,
                                  "hbm_bytes_per_step": int((2 * total_f + total_w) * 1024 / steps),
                                  "hbm_bytes_per_step_uncorrected": int((total_f + total_w) * 1024 / steps)}}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
