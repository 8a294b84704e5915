#!/bin/bash
# Effective shader clock of the bench kernels: GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (MI355X_MICROARCH.md, "DVFS
# give-back": the quotient reads high on dispatches shorter than ~0.3 ms, so the default workload here is four 8192 x 8192
# images per launch).  GPU box only:  gpurun -- 'bash scripts/collect_clock.sh [workload]'
set -e
WL=${1:-full8192x4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/clock_$WL
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p" -- python3 "$R/scripts/traffic_step.py" "$WL" > "$OUT/p.json" 2> "$OUT/p.err" || { tail -5 "$OUT/p.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
dur = {}
for f in glob.glob(os.path.join(out, "p", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), row["Kernel_Name"])
act = defaultdict(float)
for f in glob.glob(os.path.join(out, "p", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[row["Dispatch_Id"]] += float(row["Counter_Value"])
per = defaultdict(list)
for d, a in act.items():
    if d in dur and dur[d][0] > 0 and "ako::" in dur[d][1]:
        per[dur[d][1]].append((dur[d][0], a / 8.0 / dur[d][0]))
print("kernel, launches, mean duration us, GRBM_GUI_ACTIVE / 8 / duration = effective clock GHz (min .. max)")
for k, v in sorted(per.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    ds = [x[0] for x in v]; cs = [x[1] for x in v]
    print(f"{k[:90]:90s} {len(v):3d} {sum(ds)/len(ds)/1e3:9.1f} {sum(cs)/len(cs):6.3f} ({min(cs):.3f} .. {max(cs):.3f})")
PY
rm -rf "$OUT/p"
