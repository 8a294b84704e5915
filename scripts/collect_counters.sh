#!/bin/bash
# any list of PMC counters for the kernels of one bench workload, one rocprofv3 --pmc pass (GPU box only):
#   scripts/collect_counters.sh <workload> COUNTER1 COUNTER2 ...     -> per kernel, the largest dispatch's value of every counter
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=$1; shift
OUT=$R/gpurun_out/ctr_$WL
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/p1" -- python3 "$R/bench.py" --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 > "$OUT/p1.json" 2> "$OUT/p1.err" || { tail -5 "$OUT/p1.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p1", "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(lambda: defaultdict(float)); names = {}
    for row in csv.DictReader(open(f)):
        per[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"]); names[row["Dispatch_Id"]] = row["Kernel_Name"]
    for d, c in per.items():
        for k, v in c.items():
            acc[names[d]][k].append(v)
for name, c in sorted(acc.items()):
    if "ako::" not in name: continue
    print(name[:78])
    print("   " + "  ".join(f"{k} {max(v):.0f}" for k, v in sorted(c.items())))
PY
rm -rf "$OUT/p1"
