#!/bin/bash
# instruction-cache counters of the bench kernels (one --pmc pass). GPU box only.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/icache
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_IFETCH --output-format csv -d "$OUT/p1" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 > "$OUT/p1.json" 2> "$OUT/p1.err" || { tail -5 "$OUT/p1.err"; exit 1; }
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
for name, c in acc.items():
    if "ako::" not in name: continue
    m = {k: max(v) for k, v in c.items()}
    print(name[:72]); print("   ", {k: int(v) for k, v in m.items()})
PY
