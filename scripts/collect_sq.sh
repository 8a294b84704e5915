#!/bin/bash
# SQ instruction / stall counters of the bench kernels (one --pmc pass, 8 SQ slots). GPU box only.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_${1:-full8192}
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
CTRS=${SQ_COUNTERS:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY}
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/p1" -- python3 "$R/bench.py" --workload ${1:-full8192} --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 > "$OUT/p1.json" 2> "$OUT/p1.err" || { tail -5 "$OUT/p1.err"; exit 1; }
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
    w = m.get("SQ_WAVES", 1)
    print(name[:70])
    print("   waves %d  VALU/wave %.0f  SALU/wave %.0f  wave_cycles/wave %.0f (x4 = cycles)  wait_any %.2f  wait_inst_any %.2f  active_valu %.2f  active_any %.2f" % (
        w, m.get("SQ_INSTS_VALU",0)/w, m.get("SQ_INSTS_SALU",0)/w, m.get("SQ_WAVE_CYCLES",0)/w,
        m.get("SQ_WAIT_ANY",0)/max(m.get("SQ_WAVE_CYCLES",1),1), m.get("SQ_WAIT_INST_ANY",0)/max(m.get("SQ_WAVE_CYCLES",1),1),
        m.get("SQ_ACTIVE_INST_VALU",0)/max(m.get("SQ_WAVE_CYCLES",1),1), m.get("SQ_ACTIVE_INST_ANY",0)/max(m.get("SQ_WAVE_CYCLES",1),1)))
    if "SQ_INSTS_VMEM_WR" in m or "SQ_INSTS_VMEM_RD" in m:
        print("   VMEM reads/wave %.0f  VMEM writes/wave %.0f" % (m.get("SQ_INSTS_VMEM_RD",0)/w, m.get("SQ_INSTS_VMEM_WR",0)/w))
PY
rm -rf "$OUT/p1"
