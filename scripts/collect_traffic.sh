#!/bin/bash
# HBM traffic of every kernel of the bench from the PMC counters, per MI355X_MICROARCH.md (HBM section):
# FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots), so two --pmc runs; counters only together
# with --kernel-trace.  Run on the GPU box:  gpurun -- 'bash scripts/collect_traffic.sh full8192'
set -e
WL=${1:-full8192}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/traffic_$WL
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -- python3 "$R/scripts/traffic_step.py" "$WL" > "$OUT/$C.json" 2> "$OUT/$C.err" || { tail -5 "$OUT/$C.err"; exit 1; }
done
python3 "$R/scripts/parse_traffic.py" "$OUT" "$WL"
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"  # (raw per-dispatch counter files: tens of MiB; traffic_raw.json holds what was read from them)
