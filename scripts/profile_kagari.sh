#!/bin/bash
# rocprofv3 kernel stats of the device entropy stage (scripts/e2e_encode.py). GPU box only.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/kagari_prof
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/run" -- python3 "$R/scripts/e2e_encode.py" > "$OUT/e2e.json" 2> "$OUT/err.txt" || { tail -5 "$OUT/err.txt"; exit 1; }
cp "$(find "$OUT/run" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
grep -i "k_kg\|Name" "$OUT/kernel_stats.csv"
cat "$OUT/e2e.json"
