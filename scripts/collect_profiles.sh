#!/bin/bash
# Everything profiles/ holds for one round, in one GPU-box call:
#   rocprofv3 --kernel-trace --stats of the default bench command and of the one-step-in-flight mode,
#   the PMC traffic passes (scripts/collect_traffic.sh) and the SQ counter pass (scripts/collect_sq.sh).
# usage (GPU box):  bash scripts/collect_profiles.sh <tag>      -> gpurun_out/profiles_<tag>/
set -e
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for MODE in default inflight1; do
  EXTRA=""; [ $MODE = inflight1 ] && EXTRA="--inflight 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$MODE" -- python3 "$R/bench.py" --no-cpu-baseline $EXTRA > "$OUT/${MODE}_bench_under_rocprof.json" 2> "$OUT/$MODE.err" || { tail -5 "$OUT/$MODE.err"; exit 1; }
  cp "$(find "$OUT/$MODE" -name '*kernel_stats.csv' | head -1)" "$OUT/${MODE}_kernel_stats.csv"
  echo "$MODE: $(head -c 300 "$OUT/${MODE}_bench_under_rocprof.json")"
done
cd "$R"
bash scripts/collect_traffic.sh full8192 | tail -3
bash scripts/collect_sq.sh full8192 > "$OUT/sq_counters.txt"
cp gpurun_out/traffic_full8192/traffic_raw.json "$OUT/traffic_raw.json" 2>/dev/null || true
python3 bench.py > "$OUT/bench.json"
python3 bench.py --inflight 1 --no-cpu-baseline > "$OUT/bench_inflight1.json"
echo collected
