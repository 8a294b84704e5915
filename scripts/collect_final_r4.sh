#!/bin/bash
# Round-4 evidence of the final build in one GPU-box call (copy what it leaves in gpurun_out/r4f/ into profiles/):
#   1. bench lines of every workload, 2. rocprofv3 --kernel-trace --stats of the default bench and of one step in flight,
#   3. SQ counters, 4. effective clock (GRBM_GUI_ACTIVE), 5. HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of every workload
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4f
mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$R"
for WL in full8192 batch4k rgb8192 lift4096 tiles16k; do
  python3 bench.py --workload $WL $( [ $WL = full8192 ] || echo --no-cpu-baseline ) > "$OUT/r4_bench_$WL.json" 2> "$OUT/bench_$WL.err"
  echo "bench $WL done"
done
AKO_BENCH_TILES=256 python3 bench.py --workload tiles16k --no-cpu-baseline > "$OUT/r4_bench_tiles16k_256.json" 2>> "$OUT/bench_tiles16k.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_default" -- python3 "$R/bench.py" --no-cpu-baseline > "$OUT/r4_default_bench_under_rocprof.json" 2> "$OUT/stats_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_inflight1" -- python3 "$R/bench.py" --no-cpu-baseline --inflight 1 > "$OUT/r4_inflight1_bench_under_rocprof.json" 2> "$OUT/stats_inflight1.err"
for M in default inflight1; do F=$(find "$OUT/stats_$M" -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" "$OUT/r4_${M}_kernel_stats.csv"; rm -rf "$OUT/stats_$M"; done
echo "kernel stats done"
cd "$R"
bash scripts/collect_sq.sh full8192 > "$OUT/r4_sq_counters.txt" 2>&1
bash scripts/collect_clock.sh full8192x4 > "$OUT/r4_effective_clock.txt" 2>&1
for WL in full8192 rgb8192 batch4k lift4096 tiles16k_512 tiles16k_256; do
  bash scripts/collect_traffic.sh $WL > "$OUT/traffic_$WL.log" 2>&1 && cp "$R/gpurun_out/traffic_$WL/traffic_raw.json" "$OUT/r4_traffic_raw_$WL.json"
  rm -rf "$R/gpurun_out/traffic_$WL"
  echo "traffic $WL done"
done
rm -rf "$R/gpurun_out/sq_full8192" "$R/gpurun_out/clock_full8192x4"
ls "$OUT"
