#!/bin/bash
# Round-3 evidence of the final build in one GPU-box call (copy what it leaves in gpurun_out/r3f/ into profiles/):
#   1. what bounds the level-0 kernels: measurement build with loads / stores / parts of the arithmetic taken out
#   2. bench lines of every workload, 3. rocprofv3 --kernel-trace --stats (default and one step in flight), 4. SQ counters,
#   5. HBM traffic of the default workload
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3f
mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$R"
{
  echo "# scripts/dev/fwd_variants.sh -- level-0 forward strip kernel, measurement build, AKO_HIP_DBG bits: 4096 every store dropped (out of range), 16384 no store instructions, 8192 no pixel loads (constant pixels), 24576 neither, 12288 no loads + dropped stores"
  bash scripts/dev/fwd_variants.sh 2>&1 | grep -v amdgpu.ids
  echo; echo "# scripts/dev/inv_variants.sh -- level-0 inverse: 32768 no coefficient loads (constants), 65536 no pixel stores (and no colour inverse / packing), 98304 neither"
  bash scripts/dev/inv_variants.sh 2>&1 | grep -v amdgpu.ids
  echo; echo "# scripts/dev/fwd_cuts.sh -- parts of the forward arithmetic compiled out"
  bash scripts/dev/fwd_cuts.sh 2>&1 | grep -v amdgpu.ids
  echo; echo "# scripts/dev/grp_variants.sh -- column-group forward kernel (AKO_HIP_GROUP=1): 512 no barrier, 1024 no drain (no stores), 1536 neither, 2048 no row-buffer writes, 3584 none of the three"
  AKO_HIP_GROUP=1 bash scripts/dev/grp_variants.sh 2>&1 | grep -v amdgpu.ids
} > "$OUT/r3_level0_without_memory.txt"
for WL in full8192 batch4k rgb8192 lift4096 tiles16k; do
  python3 bench.py --workload $WL $( [ $WL = full8192 ] || echo --no-cpu-baseline ) > "$OUT/r3_bench_$WL.json" 2> "$OUT/bench_$WL.err"
done
AKO_BENCH_TILES=256 python3 bench.py --workload tiles16k --no-cpu-baseline > "$OUT/r3_bench_tiles16k_256.json" 2>> "$OUT/bench_tiles16k.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_default" -- python3 "$R/bench.py" --no-cpu-baseline > "$OUT/r3_default_bench_under_rocprof.json" 2> "$OUT/stats_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_inflight1" -- python3 "$R/bench.py" --no-cpu-baseline --inflight 1 > "$OUT/r3_inflight1_bench_under_rocprof.json" 2> "$OUT/stats_inflight1.err"
for M in default inflight1; do F=$(find "$OUT/stats_$M" -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" "$OUT/r3_${M}_kernel_stats.csv"; rm -rf "$OUT/stats_$M"; done
cd "$R"
bash scripts/collect_sq.sh full8192 > "$OUT/r3_sq_counters.txt" 2>&1
bash scripts/collect_traffic.sh full8192 > "$OUT/traffic_full8192.log" 2>&1 && cp "$R/gpurun_out/traffic_full8192/traffic_raw.json" "$OUT/r3_traffic_raw_full8192.json"
rm -rf "$R/gpurun_out/traffic_full8192" "$R/gpurun_out/sq_full8192"
du -sh "$R/gpurun_out" | tail -1; ls "$OUT"
