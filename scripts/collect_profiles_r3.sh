#!/bin/bash
# Round-3 evidence in one GPU-box call (copy what it leaves in gpurun_out/r3/ into profiles/):
#   1. the box's memory rates incl. the tuned copy, 2. FETCH_SIZE / WRITE_SIZE calibration on known byte counts,
#   3. HBM traffic (two --pmc passes) of every bench workload, 4. rocprofv3 --kernel-trace --stats of the default bench and of
#   one step in flight, 5. SQ counters of the shipped level-0 kernels and of their memory-only variants (measurement build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3
mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$R"
hipcc --offload-arch=gfx950 -O3 scripts/hbm_rates.hip -o /tmp/hbm && /tmp/hbm > "$OUT/r3_hbm_rates.txt" 2>&1
hipcc --offload-arch=gfx950 -O3 scripts/fetch_calib.hip -o /tmp/fc
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/calib_$C" -- /tmp/fc > "$OUT/calib_$C.out" 2>&1
done
python3 - "$OUT" > "$OUT/r3_fetch_calibration.txt" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
GiB = 1 << 30
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(os.path.join(out, "calib_" + C, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == C:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    agg = {}
    for d, n, v in rows:
        agg.setdefault(d, [n, 0.0])[1] += v
    for d in sorted(agg):
        n, v = agg[d]
        if "k_calib" in n and v > 0:
            print("%-10s dispatch %3d  %-60s counter %12.1f KiB  bytes moved / (counter * 1024) = %.3f" % (C, d, n[:60], v, GiB / (v * 1024)))
PY
rm -rf "$OUT"/calib_FETCH_SIZE "$OUT"/calib_WRITE_SIZE
cd "$R"
for WL in full8192 rgb8192 batch4k lift4096 tiles16k_512 tiles16k_256; do
  bash scripts/collect_traffic.sh $WL > "$OUT/traffic_$WL.log" 2>&1 && cp "$R/gpurun_out/traffic_$WL/traffic_raw.json" "$OUT/traffic_raw_$WL.json"
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_default" -- python3 "$R/bench.py" --no-cpu-baseline > "$OUT/r3_default_bench_under_rocprof.json" 2> "$OUT/stats_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_inflight1" -- python3 "$R/bench.py" --no-cpu-baseline --inflight 1 > "$OUT/r3_inflight1_bench_under_rocprof.json" 2> "$OUT/stats_inflight1.err"
for M in default inflight1; do F=$(find "$OUT/stats_$M" -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" "$OUT/r3_${M}_kernel_stats.csv"; rm -rf "$OUT/stats_$M"; done
cd "$R"
bash scripts/collect_sq.sh full8192 > "$OUT/r3_sq_counters.txt" 2>&1
if [ -f "$R/ako_amd/libako_meas.so" ]; then
  for D in 16 48 80; do
    echo "== measurement build, AKO_HIP_DBG=$D (16: loads and stores without arithmetic, 48: loads only, 80: stores only)" >> "$OUT/r3_sq_counters_memonly.txt"
    SQ_COUNTERS="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" AKO_LIB_OVERRIDE=$R/ako_amd/libako_meas.so AKO_HIP_DBG=$D AKO_BENCH_NOCHECK=1 bash scripts/collect_sq.sh full8192 >> "$OUT/r3_sq_counters_memonly.txt" 2>&1
  done
fi
du -sh "$R/gpurun_out" | tail -1; ls -la "$OUT" | head -40
