#!/bin/bash
# which part of the level-0 forward kernel's arithmetic costs what (measurement build, garbage output; no loads, no stores)
export TOP=2 AKO_HIP_GROUP=0
B=24576
for extra in 0 131072 262144 524288 1048576 393216 917504 1966080; do
  echo "DBG=no-memory+$extra"; AKO_LIB_OVERRIDE=ako_amd/libako_meas.so AKO_HIP_DBG=$((B+extra)) python scripts/bench_nocheck.py
done
