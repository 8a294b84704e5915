#!/bin/bash
# A/B of library variants on one box: per-kernel times (one step in flight) of the default workload; every variant twice
export TOP=2 AKO_HIP_GROUP=0
for rep in 1 2; do
for lib in "$@"; do
  echo "lib=$lib"; AKO_LIB_OVERRIDE=ako_amd/libako_$lib.so python scripts/bench_nocheck.py
done; done
