#!/bin/bash
# timing experiments on the column-group forward kernel (measurement build: garbage output on purpose)
export TOP=2 AKO_HIP_GROUP=1
for dbg in 0 512 1024 1536 2048 3584; do
  echo "lib=meas DBG=$dbg"; AKO_LIB_OVERRIDE=ako_amd/libako_meas.so AKO_HIP_DBG=$dbg python scripts/bench_nocheck.py
done
echo "group off"; AKO_HIP_GROUP=0 TOP=2 python scripts/bench_nocheck.py
