#!/bin/bash
# level-0 forward kernel with parts of its arithmetic COMPILED out (scripts/dev/build_cut.sh: -DAKO_MEASURE -DAKO_CUT=N in the
# u8 RGBA translation unit), loads and stores as shipped / taken out at run time (AKO_HIP_DBG 24576)
export TOP=6 AKO_HIP_GROUP=0
for n in 0 1 2 4 8 15; do
  echo "cut=$n (1 quantizer, 2 row pass, 4 column pass, 8 pixel decode)"; AKO_LIB_OVERRIDE=ako_amd/libako_cut$n.so python scripts/bench_nocheck.py | tr ',' '\n' | grep -A2 "fwd_stream_dd137_u8" | tr '\n' ' '; echo
  echo "cut=$n, no loads, no store instructions"; AKO_HIP_DBG=24576 AKO_LIB_OVERRIDE=ako_amd/libako_cut$n.so python scripts/bench_nocheck.py | tr ',' '\n' | grep -A2 "fwd_stream_dd137_u8" | tr '\n' ' '; echo
done
echo "shipped library"; python scripts/bench_nocheck.py
