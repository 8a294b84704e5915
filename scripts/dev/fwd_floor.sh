#!/bin/bash
# the level-0 forward kernel with neither memory operations nor arithmetic (measurement build): what is left?
export TOP=2 AKO_HIP_GROUP=0 AKO_LIB_OVERRIDE=ako_amd/libako_meas.so AKO_HIP_DBG=$((24576+1966080))
echo "default"; python scripts/bench_nocheck.py
echo "U8_WAVES=4096"; AKO_HIP_U8_WAVES=4096 python scripts/bench_nocheck.py
echo "U8_WAVES=16384"; AKO_HIP_U8_WAVES=16384 python scripts/bench_nocheck.py
echo "U8_WAVES=32768"; AKO_HIP_U8_WAVES=32768 python scripts/bench_nocheck.py
echo "LOCKSTEP=0"; AKO_HIP_LOCKSTEP=0 python scripts/bench_nocheck.py
echo "W=4096"; W=4096 python scripts/bench_nocheck.py
echo "W=2048"; W=2048 python scripts/bench_nocheck.py
