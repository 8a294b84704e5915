#!/bin/bash
# what bounds the level-0 forward strip kernel: its arithmetic, its loads, its stores? (measurement build, garbage output)
export TOP=3 AKO_HIP_GROUP=0
for w in 0 1; do
for dbg in 0 4096 16384 8192 24576 12288; do
  echo "wavelet=$w DBG=$dbg"; WAVELET=$w AKO_LIB_OVERRIDE=ako_amd/libako_meas.so AKO_HIP_DBG=$dbg python scripts/bench_nocheck.py
done; done
