#!/bin/bash
# what bounds the level-0 inverse strip kernel: its arithmetic, its loads, its stores? (measurement build, garbage output)
# bit 16 (no pixel stores) also skips the colour inverse and packing of the row, which live behind the same test
export TOP=4 AKO_HIP_GROUP=0
for w in 0 1; do
for dbg in 0 32768 65536 98304; do
  echo "wavelet=$w DBG=$dbg"; WAVELET=$w AKO_LIB_OVERRIDE=ako_amd/libako_meas.so AKO_HIP_DBG=$dbg python scripts/bench_nocheck.py
done; done
