#!/bin/bash
# level-1 segment lengths; every setting twice (identical processes alternate between two placements) (GPU box)
run() { echo "== $*"; env "$@" TOP=6 python scripts/bench_nocheck.py; }
for f in 24 18 16 12 24 18 16 12; do run AKO_HIP_FLOOR_BIG=$f; done
