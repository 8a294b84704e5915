#!/bin/bash
# level-1 segment lengths around one full round of resident waves; order chosen so that both settings meet both
# placements (identical processes alternate between two) (GPU box)
run() { echo "== $*"; env "$@" TOP=4 python scripts/bench_nocheck.py | cut -c1-260; }
for f in 24 29 29 24 24 29 29 24 26 26; do run AKO_HIP_FLOOR_BIG=$f; done
