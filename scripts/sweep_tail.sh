#!/bin/bash
# level-1 segment lengths around one full round of resident waves; every setting twice (placement alternates) (GPU box)
run() { echo "== $*"; env "$@" TOP=4 python scripts/bench_nocheck.py | cut -c1-260; }
for f in 24 30 32 40 24 30 32 40; do run AKO_HIP_FLOOR_BIG=$f; done
