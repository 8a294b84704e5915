#!/bin/bash
# per-kernel times of the bench step under the lockstep / workgroup-shape knobs (run on the GPU box)
run() { echo "== $*"; env "$@" TOP=6 python scripts/bench_nocheck.py; }
run AKO_HIP_LOCKSTEP=0 AKO_HIP_INV_PAIRS=1
run AKO_HIP_LOCKSTEP=1 AKO_HIP_INV_PAIRS=1
run AKO_HIP_LOCKSTEP=3 AKO_HIP_INV_PAIRS=1
run AKO_HIP_LOCKSTEP=3 AKO_HIP_INV_PAIRS=2
run AKO_HIP_LOCKSTEP=3 AKO_HIP_INV_PAIRS=4
run AKO_HIP_LOCKSTEP=3 AKO_HIP_INV_PAIRS=2 AKO_HIP_FWD_PAIRS=4
run AKO_HIP_LOCKSTEP=3 AKO_HIP_INV_PAIRS=2 AKO_HIP_FWD_PAIRS=1
run AKO_HIP_LOCKSTEP=2 AKO_HIP_INV_PAIRS=2
