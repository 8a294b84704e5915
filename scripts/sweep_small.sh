#!/bin/bash
# Small-level kernel times against the shortest segment length / the prefetch form. GPU box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for CFG in "6 1" "6 0" "2 1" "2 0" "1 1" "3 1"; do
  set -- $CFG
  echo "seg_rows_small=$1 deep=$2 $(AKO_HIP_SEG_ROWS_SMALL=$1 AKO_HIP_DEEP=$2 python3 $R/scripts/bench_nocheck.py 2>/dev/null)"
done
