#!/bin/bash
# Level-0 / level-1 kernel time against the row-segment length (AKO_HIP_SEG_ROWS_BIG: levels >= 1024 columns). GPU box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for SR in 0 48 66 90 96 102 126 192; do
  if [ $SR = 0 ]; then unset AKO_HIP_SEG_ROWS_BIG; else export AKO_HIP_SEG_ROWS_BIG=$SR; fi
  echo "seg_rows_big=$SR $(python3 $R/scripts/bench_nocheck.py)"
done
