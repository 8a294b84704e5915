#!/bin/bash
# bench.py's default line under environment settings, alternating, three rounds: scripts/dev/ab_bench_env.sh "" "VAR=value" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do
  for E in "$@"; do
    echo "[$E] $(env $E python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['value_inflight1'], d['roofline']['avg_launch_ms'])")"
  done
done
