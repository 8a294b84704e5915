#!/bin/bash
# bench.py's value against the steps in flight, per workload: scripts/dev/inflight_sweep.sh [workloads...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for WL in ${@:-full8192}; do for n in 4 8; do for i in 1 2; do echo "$WL inflight $n: $(python3 $R/bench.py --workload $WL --no-cpu-baseline --inflight $n 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['value_inflight1'])")"; done; done; done
