#!/bin/bash
# same-box, same-library comparison of environment settings on the bench's step (kernel table + whole step):
#   scripts/ab_envs.sh "" "AKO_HIP_INTERIOR=0" ...      (each argument: space-separated VAR=value list, "" = defaults)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do
  for E in "$@"; do
    echo "[$E] $(env $E TOP=${TOP:-4} python3 $R/scripts/bench_nocheck.py 2>/dev/null)"
  done
done
for E in "$@"; do
  env $E python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$E]', d['value'], d['value_inflight1'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
