#!/bin/bash
# same-box comparison of several builds: scripts/ab_libs.sh name1 name2 ...   ("base" = libako.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do
  for N in "$@"; do
    L=""; [ "$N" != base ] && L=$R/ako_amd/libako_$N.so
    echo "$N $(AKO_LIB_OVERRIDE=$L TOP=6 python3 $R/scripts/bench_nocheck.py 2>/dev/null)"
  done
done
for N in "$@"; do
  L=""; [ "$N" != base ] && L=$R/ako_amd/libako_$N.so
  AKO_LIB_OVERRIDE=$L python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$N', d['value'], d['value_inflight1'])"
done
