#!/bin/bash
# same-box A/B of two builds of the library: scripts/ab_lib.sh <variant-name>   (ako_amd/libako_<name>.so vs libako.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$R/ako_amd/libako_$1.so
for i in 1 2 3; do
  echo "base    $(TOP=16 python3 $R/scripts/bench_nocheck.py 2>/dev/null)"
  echo "variant $(AKO_LIB_OVERRIDE=$V TOP=16 python3 $R/scripts/bench_nocheck.py 2>/dev/null)"
done
for L in "" $V "" $V; do
  AKO_LIB_OVERRIDE=$L python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lib=${L##*/}', d['value'], d['value_inflight1'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])"
done
