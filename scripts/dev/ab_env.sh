#!/bin/bash
# A/B of environment knobs with the real bench line: scripts/dev/ab_env.sh "VAR=a" "VAR=b" ...   (LIB=<variant> optional)
for rep in 1 2; do
for kv in "$@"; do
  env $kv ${LIB:+AKO_LIB_OVERRIDE=ako_amd/libako_$LIB.so} python bench.py --no-cpu-baseline ${BENCH_ARGS} | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$kv', 'value', d['value'], 'inflight1', d['value_inflight1'], 'level 0/1 isolated ms', [(k['name'], k['isolated_ms']) for k in d['kernels'] if k['level'] in (0, 1)][:5])"
done; done
