#!/bin/bash
# A/B of library variants on a bench workload: scripts/dev/ab_lib_workload.sh <workload> <lib> <lib> ...   (env passes through)
WL=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  AKO_LIB_OVERRIDE=ako_amd/libako_$lib.so python bench.py --workload $WL --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$WL lib=$lib', 'value', d['value'], 'inflight1', d['value_inflight1'], [(k['name'], k['level'], k['isolated_ms']) for k in d['kernels'] if k['level'] <= 1][:4])"
done; done
