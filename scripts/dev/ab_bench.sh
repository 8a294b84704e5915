#!/bin/bash
# A/B of library variants with the real bench line (4 steps in flight / one), one box, every variant twice
for rep in 1 2; do
for lib in "$@"; do
  AKO_LIB_OVERRIDE=ako_amd/libako_$lib.so python bench.py --no-cpu-baseline ${BENCH_ARGS} | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
ks={(k['name'],k['level']):k['isolated_ms'] for k in d['kernels']}
print('lib=$lib', 'value', d['value'], 'inflight1', d['value_inflight1'], 'L0 fwd/inv isolated', ks.get(('fwd_stream_dd137_u8',0)), ks.get(('inv_stream_dd137_u8',0)))"
done; done
