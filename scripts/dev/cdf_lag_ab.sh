#!/bin/bash
# same-box A/B: CDF5/3 on its one-slot column pipeline (shipped) against the six-deep ring pipeline (scripts/build_rgba_variant.sh lag3 -DAKO_CDF_SHALLOW=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { env "$@" python3 $R/bench.py --workload tiles16k --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['value_inflight1'], d['verified_against_reference_checksums'], [(k['name'],k['level'],k['isolated_ms']) for k in d['kernels'][:2]])"; }
for i in 1 2; do for T in 512 256; do
echo "tiles $T one-slot pipeline: $(run AKO_BENCH_TILES=$T)"
echo "tiles $T ring pipeline:     $(run AKO_BENCH_TILES=$T AKO_LIB_OVERRIDE=$R/ako_amd/libako_lag3.so)"
done; done
