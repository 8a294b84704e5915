#!/bin/bash
# same-box A/B of the tiled workload (16384 x 16384 RGBA, CDF5/3 lossless) with the lean level-0 kernels (rows / wide strips) and
# without (AKO_HIP_LEAN=0: the general u8 kernels):  scripts/tiles_lean_ab.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { env "$@" python3 $R/bench.py --workload tiles16k --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['value_inflight1'], d['verified_against_reference_checksums'], [(k['name'],k['level'],k['isolated_ms']) for k in d['kernels'][:2]])"; }
for i in 1 2; do
for T in 512 256; do
echo "tiles $T lean:    $(run AKO_BENCH_TILES=$T)"
echo "tiles $T general: $(run AKO_BENCH_TILES=$T AKO_HIP_LEAN=0)"
done; done
