#!/bin/bash
# libako_cut<N>.so: the u8 RGBA kernels with parts of the forward arithmetic compiled out (-DAKO_MEASURE -DAKO_CUT=N), linked
# with the other translation units of the measurement build (run scripts/build_variant.sh meas --both -DAKO_MEASURE first)
set -e
cd "$(dirname "$0")/../.."
O=ako_amd/csrc/build
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -w"
for n in "$@"; do $CC -DAKO_MEASURE -DAKO_CUT=$n -c ako_amd/csrc/ako_u8_rgba.hip -o $O/ako_u8_rgba_cut$n.o & done
wait
for n in "$@"; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ako_amd/libako_cut$n.so $O/ako_plan_meas.o $O/ako_fused_meas.o $O/ako_u8_rgba_cut$n.o $O/ako_u8_rgb.hip.o $O/ako_u8_group_meas.o $O/ako_quant.c.o $O/ako_head.c.o $O/ako_misc.c.o $O/ako_kagari.c.o $O/ako_codec.c.o $O/ako_synth.c.o $O/ako_batch.c.o -lm -lpthread
done
echo built
