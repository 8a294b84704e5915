#!/bin/bash
# a variant of libako.so in which ONLY the u8 RGBA translation unit (the level-0 kernels of the default workload) is rebuilt
# with extra hipcc flags; everything else comes from the standard build:
#   scripts/build_rgba_variant.sh <name> <flags...>   -> ako_amd/libako_<name>.so     (about a minute)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
O=ako_amd/csrc/build
RGBA=$O/ako_u8_rgba_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -w "$@" -c ako_amd/csrc/ako_u8_rgba.hip -o $RGBA
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ako_amd/libako_$NAME.so $O/ako_plan.hip.o $O/ako_copy.hip.o $O/ako_u8_gray.hip.o $RGBA $O/ako_u8_rgb.hip.o $O/ako_quant.c.o $O/ako_head.c.o $O/ako_misc.c.o $O/ako_kagari.c.o $O/ako_codec.c.o $O/ako_synth.c.o $O/ako_batch.c.o -lm -lpthread
echo built ako_amd/libako_$NAME.so
