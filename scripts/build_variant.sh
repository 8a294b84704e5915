#!/bin/bash
# a variant of libako.so with extra hipcc flags in the chosen translation units, everything else from the standard build
# (python -m ako_amd.build first):
#   scripts/build_variant.sh <name> [--plan | --rgba | --all] <flags...>   -> ako_amd/libako_<name>.so
# --plan: ako_plan.hip (minutes); --rgba (default): ako_u8_rgba.hip, the level-0 kernels of the default workload (a minute);
# --all: both and ako_u8_rgb.hip (e.g. -DAKO_MEASURE, which switches AKO_HIP_DBG and the *_memonly measurement kernels on).
# The experimental routes (AKO_HIP_FUSE2, AKO_HIP_GROUP) have a build of their own: AKO_BUILD_EXPERIMENTAL=1 python -m ako_amd.build
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
WHICH=rgba
case "$1" in --plan) WHICH=plan; shift;; --rgba) WHICH=rgba; shift;; --all|--both) WHICH=all; shift;; esac
O=ako_amd/csrc/build
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -w"
PLAN=$O/ako_plan.hip.o; RGBA=$O/ako_u8_rgba.hip.o; RGB=$O/ako_u8_rgb.hip.o
if [ $WHICH != rgba ]; then PLAN=$O/ako_plan_$NAME.o; $CC "$@" -c ako_amd/csrc/ako_plan.hip -o $PLAN & fi
if [ $WHICH != plan ]; then RGBA=$O/ako_u8_rgba_$NAME.o; $CC "$@" -c ako_amd/csrc/ako_u8_rgba.hip -o $RGBA & fi
if [ $WHICH = all ]; then RGB=$O/ako_u8_rgb_$NAME.o; $CC "$@" -c ako_amd/csrc/ako_u8_rgb.hip -o $RGB & fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ako_amd/libako_$NAME.so $PLAN $O/ako_copy.hip.o $O/ako_u8_gray.hip.o $RGBA $RGB $O/ako_quant.c.o $O/ako_head.c.o $O/ako_misc.c.o $O/ako_kagari.c.o $O/ako_codec.c.o $O/ako_synth.c.o $O/ako_batch.c.o -lm -lpthread
echo built ako_amd/libako_$NAME.so
