#!/bin/bash
# build a variant of libako.so with extra hipcc flags:
#   scripts/build_variant.sh <name> [--plan | --both] <flags...>  -> ako_amd/libako_<name>.so
# The flags go to ako_fused.hip (default: the two-level workgroup kernels, 40 s), to ako_plan.hip (--plan, minutes) or to
# both and to ako_u8_group.hip + ako_u8_rgba.hip (--both: every translation unit that holds measurement code; e.g. -DAKO_MEASURE, which switches AKO_HIP_DBG and the *_memonly measurement kernels on); the other
# translation unit comes from the standard build (python -m ako_amd.build).
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
WHICH=fused
if [ "$1" = "--plan" ]; then WHICH=plan; shift; elif [ "$1" = "--both" ]; then WHICH=both; shift; fi
O=ako_amd/csrc/build
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -w"
PLAN=$O/ako_plan.hip.o; FUSED=$O/ako_fused.hip.o
if [ $WHICH != fused ]; then PLAN=$O/ako_plan_$NAME.o; $CC "$@" -c ako_amd/csrc/ako_plan.hip -o $PLAN & fi
if [ $WHICH != plan ]; then FUSED=$O/ako_fused_$NAME.o; $CC "$@" -c ako_amd/csrc/ako_fused.hip -o $FUSED & fi
GROUP=$O/ako_u8_group.hip.o; RGBA=$O/ako_u8_rgba.hip.o
if [ $WHICH = both ]; then
  GROUP=$O/ako_u8_group_$NAME.o; $CC "$@" -c ako_amd/csrc/ako_u8_group.hip -o $GROUP &
  RGBA=$O/ako_u8_rgba_$NAME.o; $CC "$@" -c ako_amd/csrc/ako_u8_rgba.hip -o $RGBA &
fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ako_amd/libako_$NAME.so $PLAN $FUSED $RGBA $O/ako_u8_rgb.hip.o $GROUP $O/ako_quant.c.o $O/ako_head.c.o $O/ako_misc.c.o $O/ako_kagari.c.o $O/ako_codec.c.o $O/ako_synth.c.o $O/ako_batch.c.o -lm -lpthread
echo built ako_amd/libako_$NAME.so
