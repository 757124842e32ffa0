#!/bin/bash
# tools/build_variant.sh NAME [-DMACRO[=V] ...] — an experimental build of the SAME sources with extra macros
# (kernel ablations / tuning knobs: LPMP_ABLATE_REDUCE, LPMP_ABLATE_LB_TRACK, LPMP_KMAX16=4, LPMP_PK_WPE=4 ...) into
# build/exp/liblpmp_engine_NAME.so (always with -DLPMP_EXPERIMENT_BUILD: kernels.hip refuses LPMP_ABLATE_* without it); select it with LPMP_ENGINE_SO=<that path> (lp_mp_amd/engine.py).  build/ is
# git-ignored but travels to the GPU box.
set -eu
NAME=$1; shift
cd "$(dirname "$0")/../lp_mp_amd/csrc"
mkdir -p ../../build/exp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-strict-aliasing -Wno-unused-function -DLPMP_EXPERIMENT_BUILD "$@" \
  -o ../../build/exp/liblpmp_engine_$NAME.so kernels.hip engine.cpp plan.cpp boundary.hip graph.cpp
echo build/exp/liblpmp_engine_$NAME.so
