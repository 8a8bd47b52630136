#!/bin/bash
# A second build of the kernels beside the product library, for A/B runs through HK_LIB_PATH (tools/gpu_ab.sh):
#   tools/build_variant.sh <name> [-DFLAG=value ...]      -> build/lib_<name>.so      (the build directory is git-ignored and travels to the GPU box)
#   tools/build_variant.sh dbg -DHK_DEBUG_UTIL             -> the lane-slot probes (DStats::dbg) that tools/gpu_util.sh prints
set -e
name=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/hikari.jl_amd/csrc"
mkdir -p "$ROOT/build"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include"
/opt/rocm/bin/hipcc $F "$@" -c hk_kernels.hip -o /tmp/hk_kernels_$name.o &
/opt/rocm/bin/hipcc $F "$@" -x hip -c hk_api.cpp -o /tmp/hk_api_$name.o &
wait
make -s bvh_build.o light_bvh.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build/lib_$name.so" /tmp/hk_kernels_$name.o /tmp/hk_api_$name.o bvh_build.o light_bvh.o
echo "$ROOT/build/lib_$name.so"
