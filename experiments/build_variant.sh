#!/bin/bash
# experiments/build_variant.sh NAME "-DFLAG ..." [SOURCE] : a variant of the library with extra flags for ONE source file
# (default kernels_gemm_x3p) -> experiments/_libs/libd3d_NAME.so (same-box A/B with experiments/ab_libs.sh on the GPU box)
set -e
name=$1; flags=$2; src=${3:-kernels_gemm_x3p}
cd "$(dirname "$0")/.."
mkdir -p experiments/_libs /tmp/d3dvar_$name
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c diff3dhpe_amd/csrc/$src.hip -o /tmp/d3dvar_$name/var.o
objs=$(ls diff3dhpe_amd/build/*.o | grep -v $src.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o experiments/_libs/libd3d_$name.so $objs /tmp/d3dvar_$name/var.o
echo built experiments/_libs/libd3d_$name.so
