#!/bin/bash
# experiments/build_variant.sh NAME "-DFLAG ..." [SOURCE] : a variant of the library with extra flags for ONE source file
# (default kernels_gemm_x3p) -> experiments/_libs/libd3d_NAME.so (same-box A/B with experiments/ab_libs.sh on the GPU box)
set -e
# SOURCE may be a comma-separated list: every file of it is rebuilt with the flags
name=$1; flags=$2; srcs=${3:-kernels_gemm_x3p}
cd "$(dirname "$0")/.."
mkdir -p experiments/_libs /tmp/d3dvar_$name
objs=$(ls diff3dhpe_amd/build/*.o)
vars=""
for src in ${srcs//,/ }; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c diff3dhpe_amd/csrc/$src.hip -o /tmp/d3dvar_$name/$src.o &
  objs=$(echo "$objs" | grep -v "/$src.o")
  vars="$vars /tmp/d3dvar_$name/$src.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o experiments/_libs/libd3d_$name.so $objs $vars
echo built experiments/_libs/libd3d_$name.so
