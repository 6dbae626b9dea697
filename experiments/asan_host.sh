#!/bin/bash
# AddressSanitizer + UBSan build of the library's HOST code (device code untouched), CPU box only -- never on the GPU pool.
#   experiments/asan_host.sh            -> experiments/_libs/libd3d_asan.so, runs tests/test_abi_host.py against it, log under profiles/
# What it covers: everything tests/test_abi_host.py reaches without a device -- d3d_engine_create / set_weight / weight_info /
# d3d_ddim_times / d3d_num_windows / every error path of the C ABI (argument checks, last-error strings), the weight-name tables.
set -eo pipefail
cd "$(dirname "$0")/.."
out=/tmp/d3d_asan; mkdir -p $out experiments/_libs
SAN="-Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -g"
objs=""
for f in diff3dhpe_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  extra=""; [ $b = kernels_elem ] && extra="-fno-slp-vectorize"
  hipcc --offload-arch=gfx950 -O1 -std=c++17 -fPIC -Wno-unused-function $extra $SAN -c $f -o $out/$b.o &
  objs="$objs $out/$b.o"
  [ $(jobs -r | wc -l) -ge 4 ] && wait -n || true
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o experiments/_libs/libd3d_asan.so $objs
rt=$(dirname $(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so))/libclang_rt.asan-x86_64.so
[ -f "$rt" ] || rt=$(find /opt/rocm/lib/llvm -name 'libclang_rt.asan*x86_64*.so' | head -1)
echo "asan runtime: $rt"
cp diff3dhpe_amd/libd3d_hip.so /tmp/_lib_cur.so
cp experiments/_libs/libd3d_asan.so diff3dhpe_amd/libd3d_hip.so
trap 'cp /tmp/_lib_cur.so diff3dhpe_amd/libd3d_hip.so' EXIT
log=profiles/r06_asan_host.log
{ echo "# experiments/asan_host.sh: libd3d_hip.so host code under -fsanitize=address,undefined, tests/test_abi_host.py (CPU box)"; date -u; } > $log
LD_PRELOAD=$rt ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_abi_host.py -q -x -p no:cacheprovider 2>&1 | tee -a $log | tail -15
