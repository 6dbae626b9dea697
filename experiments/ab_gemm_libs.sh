#!/bin/bash
# experiments/ab_gemm_libs.sh REPS NAME...  (on the GPU box): op-level GEMM timings (experiments/gemm_bench.py, the engine's own tile
# choice, real random operands) alternating over experiments/_libs/libd3d_NAME.so ("cur" = the in-tree build)
reps=$1; shift
cur=diff3dhpe_amd/libd3d_hip.so
cp $cur /tmp/_lib_cur.so
for r in $(seq $reps); do
  for n in "$@"; do
    if [ "$n" = cur ]; then cp /tmp/_lib_cur.so $cur; else cp experiments/_libs/libd3d_$n.so $cur; fi
    echo "== $n"; python experiments/gemm_bench.py 264384 0 2>&1 | grep -v "^$"
  done
done
cp /tmp/_lib_cur.so $cur
