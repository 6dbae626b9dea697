#!/bin/bash
# experiments/ab_libs.sh REPS NAME...   (on the GPU box): alternate short bench runs over experiments/_libs/libd3d_NAME.so
# ("cur" = the in-tree build; BENCH_ARGS="--frames 81 --batch 128" etc. reach bench.py); prints pose-seq/s and the per-GEMM-kind launch times of every run
reps=$1; shift
cur=diff3dhpe_amd/libd3d_hip.so
cp $cur /tmp/_lib_cur.so
for r in $(seq $reps); do
  for n in "$@"; do
    if [ "$n" = cur ]; then cp /tmp/_lib_cur.so $cur; else cp experiments/_libs/libd3d_$n.so $cur; fi
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-selfcheck --no-extras $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); g = d['roofline'].get('by_gemm', {}); k = d['roofline']['by_kernel_ms_per_step']
print('%-10s %7.2f seq/s  linear %6.1f  ' % ('$n', d['value'], k['linear']) + '  '.join('%s %.4f' % (a, b['avg_launch_ms']) for a, b in g.items()) + '  attn %.1f/%.1f  qkv_sattn %.1f  qkv_tattn %.1f' % (k.get('attn_spatial', 0), k.get('attn_temporal', 0), k.get('qkv_sattn', 0), k.get('qkv_tattn', 0)))"
  done
done
cp /tmp/_lib_cur.so $cur
