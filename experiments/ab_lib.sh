#!/bin/bash
# A/B of two builds of the library on one box:  experiments/ab_lib.sh path/to/other/libd3d_hip.so [reps]
# (the in-tree library against another build, alternating; run on the GPU box -- it swaps the in-tree file of that copy)
other=$1; reps=${2:-2}
cur=diff3dhpe_amd/libd3d_hip.so
cp $cur /tmp/_lib_cur.so
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-selfcheck | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['by_kernel_ms_per_step']; print(d['value'], k['linear'], k['attn_spatial'], k['attn_temporal'])"; }
for r in $(seq $reps); do
  cp /tmp/_lib_cur.so $cur; a=$(run)
  cp $other $cur; b=$(run)
  echo "in-tree: $a   other: $b"
done
cp /tmp/_lib_cur.so $cur
