#!/bin/bash
# A/B of an environment switch on one box:  experiments/ab_env.sh D3D_NO_PN [reps]  -> pose-seq/s with the switch off / on, alternating
var=$1; reps=${2:-2}
for r in $(seq $reps); do
  a=$(python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-selfcheck | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['by_kernel_ms_per_step']; print(d['value'], k['linear'], k['attn_spatial'], k['attn_temporal'])")
  b=$(env $var=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-selfcheck | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['by_kernel_ms_per_step']; print(d['value'], k['linear'], k['attn_spatial'], k['attn_temporal'])")
  echo "default: $a   $var=1: $b"
done
