#!/bin/bash
# A/B of an engine option on one box:  experiments/ab_option.sh fused_postnorm=0 [reps] [bench args]
#   -> pose-seq/s and per-class kernel ms with the default engine / with the option, alternating runs of bench.py
opt=$1; shift; reps=${1:-2}; [ $# -gt 0 ] && shift
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["by_kernel_ms_per_step"]; print(d["value"], {a: round(b, 1) for a, b in k.items() if b > 5}, {a: b["avg_launch_ms"] for a, b in d["roofline"].get("by_gemm", {}).items()})'
for r in $(seq $reps); do
  a=$(python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-selfcheck --no-extras "$@" | python -c "$pick")
  b=$(python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-selfcheck --no-extras --option $opt "$@" | python -c "$pick")
  echo "default: $a   $opt: $b"
done
