#!/bin/bash
# the configuration of two_rank_repeat.sh as ONE process (global batch 6), repeated: MPJPE must not change from run to run
n=${1:-3}; shift
for i in $(seq $n); do
  python bench.py --gpus 1 --batch 6 --steps 1 --warmup 0 --frames 27 --sampling 3 --no-cpu-baseline "$@" 2>/dev/null \
    | python -c 'import sys, json; [print(json.loads(l)["mpjpe_vs_synthetic_gt"]) for l in sys.stdin if l.startswith("{")]'
done
