#!/bin/bash
# two bench ranks sharing cuda:0 (gloo), repeated: the gathered MPJPE must not change from run to run
n=${1:-3}; shift
for i in $(seq $n); do
  D3D_BENCH_ONE_DEVICE=1 D3D_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port $((29600 + i)) bench.py --gpus 2 --batch 3 --steps 1 --warmup 0 --frames 27 --sampling 3 --no-cpu-baseline "$@" 2>/dev/null \
    | python -c 'import sys, json; [print(json.loads(l)["mpjpe_vs_synthetic_gt"]) for l in sys.stdin if l.startswith("{")]'
done
