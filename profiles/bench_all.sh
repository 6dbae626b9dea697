#!/bin/bash
# Run ON THE GPU BOX (through gpurun, from the repo root): the bench lines DESIGN.md section 5 quotes, one JSON file each.
#   profiles/bench_all.sh r03
set -eo pipefail
tag=${1:-r05}
out=gpurun_out
python3 bench.py --steps 20 --warmup 5 > $out/${tag}_bench_f16x3.json 2> $out/${tag}_bench_f16x3.err
python3 bench.py --steps 10 --warmup 3 --frames 81 --batch 128 --no-cpu-baseline > $out/${tag}_bench_f16x3_T81.json 2>/dev/null
python3 bench.py --steps 10 --warmup 3 --frames 27 --batch 512 --seq2frame --no-cpu-baseline > $out/${tag}_bench_f16x3_s2f.json 2>/dev/null
python3 bench.py --steps 3 --warmup 1 --sampling 50 --graph --no-cpu-baseline --no-extras > $out/${tag}_bench_f16x3_S50_graph.json 2>/dev/null
python3 bench.py --steps 3 --warmup 1 --sampling 50 --no-cpu-baseline --no-extras > $out/${tag}_bench_f16x3_S50_eager.json 2>/dev/null
python3 bench.py --steps 10 --warmup 3 --precision bf16 --no-cpu-baseline --no-extras > $out/${tag}_bench_bf16.json 2>/dev/null
python3 bench.py --steps 10 --warmup 3 --frames 81 --batch 128 --precision bf16 --no-cpu-baseline --no-extras > $out/${tag}_bench_bf16_T81.json 2>/dev/null
python3 bench.py --steps 5 --warmup 2 --precision fp32 --no-cpu-baseline --no-extras > $out/${tag}_bench_fp32.json 2>/dev/null
for f in $out/${tag}_bench_*.json; do python3 -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['unit'], d['headline_under'], 'frac', d['roofline']['frac'])"; done
