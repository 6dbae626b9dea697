#!/bin/bash
# experiments/ab_option_bf16.sh REPS OPTION VALUE... [-- bench args]  (on the GPU box): alternate bench runs of the bf16 operand mode over the values
# of one engine option (round 5: "cu_split" at BASELINE configs[1]'s own shape)
reps=$1; opt=$2; shift 2
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done; [ "$1" = "--" ] && shift
for r in $(seq $reps); do
  for v in "${vals[@]}"; do
    python bench.py --precision bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-selfcheck --no-extras --option $opt=$v "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); g = d['roofline'].get('by_gemm', {}); k = d['roofline']['by_kernel_ms_per_step']
print('$opt=$v %8.2f seq/s  %8.2f ms | one-stream pass: ' % (d['value'], d['ms_per_step']) + '  '.join('%s %.4f' % (a, b['avg_launch_ms']) for a, b in g.items()) + '  attn %.1f/%.1f' % (k.get('attn_spatial', 0), k.get('attn_temporal', 0)))"
  done
done
