#!/bin/bash
# Per-kernel launch times against the batch size (one-stream profiled pass of bench.py) beside the timed ms per sampling of the default
# (two-stream) and the one-stream engine: where the tile grids of small / mid-size batches lose CUs (NOTES round 6, small batches).
# usage: [BENCH_EXTRA="--option key=value ..."] experiments/batch_kernel_table.sh [B ...]   (on the GPU box; one line per batch size and stream count)
BS=${@:-1 2 3 4 6 8 12 16 24 32 48 64}
for b in $BS; do
  for st in 2 1; do
    python bench.py --batch $b --steps 6 --warmup 2 --streams $st --no-cpu-baseline --no-selfcheck --no-extras --profile-steps $((st == 2 ? 2 : 0)) $BENCH_EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
row = {'B': $b, 'streams': $st, 'ms': d['ms_per_step'], 'seq_per_s': d['value']}
if $st == 2:
    row['us'] = {k: round(v['avg_launch_ms'] * 1e3, 1) for k, v in r.get('by_gemm', {}).items()}
    for k in ('qkv_sattn', 'qkv_tattn'):
        if k in r: row['us'][k] = round(r[k]['avg_launch_ms'] * 1e3, 1)
print(json.dumps(row))"
  done
done
