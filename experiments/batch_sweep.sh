#!/bin/bash
# experiments/batch_sweep.sh B...   (on the GPU box): the headline shape at several batch sizes -- per-GEMM-kind launch times against the
# number of tile rounds each batch gives (round 5: what the partly filled last round costs each form, one-stream profiled pass)
for b in "$@"; do
  python bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-selfcheck --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); g = d['roofline'].get('by_gemm', {}); r = d['roofline']
M = $b * 243 * 17
print('B=%3d M=%7d  %7.2f seq/s  ms/seq %.4f | ' % ($b, M, d['value'], d['ms_per_step'] / $b) + '  '.join('%s %.4f ms (%.2f us/krow)' % (a, v['avg_launch_ms'], v['avg_launch_ms'] * 1e6 / M) for a, v in g.items())
      + ' | sattn %.4f tattn %.4f' % (r['qkv_sattn']['avg_launch_ms'], r['qkv_tattn']['avg_launch_ms'])
      + ' | rounds: proj %.2f fc1 %.2f fc2 %.2f sattn %.2f tattn %.2f' % (-(-M // 192) * 2 / 256, -(-M // 256) * 4 / 256, -(-M // 128) / 256, -(-$b * 243 // 15) * 8 / 256, $b * 17 * 8 / 256))"
done
