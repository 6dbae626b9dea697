for b in 1 2 4 8 16; do
  for g in "" "--graph"; do
    python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-selfcheck --no-extras --profile-steps 0 $g 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('B=$b', '$g' or 'eager', d['value'], 'seq/s', d['ms_per_step'], 'ms/sampling', d['headline_under'])"
  done
done
