#!/bin/bash
for rev in ed70058 372e9a1 829795e HEAD; do
  d=scratch/rev_$rev; [ "$rev" = HEAD ] && d=.
  for cfg in "64 2" "48 1"; do set -- $cfg
    extra="--init grid"; [ "$rev" = ed70058 ] && extra=""
    (cd $d && python bench.py --hidden $1 --layers $2 --no-cpu-baseline --no-roofline --steps 100 --warmup 10 $extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$rev hidden $1 x $2', {a: '%.3e' % b['value'] for a, b in d['paths'].items()})")
  done
done
