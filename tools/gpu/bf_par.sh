#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for cfg in "256 100 4 32 2" "256 200 4 32 2" "256 100 3 32 2"; do set -- $cfg; python bench.py --episodes $1 --agents $2 --taps $3 --hidden $4 --layers $5 --no-cpu-baseline --no-roofline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d['parity']
print('$1 $2 $3', {k: v for k, v in p.items() if k not in ('criterion', 'reference', 'paths')})
for k, v in p['paths'].items(): print('   ', k, v)"; done
