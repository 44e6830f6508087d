#!/bin/bash
# bench.py at other shapes, lattice resets (the r02 table's conditions) -- quick regression check of the generic builds
for cfg in "256 200 4 32 2" "256 100 4 32 2" "256 100 2 32 2" "256 125 3 32 2" "256 50 2 32 2" "256 100 3 64 2" "256 100 3 128 1" "256 100 3 32 2"; do set -- $cfg; python bench.py --episodes $1 --agents $2 --taps $3 --hidden $4 --layers $5 --no-cpu-baseline --no-roofline --steps 100 --warmup 10 --init ${INIT:-grid} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1 $2 $3 hidden $4 x $5', 'value %.3e' % d['value'], {a: '%.3e' % b['value'] for a, b in d['paths'].items()}, 'parity', d['parity']['ok'], '%.2e' % d['parity']['max_rel'])"; done
