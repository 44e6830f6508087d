#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
(timeout 600 python bench.py --dagger --episodes 16 --agents 1000 --steps 50 --updates 32 --batch-size 8 2>&1 | tail -5 | cut -c1-1500) > gpurun_out/dagger1000.log
(timeout 600 python bench.py --dagger --episodes 32 --agents 300 --steps 100 --updates 64 2>&1 | tail -3 | cut -c1-1500) >> gpurun_out/dagger1000.log
for st in 100 500; do python bench.py --episodes 64 --agents 1000 --taps 3 --no-cpu-baseline --no-roofline --steps $st --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('64 x 1000 K=3, steps $st:', 'value %.3e' % d['value'], 'us/step %.2f' % (1e3 * d['ms_per_step']), {a: '%.3e' % b['value'] for a, b in d['paths'].items()}, 'parity', d['parity']['ok'], '%.2e' % d['parity']['max_rel'])" >> gpurun_out/dagger1000.log; done
cat gpurun_out/dagger1000.log
