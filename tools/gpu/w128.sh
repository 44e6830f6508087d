#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_collect.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -15
for cfg in "128 1" "128 2" "64 2" "32 2"; do set -- $cfg; python bench.py --hidden $1 --layers $2 --no-cpu-baseline --no-roofline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('hidden $1 x $2', 'value %.3e' % d['value'], {a: '%.3e' % b['value'] for a, b in d['paths'].items()}, 'parity', d['parity']['ok'], '%.2e' % d['parity']['max_rel'], d['parity']['passed_on'])"; done
