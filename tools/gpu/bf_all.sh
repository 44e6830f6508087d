#!/bin/bash
# split-bf16 hidden layers in every build with widths <= 32: the whole GPU suite, then the shapes it touches
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/bf
{
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
for cfg in "256 100 3 32 2" "256 200 4 32 2" "256 100 4 32 2" "256 100 2 32 2" "256 125 3 32 2" "256 50 2 32 2" "256 100 3 16 1" "256 100 3 64 2"; do set -- $cfg; python bench.py --episodes $1 --agents $2 --taps $3 --hidden $4 --layers $5 --no-cpu-baseline --no-roofline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1 $2 $3 hidden $4 x $5', 'value %.3e' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'paths', {a: '%.3e' % b['value'] for a, b in d['paths'].items()}, 'parity ok', d['parity']['ok'], 'max_rel %.2e' % d['parity']['max_rel'])
"; done
python bench.py --dagger --steps 500 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('dagger collect value %.3e ms/step %.4f' % (d['value'], d['ms_per_step']), d['updates']['ms_per_update'])"
} > gpurun_out/bf/all.log 2>&1
cat gpurun_out/bf/all.log
