#!/bin/bash
# cfg/hidden_size.cfg's whole grid (n_layers 1..4 x hidden_size 4..128) at N = 100, K = 3: value and the path it ran on
cd "$GRAFT_REPO_ROOT"
for L in 1 2 3 4; do for H in 4 8 16 32 64 128; do
python bench.py --episodes 256 --agents 100 --taps 3 --hidden $H --layers $L --no-cpu-baseline --no-roofline --steps ${STEPS:-100} --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d['parity']
print('n_layers $L hidden_size %3d | value %.3e | %s | parity ok=%s %s | %s' % ($H, d['value'], ' '.join('%s %.3e' % (a, b['value']) for a, b in d['paths'].items() if a in ('two_launch', 'resident')),
      p['ok'], ' '.join('%s %.1e on %s' % (k, v['max_rel'], v['passed_on']) for k, v in p['paths'].items()), d['config']['weights'].replace('tests/golden/policies/', '')[:60]))"
done; done
