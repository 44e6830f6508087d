#!/bin/bash
# bench.py at the non-headline shapes, each on a TRAINED policy (tests/golden/policies, bench.load_weights) and the environment's
# own reset distribution (--init auto; INIT=grid for the lattice): value per path, and from the in-run parity gate per path the
# bound it passed on, the well-conditioned count and max_rel.  -> profiles/r04_other_configs.txt
# columns: episodes agents taps hidden layers env
STEPS=${STEPS:-100}
while read -r cfg; do
  [ -z "$cfg" ] && continue
  set -- $cfg
  python bench.py --episodes $1 --agents $2 --taps $3 --hidden $4 --layers $5 --env $6 --no-cpu-baseline --no-roofline --steps $STEPS --warmup 10 --init ${INIT:-auto} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
p = d['parity']
print('$1 x N=$2 K=$3 hidden $4 x $5 $6 | value %.3e |' % d['value'], ' '.join('%s %.3e' % (a, b['value']) for a, b in d['paths'].items()),
      '| weights:', d['config']['weights'].replace('tests/golden/policies/', ''))
print('    parity ok=%s well_conditioned %d/%d  reference_fp32_noise %.2e  ' % (p['ok'], p['well_conditioned_episodes'], p['checked_episodes'], p['reference_fp32_noise'])
      + '  '.join('%s: max_rel %.2e (well-conditioned %s) passed on %s' % (k, v['max_rel'], ('%.2e' % v['max_rel_well_conditioned']) if v['max_rel_well_conditioned'] is not None else '-', v['passed_on']) for k, v in p['paths'].items()))"
done <<CFGS
256 200 4 32 2 FlockingRelative-v0
256 200 4 32 2 FlockingLeader-v0
256 200 4 32 2 FlockingTwoFlocks-v0
64 1000 3 32 2 FlockingRelative-v0
256 100 4 32 2 FlockingRelative-v0
256 100 2 32 2 FlockingRelative-v0
256 100 1 32 2 FlockingRelative-v0
256 125 3 32 2 FlockingRelative-v0
256 50 2 32 2 FlockingRelative-v0
256 100 3 64 2 FlockingRelative-v0
256 100 3 128 1 FlockingRelative-v0
256 100 3 128 2 FlockingRelative-v0
256 100 3 128 3 FlockingRelative-v0
256 100 3 128 4 FlockingRelative-v0
256 100 3 32 2 FlockingStochastic-v0
256 100 1 32 2 FlockingLeader-v0
256 100 2 32 2 FlockingTwoFlocks-v0
256 100 3 32 2 FlockingRelative-v0
CFGS
