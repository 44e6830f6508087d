#!/bin/bash
# the reference's other sweeps at full size: cfg/n.cfg (n_agents 25..150 x k 1..4), cfg/n_twoflocks.cfg (50..250), cfg/rad.cfg (comm_radius 0.8..4),
# cfg/vel.cfg (v_max 0.5..5.5), cfg/dt.cfg (dt 0.0075..0.1)
cd "$GRAFT_REPO_ROOT"
run() {  # label, then bench.py arguments
  lbl=$1; shift
  python bench.py --episodes 256 --no-cpu-baseline --no-roofline --steps ${STEPS:-100} --warmup 10 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d['parity']
print('$lbl | value %.3e | %s | step_path %s | parity ok=%s %s | mean degree %s' % (d['value'], ' '.join('%s %.3e' % (a, b['value']) for a, b in d['paths'].items() if a in ('two_launch', 'resident', 'factored')),
      str(d['config'].get('step_path', ''))[:40], p['ok'], ' '.join('%s %.1e on %s' % (k, v['max_rel'], v['passed_on']) for k, v in p['paths'].items()), ('%.1f' % d['config']['mean_degree']) if d['config'].get('mean_degree') is not None else '-'))" || echo "$lbl | FAILED"
}
for N in 25 50 75 125 150; do for K in 1 2 3 4; do run "n.cfg n_agents $N k $K" --agents $N --taps $K; done; done
for N in 50 150 250; do for K in 1 4; do run "n_twoflocks.cfg n_agents $N k $K" --agents $N --taps $K --env FlockingTwoFlocks-v0; done; done
for R in 0.8 0.9 1.5 2.0 2.5 3.0 4.0; do for K in 1 3; do run "rad.cfg comm_radius $R k $K" --comm-radius $R --taps $K; done; done
for V in 0.5 1.5 2.5 3.5 4.5 5.5; do for K in 1 3; do run "vel.cfg v_max $V k $K" --v-max $V --taps $K; done; done
for D in 0.0075 0.025 0.05 0.075 0.1; do for K in 1 3; do run "dt.cfg dt $D k $K" --dt $D --taps $K; done; done
