for st in 100 500; do python bench.py --episodes 64 --agents 1000 --taps 3 --no-cpu-baseline --no-roofline --steps $st --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
pf = d['parity']['paths']['factored']
print('64 1000 3 hidden 32 x 2, $st steps per call:', 'value %.3e' % d['value'], 'us/step %.2f' % (1e3 * d['ms_per_step']), 'paths', {a: '%.3e' % b['value'] for a, b in d['paths'].items()}, 'mean degree %.2f' % d['config']['mean_degree'], '| parity gate, factored step: ok', d['parity']['ok'], 'max_rel %.2e' % pf['max_rel'], 'passed on', pf['passed_on'].split(' (')[0], '-- well conditioned %d/%d episodes, the reference\'s own fp32 noise on these states %.1e' % (d['parity']['well_conditioned_episodes'], d['parity']['checked_episodes'], d['parity']['reference_fp32_noise']))
"; done
