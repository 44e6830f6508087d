#!/bin/bash
# step 7 of tools/regen_profiles.sh alone (the factored path, N > 256) into gpurun_out/final/
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R; export TMPDIR=/tmp
{ ./scratch/sp_prof_stamps 64 1000 3 200; ./scratch/sp_prof 64 1000 3 200; ./scratch/sp_prof 256 300 3 200; ./scratch/sp_prof 64 1000 4 200; } > $O/factored_step_stamps.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/trace_sp -o sp -- $R/scratch/sp_prof 64 1000 3 200 > /dev/null 2>&1)
TS=$(find $O/trace_sp -name "*results.db" | head -1)
python tools/rocpd_stats.py $TS | head -8 > $O/factored_kernel_trace.txt 2>&1
for st in 100 500; do python bench.py --episodes 64 --agents 1000 --taps 3 --no-cpu-baseline --no-roofline --steps $st --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
p = d['parity']
print('64 1000 3 hidden 32 x 2, $st steps per call:', 'value %.3e' % d['value'], 'us/step %.2f' % (1e3 * d['ms_per_step']), 'paths', {a: '%.3e' % b['value'] for a, b in d['paths'].items()}, 'mean degree %.2f' % d['config']['mean_degree'])
print('    parity ok=%s well_conditioned %d/%d  reference_fp32_noise %.2e  ' % (p['ok'], p['well_conditioned_episodes'], p['checked_episodes'], p['reference_fp32_noise'])
      + '  '.join('%s: max_rel %.2e (well-conditioned %s) passed on %s' % (k, v['max_rel'], ('%.2e' % v['max_rel_well_conditioned']) if v['max_rel_well_conditioned'] is not None else '-', v['passed_on']) for k, v in p['paths'].items()))
" >> $O/factored_kernel_trace.txt; done
python bench.py --dagger --episodes 64 --agents 1000 --steps 200 --warmup 10 --updates 64 2> $O/dagger_round_n1000.err | grep "^{" > $O/dagger_round_n1000.json
python bench.py --dagger --episodes 256 --agents 300 --steps 200 --warmup 10 --updates 256 2> $O/dagger_round_n300.err | grep "^{" > $O/dagger_round_n300.json
python bench.py --episodes 64 --agents 1000 --taps 3 --hidden 32 --layers 2 --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = {a: (round(v['avg_launch_ms']*1e3,1), round(v['GBps'])) for a, v in d.get('kernels', {}).items()}
print('64 1000 3 hidden 32 x 2', 'value %.3e' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'paths', {a: '%.3e' % b['value'] for a, b in d['paths'].items()}, 'parity ok', d['parity']['ok'], 'max_rel %.2e' % d['parity']['max_rel'], k, d['config']['state_finite'])
" > $O/other_configs_n1000.txt
rm -rf $O/trace_sp
