#!/bin/bash
# one iteration of resident-kernel work on the GPU box: parity tests, phase stamps, the two bench forms
# (build first:  python -m multiagent_gnn_policies_amd.build && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -o scratch/ro_prof tools/harness/ro_phase_prof.hip)
O=gpurun_out/ro_iter; mkdir -p $O
TESTS=${TESTS:-tests/test_gpu_rollout.py tests/test_gpu_headline_parity.py}
timeout 1500 python -m pytest $TESTS -x -q 2>&1 | tail -15 > $O/tests.txt
python tools/dump_rollout_state.py /tmp/ro_state5.bin 5 > /dev/null 2>&1
{ RO_CARRY=1 ./scratch/ro_prof 256 100 3 200; RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 ./scratch/ro_prof 256 100 3 20 20; } > $O/stamps.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench20.json 2> $O/bench20.err
timeout 300 python bench.py --no-cpu-baseline --no-roofline > $O/bench1000.json 2> $O/bench1000.err
cat $O/tests.txt; grep -v "^  stamp\|^cycles\|^exit\|^state" $O/stamps.txt; grep "stamp" $O/stamps.txt | tail -13
python - <<'PY'
import json
for f in ('bench20', 'bench1000'):
    try:
        d = json.load(open('gpurun_out/ro_iter/%s.json' % f))
        print(f, 'value %.4g' % d['value'], 'ms/step %.5f' % d['ms_per_step'], 'deg', d['config']['mean_degree'], 'parity', d['parity']['ok'], '%.3g' % d['parity']['max_rel'])
    except Exception as e:
        print(f, 'FAILED', e)
PY
