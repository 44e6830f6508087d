#!/bin/bash
# factored path (N > 256): split-bf16 hidden layers in the policy launch -- harness A/B, tests, bench at the cfg-3 shape
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/bf
{
for rep in 1 2; do
echo "== fp32 MFMA"; ./scratch/sp_prof_f32 64 1000 3 200 | tail -4
echo "== split-bf16"; ./scratch/sp_prof 64 1000 3 200 | tail -4
done
timeout 1500 python -m pytest tests/test_gpu_sparse.py -x -q 2>&1 | tail -3
python bench.py --episodes 64 --agents 1000 --taps 3 --no-cpu-baseline --no-roofline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('64 1000 3:', 'value %.3e' % d['value'], 'us/step %.2f' % (1e3 * d['ms_per_step']), 'parity ok', d['parity']['ok'], 'max_rel %.2e' % d['parity']['max_rel'], d['parity']['passed_on'])"
} > gpurun_out/bf/sp.log 2>&1
cat gpurun_out/bf/sp.log
