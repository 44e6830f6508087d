#!/bin/bash
# fused Actor forward: phase stamps (B = 256 and 1), the tests that cover it, the two-launch bench figures
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/af
{
./scratch/af_prof 256 100 | grep -v "^block\|^#"; ./scratch/af_prof 1 100 | grep -v "^block\|^#"
timeout 1200 python -m pytest tests/test_gpu_actor.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('value %.3e' % d['value'], {k: '%.3e' % v['value'] for k, v in d['paths'].items()}, d['parity']['ok'], d['parity']['max_rel'])
print({k: (v.get('avg_launch_ms'), v.get('frac')) for k, v in d['roofline']['dense_kernels'].items() if isinstance(v, dict)})"
} > gpurun_out/af/iter.log 2>&1
cat gpurun_out/af/iter.log
