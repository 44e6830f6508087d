#!/bin/bash
# fused sim + state kernel: harness (timing, stamps) + the tests that cover it + the two-launch bench figure
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/fl
{
./scratch/fl_prof 256 100
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_sim_state or flock" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('value %.3e' % d['value'], {k: '%.3e' % v['value'] for k, v in d['paths'].items()}, d['parity']['ok'] if 'parity' in d else None)
print({k: (v.get('avg_launch_ms'), v.get('frac')) for k, v in d['roofline']['dense_kernels'].items() if isinstance(v, dict)})"
} > gpurun_out/fl/order.log 2>&1
cat gpurun_out/fl/order.log
