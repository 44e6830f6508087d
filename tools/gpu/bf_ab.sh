#!/bin/bash
# split-bf16 hidden layers of the compiled-in policy shape vs the fp32-MFMA form: harness A/B on the bench state + parity tests
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/bf
{
python tools/dump_rollout_state.py /tmp/ro_state5.bin 5 > /dev/null 2>&1
for rep in 1 2 3; do
  for b in scratch/ro_prof_f32 scratch/ro_prof_bf; do
    a=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 200 5 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    c=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 20 40 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    echo "$rep $b  T200 $a us/step   T20 $c us/step"
  done
done
RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 scratch/ro_prof_bf 256 100 3 20 20 | grep "stamp 1[2-5]\|stamp  [36] "
timeout 1500 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_headline_parity.py tests/test_gpu_collect.py -x -q 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('value %.3e' % d['value'], {k: '%.3e' % v['value'] for k, v in d['paths'].items()}, d['parity'])"
} > gpurun_out/bf/ab.log 2>&1
cat gpurun_out/bf/ab.log
