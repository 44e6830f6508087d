#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_collect.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5
INIT=auto bash tools/gpu/other_cfgs.sh 2>&1 | head -3
python bench.py --dagger --steps 500 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('dagger collect value %.3e ms/step %.4f' % (d['value'], d['ms_per_step']), d['updates']['ms_per_update'])"
