#!/bin/bash
# Two ranks sharing the one test GPU: the update time is bimodal (LAB_NOTES).  Repeated runs of the two-rank DAGGER round under
# different queue settings of the HIP runtime (every run under `timeout`: a rank that hangs at teardown must not hold the box).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() {   # label, env assignments...
  label=$1; shift
  for i in 1 2 3 4 5; do
    out=$(env MGP_DIST_BACKEND=gloo "$@" timeout 120 python bench.py --dagger --gpus 2 --steps 300 --warmup 20 --episodes 128 2>/dev/null | grep "^{" | tail -1)
    echo "$out" | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); u = d['updates']
    print('$label run $i: %.1f us per update (exchange %s)' % (1e3 * u['ms_per_update'], u['exchange']))
except Exception as e:
    print('$label run $i: no result (%s)' % type(e).__name__)
"
  done
}
run "default" X=1
run "GPU_MAX_HW_QUEUES=1" GPU_MAX_HW_QUEUES=1
run "GPU_MAX_HW_QUEUES=2" GPU_MAX_HW_QUEUES=2
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
