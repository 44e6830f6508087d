#!/bin/bash
# long launches: S1 mode statistics and time per step, candidate lists on (x1) / off (x0)
python tools/dump_rollout_state.py /tmp/ro_state5.bin 5 > /dev/null 2>&1
for b in scratch/ro_prof_x0 scratch/ro_prof_x1; do
  for T in 100 500 1000; do
    echo "== $b T=$T"; RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 $T 1 | grep -E "resident rollout|S1 modes"
  done
done
