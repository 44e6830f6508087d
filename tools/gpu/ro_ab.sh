#!/bin/bash
# A/B timing of harness variants (scratch/ro_prof_x*) on the bench state: 3 repetitions each, 200- and 20-step launches
python tools/dump_rollout_state.py /tmp/ro_state5.bin 5 > /dev/null 2>&1
for rep in 1 2 3; do
  for b in scratch/ro_prof_base scratch/ro_prof_x*; do
    a=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 200 5 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    c=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 20 40 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    echo "$rep $b  T200 $a us/step   T20 $c us/step"
  done
done
