#!/bin/bash
# round 4: A/B of the resident kernel (scratch/ro_prof_base = round 3's sources, scratch/ro_prof_x* = candidates) on bench.py's own
# state 5 steps after a disc reset, then the launch anatomy of the current sources (tools/harness/ro_launch_prof.hip)
python tools/dump_rollout_state.py /tmp/ro_state5.bin 5 > /dev/null 2>&1
for rep in 1 2 3; do
  for b in scratch/ro_prof_base scratch/ro_prof_x*; do
    a=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 200 5 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    c=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 20 40 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    d=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 1 40 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    echo "$rep $b  T200 $a us/step   T20 $c us/step   T1 $d us"
  done
done
RO_STATE=/tmp/ro_state5.bin RO_WG_DUMP=gpurun_out/wg_times.txt scratch/ro_launch 256 100 3 "1 2 3 5 10 20 40" 30
RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 scratch/ro_prof_x1 256 100 3 20 20
