#!/bin/bash
# round 5: A/B of the resident kernel with Verlet candidate lists in S1 (scratch/ro_prof_base = round 4's sources, x0 = the new sources
# with RO_VERLET=0, x1.. = skin / pass-count variants) on bench.py's own state 5 steps after a disc reset
python tools/dump_rollout_state.py /tmp/ro_state5.bin 5 > /dev/null 2>&1
for rep in 1 2 3; do
  for b in scratch/ro_prof_base scratch/ro_prof_x*; do
    a=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 200 5 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    c=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 20 40 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    d=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 1 40 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    echo "$rep $b  T200 $a us/step   T20 $c us/step   T1 $d us"
  done
done
for b in ${RO_STAMP_BINS:-scratch/ro_prof_x1 scratch/ro_st0}; do
  echo "== $b"; RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 20 20
done
# launch anatomy + per-episode durations and S1 modes, with and without the candidate lists
for b in scratch/ro_launch_0 scratch/ro_launch_v; do
  [ -x $b ] || continue
  echo "== $b"; RO_STATE=/tmp/ro_state5.bin RO_WG_DUMP=gpurun_out/wg_times_$(basename $b).txt $b 256 100 3 "1 20" 20
done
