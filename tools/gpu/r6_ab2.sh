#!/bin/bash
# round 6, second A/B: q0 = product build; q1..q3 = RO_BC_PRIO 1..3 (raised priority of the SIMDs' second tiles until the end of layer 0 /
# layer 1 / the whole policy phase); q4 = RO_MFMA_LARGEST_FIRST
python tools/dump_rollout_state.py /tmp/ro_state5.bin 5 > /dev/null 2>&1
for rep in 1 2 3; do
  for b in scratch/ro_prof_q0 scratch/ro_prof_q1 scratch/ro_prof_q2 scratch/ro_prof_q3 scratch/ro_prof_q4; do
    a=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 200 5 | grep "resident rollout\|fingerprint" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/' | tr '\n' ' ')
    c=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 20 40 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    echo "$rep $b  T200 $a us/step   T20 $c us/step"
  done
done
for b in scratch/ro_prof_q1 scratch/ro_prof_q2; do
  echo "== $b"; RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 20 20 | grep "stamp  0\|stamp  6\|stamp 1[2345]\|stamp  3"
done
