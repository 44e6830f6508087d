#!/bin/bash
# round 6: A/B of the resident kernel's policy phase on wave pairs (scratch/ro_prof_p0 = RO_PAIR=0, p1 = the product build) on bench.py's own
# state 5 steps after a disc reset; fingerprints of the final state must agree (same arithmetic, bit for bit)
python tools/dump_rollout_state.py /tmp/ro_state5.bin 5 > /dev/null 2>&1
for rep in 1 2 3; do
  for b in ${RO_AB_BINS:-scratch/ro_prof_p0 scratch/ro_prof_p1}; do
    a=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 200 5 | grep "resident rollout\|fingerprint" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/' | tr '\n' ' ')
    c=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 20 40 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    d=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 1 40 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    echo "$rep $b  T200 $a us/step   T20 $c us/step   T1 $d us"
  done
done
for b in ${RO_AB_BINS:-scratch/ro_prof_p0 scratch/ro_prof_p1}; do
  echo "== $b"; RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 20 20
done
