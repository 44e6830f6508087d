#!/bin/bash
# round 6, fourth A/B (policy phase, policy-shape build): f0 = product build; f1 / f2 = RO_BC_PREFETCH 1 / 2 (tail operands requested in front of the
# hidden layers); g1 = RO_BC_BPERM (last gather stage -> layer 0's operand by ds_bpermute); g2 / g3 = both
python tools/dump_rollout_state.py /tmp/ro_state5.bin 5 > /dev/null 2>&1
for rep in 1 2 3; do
  for b in scratch/ro_prof_f0 scratch/ro_prof_f1 scratch/ro_prof_f2 scratch/ro_prof_g1 scratch/ro_prof_g2 scratch/ro_prof_g3; do
    a=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 200 5 | grep "resident rollout\|fingerprint" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/' | tr '\n' ' ')
    c=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 20 40 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    echo "$rep $b  T200 $a us/step   T20 $c us/step"
  done
done
for b in scratch/ro_prof_f0 scratch/ro_prof_g2; do
  echo "== $b"; RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 20 20 | grep "stamp  0\|stamp  6\|stamp 1[2345]\|stamp  3"
done
