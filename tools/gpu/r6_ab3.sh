#!/bin/bash
# round 6, third A/B: (a) flock_advance_kernel with the slice's DMA waves starting FA_DMA_DELAY x 64 cycles late (scratch/fl_prof_d<delay>);
# (b) resident kernel, S2 with the gather group at raised priority (s0 = product build, s1 / s2 = RO_S2_PRIO 1 / 2)
for rep in 1 2; do for d in 0 3 6 12 24; do echo "rep $rep FA_DMA_DELAY=$d: $(./scratch/fl_prof_d$d 256 100 | grep -i "advance\|fused" | head -2 | tr '\n' ' ')"; done; done
echo "== stamps, delay 0"; ./scratch/fl_prof_d0 256 100 | grep "stamp" | head -16
echo "== stamps, delay 6"; ./scratch/fl_prof_d6 256 100 | grep "stamp" | head -16
for d in 0 6; do echo "B=2048 delay $d: $(./scratch/fl_prof_d$d 2048 100 | grep -i "advance\|fused" | head -2 | tr '\n' ' ')"; echo "N=128 delay $d: $(./scratch/fl_prof_d$d 256 128 | grep -i "advance\|fused" | head -2 | tr '\n' ' ')"; done
python tools/dump_rollout_state.py /tmp/ro_state5.bin 5 > /dev/null 2>&1
for rep in 1 2 3; do
  for b in scratch/ro_prof_s0 scratch/ro_prof_s1 scratch/ro_prof_s2; do
    a=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 200 5 | grep "resident rollout\|fingerprint" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/' | tr '\n' ' ')
    c=$(RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 20 40 | grep "resident rollout" | sed 's/.*launch, \([0-9.]*\) us per step.*/\1/')
    echo "$rep $b  T200 $a us/step   T20 $c us/step"
  done
done
for b in scratch/ro_prof_s0 scratch/ro_prof_s1; do echo "== $b"; RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 $b 256 100 3 20 20 | grep "stamp  [03478] \|stamp 2[0-4]"; done
