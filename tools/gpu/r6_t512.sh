#!/bin/bash
# round 6: two episodes per CU (csrc/rollout_t512.hip, 512-thread workgroups) against one (1024-thread workgroups) for launches with
# more episodes than CUs.  scratch/ro_prof_t = tools/harness/ro_phase_prof.hip linked with rollout_t512.hip; MGP_RO_T512 forces the build.
# Irregular harness state (7th argument), prebuilt image + carry hand-over, 100-step launches; fingerprints must agree per B.
for B in 256 512 1024 2048; do
  for t in 0 1; do
    echo "B=$B MGP_RO_T512=$t: $(MGP_RO_T512=$t RO_CARRY=1 ./scratch/ro_prof_t $B 100 3 100 5 rnd | grep "resident rollout\|fingerprint" | tr '\n' ' ')"
  done
done
for B in 512 2048; do for t in 0 1; do
  echo "bench.py --episodes $B --steps 100, MGP_RO_T512=$t: $(MGP_RO_T512=$t python bench.py --episodes $B --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.3e' % d['value'], 'us/step %.2f' % (1e3*d['ms_per_step']), 'parity', d['parity']['ok'], d['parity']['passed_on'], 'max_rel %.2e' % d['parity']['max_rel'])")"
done; done
echo "bench.py --episodes 2048 whole episodes (default 1000 steps): $(python bench.py --episodes 2048 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.3e' % d['value'], 'us/step %.2f' % (1e3*d['ms_per_step']), 'parity', d['parity']['ok'])")"
for t in 0 1; do
  echo "bench.py --dagger --episodes 1024 --steps 200, MGP_RO_T512=$t: $(MGP_RO_T512=$t python bench.py --dagger --episodes 1024 --steps 200 --warmup 10 --updates 256 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('collection %.3e agent-steps/s' % d['value'], 'us/step %.2f' % (1e3*d['ms_per_step']))")"
done
