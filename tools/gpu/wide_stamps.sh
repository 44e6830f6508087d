#!/bin/bash
# generic (run-time sized) builds: N = 104 (no compile-time instantiation), lattice harness state; KS=8 hidden 32, KS=16 hidden 64
for b in r02 now; do
  echo "=== $b KS=8 hidden 32 N=104"; RO_CARRY=1 RO_HIDDEN=32 scratch/ro_prof_${b}_ks8 256 104 3 100 5 | grep -v "^exit"
  echo "=== $b KS=16 hidden 64 N=104"; RO_CARRY=1 RO_HIDDEN=64 scratch/ro_prof_${b}_ks16 256 104 3 100 5 | grep -v "^exit"
done
