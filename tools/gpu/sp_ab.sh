#!/bin/bash
# A/B of the factored-step harness builds: scratch/sp_prof* (built in the container), kernel trace of the first one
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/sp
: > gpurun_out/sp/ab.log
for rep in 1 2; do
for f in scratch/sp_prof*; do
  [ -x "$f" ] || continue
  echo "== $f" >> gpurun_out/sp/ab.log
  timeout 120 $f ${SP_ARGS:-64 1000 3 200} >> gpurun_out/sp/ab.log 2>&1
done
done
if [ -n "$SP_TRACE" ]; then
  for f in scratch/sp_prof*; do
  [ -x "$f" ] || continue
  n=$(basename $f)
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -o t -- $GRAFT_REPO_ROOT/$f ${SP_ARGS:-64 1000 3 200} > /dev/null 2>&1)
  T=$(find /tmp/prof_$n -name "*results.db" | head -1)
  echo "== trace $f" >> gpurun_out/sp/ab.log
  python tools/rocpd_stats.py $T 2>&1 | head -8 | cut -c1-60,110-175 >> gpurun_out/sp/ab.log
  done
fi
cat gpurun_out/sp/ab.log
