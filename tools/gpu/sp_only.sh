#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/sp
tools/gpu/sp_ab.sh > /dev/null
SP_ONLY=policy ./scratch/sp_prof_stamps 64 1000 3 200 > gpurun_out/sp/only.log 2>&1
cat gpurun_out/sp/ab.log | grep -A34 "== scratch/sp_prof_stamps" | head -36
grep "factored step" gpurun_out/sp/ab.log
echo "---- policy only"; cat gpurun_out/sp/only.log
