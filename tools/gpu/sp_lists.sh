#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/sp
{
for rep in 1 2; do
echo "== lists"; ./scratch/sp_prof_lists 64 1000 3 200
echo "== lists disabled (SP_NOLISTS)"; SP_NOLISTS=1 ./scratch/sp_prof_lists 64 1000 3 200
echo "== previous build"; ./scratch/sp_prof 64 1000 3 200
done
echo "== stamps"; ./scratch/sp_prof_stamps 64 1000 3 200
} > gpurun_out/sp/lists.log 2>&1
cat gpurun_out/sp/lists.log
