#!/bin/bash
# host side of a 20-step resident launch under the HIP runtime's wait / kernarg knobs (tools/gpu/launch_probe.py)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/probe
{
for rep in 1 2; do
echo "== default";                          python tools/gpu/launch_probe.py 2>/dev/null
echo "== ROC_ACTIVE_WAIT_TIMEOUT=2000";     ROC_ACTIVE_WAIT_TIMEOUT=2000 python tools/gpu/launch_probe.py 2>/dev/null
echo "== HIP_FORCE_DEV_KERNARG=1";          HIP_FORCE_DEV_KERNARG=1 python tools/gpu/launch_probe.py 2>/dev/null
echo "== HIP_FORCE_DEV_KERNARG=0";          HIP_FORCE_DEV_KERNARG=0 python tools/gpu/launch_probe.py 2>/dev/null
echo "== both";                             ROC_ACTIVE_WAIT_TIMEOUT=2000 HIP_FORCE_DEV_KERNARG=1 python tools/gpu/launch_probe.py 2>/dev/null
echo "== ROC_SYSTEM_SCOPE_SIGNAL=0";        ROC_SYSTEM_SCOPE_SIGNAL=0 python tools/gpu/launch_probe.py 2>/dev/null
echo "== DEBUG_HIP_BLOCK_SYNC=0 / AMD_DIRECT_DISPATCH=0"; AMD_DIRECT_DISPATCH=0 python tools/gpu/launch_probe.py 2>/dev/null
done
} > gpurun_out/probe/wait.log 2>&1
cat gpurun_out/probe/wait.log
