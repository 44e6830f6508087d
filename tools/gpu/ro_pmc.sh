#!/bin/bash
# instruction mix of the resident kernel (harness, bench state): one counter pass
O=$GRAFT_REPO_ROOT/gpurun_out/ro_pmc; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT; python tools/dump_rollout_state.py /tmp/ro_state5.bin 5 > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_IFETCH"; do
  n=$(echo $set | cut -c1-20 | tr ' ' '_')
  RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 rocprofv3 --pmc $set --kernel-trace -d $O/$n -o p -- $GRAFT_REPO_ROOT/scratch/ro_prof 256 100 3 200 3 > $O/$n.log 2>&1
done
cd $GRAFT_REPO_ROOT
for db in $(find gpurun_out/ro_pmc -name "*results.db"); do python tools/pmc_sq_summary.py $db 2>&1 | grep -A30 "^rollout_kernel"; done
