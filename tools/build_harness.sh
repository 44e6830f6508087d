#!/bin/bash
# Builds the stand-alone profiling harnesses into scratch/ (they travel to the GPU box with the snapshot; hipcc cross-compiles here).
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -w"
hipcc $F -o scratch/ro_prof tools/harness/ro_phase_prof.hip multiagent_gnn_policies_amd/csrc/rollout_t512.hip &
hipcc $F -o scratch/ro_launch tools/harness/ro_launch_prof.hip multiagent_gnn_policies_amd/csrc/rollout_t512.hip &
hipcc $F -o scratch/sp_prof tools/harness/sp_step_prof.hip &
hipcc $F -DMGP_SP_PROFILE -o scratch/sp_prof_stamps tools/harness/sp_step_prof.hip &
hipcc $F -o scratch/sp_persist tools/harness/sp_persist_check.hip &
hipcc $F -DMGP_SP_PROFILE -o scratch/sp_persist_stamps tools/harness/sp_persist_check.hip &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -w -o scratch/ts_prof tools/harness/train_phase_prof.hip &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -Imultiagent_gnn_policies_amd/csrc -o scratch/stream_floor tools/harness/stream_floor.hip &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DMGP_AF_MLP_STAMPS -o scratch/af_prof tools/harness/af_phase_prof.hip &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -o scratch/fl_prof tools/harness/flock_phase_prof.hip &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o scratch/valu_rate tools/harness/valu_rate.hip &
hipcc $F -DRO_STAMP_T=0 -o scratch/ro_st0 tools/harness/ro_phase_prof.hip multiagent_gnn_policies_amd/csrc/rollout_t512.hip &
wait
ls -la scratch/ro_prof scratch/ro_launch scratch/sp_prof scratch/sp_prof_stamps scratch/ts_prof
