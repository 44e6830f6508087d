#!/bin/bash
# Regenerates every round artefact under profiles/ on the GPU box (run through gpurun from the repo root):
#   (the sections of tools/regen_profiles.sh that depend on csrc/rollout.hip only: PMC passes, bench lines, resident harnesses)
#   python tools/install_profiles.py          # copies gpurun_out/final/* into profiles/<round>_*  (ROUND=r05)
# The harnesses must have been built first (they travel with the snapshot under scratch/):  bash tools/build_harness.sh
# PMC passes use --pmc with --kernel-trace only (no sys/runtime/hip trace domains).
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# 1. PMC passes (counters only + kernel trace).  The resident kernel is profiled at two launch lengths (1000 and 20 steps):
#    bytes(T) = fixed + per_step * T
export PROBE_T=1000
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o fetch -- python $R/tools/pmc_probe.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o write -- python $R/tools/pmc_probe.py > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU --kernel-trace -d $O/pmc_sq -o sq -- python $R/tools/pmc_probe.py > $O/pmc_sq.log 2>&1
export PROBE_T=20
PROBE_ROLLOUT_ONLY=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch20 -o fetch -- python $R/tools/pmc_probe.py > $O/pmc_fetch20.log 2>&1
PROBE_ROLLOUT_ONLY=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write20 -o write -- python $R/tools/pmc_probe.py > $O/pmc_write20.log 2>&1
export PROBE_T=1000
# the factored path (BASELINE configs[2]: 64 x 1000): bytes per launch of its three kernels
PROBE_FACTORED=1 PROBE_B=64 PROBE_N=1000 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_f -o fetch -- python $R/tools/pmc_probe.py > $O/pmc_fetch_f.log 2>&1
PROBE_FACTORED=1 PROBE_B=64 PROBE_N=1000 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_f -o write -- python $R/tools/pmc_probe.py > $O/pmc_write_f.log 2>&1
F=$(find $O/pmc_fetch -name "*results.db" | head -1); W=$(find $O/pmc_write -name "*results.db" | head -1); Q=$(find $O/pmc_sq -name "*results.db" | head -1)
F2=$(find $O/pmc_fetch20 -name "*results.db" | head -1); W2=$(find $O/pmc_write20 -name "*results.db" | head -1)
cd $R
python tools/pmc_summary.py $F $W $O/pmc_traffic.json 256,100,3 $F2 $W2 20 > $O/pmc_hbm_traffic.txt 2>&1
cp $O/pmc_traffic.json $R/profiles/${ROUND:-r05}_pmc_traffic.json
FF=$(find $O/pmc_fetch_f -name "*results.db" | head -1); WF=$(find $O/pmc_write_f -name "*results.db" | head -1)
python tools/pmc_summary.py $FF $WF $O/pmc_traffic_factored.json 64,1000,3 > $O/pmc_hbm_traffic_factored.txt 2>&1
cp $O/pmc_traffic_factored.json $R/profiles/${ROUND:-r05}_pmc_traffic_factored.json
python tools/pmc_sq_summary.py $Q $O/pmc_sq.json > $O/pmc_sq.txt 2>&1
cp $O/pmc_sq.json $R/profiles/${ROUND:-r05}_pmc_sq.json
# 2. bench (traffic / sq now resolved from the files just written): the driver's command line, the default, and under rocprof
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
python bench.py > $O/bench_final.json 2> $O/bench_final.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
T=$(find $O/trace -name "*results.db" | head -1)
cd $R
python tools/rocpd_stats.py $T > $O/bench_kernel_trace.txt 2>&1
# 3. phase stamps of the resident kernel: lattice harness state and the bench's own state 5 steps after reset
RO_CARRY=1 ./scratch/ro_prof 256 100 3 200 > $O/rollout_phase_stamps.txt 2>&1
python tools/dump_rollout_state.py /tmp/ro_state5.bin 5 > /dev/null 2>&1
RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 ./scratch/ro_prof 256 100 3 20 20 >> $O/rollout_phase_stamps.txt 2>&1
{ RO_STATE=/tmp/ro_state5.bin RO_WG_DUMP=$O/rollout_wg_times.txt ./scratch/ro_launch 256 100 3 "1 2 3 5 10 20 40 100" 30; ./scratch/ro_launch 256 100 3 "1 2 5 20" 30; } > $O/rollout_launch_cost.txt 2>&1
# 3a. [r5] resident kernel A/B (scratch/ro_prof_base = round 4's sources, x0 = this round's with RO_VERLET=0, x1 = the product build) and the
#     per-episode durations / S1 modes of a 20-step launch with and without the candidate lists
RO_STAMP_BINS="scratch/ro_prof_x1 scratch/ro_st0" timeout 600 bash tools/gpu/r5_ab.sh > $O/rollout_ab.txt 2>&1
cp gpurun_out/wg_times_ro_launch_v.txt $O/rollout_wg_times_lists.txt 2>/dev/null
cp gpurun_out/wg_times_ro_launch_0.txt $O/rollout_wg_times_no_lists.txt 2>/dev/null
timeout 300 bash tools/gpu/r5_long.sh > $O/rollout_long_launches.txt 2>&1
# 3a'. [r5] mgp_flock_step_advance: the one-workgroup-per-episode kernel and the row-tiled one it replaces (tools/harness/flock_phase_prof.hip)
{ for cfg in "256 100" "2048 100" "256 128" "16 100"; do echo "== flock_advance_kernel, B N = $cfg"; ./scratch/fl_prof $cfg | grep -v "stamp [0-7] "; echo "== row-tiled kernel (MGP_FLOCK_ADVANCE_TILED=1), B N = $cfg"; MGP_FLOCK_ADVANCE_TILED=1 ./scratch/fl_prof $cfg | head -2; done; } > $O/flock_advance_stamps.txt 2>&1
# 3b'. [r5] fused Actor forward at the wide shapes (actor_fwd_wide_kernel; scratch/af_prof = tools/harness/af_phase_prof.hip with the MLP
#      stamps) and the generic fp32-MFMA chain on the same box
{ for cfg in "256 100 128" "256 128 128" "256 100 64" "1 100 128"; do echo "== actor_fwd_wide_kernel, B N hidden = $cfg"; ./scratch/af_prof $cfg | grep -v "^block"; echo "== generic chain (MGP_ACTOR_WIDE=0), B N hidden = $cfg"; MGP_ACTOR_WIDE=0 ./scratch/af_prof $cfg | head -1; done; } > $O/actor_fwd_wide_stamps.txt 2>&1
# 6. instruction mix of the resident kernel (harness, bench state)
bash tools/gpu/ro_pmc.sh > $O/rollout_inst_mix.txt 2>&1
bash tools/gpu/other_cfgs.sh > $O/other_configs.txt 2>&1
bash tools/gpu/hidden_grid.sh > $O/hidden_grid.txt 2>&1
bash tools/gpu/sweep_grid.sh > $O/sweep_grid.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_fetch20 $O/pmc_write20 $O/pmc_fetch_f $O/pmc_write_f $O/trace gpurun_out/ro_pmc
ls $O
