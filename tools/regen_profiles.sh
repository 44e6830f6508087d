#!/bin/bash
# Regenerates every round artefact under profiles/ on the GPU box (run through gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/regen_profiles.sh > gpurun_out/regen.log 2>&1'
#   python tools/install_profiles.py          # copies gpurun_out/final/* into profiles/<round>_*  (ROUND=r06)
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
# the factored path (BASELINE configs[2]: 64 x 1000): bytes per launch of spp_rollout_kernel (six calls of PROBE_FT = 200 steps)
PROBE_FACTORED=1 PROBE_B=64 PROBE_N=1000 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_f -o fetch -- python $R/tools/pmc_probe.py > $O/pmc_fetch_f.log 2>&1
PROBE_FACTORED=1 PROBE_B=64 PROBE_N=1000 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_f -o write -- python $R/tools/pmc_probe.py > $O/pmc_write_f.log 2>&1
F=$(find $O/pmc_fetch -name "*results.db" | head -1); W=$(find $O/pmc_write -name "*results.db" | head -1); Q=$(find $O/pmc_sq -name "*results.db" | head -1)
F2=$(find $O/pmc_fetch20 -name "*results.db" | head -1); W2=$(find $O/pmc_write20 -name "*results.db" | head -1)
cd $R
python tools/pmc_summary.py $F $W $O/pmc_traffic.json 256,100,3 $F2 $W2 20 > $O/pmc_hbm_traffic.txt 2>&1
cp $O/pmc_traffic.json $R/profiles/${ROUND:-r06}_pmc_traffic.json
FF=$(find $O/pmc_fetch_f -name "*results.db" | head -1); WF=$(find $O/pmc_write_f -name "*results.db" | head -1)
python tools/pmc_summary.py $FF $WF $O/pmc_traffic_factored.json 64,1000,3 > $O/pmc_hbm_traffic_factored.txt 2>&1
cp $O/pmc_traffic_factored.json $R/profiles/${ROUND:-r06}_pmc_traffic_factored.json
python tools/pmc_sq_summary.py $Q $O/pmc_sq.json > $O/pmc_sq.txt 2>&1
cp $O/pmc_sq.json $R/profiles/${ROUND:-r06}_pmc_sq.json
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
# 3a. [r6] what a wave64 vector instruction costs its SIMD, per class (tools/harness/valu_rate.hip) + what SQ_ACTIVE_INST_VALU counts per
#     instruction (counter pass over the same kernels); the step of the launch's FIRST step (a rebuild step) beside the cheap step above
(cd $R && ./scratch/valu_rate > $O/valu_rate.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES --kernel-trace -d $O/pmc_valu -o v -- $R/scratch/valu_rate pmc > $O/valu_rate_pmc.log 2>&1)
(cd $R && python tools/pmc_sq_summary.py $(find $O/pmc_valu -name "*results.db" | head -1) > $O/valu_rate_pmc.txt 2>&1; rm -rf $O/pmc_valu)
cd $R
{ echo "== step 0 of the launch (rebuild step), bench state"; RO_STATE=/tmp/ro_state5.bin RO_CARRY=1 ./scratch/ro_st0 256 100 3 20 20; } >> $O/rollout_phase_stamps.txt 2>&1
# 3a''. [r6] two episodes per CU (rollout_t512.hip) against one, B = 256 .. 2048, harness + bench.py + DAGGER collection; and the SQ shares
#       of the resident kernel at 2048 episodes (512-thread workgroups)
bash tools/gpu/r6_t512.sh > $O/two_episodes_per_cu.txt 2>&1
(cd /tmp && PROBE_B=2048 PROBE_T=200 PROBE_ROLLOUT_ONLY=1 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU --kernel-trace -d $O/pmc_sq_b2048 -o sq -- python $R/tools/pmc_probe.py > $O/pmc_sq_b2048.log 2>&1)
python tools/pmc_sq_summary.py $(find $O/pmc_sq_b2048 -name "*results.db" | head -1) > $O/pmc_sq_b2048.txt 2>&1; rm -rf $O/pmc_sq_b2048
# 3a'. [r5] mgp_flock_step_advance: the one-workgroup-per-episode kernel and the row-tiled one it replaces (tools/harness/flock_phase_prof.hip)
{ for cfg in "256 100" "2048 100" "256 128" "16 100"; do echo "== flock_advance_kernel, B N = $cfg"; ./scratch/fl_prof $cfg | grep -v "stamp [0-7] "; echo "== row-tiled kernel (MGP_FLOCK_ADVANCE_TILED=1), B N = $cfg"; MGP_FLOCK_ADVANCE_TILED=1 ./scratch/fl_prof $cfg | head -2; done; } > $O/flock_advance_stamps.txt 2>&1
# 3b'. [r5] fused Actor forward at the wide shapes (actor_fwd_wide_kernel; scratch/af_prof = tools/harness/af_phase_prof.hip with the MLP
#      stamps) and the generic fp32-MFMA chain on the same box
{ for cfg in "256 100 128" "256 128 128" "256 100 64" "1 100 128"; do echo "== actor_fwd_wide_kernel, B N hidden = $cfg"; ./scratch/af_prof $cfg | grep -v "^block"; echo "== generic chain (MGP_ACTOR_WIDE=0), B N hidden = $cfg"; MGP_ACTOR_WIDE=0 ./scratch/af_prof $cfg | head -1; done; } > $O/actor_fwd_wide_stamps.txt 2>&1
# 3b. phase stamps of the fused Actor forward (MFMA aggregation variant), B = 256 and B = 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -w -o /tmp/af_prof tools/harness/af_phase_prof.hip 2>/dev/null
{ /tmp/af_prof 256 100; /tmp/af_prof 1 100; } > $O/actor_fwd_phase_stamps.txt 2>&1
# 4. DAGGER update / collection + other configs
python tools/bench_update.py > $O/dagger_update.json 2> $O/dagger_update.err
bash tools/gpu/other_cfgs.sh > $O/other_configs.txt 2>&1
bash tools/gpu/hidden_grid.sh > $O/hidden_grid.txt 2>&1
bash tools/gpu/sweep_grid.sh > $O/sweep_grid.txt 2>&1
for cfg in "1 100 3 32 2" "2048 100 3 32 2"; do set -- $cfg; python bench.py --episodes $1 --agents $2 --taps $3 --hidden $4 --layers $5 --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = {a: (round(v['avg_launch_ms']*1e3,1), round(v['GBps'])) for a, v in d.get('kernels', {}).items()}
print('$1 x N=$2 K=$3 hidden $4 x $5', 'value %.3e' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'paths', {a: '%.3e' % b['value'] for a, b in d['paths'].items()}, 'parity', d['parity']['ok'], d['parity']['passed_on'], 'kernels (us, GB/s)', k)
" >> $O/other_configs.txt; done
for f in 8 4; do MGP_AGG_FORM=$f python tools/gpu/agg_ab.py 2>/dev/null; done > $O/agg_forms.txt
# [r6] form 43: one workgroup per episode, the three taps in sequence in every wave (a tap's sums and stores under the later taps' flight)
AGG_SHAPES="256,100,3 512,100,3 1024,100,3 2048,100,3" MGP_AGG_FORM=43 python tools/gpu/agg_ab.py 2>/dev/null | grep agg_fwd >> $O/agg_forms.txt
# [r6] forms 44 / 45: 7 / 10 of a wave's 14 requests up front at every batch size (the default does that from B K >= 3072 on)
for f in 44 45; do AGG_SHAPES="256,100,3 512,100,3 1024,100,3 2048,100,3" MGP_AGG_FORM=$f python tools/gpu/agg_ab.py 2>/dev/null | grep agg_fwd >> $O/agg_forms.txt; done
bash tools/gpu/update_slots_ab.sh > $O/dagger_update_slots.txt 2>&1
./scratch/stream_floor > $O/stream_floor.txt 2>&1
python tools/gpu/train_wall.py 2>/dev/null | tail -1 > $O/train_wall.json
python tools/gpu/train_wall.py --agents 200 --taps 4 2>/dev/null | tail -1 > $O/train_wall_n200_k4.json
{ echo '# tools/harness/train_phase_prof.hip on MI355X: the two-launch DAGGER update, B = 20, 6-32-32-2, K = 3 -- wall time per update (back to back, no graph) and in-kernel'
  echo '# cycle stamps of workgroup (0,0) of train_tile_kernel.  Blocks: dense (X, G) at N = 100 | aggregated input at N = 100 | the same with the generic'
  echo '# kernel (MGP_TRAIN_CS=0: run-time widths) | aggregated input at N = 1000'
  ./scratch/ts_prof 20 100 3 0; ./scratch/ts_prof 20 100 3 1; MGP_TRAIN_CS=0 ./scratch/ts_prof 20 100 3 1; ./scratch/ts_prof 20 1000 3 1; } > $O/train_phase_stamps.txt 2>&1
# 4b. degree sweep on the environment's own (disc) resets: the communication radius sets the mean degree (~ R^2)
for R_ in 0.83 0.95 1.0 1.05 1.15 1.3; do python bench.py --comm-radius $R_ --no-cpu-baseline --no-roofline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('degree sweep: comm_radius $R_', 'mean degree at reset %.2f' % d['config']['mean_degree_at_reset'], 'after the timed region %.2f' % d['config']['mean_degree'], 'value %.3e' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'paths', {a: '%.3e' % b['value'] for a, b in d['paths'].items()}, 'parity ok', d['parity']['ok'], 'max_rel %.2e' % d['parity']['max_rel'])
" >> $O/other_configs.txt; done
# 5. the DAGGER round (BASELINE configs[3]): one rank, and two ranks sharing this GPU (gloo carries the IPC handles; the gradient
#    goes through the one-shot exchange), plus the exchange's own latency for 2 / 3 / 4 ranks on the one device
python bench.py --dagger --steps 500 --warmup 20 > $O/dagger_round_1rank.json 2> $O/dagger_round_1rank.err
MGP_DIST_BACKEND=gloo timeout 300 python bench.py --dagger --gpus 2 --steps 500 --warmup 20 --episodes 128 2> $O/dagger_round_2ranks.err | grep "^{" > $O/dagger_round_2ranks_shared_gpu.json
timeout 900 python - > $O/p2p_exchange_latency.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, 'tests')
import test_gpu_p2p as t
for w in (2, 3, 4, 8):
    r = t.run_ranks('allreduce', world=w, timeout=900)
    print('ranks %d (one MI355X, IPC between processes): %.2f us per exchange of 1,731 floats inside a 32-exchange HIP graph (launch of the stand-alone kernel included), mailbox memory kind %d (2 = uncached), %d exchanges checked bit-exact' % (w, r['exchange_us_in_graph'], r['mem_kind'], r['exchanges']))
PY
DRY=1 timeout 600 python tools/first_multi_gpu.py > $O/first_multi_gpu_dry.json 2> $O/first_multi_gpu_dry.err
# 6. instruction mix of the resident kernel (harness, bench state)
bash tools/gpu/ro_pmc.sh > $O/rollout_inst_mix.txt 2>&1
# 7. the factored path (N > 256; cfg-3 shape 64 x 1000, K = 3): harness stamps of its three kernels, their kernel trace, the
#    bench at 100 / 500 steps per call, and DAGGER rounds collecting on the factored state (N = 1000, N = 300)
#    (harness: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DMGP_SP_PROFILE] -o scratch/sp_prof[_stamps] tools/harness/sp_step_prof.hip)
#    [r6] + the persistent form (tools/harness/sp_persist_check.hip: every output against the K-launch form bit for bit, both timed,
#    stamps of a steady-state step; 64 x 1000, 256 x 300 in two launches of 128 episodes, a contracted flock on the bit-row fallback)
{ ./scratch/sp_persist_stamps 64 1000 100; ./scratch/sp_persist 64 1000 100; ./scratch/sp_persist 64 1000 500; ./scratch/sp_persist 256 300 100; ./scratch/sp_persist 64 1000 50 0.3;
  echo "== the K-launch form, one call per step (tools/harness/sp_step_prof.hip)"
  ./scratch/sp_prof_stamps 64 1000 3 200; ./scratch/sp_prof 64 1000 3 200; ./scratch/sp_prof 256 300 3 200; ./scratch/sp_prof 64 1000 4 200; } > $O/factored_step_stamps.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/trace_spp -o spp -- $R/scratch/sp_persist 64 1000 100 > /dev/null 2>&1)
python tools/rocpd_stats.py $(find $O/trace_spp -name "*results.db" | head -1) | head -9 > $O/factored_kernel_trace.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/trace_sp -o sp -- $R/scratch/sp_prof 64 1000 3 200 > /dev/null 2>&1)
TS=$(find $O/trace_sp -name "*results.db" | head -1)
python tools/rocpd_stats.py $TS | head -8 >> $O/factored_kernel_trace.txt 2>&1
for st in 100 500; do python bench.py --episodes 64 --agents 1000 --taps 3 --no-cpu-baseline --no-roofline --steps $st --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
pf = d['parity']['paths']['factored']
print('64 1000 3 hidden 32 x 2, $st steps per call:', 'value %.3e' % d['value'], 'us/step %.2f' % (1e3 * d['ms_per_step']), 'paths', {a: '%.3e' % b['value'] for a, b in d['paths'].items()}, 'mean degree %.2f' % d['config']['mean_degree'], '| parity gate, factored step: ok', d['parity']['ok'], 'max_rel %.2e' % pf['max_rel'], 'passed on', pf['passed_on'].split(' (')[0], '-- well conditioned %d/%d episodes, the reference\'s own fp32 noise on these states %.1e' % (d['parity']['well_conditioned_episodes'], d['parity']['checked_episodes'], d['parity']['reference_fp32_noise']))
" >> $O/factored_kernel_trace.txt; done
python bench.py --dagger --episodes 64 --agents 1000 --steps 200 --warmup 10 --updates 64 2> $O/dagger_round_n1000.err | grep "^{" > $O/dagger_round_n1000.json
python bench.py --dagger --episodes 256 --agents 300 --steps 200 --warmup 10 --updates 256 2> $O/dagger_round_n300.err | grep "^{" > $O/dagger_round_n300.json
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_fetch20 $O/pmc_write20 $O/pmc_fetch_f $O/pmc_write_f $O/trace $O/trace_sp $O/trace_spp gpurun_out/ro_pmc
ls -la $O
