"""Copy the outputs of tools/regen_profiles.sh (gpurun_out/final/) into profiles/<round>_* (run from the repo root;
ROUND=r06 by default)."""
import json
import os
import shutil
import sys
sys.path.insert(0, ".")
import os
O, R = 'gpurun_out/final', os.environ.get('ROUND', 'r06')


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


d20 = last_json(O + '/bench_steps20.json')
d = last_json(O + '/bench_final.json')
d2 = last_json(O + '/bench_under_rocprof.json')
for name, x in (('--steps 20 --warmup 5', d20), ('default', d), ('--steps 20 under rocprofv3', d2)):
    r = x['roofline']
    print('%-28s value %.4g ms/step %.5f | roofline %s frac %.3f traffic %s | parity %s %.2e | mean degree %.2f' % (
        name, x['value'], x['ms_per_step'], r['bound'], r['frac'] if r['frac'] is not None else float('nan'), r['traffic'], x['parity']['ok'], x['parity']['max_rel'],
        x['config']['mean_degree']))
    print('   dense kernels:', {k: (round(v['avg_launch_ms'] * 1e3, 2), round(v['frac'], 3), v['traffic']) for k, v in r['dense_kernels'].items()
                                if isinstance(v, dict)})
shutil.copy(O + '/pmc_traffic.json', 'profiles/%s_pmc_traffic.json' % R)
shutil.copy(O + '/pmc_hbm_traffic.txt', 'profiles/%s_pmc_hbm_traffic.txt' % R)
shutil.copy(O + '/pmc_sq.json', 'profiles/%s_pmc_sq.json' % R)
shutil.copy(O + '/pmc_sq.txt', 'profiles/%s_pmc_sq.txt' % R)
open('profiles/%s_bench_steps20.json' % R, 'w').write(json.dumps(d20) + '\n')
open('profiles/%s_bench_default_1000steps.json' % R, 'w').write(json.dumps(d) + '\n')
open('profiles/%s_bench_steps20_under_rocprof.json' % R, 'w').write(json.dumps(d2) + '\n')
kt = open(O + '/bench_kernel_trace.txt').read().splitlines(True)
note = ("# rocprofv3 --kernel-trace --stats of `python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command line).\n"
        "# rollout_kernel<100,3,..>: bench.py times the resident region twice over the same steps (rewind to the reset + pre-roll launch, 5-step\n"
        "#      warm-up launch, 20-step TIMED launch; the first pass is `value`, the second carries the kernel-stamped HIP events behind\n"
        "#      roofline.avg_launch_ms), then the same on the jittered lattice, then the parity gate's one-step launch.  The other kernels belong to the\n"
        "#      two-launch path (timed in the same run) and to the stand-alone dense-kernel roofline leg.\n")
open('profiles/%s_bench_kernel_trace.txt' % R, 'w').write(''.join(kt[:2]) + note + ''.join(kt[2:]))
shutil.copy(O + '/dagger_update.json', 'profiles/%s_dagger_update.json' % R)
hdr = ("# tools/gpu/other_cfgs.sh: bench.py at the non-headline shapes, each on a TRAINED policy (tests/golden/policies, tools/train_policies.py;\n"
       "# the reference's shipped checkpoint where its shape fits) and the environment's own reset distribution, 100-step launches right after\n"
       "# reset: value, per-path throughput; second line: the in-run parity gate -- well-conditioned episodes of the 16 checked, the\n"
       "# reference's own fp32 noise, and per path max_rel and the bound it passed on.  Then B = 1 / B = 2048 with per-kernel (us, GB/s), then the degree sweep.\n")
open('profiles/%s_other_configs.txt' % R, 'w').write(hdr + open(O + '/other_configs.txt').read())
open('profiles/%s_rollout_phase_stamps.txt' % R, 'w').write(
    "# tools/harness/ro_phase_prof.hip on MI355X (RO_CARRY=1: prebuilt weight image + factored hand-over, the repeated-launch form):\n"
    "# in-kernel s_memtime stamps of workgroup 0, lane 0 of eleven waves, step 5 of the launch (shader cycles).  First block: regular-lattice\n"
    "# harness state, 200-step launch; second block: bench.py's own state 5 steps after reset (irregular degrees), 20-step launches.\n"
    "# Schedule: B/C (gather stage K-1 + MLP + output layer on registers + per-axis integration, waves 0-6) | S1 (membership: the cheap pass\n"
    "# over the row's Verlet candidates on the stamped step 5 -- stamps 16-19 belong to the exact / rebuild pass and stay empty there --; 13 waves)\n"
    "# | S2 (fp64 features, waves 0-6 || gather stage 1 of the next step, waves 7-13 || the Verlet helper on wave 14); 3 barriers.\n"
    + open(O + '/rollout_phase_stamps.txt').read())
open('profiles/%s_rollout_launch_cost.txt' % R, 'w').write(
    "# tools/harness/ro_launch_prof.hip: anatomy of a resident launch (prebuilt weight image, factored hand-over), B=256 N=100 K=3.  Every launch\n"
    "# of a row starts from the SAME saved state; kernel us = the launch's own begin/end events; the other columns from the 100 MHz wall clock every\n"
    "# workgroup stamps (medians over workgroups and 30 launches): dispatch ramp, entry (state -> LDS, lists from the carry, gather stage 1 of the\n"
    "# first step), the first three steps, the later steps, exit, and the spread of the workgroups' END times (the launch lasts as long as its\n"
    "# slowest episode).  First block: bench.py's state 5 steps after a disc reset; second: the regular lattice of the harness.\n"
    + open(O + '/rollout_launch_cost.txt').read())
if os.path.exists(O + '/rollout_wg_times.txt'):
    shutil.copy(O + '/rollout_wg_times.txt', 'profiles/%s_rollout_wg_times.txt' % R)
open('profiles/%s_actor_fwd_phase_stamps.txt' % R, 'w').write(
    "# tools/harness/af_phase_prof.hip on MI355X: in-kernel s_memtime stamps of workgroup 0 of actor_fwd_mfma_kernel<28, 2> (shader cycles),\n"
    "# B = 256 (one workgroup per CU) and B = 1.  Thread 0 (a streaming wave): 1 = its X + G requests issued | 2 = its 4x4x1 MFMAs done\n"
    "# (consumed as the rows land) | 3 = row classes added, aggregation tile written | 4 = THE barrier passed (all streaming waves + the\n"
    "# staging waves) | 5 = MLP start | 6, 7, 8 = end of layers 0, 1, 2 (register-chained).  First thread of staging wave 0: 16 = weight area\n"
    "# zeroed, all LDS-DMA rows issued | 19 = they have landed (17, if printed by an older harness build: same point as 16).\n"
    + open(O + '/actor_fwd_phase_stamps.txt').read())
from multiagent_gnn_policies_amd import build
print('hash ok', build.source_hash() == json.load(open('profiles/%s_pmc_traffic.json' % R))['_meta']['source_hash'])

for name in ('dagger_round_1rank.json', 'dagger_round_2ranks_shared_gpu.json', 'p2p_exchange_latency.txt', 'rollout_inst_mix.txt',
             'agg_forms.txt', 'valu_rate.txt', 'valu_rate_pmc.txt', 'two_episodes_per_cu.txt', 'pmc_sq_b2048.txt', 'dagger_update_slots.txt', 'train_phase_stamps.txt', 'stream_floor.txt', 'train_wall.json', 'train_wall_n200_k4.json', 'first_multi_gpu_dry.json', 'pmc_traffic_factored.json', 'pmc_hbm_traffic_factored.txt'):
    if os.path.exists(O + '/' + name):
        shutil.copy(O + '/' + name, 'profiles/%s_%s' % (R, name))

if os.path.exists(O + '/factored_step_stamps.txt'):
    open('profiles/%s_factored_step_stamps.txt' % R, 'w').write(
        "# tools/harness/sp_persist_check.hip on MI355X: the factored-state rollout (N > 256) as ONE launch of persistent workgroups\n"
        "# (csrc/sparse_persist.hip) against the K-launch form on the same state -- every output buffer compared bit for bit, both timed,\n"
        "# in-kernel stamps of a steady-state step; args: episodes agents steps-per-call [lattice pitch / radius: 0.3 = a contracted flock].\n"
        "# Below the separator, tools/harness/sp_step_prof.hip: the K-launch form one call per step -- gather stage(s), policy tail, cell-list simulator --\n"
        "# on a hash-permuted jittered lattice (neighbours not adjacent in index).  First block: in-kernel s_memtime stamps (shader cycles) of\n"
        "# workgroup (tile 1, episode 3), lane 0 of waves 0 / 5 / 10 / 15, last step; then the unstamped build at 64 x 1000 (K = 3), 256 x 300, 64 x 1000 (K = 4).\n"
        "# Round 2's kernels on the same harness state: 49.6 us per step (simulator 25.2, policy tail 13.7, gather 10.5).\n"
        + open(O + '/factored_step_stamps.txt').read())
    open('profiles/%s_factored_kernel_trace.txt' % R, 'w').write(
        "# rocprofv3 --kernel-trace --stats of `scratch/sp_persist 64 1000 100` (both forms; spp_rollout_kernel = 100 steps per launch) and of\n"
        "# `scratch/sp_prof 64 1000 3 200` (the K-launch form, one call per step), then bench.py at the cfg-3 shape\n"
        "# (jittered-lattice resets ordered by radius: FlockParams.init_mode 'auto' at N > 100) at 100 and 500 steps per policy_rollout call\n"
        + open(O + '/factored_kernel_trace.txt').read())
    for n in ('n1000', 'n300'):
        shutil.copy(O + '/dagger_round_%s.json' % n, 'profiles/%s_dagger_round_%s_factored.json' % (R, n))

for src, hdr in (
        ('rollout_ab.txt',
         "# tools/gpu/r5_ab.sh: the resident kernel on bench.py's own state 5 steps after a disc reset, B=256 N=100 K=3, three repetitions:\n"
         "# ro_prof_base = round 4's sources, x0 = this round's sources with RO_VERLET=0 (no candidate lists), x1 = the product build (skin 0.3,\n"
         "# two passes of eight per trip); T200 / T20 / T1 = 200-, 20- and one-step launches back to back (us per step, us per launch for T1).\n"
         "# Then the phase stamps of x1 (step 5: a cheap step) and of the same build stamped at step 0 (a rebuild step), then the launch anatomy\n"
         "# (tools/harness/ro_launch_prof.hip) without (ro_launch_0) and with (ro_launch_v) the lists.\n"),
        ('flock_advance_stamps.txt',
         "# tools/harness/flock_phase_prof.hip: mgp_flock_step (ping-pong sim step) and mgp_flock_step_advance (sim + state transition, K = 3) back to\n"
         "# back; stamps of workgroup 0 of flock_advance_kernel (shader cycles); below each block the row-tiled kernel of rounds 1-4 on the same box.\n"),
        ('rollout_long_launches.txt',
         "# tools/gpu/r5_long.sh: 100 / 500 / 1000-step launches from the bench state (no resets: the flock of a 1000-step launch is 1000+ steps old),\n"
         "# candidate lists off (x0) and on (x1), with the S1 modes the helper wave chose over all workgroups.\n"),
        ('actor_fwd_wide_stamps.txt',
         "# tools/harness/af_phase_prof.hip B N hidden (-DMGP_AF_MLP_STAMPS): mgp_actor_fwd, inference, two hidden layers of `hidden` channels, back to back on\n"
         "# rotating input sets; stamps of workgroup 0 (shader cycles): thread 0 (a streaming wave, then column tile 0): 1 = its X + G requests issued |\n"
         "# 2 = its 4x4x1 MFMAs done | 3 = aggregation tile written | 4 = first barrier passed | 6 = layer 0 done | 7 = layer 1 done (incl. the second\n"
         "# barrier at [128, 128]) | 8 = action written; first thread of staging wave 0: 19 = its records are in LDS.  Below each block the generic\n"
         "# fp32-MFMA chain (MGP_ACTOR_WIDE=0) on the same box.\n"),
        ('hidden_grid.txt',
         "# tools/gpu/hidden_grid.sh: every (n_layers, hidden_size) of the reference's cfg/hidden_size.cfg at N = 100, K = 3, 256 episodes, 100-step\n"
         "# policy_rollout calls after a disc reset: agent-steps/s (value = the resident kernel where it covers the shape), the two paths, the in-run\n"
         "# parity gate per path (max_rel against the CPU port's Actor forward and the bound it passed on), the weights (trained where a fixture exists).\n"),
        ('sweep_grid.txt',
         "# tools/gpu/sweep_grid.sh: the reference's other sweeps at full size, 256 episodes, 100-step policy_rollout calls after a reset, reference checkpoint\n"
         "# or trained policy where one exists for (env, K): cfg/n.cfg (n_agents 25..150 x k 1..4), cfg/n_twoflocks.cfg (50..250), cfg/rad.cfg (comm_radius\n"
         "# 0.8..4: mean degree 3..79), cfg/vel.cfg (v_max 0.5..5.5), cfg/dt.cfg (dt 0.0075..0.1).  One workgroup per episode: at 256 episodes small flocks leave the CUs' other workgroup slots empty.\n"),
        ('rollout_wg_times_lists.txt', ''), ('rollout_wg_times_no_lists.txt', '')):
    if os.path.exists(O + '/' + src):
        body = ''.join(l for l in open(O + '/' + src).read().splitlines(True) if not l.startswith('+ '))     # (the regen script runs under set -x)
        open('profiles/%s_%s' % (R, src), 'w').write(hdr + body)
print('installed profiles/%s_*' % R)
