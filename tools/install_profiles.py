"""Copy the outputs of tools/regen_profiles.sh (gpurun_out/final/) into profiles/r01_* (run from the repo root)."""
import sys
sys.path.insert(0, ".")
import json, shutil
O='gpurun_out/final'
d=json.loads(open(O+'/bench_final.json').read().strip().splitlines()[-1])
print('value %.4g ms/step %.5f' % (d['value'], d['ms_per_step']), 'traffic', d['roofline']['traffic'], 'frac %.3f' % d['roofline']['frac'], 'paths', {k: '%.3g' % v['value'] for k,v in d['paths'].items()}, 'cpu %.3g' % d['cpu_baseline']['value'])
d2=json.loads(open(O+'/bench_under_rocprof.json').read().strip().splitlines()[-1])
print('under rocprof %.4g' % d2['value'])
shutil.copy(O+'/pmc_traffic.json','profiles/r01_pmc_traffic.json')
shutil.copy(O+'/pmc_hbm_traffic.txt','profiles/r01_pmc_hbm_traffic.txt')
import os as _os
if _os.path.exists(O+'/pmc_sq.txt'):
    shutil.copy(O+'/pmc_sq.txt','profiles/r01_pmc_sq.txt')
open('profiles/r01_bench_final.json','w').write(json.dumps(d)+'\n')
open('profiles/r01_bench_final_under_rocprof.json','w').write(json.dumps(d2)+'\n')
_kt = open(O+'/bench_kernel_trace.txt').read().splitlines(True)
_note = ("# NOTE rollout_kernel<100, 3>: 2 launches = the 100-step warm-up launch (min_us) and the 1000-step TIMED launch (max_us);\n"
         "#      bench.py's roofline.avg_launch_ms (HIP events around the timed launch) is the max_us figure, not avg_us.\n"
         "#      The other kernels belong to the two-launch path (timed in the same run) and to the stand-alone roofline leg.\n")
open('profiles/r01_bench_kernel_trace_final.txt','w').write(''.join(_kt[:2]) + _note + ''.join(_kt[2:]))
shutil.copy(O+'/dagger_update.json','profiles/r01_dagger_update.json')   # produced by `python bench.py --dagger-update`
hdr = "# bench.py at other shapes (B N K): value, per-path throughput, per-kernel (avg launch us, GB/s), state finite\n# `resident` = mgp_rollout_steps (covered: N <= 256, widths <= 64); `factored` = HBM bit-row state (N > 256); otherwise the two-launch path is `value`\n"
open('profiles/r01_other_configs.txt','w').write(hdr+open(O+'/other_configs.txt').read())
import os
# phase-stamp files keep their hand-written legend (leading '#' lines); the body is the harness output of this run
for src, dst in [('rollout_phase_stamps.txt', 'profiles/r01_rollout_phase_stamps.txt'),
                 ('flock_phase_stamps.txt', 'profiles/r01_flock_step_phase_stamps.txt'),
                 ('af_phase_stamps.txt', 'profiles/r01_actor_fwd_phase_stamps.txt')]:
    if os.path.exists(O + '/' + src) and os.path.exists(dst):
        lead = []
        for line in open(dst).read().splitlines(True):
            if not line.startswith('#'):
                break
            lead.append(line)
        open(dst, 'w').write(''.join(lead) + open(O + '/' + src).read())
if os.path.exists(O+'/train_phase_stamps.txt'):
    open('profiles/r01_train_step_phase_stamps.txt','w').write(
        "# tools/harness/train_phase_prof.hip on MI355X: in-kernel s_memtime stamps of workgroup (0,0), thread 0 of\n"
        "# train_tile_kernel (shader cycles, ~2.1 GHz); the per-update time is both launches of mgp_train_step\n"
        + open(O+'/train_phase_stamps.txt').read())
from multiagent_gnn_policies_amd import build
print('hash ok', build.source_hash() == json.load(open('profiles/r01_pmc_traffic.json'))['_meta']['source_hash'])
