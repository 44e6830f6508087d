"""bench.py legs: constants and the readers of the committed profile files (profiles/<round>_*.json)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
MFMA_BF16_PEAK_TFLOPS = 2500.0                     # dense bf16 (MI355X_MICROARCH.md: ~2.5 PFLOP/s; no sparsity)
N_CUS = 256               # MI355X: 256 CUs in 8 XCDs
CLOCK_GHZ = 2.4           # peak engine clock (MI355X_MICROARCH.md); the peaks above are quoted at it
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_16x16x4_f32 / 32x32x2 dense peak (MI355X_MICROARCH.md: = the fp32 vector rate)
PARITY_TOL = 1e-5
NOISE_FACTOR = 2.0          # allowance on ill-conditioned episodes: tol + NOISE_FACTOR x the reference's own fp32 noise (round 3: 10)
PROFILE_ROUND = 'r06'
F_FEAT, N_ACT = 6, 2


def _profile_json(name):
    path = os.path.join(ROOT, 'profiles', '%s_%s' % (PROFILE_ROUND, name))
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def pmc_traffic(kernel, B, N, K, steps_per_launch=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/<round>_pmc_traffic.json:
    separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 correction; tools/pmc_summary.py).
    The episode-resident kernel is profiled at two launch lengths, which gives bytes(T) = fixed + per_step * T for any
    --steps.  Returns (bytes or None, note): None when no pass on these shapes is committed; the note says whether the pass
    was taken on this very build of the kernels (source hash) or on an earlier one."""
    d = _profile_json('pmc_traffic.json')
    if d is None:
        return None, 'no committed PMC pass'
    meta = d.get('_meta', {})
    if meta.get('shape') != [B, N, K]:
        return None, 'committed PMC pass is for shape %s' % (meta.get('shape'),)
    from multiagent_gnn_policies_amd import build as mgp_build
    note = 'profiles/%s_pmc_traffic.json (%s)' % (PROFILE_ROUND, 'this build' if meta.get('source_hash') == mgp_build.source_hash()
                                                  else 'taken on an earlier build of the kernels')
    try:
        if steps_per_launch is None:
            return d[kernel]['total_bytes'], note
        m = d[kernel + '_model']
        return m['fixed_bytes'] + m['bytes_per_step'] * steps_per_launch, note + '; fixed %.0f B + %.0f B/step per launch' % (
            m['fixed_bytes'], m['bytes_per_step'])
    except KeyError:
        return None, 'kernel not in the committed PMC pass'


def pmc_traffic_factored(B, N, K):
    """HBM bytes per ENV STEP of the factored path (simulator + gather stage(s) + policy tail) from the committed PMC passes
    (profiles/<round>_pmc_traffic_factored.json: tools/pmc_probe.py with PROBE_FACTORED=1), or (None, why)."""
    d = _profile_json('pmc_traffic_factored.json')
    if d is None:
        return None, 'no committed PMC pass of the factored kernels'
    if d.get('_meta', {}).get('shape') != [B, N, K]:
        return None, 'committed PMC pass is for shape %s' % (d.get('_meta', {}).get('shape'),)
    if 'spp_rollout_kernel' in d:                               # the persistent form: one launch per call of FT steps
        ft = float(d['_meta'].get('factored_steps_per_launch', 200))
        k = d['spp_rollout_kernel']
        return k['total_bytes'] / ft, ('profiles/%s_pmc_traffic_factored.json (spp_rollout_kernel: %.1f MB read + %.1f MB written per '
                                       'launch of %d steps, entry and exit included)' % (PROFILE_ROUND, k['read_bytes'] / 1e6,
                                                                                         k['write_bytes'] / 1e6, ft))
    tot, parts = 0.0, []
    for k, per_step in (('sp_sim_kernel', 1), ('spl_gather_kernel', max(K - 2, 0)), ('spl_policy_kernel', 1)):
        if per_step and k in d:
            tot += d[k]['total_bytes'] * per_step
            parts.append('%s %.2f MB' % (k, d[k]['total_bytes'] / 1e6))
    if not parts:
        return None, 'kernels not in the committed PMC pass'
    return tot, 'profiles/%s_pmc_traffic_factored.json (per launch: %s)' % (PROFILE_ROUND, ', '.join(parts))


def pmc_sq(kernel):
    """Wave-cycle breakdown and matrix-pipe occupancy of `kernel` from the committed SQ-counter pass
    (profiles/<round>_pmc_sq.json, tools/pmc_sq_summary.py), or None."""
    d = _profile_json('pmc_sq.json')
    if d is None or kernel not in d:
        return None
    v = dict(d[kernel])
    v['source'] = 'profiles/%s_pmc_sq.json' % PROFILE_ROUND
    return v
