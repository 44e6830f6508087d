"""bench.py leg: the `roofline` and `kernels` objects of the JSON line.

Three shapes of the timed region, three dominant kernels:
  resident   rollout_kernel: nothing streams from HBM; its nearest ceiling is vector-instruction issue.  The peak is the
             instruction-mix-weighted issue rate: CUs x 4 SIMDs x clock / (mean cycles per wave64 VALU instruction), the mean
             taken over the static mix of the kernel's step loop (tools/isa_mix.py -> profiles/<round>_isa_mix.json) with the
             per-class costs tools/harness/valu_rate.hip measured on MI355X (profiles/<round>_valu_rate.txt: 2.4 cycles for
             fp32 fma / mul / add, 32-bit and / or / add / mov; 4.3 for nearly everything else; 8.2 transcendental).  achieved =
             SQ_INSTS_VALU per episode-step (committed counter pass of THIS build) x live episode-steps per second.
  factored   N > 256: sp_sim_kernel + gather + policy launches; required bytes per step against the HBM rate.
  dense      the two-launch path: the fused Actor forward (or the aggregation alone) against the HBM rate.
`dense_kernels` (the HBM-roofline figures of the kernels that stream the dense operator) rides on all three."""
from . import kernels as _k
from .common import (CLOCK_GHZ, F_FEAT, HBM_PEAK_GBS, MFMA_BF16_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS, N_CUS, PROFILE_ROUND,
                     _profile_json, pmc_sq, pmc_traffic, pmc_traffic_factored)


def valu_peak():
    """(G wave-instructions / s, mean cycles per VALU instruction, note) from the committed ISA mix of the headline kernel; the
    flat 4-cycle figure of rounds 1-5 if no mix is committed."""
    mix = _profile_json('isa_mix.json')
    if mix and mix.get('mean_cycles_per_valu_instruction'):
        from multiagent_gnn_policies_amd import build as mgp_build
        c = float(mix['mean_cycles_per_valu_instruction'])
        same = mix.get('source_hash') == mgp_build.source_hash()
        return (N_CUS * 4 * CLOCK_GHZ / c, c,
                "mix-weighted issue cost %.2f cycles per wave64 VALU instruction: static mix of the step loop (%d instructions: %s) "
                "priced with the measured per-class costs %s (profiles/%s_isa_mix.json, %s; rates %s)"
                % (c, mix.get('static_valu_instructions', 0), mix.get('by_class'), mix.get('cycles_per_wave_instruction'),
                   PROFILE_ROUND, 'this build' if same else 'taken on an earlier build of the kernels', mix.get('rates_file')))
    return N_CUS * 4 * CLOCK_GHZ / 4.0, 4.0, "no committed ISA mix: 4 cycles per wave64 VALU instruction assumed"


def hbm_block(res, key, kname, pmc_names, B, N, K):
    r = res[key]
    tr, tr_note, sq = None, 'not profiled', None
    for pmc_name in pmc_names:                               # the variant the library picked for this shape comes first
        tr, tr_note = pmc_traffic(pmc_name, B, N, K)
        sq = pmc_sq(pmc_name)
        if tr is not None:
            break
    return {"kernel": kname, "bound": "hbm", "achieved": r['gbs'], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": r['gbs'] / HBM_PEAK_GBS, "traffic": tr, "traffic_source": tr_note,
            "algorithmic_bytes_per_launch": r['bytes'], "avg_launch_ms": r['ms'], "sq": sq}


def roofline_blocks(device, B, N, K, hidden, actor, flock_c, mode, steps, res_launches=None, res_launch_ms=None, el_fact=None,
                    factored_persistent=None):
    """-> (roofline, kernels) for the JSON line.  mode: 'resident' | 'factored' | 'dense' (which path is `value`)."""
    res, n_sets = _k.kernel_rooflines(device, B, N, K, actor, flock_c)
    fused = getattr(actor, 'use_fused', False) and actor.ind_agg == 0
    # HBM-roofline figures of the kernels that stream the dense operator G (B,K,N,N) from HBM on every launch --
    # north_star's "fraction of HBM roofline for the S^k X aggregation" -- measured live with HIP events on the
    # launch stream over rotating input sets larger than the Infinity Cache; algorithmic bytes per SURVEY.md 8(d)
    dense = {
        "actor_fwd": hbm_block(res, 'actor_fwd', "mgp_actor_fwd: for N <= 128 actor_fwd_pol_kernel (the reference's policy shape [32, 32] compiled in: "
                               "aggregation on 4x4x1 fp32 MFMA, hidden layers on split-bf16 MFMA) or actor_fwd_mfma_kernel (any "
                               "widths <= 128, fp32 MFMA), actor_fwd_kernel otherwise: 4KN^2 + 4KFN + 4 nA N bytes per "
                               "episode", ('actor_fwd_pol_kernel', 'actor_fwd_mfma_kernel', 'actor_fwd_kernel')
                               if hidden == [32, 32] else ('actor_fwd_mfma_kernel', 'actor_fwd_kernel'), B, N, K),
        "agg_fwd": hbm_block(res, 'agg_fwd', "mgp_agg_fwd: agg_fwd_mfma4_kernel for N <= 128 (four waves per (episode, tap): a wave "
                             "streams half the rows of its column block), agg_fwd_kernel otherwise (aggregation "
                             "X.G alone): 4KN^2 + 8KFN bytes per episode", ('agg_fwd_mfma4_kernel', 'agg_fwd_mfma_kernel', 'agg_fwd_kernel'),
                             B, N, K),
        "sim_state_step": hbm_block(res, 'sim_state_step', "mgp_flock_step_advance: flock_advance_kernel for N <= 128 (one workgroup per "
                                    "episode: sim step + delayed-GSO / delay-line transition, source slice staged in LDS by "
                                    "LDS-DMA), the row-tiled flock_step_kernel<advance> otherwise",
                                    ('flock_advance_kernel',) if (N <= 128 and N % 4 == 0) else ('flock_step_kernel',), B, N, K),
        "rotating_input_sets": n_sets,
    }
    if mode == 'resident':
        # dominant (only) kernel of the timed region.  Nothing streams from HBM; the matrix pipe is a few per cent busy.
        n_launch = res_launches                              # launches of the timed region (split at episode ends / 2000 steps)
        spl = steps / float(n_launch)
        dims = [F_FEAT * K] + hidden
        flops_unit = 2.0 * N * sum(a * b_ for a, b_ in zip(dims[:-1], dims[1:]))
        flops = flops_unit * B * spl
        ms = res_launch_ms / n_launch
        alg = (4 * K * N * N + 8 * K * F_FEAT * N) * B * spl
        # which instructions run those flops: layers whose inputs have <= 32 channels run on split-bf16 MFMA (three bf16
        # pieces per fp32 operand, six 16x16x32 products per fp32 product: rollout_common.h ro_layer_bf16), the 64-wide
        # build on fp32 MFMA 16x16x4
        split = N <= 128 or max(hidden) <= 32        # every build of the N <= 128 kernel; beyond, widths <= 32 only
        matrix_note = ({"form": "split-bf16: v_mfma_f32_16x16x32_bf16, 6 bf16 products per fp32 product, K padded to 32",
                        "bf16_flops_per_episode_step": 6.0 * 2.0 * N * sum(32 * 16 * ((b_ + 15) // 16) for b_ in dims[1:]),
                        "bf16_peak_TFLOPs": MFMA_BF16_PEAK_TFLOPS}
                       if split else {"form": "fp32: v_mfma_f32_16x16x4_f32"})
        if split:
            matrix_note["frac_of_bf16_peak"] = (matrix_note["bf16_flops_per_episode_step"] * B * spl / ms / 1e9 /
                                                MFMA_BF16_PEAK_TFLOPS)
        tr, tr_note = pmc_traffic('rollout_kernel', B, N, K, steps_per_launch=spl)
        mfma = {"achieved": flops / ms / 1e9, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": flops / ms / 1e9 / MFMA_F32_PEAK_TFLOPS,
                "note": "algorithmic flops of the MFMA-run layers against the fp32 matrix peak (rounds 1-4 reported this as "
                        "`frac`); the pipe itself is sq.mfma_busy busy"}
        # The SQ counter pass and the ISA mix are those of the headline shape (tools/pmc_probe.py, tools/isa_mix.py).  Any other
        # shape has no VALU evidence of its own: its line carries the algorithmic matrix-pipe figure as `frac` and says so.
        sq = pmc_sq('rollout_kernel')
        sq_meta = (_profile_json('pmc_sq.json') or {}).get('_meta', {})
        profiled_shape = (N == 100 and K == 3 and hidden == [32, 32])
        if not profiled_shape:
            sq = None
        valu_unit = None
        if sq is not None and sq.get('valu_insts') and sq_meta.get('rollout_episode_steps_per_launch'):
            valu_unit = sq['valu_insts'] / float(sq_meta['rollout_episode_steps_per_launch'])
        vpeak, vcyc, vnote = valu_peak()
        valu_ach = (valu_unit * B * spl / ms / 1e6) if valu_unit else None
        common = {
            "kernel": "rollout_kernel (episode-resident: power-iterated aggregation along neighbour lists + split-bf16 MFMA "
                      "filter/MLP + sim step + Verlet-listed neighbour search, %.0f steps per launch on average)" % spl,
            "mfma": mfma, "traffic": tr, "traffic_source": tr_note,
            "algorithmic_flops_per_launch": flops, "algorithmic_flops_per_episode_step": flops_unit,
            "avg_launch_ms": ms, "steps_per_launch": spl, "resident": True,
            "matrix_instructions": matrix_note,
            "launch_timing": "HIP events stamped by the launch itself (mgp_set_launch_events) in a second pass over the same "
                             "steps of the same episodes (paths.resident.ms_per_step_event_pass); the pass `value` is taken "
                             "from carries no events",
            "sq": sq,
            "equivalent_hbm": {"GBps": alg / ms / 1e6, "frac_of_peak": alg / ms / 1e6 / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_launch": alg,
                               "note": "NOT a roofline fraction: the bytes the dense-contract aggregation (4KN^2 + 8KFN "
                                       "per episode-step, SURVEY.md 8d) WOULD stream for these steps / launch time; "
                                       "inside the launch the operator exists only as neighbour lists in LDS"},
            "dense_kernels": dense}
        if valu_ach is not None:
            roof = dict(common, bound="valu", achieved=valu_ach, peak=vpeak, unit="G wave-instructions/s", frac=valu_ach / vpeak,
                        valu_insts_per_episode_step=valu_unit, mean_cycles_per_valu_instruction=vcyc,
                        bound_note="vector-ALU issue: %d CUs x 4 SIMDs x %.1f GHz / %.2f cycles; %s; instruction count from the "
                                   "committed SQ counter pass (profiles/%s_pmc_sq.json, source hash %s), rate from this run's "
                                   "event-stamped launches" % (N_CUS, CLOCK_GHZ, vcyc, vnote, PROFILE_ROUND,
                                                               (sq_meta.get('source_hash') or 'not recorded')[:12]),
                        limiter="the vector pipes are the resource nearest their ceiling (frac), but a step is a chain of "
                                "dependent phases of ONE episode per CU: a wave alone on its SIMD issues one instruction per "
                                ">= 5.2 cycles whatever the class and a dependent one per 8.4 (profiles/%s_valu_rate.txt), waves "
                                "are parked at s_waitcnt / barriers sq.wait_any of their cycles, and the policy phase keeps 7 of "
                                "16 waves busy -- issue, LDS latency and barriers add up to the rest.  Neither HBM (traffic = "
                                "state in/out + 8 B of reward per step) nor the matrix pipe (sq.mfma_busy) is near saturation"
                                % PROFILE_ROUND)
        else:
            roof = dict(common, bound="mfma", achieved=mfma["achieved"], peak=mfma["peak"], unit=mfma["unit"], frac=mfma["frac"],
                        bound_note="algorithmic flops of the MFMA-run layers / launch duration against the fp32 matrix peak.  "
                                   "The SQ counter pass and the ISA mix behind the vector-issue figure exist for the headline shape "
                                   "(N = 100, K = 3, hidden [32, 32]) only; this shape's limiter was not measured (on the headline "
                                   "shape it is vector-instruction issue and the phase chain, not the matrix pipe)")
        return roof, _kernels(res)
    if mode == 'factored':
        # N > 256: the factored state in HBM, K launches per env step (simulator, K - 2 gather stages, policy tail).  No
        # dense operator exists; the bytes a step REQUIRES (each array once) per episode:
        #   simulator    x in + out (2 x 32 N), bit rows (8 NW N), row weights (4 N), feature rows (32 N), lists (32 N)
        #   gather q     lists of A_{t-q+1} (32 N) + row weights (4 N), source rows of taps >= q in (32 N each) and out
        #   policy tail  lists + weights of the last factor, its source rows, the K finished taps (32 N each), action (8 N)
        NW = (N + 63) // 64
        sim_b = (64 + 8 * NW + 4 + 32 + 32) * N
        gather_b = sum((36 + 64 * (K - q)) * N for q in range(1, K - 1))          # stages 1 .. K-2: taps q .. K-1 in and out
        policy_b = ((36 + 32) * (1 if K >= 2 else 0) + 32 * K + 8) * N
        req = (sim_b + gather_b + policy_b) * B
        ms_step = 1e3 * el_fact / steps
        trf, trf_note = pmc_traffic_factored(B, N, K)
        # which form ran: the caller's query of the library (mgp_sparse_rollout_persistent); without it, what the PMC file profiled
        persistent = factored_persistent if factored_persistent is not None else 'spp_rollout_kernel' in (trf_note or '')
        roof = {
            "kernel": ("factored step inside spp_rollout_kernel (one launch of persistent workgroups per call; gather stage, policy "
                       "tail and cell-list simulator as phases behind three sibling exchanges per step)" if persistent else
                       "factored step: sp_sim_kernel + %d x spl_gather_kernel + spl_policy_kernel (%d launches per env step)"
                       % (max(K - 2, 0), max(K, 2))),
            "bound": "hbm", "achieved": req / ms_step / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": req / ms_step / 1e6 / HBM_PEAK_GBS, "traffic": trf, "traffic_source": trf_note,
            "required_bytes_per_step": req,
            "required_bytes_per_episode_step": {"simulator": sim_b, "gather_stages": gather_b, "policy_tail": policy_b},
            "avg_step_ms": ms_step,
            "note": "bytes the factored state REQUIRES per env step (bit rows, lists, row weights, feature rings, agent "
                    "states: each array once, as the K-launch form defines them) / wall time per step of the timed region / 8 TB/s.  "
                    "The path is latency-, not bandwidth-bound: three sibling exchanges and a dependent chain of phases per "
                    "step (profiles/%s_factored_step_stamps.txt); `traffic` = PMC bytes per env step; the dense-contract "
                    "bytes of this shape would be %.0f MB per step" % (PROFILE_ROUND, (4 * K * N * N + 8 * K * F_FEAT * N) * B / 1e6),
            "dense_kernels": dense}
        return roof, _kernels(res)
    return dict(dense["actor_fwd" if fused else "agg_fwd"], dense_kernels=dense), _kernels(res)


def _kernels(res):
    return {k: {"avg_launch_ms": v['ms'], "algorithmic_bytes": v['bytes'], "GBps": v['gbs']} for k, v in res.items()}
