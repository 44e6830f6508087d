"""bench.py leg: the in-run parity gate (the checker side of oracle/: torch-CPU port of the reference forward)."""
import numpy as np
import torch

from .common import N_ACT, NOISE_FACTOR, PARITY_TOL


def parity_gate(ro, n_check=16):
    """In-run parity gate, part of the cpu_baseline leg (the only place besides cpu_baseline() where bench.py touches
    oracle/, and only as the checker): the reference op sequence of actor.py:63-82 in PyTorch-CPU fp32
    (oracle/torch_port.actor_forward -- the very port that is timed as cpu_baseline, itself pinned to the reference by the
    goldens) on the identical (S, X) = (delay_gso, delay_state) the HIP kernels consume, for `n_check` sampled episodes:
      two_launch  mgp_actor_fwd on the current state
      resident    the action of a one-step mgp_rollout_steps launch from the same state (when the shape is covered)
      factored    N > 256: the action of one step of the factored path (mgp_sparse_rollout) from the same state
    max_rel is elementwise |gpu - cpu| / max(1, |cpu|); the gate is max_rel <= 1e-5.  Runs after the timed regions."""
    from oracle import torch_port
    from multiagent_gnn_policies_amd.learner.rollouts import policy_rollout
    B = ro.B
    idx = sorted(set(int(i) for i in np.linspace(0, B - 1, min(n_check, B))))
    G = ro.state.delay_gso[idx].cpu()
    X = ro.state.delay_state[idx].cpu()
    Ws = [c.weight.detach().cpu() for c in ro.actor.conv_layers]
    bs = [c.bias.detach().cpu() for c in ro.actor.conv_layers]
    with torch.no_grad():
        ref = torch_port.actor_forward(X, G, Ws, bs, 0, ro.K).double()
        # the same op sequence in fp64 on the same fp32 inputs: how far the fp32 REFERENCE itself is from the exact result
        # on this state (crowded flocks make 1/r^4 features O(1e4) and the policy ill-conditioned: two fp32 evaluations
        # of the reference -- numpy vs torch op order -- then differ by more than 1e-5 from each other)
        exact = torch_port.actor_forward(X.double(), G.double(), [w.double() for w in Ws], [b_.double() for b_ in bs], 0, ro.K)
        two = ro.actor(ro.state.delay_state, ro.state.delay_gso)[idx].cpu().double()
    res = {}

    def rel_b(u, r):                                          # per sampled episode: max over (action axis, agent)
        return ((u - r).abs() / r.abs().clamp(min=1.0)).flatten(1).max(dim=1).values
    # second witness of how well fp32 determines the result on this state: the exact evaluation of inputs moved by ONE fp32
    # rounding (every element of S and X times (1 +- 2^-24), fixed seed) -- what a single rounding of the operands does
    gen = torch.Generator().manual_seed(12345)
    sgn = lambda t: (torch.randint(0, 2, t.shape, generator=gen).double() * 2.0 - 1.0) * 2.0 ** -24
    with torch.no_grad():
        moved = torch_port.actor_forward(X.double() * (1.0 + sgn(X)), G.double() * (1.0 + sgn(G)), [w.double() for w in Ws],
                                         [b_.double() for b_ in bs], 0, ro.K)
    noise_b = torch.maximum(rel_b(ref, exact), rel_b(moved, exact))
    well = noise_b <= 0.5 * PARITY_TOL                        # episodes where the fp32 reference is determined to < tol
    paths = {'two_launch': two}
    if ro.resident_supported() or ro.factored_supported():
        # one step of the path that is `value`, from the very state whose (S, X) the reference was evaluated on: the
        # episode-resident kernel, or -- N > 256 -- the factored path's K launches (policy_rollout continues the factored state the
        # timed region left; without one it would fall back to the two-launch step and return False)
        action = torch.zeros((B, 1, N_ACT, ro.N), device=ro.sim.device)
        if policy_rollout(ro.actor, ro.sim, ro.state, 1, action=action):
            paths['resident' if ro.resident_supported() else 'factored'] = action[idx].cpu().double()
    ok = True
    for name, u in paths.items():
        r_ref, r_ex = rel_b(u, ref), rel_b(u, exact)
        res[name] = {"max_abs": float((u - ref).abs().max()), "max_rel": float(r_ref.max()),
                     "max_rel_vs_exact": float(r_ex.max()),
                     "max_rel_well_conditioned": float(r_ref[well].max()) if bool(well.any()) else None}
        plain = r_ref <= PARITY_TOL
        res[name]["episodes_within_plain_tol"] = int(plain.sum())
        relaxed = plain | (r_ex <= PARITY_TOL + NOISE_FACTOR * noise_b)
        res[name]["passed_on"] = "plain bound" if bool(plain.all()) else ("relaxed bound (see criterion)" if bool(relaxed.all())
                                                                          else "FAILED")
        # an episode passes on the plain bound, or -- where the reference's own fp32 evaluation is not determined to that
        # accuracy -- by staying within tol + NOISE_FACTOR x that episode's reference noise of the fp64 evaluation
        ok = ok and bool((plain | (r_ex <= PARITY_TOL + NOISE_FACTOR * noise_b)).all())
        if 'reference checkpoint' in ro.weights and bool(well.all()):
            ok = ok and bool(plain.all())                    # the shipped policy on well-conditioned states: plain bound only
    worst = max((v["max_rel_well_conditioned"] for v in res.values() if v["max_rel_well_conditioned"] is not None),
                default=None)
    return {"ok": ok, "tol": PARITY_TOL, "max_abs": max(v['max_abs'] for v in res.values()),
            "max_rel": max(v['max_rel'] for v in res.values()),            # over ALL checked episodes and paths
            "max_rel_well_conditioned": worst,
            "passed_on": {k_: v["passed_on"] for k_, v in res.items()},
            "reference_fp32_noise": float(noise_b.max()), "max_abs_reference_output": float(ref.abs().max()),
            "checked_episodes": len(idx), "well_conditioned_episodes": int(well.sum()), "paths": res,
            "criterion": "per sampled episode: elementwise |gpu - cpu| / max(1, |cpu|) <= tol against the fp32 CPU reference "
                         "(paths.*.episodes_within_plain_tol counts these), or, failing that, within tol + 2 x the "
                         "episode's reference_fp32_noise of the fp64 evaluation of the same op sequence on the same fp32 "
                         "inputs (reference_fp32_noise = how far fp32 evaluations of the REFERENCE are from that evaluation -- "
                         "the larger of two witnesses: the PyTorch-CPU fp32 op sequence, and the exact evaluation of inputs "
                         "moved by one fp32 rounding: "
                         "colliding agents drive 1/r^4 features to 1e6, random-init wide networks amplify them, and any "
                         "two fp32 evaluations then differ by more than tol).  max_rel is over all checked episodes, "
                         "max_rel_well_conditioned over those where the reference is determined to tol/2; passed_on says "
                         "per path which bound it passed on.  The shipped reference checkpoint on well-conditioned states is "
                         "held to the plain bound only",
            "reference": "oracle/torch_port.actor_forward: PyTorch-CPU fp32, the op sequence of reference actor.py:63-82, "
                         "on the identical (delay_gso, delay_state) of the sampled episodes"}
