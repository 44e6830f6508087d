#!/usr/bin/env python3
"""bench.py -- agent-steps/sec of the per-step hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" advances EVERY episode of the batch by one environment step through the whole path
    Actor forward (aggregation X.G + filter GEMM + tanh MLP)  ->  action  ->  sim step
    ->  delayed-GSO / delay-line update,
all state resident on the GPU, no host round trip.  Up to three implementations of the same step are timed in the same run:
  resident    mgp_rollout_steps: ALL timed steps in one launch of the episode-resident kernel (one workgroup per
              episode; delay line / agent states / neighbour lists of the last K-1 networks / weights in LDS, the
              aggregation power-iterated along those lists; HBM sees the state on entry and exit).  This is `value`
              when the shape is covered (N <= 256, widths <= 64; one hidden layer up to 128 wide at N <= 128).
  factored    N > 256 only: the same factored state kept in HBM as bit rows / feature rings, K launches per step
              (mgp_sparse_rollout); the dense delay_gso of the contract is rebuilt on first read, outside the timed region.
  two_launch  mgp_actor_fwd + mgp_flock_step_advance per step (dense operator streamed from HBM every step),
              replayed from a captured HIP graph; reported next to it, and `value` for shapes the resident kernel
              does not cover.
Workload = BASELINE.json configs[1]: FlockingRelative-v0, N=100 agents, K=3 taps, 256 parallel episodes
PER GPU (weak scaling: episodes are independent, ranks never communicate in the rollout).
value = (episodes * agents * steps * ranks) / max-over-ranks wall time.

Also reported on the same JSON line:
  roofline      the dominant kernel of the timed region.  Resident path: rollout_kernel, bound = "mfma" -- algorithmic flops of
                the MFMA-run layers x episode-steps per launch / launch duration (fp32 MFMA peak; nothing streams from HBM, the
                `sq` fractions say what bounds it) with `traffic` = what HBM really moved (PMC) and `equivalent_hbm` = the
                dense-contract bytes (4KN^2 + 8KFN per episode-step) over the same duration, labelled as NOT a
                roofline fraction.  The duration comes from HIP events the launch stamps itself (mgp_set_launch_events); they
                cost a launch ~11 us of wall time, so the resident region is timed TWICE over the same steps of the same
                episodes (rewind to the reset, roll forward): the plain pass is `value`, the stamped pass is the roofline's
                duration (paths.resident.ms_per_step_event_pass is its wall time).  `dense_kernels` holds the HBM-roofline
                figures of the kernels that do stream the dense operator from HBM (fused Actor forward, aggregation alone, fused
                sim + state step), measured live on rotating input sets larger than the 256 MiB Infinity Cache.
  kernels       same measurement for the other stand-alone kernels.
  cpu_baseline  the PyTorch-CPU port of the reference op sequence (oracle/torch_port.py, "kind": "port"),
                reference-style B=1 loop, timed on this host for a bounded sample (rank 0, N=1 only).
  parity        the in-run parity gate (north_star "within 1e-5 fp32 ... in the same run"): the
                action of one resident step and of one two-launch step against the SAME CPU port's Actor forward on the
                identical (S, X) the kernels consumed, for 16 sampled episodes: {ok, tol, max_abs, max_rel, ...}.
                A failed gate prints the line with "ok": false and exits with status 3.
Which implementation is `value` is a function of the SHAPE only (never of --steps): resident where mgp_rollout_supported
says so (N <= 256, widths <= 64; one hidden layer up to 128 wide at N <= 128), factored for N > 256 where mgp_sparse_policy_supported, else two_launch; config.step_path
names it and paths.* carries every implementation that was timed.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from multiagent_gnn_policies_amd import ops, parallel  # noqa: E402
from multiagent_gnn_policies_amd.envs import FlockParams, VecFlock  # noqa: E402
from multiagent_gnn_policies_amd.envs.flocking import use_grid  # noqa: E402
from multiagent_gnn_policies_amd.learner import Actor  # noqa: E402
from multiagent_gnn_policies_amd.learner.state_with_delay import BatchedDelayState  # noqa: E402

DEFAULT_STEPS = 1000      # env steps in the timed region (resident path: one launch; ~10 ms)
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
MFMA_BF16_PEAK_TFLOPS = 2500.0                     # dense bf16 (MI355X_MICROARCH.md: ~2.5 PFLOP/s; no sparsity)
N_CUS = 256               # MI355X: 256 CUs in 8 XCDs
CLOCK_GHZ = 2.4           # peak engine clock (MI355X_MICROARCH.md); the peaks above are quoted at it
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_16x16x4_f32 / 32x32x2 dense peak (MI355X_MICROARCH.md: = the fp32 vector rate)
PARITY_TOL = 1e-5
NOISE_FACTOR = 2.0          # allowance on ill-conditioned episodes: tol + NOISE_FACTOR x the reference's own fp32 noise (round 3: 10)
PROFILE_ROUND = 'r05'
F_FEAT, N_ACT = 6, 2


POLICY_DIR = os.path.join(ROOT, 'tests', 'golden', 'policies')
ENV_TAGS = {'FlockingRelative-v0': 'relative', 'FlockingLeader-v0': 'leader', 'FlockingTwoFlocks-v0': 'twoflocks',
            'FlockingStochastic-v0': 'stochastic'}


def _load_npz_policy(actor, path):
    """All or nothing: load_state_dict copies the matching tensors before it raises on a mismatch, which would leave a
    hybrid (e.g. K = 4: default-init filter, checkpoint readout) -- a policy nobody trained, whose flocks collapse."""
    with np.load(path) as z:
        sd = {k.replace('__', '.'): torch.from_numpy(z[k]) for k in z.files if k != 'meta'}
    own = actor.state_dict()
    if set(own) == set(sd) and all(tuple(own[k].shape) == tuple(sd[k].shape) for k in own):
        actor.load_state_dict(sd)
        return True
    return False


def load_weights(actor, env='FlockingRelative-v0', n_agents=100):
    """A TRAINED policy for every shape that is benched or parity-gated (a random network drives a freshly reset flock into
    itself: 1/r^4 features of 1e4 and more, an ill-conditioned forward):
      1. the reference's shipped checkpoint (tests/golden/ckpt_dagger_k3.npz: plain arrays) -- FlockingRelative-v0, K = 3,
         hidden [32, 32]; it is also the reference's own transfer policy for larger flocks (test_model_transfer.py);
      2. tests/golden/policies/policy_<env>_k<K>_h<H>x<L>_n<N>.npz -- trained with this package's own vectorised DAGGER loop
         on the reference's schedule (tools/train_policies.py; rewards next to the teacher's in summary.json): same
         environment, K and hidden sizes, the file whose N is nearest; then the plain environment's policy of that shape
         (the variants change resets / leaders / links, not the observation);
      3. torch default init under seed 11 (cfg/dagger.cfg:9), said so in the returned description."""
    import glob
    import re
    layers = [int(v) for v in actor.layers]
    K, hidden = int(actor.k), layers[1:-1]
    ck = os.path.join(ROOT, 'tests', 'golden', 'ckpt_dagger_k3.npz')
    if env == 'FlockingRelative-v0' and os.path.exists(ck) and _load_npz_policy(actor, ck):
        return 'reference checkpoint actor_FlockingRelative-v0_dagger_k3'
    if hidden and len(set(hidden)) == 1:
        for tag in dict.fromkeys([ENV_TAGS.get(env, 'relative'), 'relative']):
            pat = os.path.join(POLICY_DIR, 'policy_%s_k%d_h%dx%d_n*.npz' % (tag, K, hidden[0], len(hidden)))
            cands = sorted(glob.glob(pat), key=lambda f: (abs(int(re.search(r'_n(\d+)\.npz$', f).group(1)) - n_agents), f))
            for f in cands:
                if _load_npz_policy(actor, f):
                    return 'trained policy tests/golden/policies/%s (tools/train_policies.py)' % os.path.basename(f)
    if os.path.exists(ck) and _load_npz_policy(actor, ck):          # a variant at the checkpoint's shape with no policy of its own
        return 'reference checkpoint actor_FlockingRelative-v0_dagger_k3'
    return 'default init (seed 11)'


class Rollout(object):
    """Device-resident vectorised rollout; one `step()` = one env step for all B episodes."""

    def __init__(self, device, B, N, K, hidden, seed, init_mode='auto', comm_radius=1.0, env=None, **variant):
        self.B, self.N, self.K = B, N, K
        if env is None:                                        # callers that pass the variant's fields instead of its id
            env = ('FlockingLeader-v0' if variant.get('n_leaders') else 'FlockingTwoFlocks-v0' if variant.get('two_flocks')
                   else 'FlockingStochastic-v0' if variant.get('link_drop') else 'FlockingRelative-v0')
        else:
            from multiagent_gnn_policies_amd.envs.flocking import _REGISTRY
            variant = dict(getattr(_REGISTRY[env], 'variant', {}), **variant)
        self.env = env
        # variant: FlockParams fields of the environment variants (n_leaders = 2: FlockingLeader-v0, two_flocks = True:
        # FlockingTwoFlocks-v0, link_drop: FlockingStochastic-v0)
        self.params = FlockParams(n_agents=N, init_mode=init_mode, comm_radius=comm_radius, **variant)
        self.sim = VecFlock(B, self.params, device)
        torch.manual_seed(11)
        self.actor = Actor(F_FEAT, N_ACT, hidden, K, 0).to(device)
        self.weights = load_weights(self.actor, env, N)
        self.actor.eval()
        self.state = BatchedDelayState(device, B, K, F_FEAT, N)
        self.sim.reset(np.random.RandomState(seed))
        self.state.push(self.sim.network, self.sim.features)
        self._rw = None
        self._plan = None         # ResidentPlan: prebuilt weight image + bound argument list (the weights are fixed)
        self._rws = {}            # reward buffers by launch length (allocated outside the timed region by the warm-up)

    def step(self):
        with torch.no_grad():
            out = self.actor(self.state.delay_state, self.state.delay_gso)      # (B,1,2,N)
            # action consumed as (B,1,2,N); sim step + delayed-GSO / delay-line transition in one fused kernel when
            # the shape allows it (N % 4 == 0, N <= 128), else mgp_flock_step + mgp_gso_advance
            self.sim.step_advance(out, self.state)

    def resident_supported(self):
        return ops.rollout_supported(tuple(self.actor.layers), self.K, self.N)

    def factored_supported(self):
        """N > 256: the factored state in HBM (learner/sparse_rollout.py), K launches per step."""
        from multiagent_gnn_policies_amd.learner.sparse_rollout import sparse_supported
        return self.N > 256 and sparse_supported(self.actor, self.K, self.N)

    def restart(self, seed):
        """Back to a reset observation (the factored path starts where the history is known)."""
        self.sim.reset(np.random.RandomState(seed))
        self.state.reset()
        self.state.push(self.sim.network, self.sim.features)

    def presample_resets(self, n_sets, seed):
        """`n_sets` batches of reset states drawn on the host from the environment's reset distribution (control logic, once
        per episode: outside every timed region) and parked on the device: device_reset() installs one without host work."""
        rng = np.random.RandomState(seed)
        from multiagent_gnn_policies_amd.envs.flocking import sample_initial_states   # (the sequential sampler's states, see there)
        return [torch.from_numpy(sample_initial_states(rng, self.params, self.B, self.sim.device)).to(self.sim.device)
                for _ in range(n_sets)]

    def device_reset(self, x_dev, align=None):
        """Episode end (TimeLimit, FLOCK-SPEC item 6) for every lane, entirely on the device and asynchronous: install the
        pre-sampled states, recompute the observation, restart the delay line from it (reference gnn_dagger.py:150: a fresh
        MultiAgentStateWithDelay without prev_state).  `align` = (x buffer, state buffer index) a captured HIP graph of steps
        expects at its start: the ping-pong roles are arranged so that the graph stays replayable."""
        self.sim.x.copy_(x_dev)
        self.sim.refresh()
        self.state.reset()
        if align is not None:
            cap_x, cap_cur = align
            self.state._cur = (1 - cap_cur) if self.sim.x.data_ptr() == cap_x else cap_cur
        self.state.push(self.sim.network, self.sim.features)

    def prepare_resident(self, lengths, chunk=2000):
        """Allocate the per-step reward buffers of the launches a timed region will issue (outside that region)."""
        for n in lengths:
            for t in {min(chunk, n - d) for d in range(0, n, chunk)}:
                if t > 0 and t not in self._rws:
                    self._rws[t] = torch.zeros((self.B, t), device=self.sim.device, dtype=torch.float64)

    def run_resident(self, n_steps, chunk=2000):
        """n_steps env steps on the episode-resident kernel (launches of <= chunk steps)."""
        from multiagent_gnn_policies_amd.learner.rollouts import policy_rollout, ResidentPlan
        if self.N > 256:                                         # factored state in HBM: policy_rollout selects it
            done = 0
            while done < n_steps:
                t = min(chunk, n_steps - done)
                if self._rw is None or self._rw.shape[1] != t:
                    self._rw = torch.zeros((self.B, t), device=self.sim.device, dtype=torch.float64)
                assert policy_rollout(self.actor, self.sim, self.state, t, rewards=self._rw)
                done += t
            return
        if self._plan is None:                                   # host side of the repeated launch, bound once
            self._plan = ResidentPlan(self.actor, self.sim, self.state)
        done = 0
        while done < n_steps:
            t = min(chunk, n_steps - done)
            rw = self._rws.get(t)
            if rw is None:
                rw = self._rws[t] = torch.zeros((self.B, t), device=self.sim.device, dtype=torch.float64)
            self._rw = rw
            self._plan.run(t, rewards=rw, update_sim_reward=False, lazy_dense=getattr(self, 'lazy_dense', True))
            done += t


class Episodes(object):
    """The timed regions honour the environment's time limit: every `episode_steps` env steps since the last reset all lanes
    are reset ON THE DEVICE inside the region (Rollout.device_reset on pre-sampled states) -- a long region then averages
    over whole episodes (dense start, aligned flock) instead of over one flock that spreads for thousands of steps (mean
    degree 8.5 at reset, 2.5 after 1100 steps without resets).  episode_steps = 0: never reset."""

    def __init__(self, ro, episode_steps, presets):
        self.ro, self.E, self.presets = ro, int(episode_steps), presets
        self.since, self.count, self.align = 0, 0, None

    def rewind(self, seed):
        self.ro.restart(seed)
        self.since, self.count = 0, 0

    def chunks(self, seq):
        """Launch lengths advance() will issue for the calls `seq` (from a rewind): for pre-allocating per-length buffers."""
        out, since = set(), 0
        for n in seq:
            while n > 0:
                t = n if not self.E else min(n, self.E - since)
                out.add(t)
                since = (since + t) % self.E if self.E else since + t
                n -= t
        return sorted(out)

    def advance(self, step_fn, n):
        """n env steps through step_fn(t), split at episode ends."""
        while n > 0:
            t = n if not self.E else min(n, self.E - self.since)
            step_fn(t)
            self.since += t
            n -= t
            if self.E and self.since >= self.E:
                self.ro.device_reset(self.presets[self.count % len(self.presets)], self.align)
                self.count += 1
                self.since = 0


def time_kernel(fn, n_sets, iters):
    """Average duration (ms) of one launch of fn(i): HIP events on the launch stream around a captured HIP graph of
    `n_sets` back-to-back launches (one per rotating input set), replayed until `iters` launches have run.  The graph
    keeps the measurement GPU-bound (an eager Python loop is host-bound below ~25 us per launch); what remains on
    top of the kernel is the ~1.5 us dependent-kernel boundary, so the figure is slightly conservative."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(min(n_sets, 3)):
            fn(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(n_sets):
            fn(i)
    reps = max(2, (iters + n_sets - 1) // n_sets)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (reps * n_sets)
    del graph
    return ms


def kernel_rooflines(device, B, N, K, actor, flock_c):
    """Per-kernel live measurements on rotating buffers (working set > 256 MiB Infinity Cache)."""
    g_bytes = 4 * K * N * N * B
    n_sets = max(2, int(np.ceil(320 * 2 ** 20 / g_bytes)))
    n_sets = min(n_sets, 24)
    gen = torch.Generator(device=device).manual_seed(1)
    Gs = [torch.rand((B, K, N, N), device=device, generator=gen) for _ in range(n_sets)]
    Xs = [torch.randn((B, K, F_FEAT, N), device=device, generator=gen) for _ in range(n_sets)]
    res = {}
    # --- aggregation (the roofline kernel)
    iters = max(50, 4 * n_sets)
    ms = time_kernel(lambda i: ops.agg_fwd(Xs[i].permute(0, 2, 1, 3), Gs[i]), n_sets, iters)
    agg_bytes = (4 * K * N * N + 8 * K * F_FEAT * N) * B
    res['agg_fwd'] = dict(ms=ms, bytes=agg_bytes, gbs=agg_bytes / ms / 1e6)
    # --- whole Actor forward (aggregation + filter GEMM + MLP readout), fused kernel
    with torch.no_grad():
        ms = time_kernel(lambda i: actor(Xs[i], Gs[i]), n_sets, iters)
    n_params = sum(p.numel() for p in actor.parameters())
    act_bytes = (4 * K * N * N + 4 * K * F_FEAT * N + 4 * N_ACT * N) * B + 4 * n_params
    res['actor_fwd'] = dict(ms=ms, bytes=act_bytes, gbs=act_bytes / ms / 1e6)
    # --- delayed-GSO update: read A, G_prev[1..K-2]; write K slices
    As = [torch.zeros((B, N, N), device=device) for _ in range(n_sets)]
    for a in As:
        mask = torch.rand((B, N, N), device=device, generator=gen) < (8.0 / N)
        a.copy_(mask.float() / mask.float().sum(-1, keepdim=True).clamp(min=1))
    Gn = [torch.empty((B, K, N, N), device=device) for _ in range(2)]
    Xn = torch.empty((B, K, F_FEAT, N), device=device)
    Xt = torch.randn((B, F_FEAT, N), device=device, generator=gen)
    ms = time_kernel(lambda i: ops.gso_update_into(As[i], Gs[i], Gn[i % 2], Xt, Xs[i], Xn, True), n_sets, iters)
    gso_bytes = 4 * (2 * K - 1) * N * N * B
    res['gso_update'] = dict(ms=ms, bytes=gso_bytes, gbs=gso_bytes / ms / 1e6)
    # --- sim step: writes the dense N x N network matrix
    xs = torch.randn((B, N, 4), device=device, dtype=torch.float64, generator=gen) * 3.0
    u = torch.zeros((B, N, 2), device=device)
    feat = torch.empty((B, F_FEAT, N), device=device)
    rew = torch.empty((B,), device=device, dtype=torch.float64)
    ms = time_kernel(lambda i: ops.flock_step(xs, u, flock_c, A=As[i], feat=feat, reward=rew), n_sets, iters)
    sim_bytes = (4 * N * N + 8 * 4 * N * 2 + 4 * (2 + 6) * N) * B
    res['flock_step'] = dict(ms=ms, bytes=sim_bytes, gbs=sim_bytes / ms / 1e6)
    # --- fused sim step + delayed-GSO / delay-line transition (what the timed step actually launches):
    #     writes A_t (N^2) and the products (K-2) N^2, gathers (K-2) N^2 of G_prev, features/labels/state are small
    try:
        from multiagent_gnn_policies_amd.envs import VecFlock
        params = FlockParams(n_agents=N, init_mode='grid')
        sim = VecFlock(B, params, device, with_expert=True)
        sim.reset(np.random.RandomState(3))
        states = [BatchedDelayState(device, B, K, F_FEAT, N) for _ in range(min(n_sets, 6))]
        for st_ in states:
            st_.push(sim.network, sim.features)
            sim.step_advance(u.view(B, N, 2), st_)
        ms = time_kernel(lambda i: sim.step_advance(u.view(B, N, 2), states[i % len(states)]), len(states), iters)
        ss_bytes = (4 * N * N * (1 + 2 * max(K - 2, 0)) + 8 * 4 * N * 2 + 4 * (2 + 6 + 2) * N + 4 * 2 * (K - 1) * F_FEAT * N) * B
        res['sim_state_step'] = dict(ms=ms, bytes=ss_bytes, gbs=ss_bytes / ms / 1e6)
    except Exception as e:                               # shapes the fused kernel does not cover
        res['sim_state_step'] = dict(ms=float('nan'), bytes=0, gbs=0.0, note=str(e))
    del Gs, Xs, As
    torch.cuda.empty_cache()
    return res, n_sets


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


def mean_degree(state):
    """Mean number of neighbours per agent in the current networks (rows of delay_gso[:, 1]), over all episodes."""
    if state.K < 2:
        return None
    return float((state.delay_gso[:, 1] != 0).sum(dim=-1).double().mean().item())


def cpu_baseline(N, K, hidden, budget_s=12.0, init_mode='auto', actor=None, variant=None):
    """Reference-style single-episode CPU loop (oracle/torch_port.py + numpy sim), bounded sample.  `actor`: the policy the
    GPU legs ran (its weights are copied to the host); `variant`: FlockParams fields of the environment variant."""
    from oracle import flock as ofl, torch_port
    path = os.path.join(ROOT, 'tests', 'golden', 'ckpt_dagger_k3.npz')
    torch.manual_seed(11)
    if actor is not None:
        Ws = [c.weight.detach().cpu().clone() for c in actor.conv_layers]
        bs = [c.bias.detach().cpu().clone() for c in actor.conv_layers]
    elif os.path.exists(path) and K == 3 and hidden == [32, 32]:
        with np.load(path) as z:
            Ws = [torch.from_numpy(z[f'conv_layers__{i}__weight']) for i in range(3)]
            bs = [torch.from_numpy(z[f'conv_layers__{i}__bias']) for i in range(3)]
    else:
        dims = [F_FEAT] + hidden + [N_ACT]
        Ws = [torch.randn(dims[i + 1], dims[i], K if i == 0 else 1, 1) * 0.1 for i in range(len(dims) - 1)]
        bs = [torch.zeros(dims[i + 1]) for i in range(len(dims) - 1)]
    p = ofl.FlockParams(n_agents=N, init_mode=init_mode, **(variant or {}))
    x0 = ofl.reset(np.random.RandomState(0), p)
    default_threads = torch.get_num_threads()
    runs = []
    for threads in sorted({1, default_threads}):
        torch.set_num_threads(threads)
        x = x0
        torch_port.rollout_steps(x, p, Ws, bs, K, 5)                     # warm-up
        chunk, done = 50, 0
        t0 = time.perf_counter()
        while True:
            _, x = torch_port.rollout_steps(x, p, Ws, bs, K, chunk)
            done += chunk
            el = time.perf_counter() - t0
            if el >= budget_s / 2 or done >= 20000:
                break
        runs.append((N * done / el, threads, done, el))
    torch.set_num_threads(default_threads)
    best = max(runs)
    return dict(value=best[0], unit='agent-steps/s', cores=best[1], kind='port',
                sample='1 episode x %d steps (%.1f s) at %d torch thread(s), reference-style B=1 loop: numpy fp64 '
                       'sim + torch-CPU state update (incl. curr_gso) + Actor forward; all thread settings tried: '
                       '%s; host has %d logical cores'
                       % (best[2], best[3], best[1],
                          ', '.join('%d thr -> %.3g agent-steps/s' % (r[1], r[0]) for r in runs), os.cpu_count() or 0),
                ms_per_env_step=1e3 * best[3] / best[2])


def dagger_update_bench():
    """Secondary measurement (`bench.py --dagger-update`): one DAGGER gradient_step at the reference's training
    shape (cfg/dagger.cfg: B=20, N=100, K=3) on the HIP path -- fused forward + MSE gradient + fused backward + flat
    Adam, replayed from one HIP graph -- next to the same op sequence of the CPU port (torch autograd + Adam; this is
    the cpu_baseline leg of the update measurement: the only place this function touches oracle/)."""
    import configparser
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    B, N, K = 20, 100, 3
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(n_states='6', n_actions='2', k=str(K), hidden_size='32', gamma='0.99', tau='0.5',
                         n_agents=str(N), actor_lr='5e-5')
    cp['t'] = {}
    torch.manual_seed(11)
    dev = torch.device('cuda:0')
    learner = DAGGER(dev, cp['t'])
    gen = torch.Generator(device=dev).manual_seed(0)
    xd = torch.randn((B, K, F_FEAT, N), device=dev, generator=gen)
    mask = torch.rand((B, K, N, N), device=dev, generator=gen) < (8.0 / N)
    gd = mask.float() / mask.float().sum(-1, keepdim=True).clamp(min=1)
    gd[:, 0] = torch.eye(N, device=dev)
    yd = torch.randn((B, 1, N_ACT, N), device=dev, generator=gen)
    for _ in range(20):
        learner.gradient_step_tensors(xd, gd, yd)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        learner.gradient_step_tensors(xd, gd, yd)                 # drop-in semantics: the host reads every loss
    torch.cuda.synchronize()
    gpu_ms = 1e3 * (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        learner.gradient_step_tensors(xd, gd, yd, sync=False)     # vectorised DAGGER: losses stay on the device
    torch.cuda.synchronize()
    gpu_ms_pipe = 1e3 * (time.perf_counter() - t0) / n
    # vectorised DAGGER's round of updates: minibatches gathered from a device replay inside the kernel, index table
    # uploaded once, one graph replay per update (sampling on the host included: random.sample per update)
    from multiagent_gnn_policies_amd.learner.vec_dagger import DeviceReplay, IndexedUpdates
    cap, U = 4096, 2000
    rb = DeviceReplay(cap, K, F_FEAT, N, N_ACT, dev)
    for i0 in range(0, cap, B):
        rb.insert_batch(xd, gd, yd)
    iu = IndexedUpdates(learner, rb, B, U)
    iu.run_sampled(64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss_round = iu.run_sampled(U).item()
    gpu_ms_idx = 1e3 * (time.perf_counter() - t0) / U
    assert np.isfinite(loss_round)
    from oracle import torch_port                          # CPU leg
    res = {}
    xc, gc, yc = xd.cpu(), gd.cpu(), yd.cpu()
    for thr in sorted({1, torch.get_num_threads()}):
        torch.set_num_threads(thr)
        Ws = [torch.nn.Parameter(c.weight.detach().cpu().clone()) for c in learner.actor.conv_layers]
        bs = [torch.nn.Parameter(c.bias.detach().cpu().clone()) for c in learner.actor.conv_layers]
        opt = torch.optim.Adam(Ws + bs, lr=5e-5)

        def step():
            opt.zero_grad()
            out = torch_port.actor_forward(xc, gc, Ws, bs, 0, K)
            loss = torch.nn.functional.mse_loss(out, yc)
            loss.backward()
            opt.step()
            return loss.item()
        for _ in range(5):
            step()
        t0 = time.perf_counter()
        m = 100
        for _ in range(m):
            step()
        res[thr] = 1e3 * (time.perf_counter() - t0) / m
    # DAGGER data collection (BASELINE.json configs[3], gnn_dagger.py:154-178): rollouts with expert labels, beta coin and
    # replay insert -- on the collecting build of the resident kernel (one launch per round) vs the host-stepped two-launch
    # loop of round 1 (>= 5 launches + host RNG + H2D per step)
    from multiagent_gnn_policies_amd.learner.vec_dagger import FrameReplay, collect_round, FrameUpdates
    lanes, Tc = 256, 500
    pcol = FlockParams(n_agents=N, init_mode='grid')
    simc = VecFlock(lanes, pcol, dev, with_expert=True)
    stc = BatchedDelayState(dev, lanes, K, F_FEAT, N)
    memc = FrameReplay(lanes, lanes * Tc, K, N, dev)
    beta_t = torch.full((lanes,), 0.75, device=dev)
    eps = torch.arange(lanes, dtype=torch.int32, device=dev)
    np.random.seed(3)
    collect_round(learner, simc, stc, memc, beta_t, eps, 11, 20)                 # warm-up (also the reset sampling cache)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    collect_round(learner, simc, stc, memc, beta_t, eps, 11, Tc)
    torch.cuda.synchronize()
    t_round = time.perf_counter() - t0
    e0c, e1c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    from multiagent_gnn_policies_amd.learner.rollouts import _actor_params
    wsc, bsc = _actor_params(learner.actor)
    img = ops.rollout_image(wsc, bsc, tuple(learner.actor.layers), K, N)
    exp_io = simc.controller().permute(0, 2, 1).contiguous()
    e0c.record()
    ops.rollout_collect(simc.x, stc._G[stc._cur], stc.delay_state, tuple(learner.actor.layers), simc._c, Tc, memc, exp_io, beta_t,
                        eps, 11, age0=Tc, ring_step0=memc.head, carry=stc.carry_buffer(),
                        flags=ops.RO_ENTER_CARRY | ops.RO_EXIT_CARRY | ops.RO_SKIP_DENSE, image=img)
    e1c.record()
    torch.cuda.synchronize()
    collect_kernel_ms = e0c.elapsed_time(e1c)
    fu = FrameUpdates(learner, memc, B, 2000, True)
    fu.run_sampled(64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fu.run_sampled(2000).item()                                # random.sample per update, overlapped with the GPU's replays
    frame_update_ms = 1e3 * (time.perf_counter() - t0) / 2000

    return {"update": "DAGGER gradient_step B=20 N=100 K=3", "hip_ms": gpu_ms, "hip_updates_per_s": 1e3 / gpu_ms,
            "collect": {"lanes": lanes, "steps": Tc, "kernel_ms": collect_kernel_ms,
                        "kernel_agent_steps_per_s": lanes * N * Tc / (1e-3 * collect_kernel_ms),
                        "round_wall_s_incl_host_reset_sampling": t_round,
                        "replay_bytes_per_transition": memc.bytes_per_transition(),
                        "hip_ms_frame_update_round": frame_update_ms},
            "hip_ms_pipelined": gpu_ms_pipe, "hip_updates_per_s_pipelined": 1e3 / gpu_ms_pipe,
            "hip_ms_indexed_round": gpu_ms_idx, "hip_updates_per_s_indexed_round": 1e3 / gpu_ms_idx,
            "cpu_port_ms_by_threads": res, "host_cores": os.cpu_count()}


def self_launch(n):
    """`python bench.py --gpus N` with no launcher: start N ranks of this very command (one per GPU; RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* as torch.distributed.run would set them), pass rank 0's JSON line through, return the worst exit
    status.  Refuses -- loudly -- to run more RCCL ranks than there are devices."""
    import socket
    import subprocess
    backend = os.environ.get('MGP_DIST_BACKEND') or 'nccl'
    have = torch.cuda.device_count()
    if backend == 'nccl' and n > have:
        sys.stderr.write("bench.py: --gpus %d but only %d device(s) visible: RCCL needs one GPU per rank "
                         "(MGP_DIST_BACKEND=gloo lets ranks share a device, for tests)\n" % (n, have))
        return 2
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for rk in range(n):
        env = dict(os.environ, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if rk == 0 else subprocess.DEVNULL))
    worst = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0:
                    worst = worst or rc
                    for q in pending:                            # a dead rank leaves the others in a collective: stop them
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return worst


def check_one_device_per_rank(world):
    if world > 1 and torch.distributed.get_backend() == 'nccl' and world > torch.cuda.device_count():
        raise SystemExit("bench.py: %d RCCL ranks but %d device(s) visible: one GPU per rank" % (world, torch.cuda.device_count()))


def dist_record():
    """What the process group really was (the record shows that RCCL saw N ranks)."""
    d = torch.distributed
    if d.is_available() and d.is_initialized():
        return {"backend": d.get_backend(), "world_size": d.get_world_size(), "devices_visible": torch.cuda.device_count()}
    return {"backend": None, "world_size": 1, "devices_visible": torch.cuda.device_count()}


def dagger_round_bench(args, device, rank, world):
    """BASELINE.json configs[3] (reference gnn_dagger.py:126-243, one device, one env): one DAGGER round on every rank --
      collection  --episodes lanes x --steps env steps inside ONE mgp_rollout_collect launch per rank (policy forward, expert
                  label, beta coin, simulator step, state transition, frame filed into the replay ring); ranks never talk
      updates     --updates minibatch updates of --batch-size samples PER RANK from the rank's own replay, captured 32 to a
                  HIP graph; the ranks' gradients (1,730 floats + the loss) are exchanged inside every update -- the one-shot
                  IPC exchange (csrc/p2p_device.h) or, without it, the RCCL all-reduce captured in the graph
    Timed with the contract's barrier + synchronize bracketing, MAX over ranks; the weights must be bit-identical on every
    rank at the end (the run fails otherwise)."""
    import configparser
    from multiagent_gnn_policies_amd.learner.gnn_dagger import DAGGER
    from multiagent_gnn_policies_amd.learner.vec_dagger import (FrameReplay, FrameUpdates, collect_round, collect_supported,
                                                                _dp_mode)
    dist = torch.distributed
    lanes, N, K, T, U, Bt = args.episodes, args.agents, args.taps, args.steps, args.updates, args.batch_size
    cp = configparser.ConfigParser()
    cp['DEFAULT'] = dict(n_states=str(F_FEAT), n_actions=str(N_ACT), k=str(K), hidden_size=str(args.hidden),
                         n_layers=str(args.layers), gamma='0.99', tau='0.5', n_agents=str(N), actor_lr='5e-5')
    cp['t'] = {}
    torch.manual_seed(11)
    learner = DAGGER(device, cp['t'])
    if not (collect_supported(learner, K, N) and FrameUpdates.supported(learner, Bt, N)):
        raise SystemExit("bench.py --dagger: shape outside mgp_rollout_collect / the graph-captured update path")
    p = FlockParams(n_agents=N, init_mode=args.init)
    sim = VecFlock(lanes, p, device, with_expert=True)
    state = BatchedDelayState(device, lanes, K, F_FEAT, N)
    memory = FrameReplay(lanes, lanes * max(T, args.warmup, 1), K, N, device)
    beta = torch.full((lanes,), 0.75, device=device)
    eps = torch.arange(rank * lanes, (rank + 1) * lanes, dtype=torch.int32, device=device)
    np.random.seed(1000 + rank)
    import random
    random.seed(1000 + rank)

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    # ---- collection: reset sampling (host, once per round) is outside the timed region, the launch inside
    from multiagent_gnn_policies_amd.learner.rollouts import _actor_params

    def collect_factored(steps):
        """N > 256: the same round on the factored state in HBM (K launches per env step enqueued by one library call;
        frame, label and coin inside the policy launch: mgp_sparse_policy_collect)."""
        from multiagent_gnn_policies_amd.learner.sparse_rollout import SparseFlockState, sparse_collect
        sim.reset(np.random)
        state.reset()
        state.push(sim.network, sim.features)
        sp = SparseFlockState(sim, K)
        sp.observe_reset(sim)
        gc.disable()                                             # (see timed(): no interpreter GC pass inside the timed region)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sparse_collect(learner.actor, sim, sp, memory, beta, eps, 11, 0, steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        gc.enable()
        barrier()
        return max_over_ranks(el)

    def collect(steps):
        if N > 256:
            return collect_factored(steps)
        sim.reset(np.random)
        state.reset()
        state.push(sim.network, sim.features)
        expert_io = sim.controller().permute(0, 2, 1).contiguous()
        ws, bs = _actor_params(learner.actor)
        image = ops.rollout_image(ws, bs, tuple(learner.actor.layers), K, N)
        carry = state.carry_buffer()
        gc.disable()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ok = ops.rollout_collect(sim.x, state._G[state._cur], state.delay_state, tuple(learner.actor.layers), sim._c, steps,
                                 memory, expert_io, beta, eps, 11, age0=0, ring_step0=memory.head, carry=carry,
                                 flags=ops.RO_ENTER_CARRY | ops.RO_EXIT_CARRY | ops.RO_SKIP_DENSE, image=image)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        gc.enable()
        barrier()
        assert ok
        memory.advance(steps)
        state._pushes += steps
        state._dense_stale = True
        return max_over_ranks(el)
    gc.freeze()
    collect(max(args.warmup, K))
    t_collect = collect(T)
    # ---- updates
    fu = FrameUpdates(learner, memory, Bt, max(U, 64), p.mean_pooling)
    learner.begin_updates()
    gc.freeze()
    fu.run_sampled(64)                                           # warm-up: captures both graphs
    learner.end_updates()
    gc.disable()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss_sum = fu.run_sampled(U)
    torch.cuda.synchronize()
    t_upd = time.perf_counter() - t0
    gc.enable()
    barrier()
    t_upd = max_over_ranks(t_upd)
    learner.end_updates()
    loss_mean = float(loss_sum.item()) / U
    # ---- every rank must hold the same weights, bit for bit
    identical = True
    if world > 1:
        cdev = device if dist.get_backend() == 'nccl' else torch.device('cpu')
        mine = learner.actor_optim.flat.detach().to(cdev)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        identical = all(torch.equal(parts[0], q) for q in parts[1:])
    if rank == 0:
        out = {
            "metric": "agent-steps/sec of DAGGER data collection, FlockingRelative-v0 N=%d K=%d" % (N, K),
            "value": world * lanes * N * T / t_collect, "unit": "agent-steps/s", "n_gpus": world, "steps": T,
            "warmup": args.warmup, "ms_per_step": 1e3 * t_collect / T, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DAGGER round (BASELINE.json configs[3]): %d lanes x %d steps of data collection per rank "
                                   "(%s: policy forward, expert label, beta coin, sim step, frame insert), "
                                   "then %d updates of %d samples per rank with the gradient exchanged between %d rank(s)"
                                   % (lanes, T, "mgp_rollout_collect" if N <= 256 else "factored state, mgp_sparse_policy_collect",
                                      U, Bt, world),
                       "episodes_per_gpu": lanes, "agents": N, "taps": K, "hidden": [args.hidden] * args.layers,
                       "init": args.init, "beta": 0.75,
                       "parallelism": "episodes sharded x%d; one exchange of %d floats per update"
                                      % (world, learner.actor_optim.flat.numel() + 1)},
            "updates": {"count": U, "batch_size_per_rank": Bt, "ms_per_update": 1e3 * t_upd / U,
                        "updates_per_s": U / t_upd, "samples_per_s": U * Bt * world / t_upd, "mean_loss": loss_mean,
                        "exchange": (fu.dp or "none (single process)"),
                        "exchange_mem_kind": getattr(learner.p2p, 'mem_kind', None),
                        "exchange_bringup": parallel.P2PExchange.last_bringup,
                        "updates_per_graph": 32,
                        # aggregated: mgp_replay_aggregate + mgp_train_step_agg (the K-hop products along the frames' bit rows,
                        # operator slices never formed); dense: mgp_replay_gather_many / _rows + mgp_train_step_indexed
                        "slots": "aggregated" if fu.aggregated else "dense"},
            "round_s": t_collect + t_upd,
            "weights_bit_identical_across_ranks": identical,
            "dist": dist_record(),
        }
        emit_json(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not identical:
        sys.stderr.write("bench.py --dagger: the ranks' weights differ\n")
        sys.exit(4)


_JSON_FD = [None]


def emit_json(obj):
    """The result line, on the process's ORIGINAL stdout (see main: fd 1 is pointed at stderr while the bench runs)."""
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD[0] is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_JSON_FD[0], line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=DEFAULT_STEPS)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--episodes', type=int, default=256, help='parallel episodes per GPU')
    ap.add_argument('--agents', type=int, default=100)
    ap.add_argument('--taps', type=int, default=3)
    ap.add_argument('--hidden', type=int, default=32)
    ap.add_argument('--layers', type=int, default=2)
    ap.add_argument('--graph-steps', type=int, default=10, help='env steps captured per HIP graph (0 = eager)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-parity', action='store_true', help='skip the in-run parity gate against the CPU port')
    ap.add_argument('--no-resident', action='store_true', help='time only the two-launch dense path')
    ap.add_argument('--init', default='auto', choices=['auto', 'disc', 'grid'],
                    help="reset distribution of the timed episodes: 'auto' = the environment's own (FlockParams.init_mode: "
                         "uniform disc up to N = 100, jittered lattice beyond); 'grid' = the lattice at any N")
    ap.add_argument('--env', default='FlockingRelative-v0', choices=sorted(ENV_TAGS),
                    help='environment id (reference cfg key `env`): the variants of FLOCK-SPEC v1')
    ap.add_argument('--episode-steps', type=int, default=FlockParams().max_episode_steps,
                    help='time limit of an episode (FLOCK-SPEC item 6: 500): all lanes are reset on the device inside the '
                         'timed region every this many env steps; 0 = never (rounds 1-3)')
    ap.add_argument('--comm-radius', type=float, default=1.0,
                    help='communication radius R (FlockParams.comm_radius; the mean degree of a reset state goes with R^2)')
    ap.add_argument('--v-max', type=float, default=None,
                    help='reset velocity range (FlockParams.v_max and v_bias, as the cfg key v_max sets both: cfg/vel.cfg sweeps 0.5 .. 5.5)')
    ap.add_argument('--dt', type=float, default=None, help='integration step (FlockParams.dt: cfg/dt.cfg sweeps 0.0075 .. 0.1)')
    ap.add_argument('--dagger', action='store_true',
                    help='BASELINE configs[3]: one DAGGER round per rank -- data collection (--episodes lanes x --steps env '
                         'steps) then --updates minibatch updates of --batch-size per rank, gradients exchanged between the '
                         'ranks; prints its own JSON line')
    ap.add_argument('--updates', type=int, default=2048, help='--dagger: timed updates')
    ap.add_argument('--batch-size', type=int, default=20, help='--dagger: per-rank minibatch (cfg/dagger.cfg:6)')
    ap.add_argument('--dagger-update', action='store_true',
                    help='secondary measurement: DAGGER updates/s at B=20 (prints its own JSON line and exits)')
    args = ap.parse_args()
    if args.dagger_update:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU path)")
        print(json.dumps(dagger_update_bench()))
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))                         # `python bench.py --gpus N` starts its own N ranks
    # ONE line on stdout: libraries that print there from C (gloo's "[Gloo] Rank 0 is connected ..." when several ranks share
    # a GPU in tests) are sent to stderr for the rest of the process; the JSON line goes to the saved descriptor (emit_json)
    sys.stdout.flush()
    _JSON_FD[0] = os.dup(1)
    os.dup2(2, 1)
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: one rank per GPU, launch with --nproc-per-node %d (or without a "
                         "launcher: bench.py starts the ranks itself)" % (args.gpus, world, args.gpus))
    check_one_device_per_rank(world)
    dev_index = parallel.local_device_index(local)
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    B, N, K = args.episodes, args.agents, args.taps
    hidden = [args.hidden] * args.layers
    if args.dagger:
        dagger_round_bench(args, device, rank, world)
        return

    phys = {}
    if args.v_max is not None:
        phys.update(v_max=args.v_max, v_bias=args.v_max)
    if args.dt is not None:
        phys.update(dt=args.dt)
    ro = Rollout(device, B, N, K, hidden, seed=1000 + rank, init_mode=args.init, comm_radius=args.comm_radius, env=args.env, **phys)
    deg_start = float((ro.sim.network != 0).sum(dim=-1).double().mean().item())
    init_name = ('jittered lattice' if use_grid(ro.params) else 'uniform disc') + " (FlockParams.init_mode='%s')" % args.init

    executed = [0]                                               # env steps actually run since the reset (a graph capture runs none)
    _t_trace = [time.perf_counter()]

    def trace(msg):                                              # MGP_BENCH_TRACE=1: where the wall time of a bench run goes (stderr)
        if os.environ.get('MGP_BENCH_TRACE'):
            now = time.perf_counter()
            sys.stderr.write('[bench %7.2f s] %s\n' % (now - _t_trace[0], msg))
            _t_trace[0] = now
    # episode ends inside the timed regions: pre-sampled reset states (host RNG: outside every region)
    n_resets = (2 + 2 * (args.warmup + args.steps)) // args.episode_steps if args.episode_steps > 0 else 0
    ep = Episodes(ro, args.episode_steps, ro.presample_resets(min(n_resets, 2), 2000 + rank) if n_resets else [])
    trace('reset states pre-sampled (%d episode ends on the longest timeline)' % n_resets)
    # ---- capture `gs` consecutive env steps into one HIP graph (even count: ping-pong buffers realign)
    gs = args.graph_steps
    if gs > 0:
        gs = gs + (gs % 2)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                ro.step()
        executed[0] += 2
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ep.since = 2
        # the graph bakes in which ping-pong buffers hold the state at its start: replayed only when the roles match
        # (an odd number of eager steps, or an episode end, re-aligns them: Rollout.device_reset(align=))
        ep.align = (ro.sim.x.data_ptr(), ro.state._cur)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(gs):
                ro.step()

        def run_raw(n_steps):
            n = n_steps
            while n > 0:
                if n >= gs and (ro.sim.x.data_ptr(), ro.state._cur) == ep.align:
                    graph.replay()
                    n -= gs
                else:
                    ro.step()
                    n -= 1
            executed[0] += n_steps
    else:
        def run_raw(n_steps):
            for _ in range(n_steps):
                ro.step()
            executed[0] += n_steps

    def run(n_steps):
        ep.advance(run_raw, n_steps)

    def run_resident_eps(n_steps):
        ep.advance(ro.run_resident, n_steps)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def timed(fn):
        """warm-up, then EXACTLY args.steps steps bracketed by barrier + synchronize on both sides; the clock is read
        between the closing synchronize and the closing barrier (the collective's own latency is not part of a step),
        and the MAX over ranks is taken below."""
        # no cyclic-GC pass of the interpreter inside the timed region: a generation-2 collection over torch's ~10^6 objects
        # takes 35-55 ms, and WHERE it lands is a deterministic function of the allocation count -- it sat inside the 3.4 ms
        # timed call of one build of this file and outside it in the previous one (12x on the reported figure).  Collected
        # gc.freeze() moves everything allocated so far out of the collector's reach (no traversal: a gc.collect() here
        # instead cost the 20-step launch +13 to +55 us of host time after 40 ms of GPU idling), and the collector stays off
        # until the region is over.
        gc.freeze()
        gc.disable()
        try:
            fn(args.warmup)
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(args.steps)
            torch.cuda.synchronize()
            el_ = time.perf_counter() - t0
        finally:
            gc.enable()
        barrier()
        if world > 1:
            t = torch.tensor([el_], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el_ = float(t.item())
        return el_

    trace('setup + graph capture')
    el_two = timed(run)
    trace('two-launch pass')
    resident = ro.resident_supported() and not args.no_resident
    el_res, res_launch_ms = None, None
    if resident:
        # HIP events stamped by the kernel launch itself (mgp_set_launch_events -> hipExtLaunchKernel: the kernel's own begin /
        # end on its stream): an event RECORDED in front of the launch makes the host wait tens of microseconds on an idle
        # stream before it can enqueue the kernel -- a fifth of a 20-step region.  One launch per timed region
        # (--steps <= 2000); longer regions sum their launches.
        from multiagent_gnn_policies_amd import _lib as mgp_lib
        n_l = (args.steps + 1999) // 2000 + (args.steps // args.episode_steps + 1 if args.episode_steps > 0 else 0)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_l)]
        for a_, b_ in evs:
            a_.record(); b_.record()                             # creates the underlying hipEvent_t handles (outside the timed region)
        torch.cuda.synchronize()
        ro.prepare_resident(ep.chunks([executed[0], args.warmup, args.steps]) + [args.warmup, args.steps, max(executed[0], 1)])
        timing_on = [False]
        pre_roll = executed[0]                                   # both resident passes time the SAME steps of the SAME episodes:

        def rewind():                                            # back to the reset, forward to where the two-launch pass ended
            ep.rewind(1000 + rank)                               # (the same episode ends at the same steps, the same reset states)
            ep.align = None
            if pre_roll > 0:
                run_resident_eps(pre_roll)
            torch.cuda.synchronize()

        ev_used = [0]

        def run_res_stamped(t):                                  # every launch stamped with its own begin / end
            while t > 0:
                c = min(2000, t)
                a_, b_ = evs[ev_used[0]]
                ev_used[0] += 1
                mgp_lib.lib().mgp_set_launch_events(a_.cuda_event, b_.cuda_event)
                ro.run_resident(c)
                t -= c

        def run_res_timed(n_steps):
            if n_steps != args.steps or not timing_on[0]:
                run_resident_eps(n_steps)
                return
            ev_used[0] = 0
            ep.advance(run_res_stamped, n_steps)
        # pass 1 (`value`): the launch as a caller issues it.  pass 2: the identical region with HIP events stamped by the launch
        # (roofline.avg_launch_ms) -- the stamped form costs the launch ~11 us of wall time (5 on the host before the doorbell, 6
        # until the completion is seen: tools/gpu/launch_probe.py), a fifteenth of a 20-step region, so it is not what `value` times
        rewind()
        trace('rewind')
        el_res = timed(run_resident_eps)
        trace('resident pass 1')
        resets_in_region = None
        rewind()
        trace('rewind')
        timing_on[0] = True
        c0 = None

        def run_res_counted(n_steps):
            nonlocal c0
            if n_steps == args.steps:
                c0 = ep.count
            run_res_timed(n_steps)
        el_res_ev = timed(run_res_counted)
        resets_in_region = ep.count - c0
        trace('resident pass 2')
        executed[0] = pre_roll + args.warmup + args.steps
        res_launches = ev_used[0]
        res_launch_ms = sum(a_.elapsed_time(b_) for a_, b_ in evs[:res_launches])   # the timed launches' own durations (this rank)
        timing_on[0] = False
        # pass 3: the same region with the dense operator slices of the contract rebuilt INSIDE every launch (lazy_dense=False:
        # no MGP_RO_SKIP_DENSE) -- what a caller pays who reads delay_gso after every launch; `value` defers them (step_path)
        rewind()
        ro.lazy_dense = False
        try:
            el_res_dense = timed(run_resident_eps)
        finally:
            ro.lazy_dense = True
        trace('resident pass 3 (dense exit)')
        # passes 4..: pass 1 again, for a median next to the single sample the contract asks for (`value` stays pass 1)
        res_repeats = [el_res]
        n_rep = 8 if el_res < 0.02 else (2 if el_res < 0.5 else 0)
        for _ in range(n_rep):
            rewind()
            res_repeats.append(timed(run_resident_eps))
        trace('resident repeats')
    # the same launch on the jittered lattice (rounds 1-2 timed this state: sparser, mean degree 6.8 at reset against 8.5)
    el_grid, deg_grid = None, None
    if resident and not use_grid(ro.params):
        ro_g = Rollout(device, B, N, K, hidden, seed=1000 + rank, init_mode='grid', comm_radius=args.comm_radius, env=args.env, **phys)
        deg_grid = float((ro_g.sim.network != 0).sum(dim=-1).double().mean().item())
        ep_g = Episodes(ro_g, args.episode_steps, ro_g.presample_resets(1, 3000 + rank) if n_resets else [])
        ro_g.prepare_resident(ep_g.chunks([args.warmup, args.steps]) + [args.warmup, args.steps])
        el_grid = timed(lambda n_: ep_g.advance(ro_g.run_resident, n_))
        trace('lattice pass')
        del ro_g
    el_fact = None
    if ro.factored_supported() and not args.no_resident:
        ep.rewind(1000 + rank)
        ep.align = None
        el_fact = timed(run_resident_eps)                        # policy_rollout: factored path, state carried between calls
    # every path is a complete implementation of the same step; which one is `value` depends on the SHAPE only
    timed_resident = resident
    factored = el_fact is not None
    el = el_fact if factored else (el_res if resident else el_two)
    finite = bool(torch.isfinite(ro.sim.x).all().item())
    deg = mean_degree(ro.state)
    # mean degree AVERAGED over the timed region of the path that is `value` (un-timed extra pass over the same steps of the
    # same episodes, sampled every 50 steps): what density the figure was measured at
    deg_region = None
    if (resident or factored) and args.steps >= 100 and K >= 2:
        ep.rewind(1000 + rank)
        ep.align = None
        run_resident_eps((pre_roll if resident else 0) + args.warmup)
        samples, left = [], args.steps
        while left > 0:
            t = min(50, left)
            run_resident_eps(t)
            left -= t
            samples.append(mean_degree(ro.state))
        deg_region = float(np.mean(samples))
        trace('degree pass')

    out = None
    if rank == 0:
        total_eps = B * world
        value = total_eps * N * args.steps / el
        out = {
            "metric": "agent-steps/sec, %s N=%d K=%d" % (args.env, N, K),
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s N=%d K=%d, %d parallel episodes per MI355X "
                                   "(BASELINE.json configs[1]); per step: Actor forward (hidden %s) -> action -> "
                                   "sim step -> delayed-GSO / delay-line update" % (args.env, N, K, B, hidden),
                       "step_path": ("factored: state as bit rows / feature ring in HBM, K launches per step "
                                     "(mgp_sparse_rollout: simulator + gather + policy launches); the dense delay_gso of the contract is "
                                     "rebuilt on first read (mgp_sparse_to_dense, ~180 us per 64 x 1000 state), i.e. AFTER and outside "
                                     "the timed region -- as the resident path defers its dense slices (RO_SKIP_DENSE)")
                                    if factored else
                                    ("resident: all %d timed steps in one mgp_rollout_steps launch per GPU (episode state in LDS; entered "
                                     "from and left as the factored hand-over: membership bits + row weights of the last K - 1 networks).  "
                                     "The dense delay_gso slices of the contract are DEFERRED (MGP_RO_SKIP_DENSE): rebuilt on first read "
                                     "(mgp_rollout_carry_to_dense, ~37 us per 256 episodes), i.e. outside the timed region; "
                                     "paths.resident_dense_exit times the same steps with the slices rebuilt inside every launch" % args.steps)
                                    if resident else
                                    "two_launch: mgp_actor_fwd + mgp_flock_step_advance per step (HIP graph)",
                       "step_path_rule": "by shape: resident if mgp_rollout_supported (N <= 256, widths <= 64; one hidden layer up to 128 wide at N <= 128), factored if "
                                         "N > 256 and mgp_sparse_policy_supported, else two_launch",
                       "episodes_per_gpu": B, "episodes_total": total_eps, "agents": N, "taps": K,
                       "graph_steps": gs, "weights": ro.weights, "parallelism": "episodes sharded x%d, no "
                       "data-path collective" % world, "state_finite": finite,
                       "mean_degree": deg, "mean_degree_at_reset": deg_start, "mean_degree_over_timed_region": deg_region,
                       "episode_steps": args.episode_steps, "comm_radius": ro.params.comm_radius, "v_max": ro.params.v_max, "dt": ro.params.dt,
                       "episode_ends_in_timed_region": (resets_in_region if resident else None),
                       "init": "%s; every lane is reset on the device every %d env steps (time limit) inside the timed region; "
                               "%d steps since the last reset at the end of it" % (init_name, args.episode_steps, ep.since)
                               if args.episode_steps > 0 else
                               "%s, never reset: %d steps since reset at the end of the timed region" % (init_name, executed[0])},
            "dist": dist_record(),
            "paths": {"two_launch": {"ms_per_step": 1e3 * el_two / args.steps,
                                     "value": total_eps * N * args.steps / el_two, "graph_steps": gs}},
        }
        if el_fact is not None:
            out["paths"]["factored"] = {"ms_per_step": 1e3 * el_fact / args.steps,
                                        "value": total_eps * N * args.steps / el_fact}
        if timed_resident:
            out["paths"]["resident"] = {"ms_per_step": 1e3 * el_res / args.steps,
                                        "value": total_eps * N * args.steps / el_res,
                                        "launch_ms_hip_events": res_launch_ms,
                                        "ms_per_step_event_pass": 1e3 * el_res_ev / args.steps,
                                        "passes": "value: the plain launch; launch_ms_hip_events: a second pass over the same %d steps "
                                                  "of the same episodes with kernel-stamped HIP events (mgp_set_launch_events), whose "
                                                  "wall time is ms_per_step_event_pass" % args.steps}
            out["paths"]["resident_dense_exit"] = {
                "ms_per_step": 1e3 * el_res_dense / args.steps, "value": total_eps * N * args.steps / el_res_dense,
                "note": "lazy_dense=False: every launch of the region writes the dense delay_gso (B,K,N,N) of its final state before it "
                        "returns (no MGP_RO_SKIP_DENSE)"}
            med = float(np.median(res_repeats))
            out["value_median_of"] = {"n": len(res_repeats), "value": total_eps * N * args.steps / med,
                                      "ms_per_step": 1e3 * med / args.steps,
                                      "samples_ms_per_step": [1e3 * e_ / args.steps for e_ in res_repeats],
                                      "note": "median over repeats of the SAME timed region (rewind to the reset, roll forward, warm-up, "
                                              "barrier + synchronize, %d steps, synchronize); `value` is the first sample (this rank's "
                                              "clock; world > 1: max over ranks per sample)" % args.steps}
            if el_grid is not None:
                out["paths"]["resident_grid"] = {"ms_per_step": 1e3 * el_grid / args.steps,
                                                 "value": total_eps * N * args.steps / el_grid,
                                                 "init": "jittered lattice (FlockParams.init_mode='grid')",
                                                 "mean_degree_at_reset": deg_grid}
    if rank == 0 and not args.no_roofline:
        trace('-')
        res, n_sets = kernel_rooflines(device, B, N, K, ro.actor, ro.sim._c)
        trace('kernel_rooflines')
        fused = getattr(ro.actor, 'use_fused', False) and ro.actor.ind_agg == 0

        def hbm_block(key, kname, pmc_names):
            r = res[key]
            tr, tr_note, sq = None, 'not profiled', None
            for pmc_name in pmc_names:                       # the variant the library picked for this shape comes first
                tr, tr_note = pmc_traffic(pmc_name, B, N, K)
                sq = pmc_sq(pmc_name)
                if tr is not None:
                    break
            return {"kernel": kname, "bound": "hbm", "achieved": r['gbs'], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": r['gbs'] / HBM_PEAK_GBS, "traffic": tr, "traffic_source": tr_note,
                    "algorithmic_bytes_per_launch": r['bytes'], "avg_launch_ms": r['ms'], "sq": sq}
        # HBM-roofline figures of the kernels that stream the dense operator G (B,K,N,N) from HBM on every launch --
        # north_star's "fraction of HBM roofline for the S^k X aggregation" -- measured live with HIP events on the
        # launch stream over rotating input sets larger than the Infinity Cache; algorithmic bytes per SURVEY.md 8(d)
        dense = {
            "actor_fwd": hbm_block('actor_fwd', "mgp_actor_fwd: for N <= 128 actor_fwd_pol_kernel (the reference's policy shape [32, 32] compiled in: "
                                   "aggregation on 4x4x1 fp32 MFMA, hidden layers on split-bf16 MFMA) or actor_fwd_mfma_kernel (any "
                                   "widths <= 128, fp32 MFMA), actor_fwd_kernel otherwise: 4KN^2 + 4KFN + 4 nA N bytes per "
                                   "episode", ('actor_fwd_pol_kernel', 'actor_fwd_mfma_kernel', 'actor_fwd_kernel')
                                   if hidden == [32, 32] else ('actor_fwd_mfma_kernel', 'actor_fwd_kernel')),
            "agg_fwd": hbm_block('agg_fwd', "mgp_agg_fwd: agg_fwd_mfma4_kernel for N <= 128 (four waves per (episode, tap): a wave "
                                 "streams half the rows of its column block), agg_fwd_kernel otherwise (aggregation "
                                 "X.G alone): 4KN^2 + 8KFN bytes per episode", ('agg_fwd_mfma4_kernel', 'agg_fwd_mfma_kernel', 'agg_fwd_kernel')),
            "sim_state_step": hbm_block('sim_state_step', "mgp_flock_step_advance: flock_advance_kernel for N <= 128 (one workgroup per "
                                        "episode: sim step + delayed-GSO / delay-line transition, source slice staged in LDS by "
                                        "LDS-DMA), the row-tiled flock_step_kernel<advance> otherwise",
                                        ('flock_advance_kernel',) if (N <= 128 and N % 4 == 0) else ('flock_step_kernel',)),
            "rotating_input_sets": n_sets,
        }
        if resident:
            # dominant (only) kernel of the timed region.  Its one roofline-shaped resource is the matrix pipe (filter GEMM +
            # hidden layers on fp32 MFMA); nothing streams from HBM.  achieved = ALGORITHMIC flops of the MFMA-run layers
            # (2 N sum_l cin_l cout_l per episode-step, hidden layers only) / launch duration.
            n_launch = res_launches                              # launches of the timed region (split at episode ends / 2000 steps)
            spl = args.steps / float(n_launch)
            dims = [F_FEAT * K] + hidden
            flops_unit = 2.0 * N * sum(a * b_ for a, b_ in zip(dims[:-1], dims[1:]))
            flops = flops_unit * B * spl
            ms = res_launch_ms / n_launch
            alg = (4 * K * N * N + 8 * K * F_FEAT * N) * B * spl
            # which instructions run those flops: layers whose inputs have <= 32 channels run on split-bf16 MFMA (three bf16
            # pieces per fp32 operand, six 16x16x32 products per fp32 product: rollout_common.h ro_layer_bf16), the 64-wide
            # build on fp32 MFMA 16x16x4.  `frac` above stays against the fp32 matrix peak -- the rate a plain fp32
            # implementation of the same flops is bounded by; the bf16 figures are here for the pipe's own occupancy
            split = N <= 128 or max(hidden) <= 32        # every build of the N <= 128 kernel; beyond, widths <= 32 only
            matrix_note = ({"form": "split-bf16: v_mfma_f32_16x16x32_bf16, 6 bf16 products per fp32 product, K padded to 32",
                            "bf16_flops_per_episode_step": 6.0 * 2.0 * N * sum(32 * 16 * ((b_ + 15) // 16) for b_ in dims[1:]),
                            "bf16_peak_TFLOPs": MFMA_BF16_PEAK_TFLOPS}
                           if split else {"form": "fp32: v_mfma_f32_16x16x4_f32"})
            if split:
                matrix_note["frac_of_bf16_peak"] = (matrix_note["bf16_flops_per_episode_step"] * B * spl / ms / 1e9 /
                                                    MFMA_BF16_PEAK_TFLOPS)
            tr, tr_note = pmc_traffic('rollout_kernel', B, N, K, steps_per_launch=spl)
            # The MEASURED limiter of this kernel is vector-instruction issue (its fullest SIMD's VALU pipe), not the matrix pipe
            # (sq.mfma_busy ~ 0.07) and not HBM (traffic = state in / out): `bound` names it.  achieved = VALU wave-instructions
            # per second = (SQ_INSTS_VALU per episode-step of the committed counter pass of THIS build, profiles/<round>_pmc_sq.json)
            # x the episode-steps per second of the live, event-stamped launches; peak = CUs x 4 SIMDs x clock / 4 cycles (a wave64
            # vector instruction occupies its SIMD for at least four cycles; fp64 pairs, transcendentals and MFMAs longer, so the
            # fraction is a LOWER bound on how busy the pipes are).  sq.valu_issue is the same quantity over the SIMDs' busy
            # cycles of the profiled launches.  The matrix-pipe figure stays beside it under `mfma`.
            sq = pmc_sq('rollout_kernel')
            sq_meta = (_profile_json('pmc_sq.json') or {}).get('_meta', {})
            valu_unit = None
            profiled_shape = (N == 100 and K == 3 and hidden == [32, 32])          # tools/pmc_probe.py profiles the headline shape
            if not profiled_shape:
                sq = None
            if sq is not None and sq.get('valu_insts') and sq_meta.get('rollout_episode_steps_per_launch'):
                valu_unit = sq['valu_insts'] / float(sq_meta['rollout_episode_steps_per_launch'])
            valu_peak = N_CUS * 4 * CLOCK_GHZ / 4.0                                  # G wave-instructions / s
            valu_ach = (valu_unit * B * spl / ms / 1e6) if valu_unit else None
            out["roofline"] = {
                "kernel": "rollout_kernel (episode-resident: power-iterated aggregation along neighbour lists + split-bf16 MFMA "
                          "filter/MLP + sim step + Verlet-listed neighbour search, %.0f steps per launch on average)" % spl,
                "bound": "valu", "achieved": valu_ach, "peak": valu_peak, "unit": "G wave-instructions/s",
                "frac": (valu_ach / valu_peak) if valu_ach else (sq or {}).get('valu_issue'),
                "valu_insts_per_episode_step": valu_unit,
                "bound_note": "vector-ALU issue: %d CUs x 4 SIMDs x %.1f GHz / 4 cycles per wave64 instruction; instruction count "
                              "from the committed SQ counter pass of this build, rate from this run's event-stamped launches" % (N_CUS, CLOCK_GHZ),
                "mfma": {"achieved": flops / ms / 1e9, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / ms / 1e9 / MFMA_F32_PEAK_TFLOPS,
                         "note": "algorithmic flops of the MFMA-run layers against the fp32 matrix peak (rounds 1-4 reported this as "
                                 "`frac`); the pipe itself is sq.mfma_busy busy"},
                "traffic": tr, "traffic_source": tr_note,
                "algorithmic_flops_per_launch": flops, "algorithmic_flops_per_episode_step": flops_unit,
                "avg_launch_ms": ms, "steps_per_launch": spl, "resident": True,
                "matrix_instructions": matrix_note,
                "launch_timing": "HIP events stamped by the launch itself (mgp_set_launch_events) in a second pass over the same "
                                 "steps of the same episodes (paths.resident.ms_per_step_event_pass); the pass `value` is taken "
                                 "from carries no events",
                "limiter": "vector-ALU instruction issue: sq.valu_issue (>= 0.6: SQ_INSTS_VALU x 4 cycles over the four SIMDs' busy "
                           "cycles, a lower bound -- fp64, transcendental and MFMA instructions hold the pipe longer) is the "
                           "resource nearest its ceiling; the rest of a step is LDS latency and workgroup barriers with one "
                           "workgroup per CU.  Neither HBM (traffic = state in/out + 8 B of reward per step) nor the matrix "
                           "pipe (sq.mfma_busy) is saturated -- the fractions in `sq` are the evidence",
                "sq": sq,
                "equivalent_hbm": {"GBps": alg / ms / 1e6, "frac_of_peak": alg / ms / 1e6 / HBM_PEAK_GBS,
                                   "algorithmic_bytes_per_launch": alg,
                                   "note": "NOT a roofline fraction: the bytes the dense-contract aggregation (4KN^2 + 8KFN "
                                           "per episode-step, SURVEY.md 8d) WOULD stream for these steps / launch time; "
                                           "inside the launch the operator exists only as neighbour lists in LDS"},
                "dense_kernels": dense}
        elif factored:
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
            ms_step = 1e3 * el_fact / args.steps
            trf, trf_note = pmc_traffic_factored(B, N, K)
            out["roofline"] = {
                "kernel": "factored step: sp_sim_kernel + %d x spl_gather_kernel + spl_policy_kernel (%d launches per env step)"
                          % (max(K - 2, 0), max(K, 2)),
                "bound": "hbm", "achieved": req / ms_step / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": req / ms_step / 1e6 / HBM_PEAK_GBS, "traffic": trf, "traffic_source": trf_note,
                "required_bytes_per_step": req,
                "required_bytes_per_episode_step": {"simulator": sim_b, "gather_stages": gather_b, "policy_tail": policy_b},
                "avg_step_ms": ms_step,
                "note": "bytes the factored state REQUIRES per env step (bit rows, lists, row weights, feature rings, agent "
                        "states: each array once) / wall time per step of the timed region / 8 TB/s.  The path is latency-, not "
                        "bandwidth-bound: every launch starts with the wait for the rows the previous launch wrote on other "
                        "XCDs (profiles/%s_factored_step_stamps.txt); the dense-contract bytes of this shape would be %.0f MB "
                        "per step" % (PROFILE_ROUND, (4 * K * N * N + 8 * K * F_FEAT * N) * B / 1e6),
                "dense_kernels": dense}
        else:
            out["roofline"] = dict(dense["actor_fwd" if fused else "agg_fwd"], dense_kernels=dense)
        out["kernels"] = {k: {"avg_launch_ms": v['ms'], "algorithmic_bytes": v['bytes'], "GBps": v['gbs']}
                          for k, v in res.items()}
    parity = None
    if rank == 0 and not args.no_parity:
        trace('roofline blocks')
        parity = parity_gate(ro)
        trace('parity gate')
        out["parity"] = parity
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from multiagent_gnn_policies_amd.envs.flocking import _REGISTRY
        out["cpu_baseline"] = cpu_baseline(N, K, hidden, init_mode=args.init, actor=ro.actor,
                                           variant=dict(getattr(_REGISTRY[args.env], 'variant', {})))
    if rank == 0:
        emit_json(out)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if parity is not None and not parity["ok"]:
        sys.stderr.write("bench.py: PARITY GATE FAILED: max_rel %.3g > %.1g\n" % (parity["max_rel"], PARITY_TOL))
        sys.exit(3)


if __name__ == '__main__':
    main()
