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
              when the shape is covered (N <= 256, widths <= 64; one hidden layer up to 128 wide at N <= 128; two at N = 100, K = 3).
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
says so (N <= 256, widths <= 64; one hidden layer up to 128 wide at N <= 128; two at N = 100, K = 3), factored for N > 256 where mgp_sparse_policy_supported, else two_launch; config.step_path
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
# the legs that are not the contract line itself live in tools/bench_legs/ (same JSON as before the split)
from tools.bench_legs.common import F_FEAT, N_ACT, PARITY_TOL, PROFILE_ROUND  # noqa: E402,F401
from tools.bench_legs.cpu import cpu_baseline  # noqa: E402
from tools.bench_legs.dagger import dagger_round_bench, dagger_update_bench  # noqa: E402,F401
from tools.bench_legs.kernels import kernel_rooflines, time_kernel  # noqa: E402,F401
from tools.bench_legs.launch import _JSON_FD, check_one_device_per_rank, dist_record, emit_json, self_launch  # noqa: E402
from tools.bench_legs.parity import parity_gate  # noqa: E402
from tools.bench_legs.roofline import roofline_blocks  # noqa: E402

DEFAULT_STEPS = 1000      # env steps in the timed region (resident path: one launch; ~10 ms)


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
        """N > 256: the factored state in HBM (learner/sparse_rollout.py); one persistent launch per call where
        factored_persistent(), else K launches per step."""
        from multiagent_gnn_policies_amd.learner.sparse_rollout import sparse_supported
        return self.N > 256 and sparse_supported(self.actor, self.K, self.N)

    def factored_persistent(self):
        """Does mgp_sparse_rollout run this shape as ONE launch of persistent workgroups (csrc/sparse_persist.hip)?"""
        import ctypes
        from multiagent_gnn_policies_amd import _lib
        dims = tuple(self.actor.layers)
        cd = (ctypes.c_int * len(dims))(*dims)
        return bool(_lib.lib().mgp_sparse_rollout_persistent(cd, len(dims) - 1, self.K, self.N, ctypes.byref(self.sim._c)))

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


def mean_degree(state):
    """Mean number of neighbours per agent in the current networks (rows of delay_gso[:, 1]), over all episodes."""
    if state.K < 2:
        return None
    return float((state.delay_gso[:, 1] != 0).sum(dim=-1).double().mean().item())


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
        sys.exit(self_launch(args.gpus, os.path.abspath(__file__)))                        # `python bench.py --gpus N` starts its own N ranks
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
        with ops.graph_capture(graph):
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
        # which build of the resident kernel the library's dispatch picked for this episode count (csrc/rollout.hip: ro_use_t512)
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        forced = os.environ.get('MGP_RO_T512', '')
        t512 = (forced != '0') if forced else (B > cus)
        t512 = t512 and N == 100 and K == 3 and hidden == [32, 32]
        resident_build = ("512 threads, two episodes per CU (rollout_t512.hip: %d episodes > %d CUs)" % (B, cus) if t512 else
                          "1024 threads, one episode per CU")
        value = total_eps * N * args.steps / el
        out = {
            "metric": "agent-steps/sec, %s N=%d K=%d" % (args.env, N, K),
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s N=%d K=%d, %d parallel episodes per MI355X "
                                   "(BASELINE.json configs[1]); per step: Actor forward (hidden %s) -> action -> "
                                   "sim step -> delayed-GSO / delay-line update" % (args.env, N, K, B, hidden),
                       "step_path": (("factored, persistent form: state as bit rows / list rows / feature ring in HBM, every "
                                      "mgp_sparse_rollout call ONE launch of workgroups that stay resident for its steps (an episode's "
                                      "feature rows, row weights and own list rows in LDS; siblings exchange through the state buffers "
                                      "behind arrival counters; bit-identical to the K-launch form); the dense delay_gso of the contract is "
                                      if ro.factored_persistent() else
                                      "factored: state as bit rows / feature ring in HBM, K launches per step "
                                      "(mgp_sparse_rollout: simulator + gather + policy launches); the dense delay_gso of the contract is ") +
                                     "rebuilt on first read (mgp_sparse_to_dense, ~180 us per 64 x 1000 state), i.e. AFTER and outside "
                                     "the timed region -- as the resident path defers its dense slices (RO_SKIP_DENSE)")
                                    if factored else
                                    ("resident: all %d timed steps in one mgp_rollout_steps launch per GPU (episode state in LDS; entered "
                                     "from and left as the factored hand-over: membership bits + row weights of the last K - 1 networks).  "
                                     "The dense delay_gso slices of the contract are DEFERRED (MGP_RO_SKIP_DENSE): rebuilt on first read "
                                     "(mgp_rollout_carry_to_dense, ~37 us per 256 episodes), i.e. outside the timed region; "
                                     "paths.resident_dense_exit times the same steps with the slices rebuilt inside every launch.  "
                                     "Workgroups: %s" % (args.steps, resident_build))
                                    if resident else
                                    "two_launch: mgp_actor_fwd + mgp_flock_step_advance per step (HIP graph)",
                       "step_path_rule": "by shape: resident if mgp_rollout_supported (N <= 256, widths <= 64; one hidden layer up to 128 wide at N <= 128; two at N = 100, K = 3), factored if "
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
        out["roofline"], out["kernels"] = roofline_blocks(
            device, B, N, K, hidden, ro.actor, ro.sim._c, 'resident' if resident else ('factored' if factored else 'dense'), args.steps,
            res_launches=res_launches if resident else None, res_launch_ms=res_launch_ms, el_fact=el_fact,
            factored_persistent=ro.factored_persistent() if factored else None)
        trace('roofline blocks')
    parity = None
    if rank == 0 and not args.no_parity:
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
