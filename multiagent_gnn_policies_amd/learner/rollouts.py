"""Episode-rollout helpers shared by the DAGGER / cloning / baseline loops (gym-style env, one episode at a time)."""
import numpy as np

from .state_with_delay import MultiAgentStateWithDelay


def run_episode(env, act, on_step=None):
    """Roll one episode: `act(obs)` -> action array; returns the summed reward.
    `on_step(obs, action, next_obs, reward, done)` is called after every env.step (used to fill replay memories)."""
    obs = env.reset()
    total, done = 0.0, False
    while not done:
        action = act(obs)
        next_obs, reward, done, _ = env.step(action)
        if on_step is not None:
            on_step(obs, action, next_obs, reward, done)
        total = total + reward                       # (a 0-d device tensor in the environment's fast loop mode: no sync)
        obs = next_obs
    return float(total)


def enable_fast_loop(env):
    """Put one of this package's gym-style environments into its fast loop mode (no host<->device traffic per step);
    returns True if the environment supports it (any other environment object is left alone).  Callers that borrow the
    user's environment use `fast_loop_mode(env)` instead, which puts the previous mode back."""
    from ..envs.flocking import FlockingRelativeEnv
    raw = getattr(env, 'env', None)
    if isinstance(raw, FlockingRelativeEnv):
        raw.fast_loop = True
        return True
    return False


class fast_loop_mode(object):
    """`with fast_loop_mode(env) as fast:` -- the environment runs in its fast loop mode inside the block (fast = True if it
    is one of this package's environments) and is handed back in the mode it came in: code written against gym_flock that
    reuses the env after one of this package's loops keeps getting numpy observations and float rewards."""

    def __init__(self, env):
        from ..envs.flocking import FlockingRelativeEnv
        raw = getattr(env, 'env', None)
        self.raw = raw if isinstance(raw, FlockingRelativeEnv) else None
        self.prev = None

    def __enter__(self):
        if self.raw is None:
            return False
        self.prev = self.raw.fast_loop
        self.raw.fast_loop = True
        return True

    def __exit__(self, *exc):
        if self.raw is not None:
            self.raw.fast_loop = self.prev
        return False


def reward_stats(rewards):
    return {'mean': np.mean(rewards), 'std': np.std(rewards)}


class PolicyRunner(object):
    """Carries the delayed state of ONE environment across steps so a learner can be used as `act(obs)`."""

    def __init__(self, learner, device, args):
        self.learner, self.device, self.args = learner, device, args
        self.state = None
        self._fresh = True

    def reset(self):
        self.state = None

    def observe(self, obs):
        """Fold a new observation into the delay line (reference gnn_dagger.py:150,165)."""
        self.state = MultiAgentStateWithDelay(self.device, self.args, obs, prev_state=self.state)
        return self.state

    def act(self, obs):
        a = self.learner.select_action(self.observe(obs))
        return a if hasattr(obs[0], 'device32') and obs[0]._dev64 is None else a.cpu().numpy()   # fast loop: stays on the device


def policy_episode_reward(env, learner, device, args):
    """One policy-only episode (the reference's test loop, gnn_dagger.py:194-203)."""
    with fast_loop_mode(env):
        runner = PolicyRunner(learner, device, args)
        return run_episode(env, runner.act)


def _actor_params(actor):
    ws, bs = [], []
    for conv in actor.conv_layers:
        ws.append(conv.weight.detach().view(conv.weight.shape[0], -1))
        bs.append(conv.bias.detach())
    return ws, bs


def rollout_image_for(actor, K, N):
    """Prebuilt weight image of the episode-resident kernel for `actor` (None: shape not covered).  Valid until the
    weights change; pass it to policy_rollout(image=...) when the same policy is rolled out in several launches."""
    from .. import ops
    ws, bs = _actor_params(actor)
    return ops.rollout_image(ws, bs, tuple(actor.layers), K, N)


class ResidentPlan(object):
    """A repeated resident launch with its host side bound ONCE: weight image, layer table, flock parameters and the argument
    list of mgp_rollout_steps_ex are prepared here, so that `run(T)` is one ctypes call plus flag bookkeeping (~10 us of host
    time instead of ~50: the chunked evaluation loops -- gnn_dagger.py:190-232, test_model.py, bench.py -- launch every few
    hundred microseconds, and whatever the host does before the launch is time the GPU idles).
    Valid while the actor's weights and the `sim` / `state` objects stay the same; `refresh()` after a weight update."""

    def __init__(self, actor, sim, state):
        import ctypes
        from .. import _lib, ops
        if not (actor.ind_agg == 0 and state.F == 6 and sim.network64 is None and sim.features64 is None
                and ops.rollout_supported(tuple(actor.layers), state.K, sim.N)):
            raise ops.MgpError("the episode-resident kernel does not cover this shape (use policy_rollout)")
        self.actor, self.sim, self.state = actor, sim, state
        self._L, self._ops, self._ct = _lib.lib(), ops, ctypes
        dims = tuple(actor.layers)
        self._cd = (ctypes.c_int * len(dims))(*dims)
        self._nl = len(dims) - 1
        self._params = ctypes.byref(sim._c)
        self._carry = state.carry_buffer()
        self._carry_p = ops._ptr(self._carry)
        self.refresh()

    def refresh(self):
        """Rebuild the weight image (call after the policy's weights changed)."""
        self.image = rollout_image_for(self.actor, self.state.K, self.sim.N)
        self._image_p = self._ops._ptr(self.image)

    def run(self, T, rewards=None, action=None, lazy_dense=True, update_sim_reward=True):
        """T closed-loop policy steps of every lane in one launch; same state bookkeeping as policy_rollout."""
        ops, state, sim = self._ops, self.state, self.sim
        flags = 0
        if self._carry is not None:
            if state._carry_valid:
                flags = ops.RO_ENTER_CARRY
            if flags or T >= state.K - 1:
                flags |= ops.RO_EXIT_CARRY | (ops.RO_SKIP_DENSE if lazy_dense else 0)
        if not (flags & ops.RO_ENTER_CARRY):
            state._ensure_dense()
        B, N, K = sim.B, sim.N, state.K
        if rewards is not None:
            assert rewards.shape == (B, T) and rewards.dtype.is_floating_point and rewards.element_size() == 8
        rc = self._L.mgp_rollout_steps_ex(sim.x.data_ptr(), state._G[state._cur].data_ptr(), state._X[state._cur].data_ptr(),
                                          None, None, self._cd, self._nl, ops._ptr(action), ops._ptr(rewards), self._params,
                                          B, K, N, int(T), self._image_p, self._carry_p, flags, ops._stream())
        if rc != 0:
            from .. import _lib
            _lib.check(rc, 'mgp_rollout_steps_ex')
        state._carry_valid = bool(flags & ops.RO_EXIT_CARRY)
        state._dense_stale = bool(flags & ops.RO_SKIP_DENSE)
        if K > 1:
            sim._network, sim._network_lazy = None, self._lazy_network
        sim.features = state._X[state._cur][:, 0]
        if rewards is not None and update_sim_reward:
            sim.reward.copy_(rewards[:, T - 1])
        state._pushes += T
        return True

    def _lazy_network(self):
        return self.state.delay_gso[:, 1]


def policy_rollout(actor, sim, state, T, rewards=None, action=None, resident=True, image=None, lazy_dense=True):
    """T closed-loop policy steps of every episode lane of `sim` (VecFlock) / `state` (BatchedDelayState): the batched
    form of the reference's evaluation loop (test_model.py:38-44).  `rewards` (B,T) fp64 receives every step's reward.

    When the shape is covered, the whole call is ONE launch of the episode-resident kernel (mgp_rollout_steps: state in
    LDS for all T steps); otherwise -- or with resident=False -- each step is the two-launch path (fused Actor forward +
    fused simulator/state kernel).  Flocks beyond that kernel (N > 256) run on the factored state in HBM when the call
    starts at a reset observation or continues such a rollout (sparse_rollout.py).  Either way sim.x, state.delay_gso,
    state.delay_state hold the state T steps later.  Returns True unless the two-launch path ran."""
    import torch
    from .. import ops
    assert state.has_prev, "push the reset observation into the delay state first"
    if T <= 0:
        return False
    if (resident and actor.ind_agg == 0 and state.F == 6 and sim.network64 is None and sim.features64 is None
            and ops.rollout_supported(tuple(actor.layers), state.K, sim.N)):
        ws, bs = _actor_params(actor)
        # Hand-over between launches in factored form: when the state's operator history is known as bit rows (it was left
        # by the previous resident launch, or the state is a reset observation) the launch enters from it, leaves it for
        # the next one, and the dense slices delay_gso[:, 1:] are materialised only if somebody reads them
        # (state.delay_gso / sim.network) -- chunked launches are then bit-identical to one long launch.
        carry = state.carry_buffer()
        flags = 0
        if carry is not None:
            if state._carry_valid:
                flags |= ops.RO_ENTER_CARRY
            if (flags & ops.RO_ENTER_CARRY) or T >= state.K - 1:
                flags |= ops.RO_EXIT_CARRY | (ops.RO_SKIP_DENSE if lazy_dense else 0)
        if not (flags & ops.RO_ENTER_CARRY):
            state._ensure_dense()                              # the launch reads the dense slices
        G_cur = state._G[state._cur]
        if ops.rollout_steps(sim.x, G_cur, state.delay_state, ws, bs, tuple(actor.layers), sim._c, T,
                             action=action, rewards=rewards, image=image, carry=carry, flags=flags):
            state._carry_valid = bool(flags & ops.RO_EXIT_CARRY)
            state._dense_stale = bool(flags & ops.RO_SKIP_DENSE)
            if state.K > 1:
                sim._network, sim._network_lazy = None, (lambda: state.delay_gso[:, 1])
            sim.features = state.delay_state[:, 0]
            if rewards is not None:
                sim.reward.copy_(rewards[:, T - 1])
            state._pushes += T
            return True
    if (resident and sim.N > 256 and actor.ind_agg == 0 and state.F == 6 and sim.network64 is None
            and sim.features64 is None):
        # beyond the LDS-resident kernel: the factored state in HBM (sparse_rollout.py), K launches per step.  It can be
        # started at a reset observation or carried over from the previous call; the dense state is refreshed at the end.
        from .sparse_rollout import SparseFlockState, sparse_policy_rollout, sparse_supported
        if sparse_supported(actor, state.K, sim.N):
            sp = getattr(state, '_sparse', None)
            if not (sp is not None and sp.owner is sim.x and sp.K == state.K and sp.at_push == state._pushes):
                sp = None
                if state._pushes == 1:
                    sp = SparseFlockState(sim, state.K)
                    sp.observe_reset(sim)
            if sp is not None:
                if state._dense_from is not None:              # a pending lazy rebuild refers to the rings about to move; the
                    state._dense_from, state._dense_stale = None, False       # dense slices are rebuilt from the NEW state on demand
                sparse_policy_rollout(actor, sim, sp, T, rewards=rewards, action=action)
                sp.to_dense(sim, state, lazy=lazy_dense)
                state._pushes += T
                sp.at_push = state._pushes
                state._sparse = sp
                return True
    with torch.no_grad():
        for t in range(T):
            out = actor(state.delay_state, state.delay_gso)
            sim.step_advance(out, state)
            if rewards is not None:
                rewards[:, t].copy_(sim.reward)
        if action is not None:
            action.copy_(out)
    return False


def policy_episode_rewards(env, learner, device, args, n_episodes):
    """Reward sums of `n_episodes` policy-only test episodes (the reference's test loop, gnn_dagger.py:194-203, run
    `n_test_episodes` times).  With this package's flocking environments all episodes run side by side: their reset
    states are drawn one after the other from the environment's own RNG -- exactly what the sequential loop would draw,
    since policy rollouts consume no randomness -- and every lane is stepped to the time limit by `policy_rollout`
    (one launch of the episode-resident kernel when the shape is covered).  Rewards agree with the one-at-a-time loop
    up to the closed-loop amplification of fp32 rounding (the summation order of the aggregation differs).
    Any other environment object falls back to the sequential loop."""
    import torch
    from ..envs.flocking import FlockingRelativeEnv, VecFlock
    from .state_with_delay import BatchedDelayState
    raw = getattr(env, 'env', None)
    steps = getattr(env, '_max_episode_steps', None)
    if n_episodes <= 0:
        return []
    if not isinstance(raw, FlockingRelativeEnv) or steps is None or learner.actor.ind_agg != 0:
        return [policy_episode_reward(env, learner, device, args) for _ in range(n_episodes)]
    p = raw.params
    sim = VecFlock(n_episodes, p, device)
    sim.reset(raw._rng)                                          # the sequential sampler's states (sample_initial_states)
    state = BatchedDelayState(device, n_episodes, learner.actor.k, learner.n_states, p.n_agents)
    state.push(sim.network, sim.features)
    per_step = torch.zeros((n_episodes, steps), device=sim.device, dtype=torch.float64)
    policy_rollout(learner.actor, sim, state, steps, rewards=per_step)
    return per_step.sum(dim=1).cpu().tolist()
