"""Vectorised, device-resident DAGGER: many episodes per GPU, nothing leaves HBM between the simulator, the state
update, the policy and the replay memory.  An extension of reference learner/gnn_dagger.py:126-243 (which steps ONE
environment through Python objects); the schedule is the reference's, applied per episode:

  * episode e (global index over rounds, ranks and lanes) uses beta_e = max(beta_coeff^(e+1), 0.5)   (:148)
  * every step stores (state, expert label) and steps the env with the expert w.p. beta_e else the policy  (:156-178)
  * after a round of `n_envs` episodes: `updates_per_step` updates per episode of the round            (:182-188)
  * final statistics = mean / std of `n_test_episodes` policy-only episode rewards                     (:221-237)

`DeviceReplay` keeps the reference's ring + sample-without-replacement semantics (replay_buffer.py:6-49, Python's
`random` RNG) but stores dense states in HBM: 10,000 transitions of N=100,K=3 are 1.3 GB of 288 GB.
Ranks shard episodes (parallel.py); the only collective is the flat-gradient all-reduce inside gradient_step.
"""
import random

import numpy as np
import torch

from .. import parallel
from ..envs import FlockParams, VecFlock
from ..envs.flocking import _REGISTRY
from .rollouts import policy_rollout
from .gnn_dagger import DAGGER
from .state_with_delay import BatchedDelayState


class DeviceReplay(object):
    """Ring buffer of transitions resident on the device (state = delay_state + delay_gso, label = expert action)."""

    def __init__(self, max_size, K, F, N, n_a, device):
        self.max_size = max_size
        kw = dict(device=device, dtype=torch.float32)
        self.delay_state = torch.empty((max_size, K, F, N), **kw)
        self.delay_gso = torch.empty((max_size, K, N, N), **kw)
        self.action = torch.empty((max_size, 1, n_a, N), **kw)
        self.curr_size = 0
        self.position = 0
        self.device = device

    def insert_batch(self, delay_state, delay_gso, action):
        """Append B transitions (oldest overwritten), same order as B consecutive `insert` calls."""
        B = delay_state.shape[0]
        idx = (torch.arange(B, device=self.device) + self.position) % self.max_size
        self.delay_state.index_copy_(0, idx, delay_state)
        self.delay_gso.index_copy_(0, idx, delay_gso)
        self.action.index_copy_(0, idx, action)
        self.position = (self.position + B) % self.max_size
        self.curr_size = min(self.max_size, self.curr_size + B)

    def sample(self, num_samples, out=None):
        """Without replacement, Python `random` RNG (reference replay_buffer.py:40).  `out` = (X, G, Y) destination
        tensors (e.g. the static buffers of the HIP-graph update): the gather is then the only copy of the batch."""
        ids = random.sample(range(self.curr_size), num_samples)
        idx = torch.tensor(ids, device=self.device, dtype=torch.long)
        if out is not None:
            torch.index_select(self.delay_state, 0, idx, out=out[0])
            torch.index_select(self.delay_gso, 0, idx, out=out[1])
            torch.index_select(self.action, 0, idx, out=out[2])
            return out
        return (self.delay_state.index_select(0, idx), self.delay_gso.index_select(0, idx),
                self.action.index_select(0, idx))

    def clear(self):
        self.curr_size = 0
        self.position = 0


class IndexedUpdates(object):
    """A round of DAGGER updates with no per-update host work: the minibatch indices of the whole round are uploaded once,
    and every update is one replay of a two-launch HIP graph (mgp_train_step_indexed) that gathers its batch from the
    replay arrays inside the kernel, applies Adam, advances the device-side update cursor and files its loss.
    Sampling is the reference's (`random.sample` without replacement per update, replay_buffer.py:40)."""

    def __init__(self, learner, memory, batch_size, max_updates):
        import ctypes
        from .. import _lib
        actor, opt = learner.actor, learner.actor_optim
        dev = opt.flat.device
        dims = tuple(actor.layers)
        self.cdims = (ctypes.c_int * len(dims))(*dims)
        self.nl, self.K, self.N, self.B = actor.n_layers, actor.k, learner.n_agents, batch_size
        self.learner, self.memory, self.cap = learner, memory, max_updates
        L = _lib.lib()
        self.idx = torch.zeros((max_updates, batch_size), device=dev, dtype=torch.long)
        self.cursor = torch.zeros((1,), device=dev, dtype=torch.int32)
        self.loss_hist = torch.zeros((max_updates,), device=dev, dtype=torch.float32)
        self.step_dev = opt.step_dev                # FlatAdam's shared device step counter
        self.ws = torch.zeros((L.mgp_train_workspace(self.cdims, self.nl, batch_size, self.K, self.N),), device=dev)
        self.graph = None

    @staticmethod
    def supported(learner, batch_size, N):
        import ctypes
        from .. import _lib
        if not (learner.use_graphed_update and learner.use_train_step) or parallel.is_distributed():
            return False
        dims = tuple(learner.actor.layers)
        cd = (ctypes.c_int * len(dims))(*dims)
        return learner.actor.ind_agg == 0 and bool(_lib.lib().mgp_train_supported(cd, learner.actor.n_layers, batch_size,
                                                                                 learner.actor.k, N))

    def _enqueue(self):
        from .. import _lib, ops
        L, o, m = _lib.lib(), self.learner.actor_optim, self.memory
        _lib.check(L.mgp_train_step_indexed(
            ops._ptr(m.delay_state), ops._ptr(m.delay_gso), ops._ptr(m.action), self.idx.data_ptr(), self.cursor.data_ptr(),
            ops._ptr(self.loss_hist), self.cap, ops._ptr(o.flat), ops._ptr(o.flat_grad), ops._ptr(o.m), ops._ptr(o.v),
            self.cdims, self.nl, o.lr, o.betas[0], o.betas[1], o.eps, self.step_dev.data_ptr(), ops._ptr(self.ws),
            self.B, self.K, self.N, ops._stream()), 'mgp_train_step_indexed')

    def run(self, ids):
        """ids: one list of `batch_size` replay rows per update.  Returns the sum of the updates' losses (device tensor)."""
        U = len(ids)
        assert 0 < U <= self.cap
        opt = self.learner.actor_optim
        self.idx[:U].copy_(torch.tensor(ids, dtype=torch.long), non_blocking=False)
        self.cursor.zero_()
        if self.graph is None:
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._enqueue()
        for _ in range(U):
            self.graph.replay()
        opt.step_count += U
        return self.loss_hist[:U].sum()


def _params_from_args(args):
    env_cls = _REGISTRY.get(args.get('env'), None)
    variant = getattr(env_cls, 'variant', {}) if env_cls is not None else {}
    kw = dict(n_agents=args.getint('n_agents'), comm_radius=args.getfloat('comm_radius'),
              v_max=args.getfloat('v_max'), v_bias=args.getfloat('v_max'))
    # same reset distribution as the gym-style environment built from the same cfg section (FlockParams.init_mode 'auto':
    # disc sampling up to N = 100, jittered lattice beyond), so `alg = dagger_vec` and `alg = dagger` statistics are
    # comparable; `init_mode = grid` in the cfg selects the lattice explicitly (cheaper resets for many lanes)
    if args.get('init_mode') is not None:
        kw['init_mode'] = args.get('init_mode')
    if args.get('dt') is not None:
        kw['dt'] = args.getfloat('dt')
    kw.update(variant)
    if args.get('link_drop') is not None:
        kw['link_drop'] = args.getfloat('link_drop')
    if kw.get('link_drop', 0.0) > 0.0:
        kw['link_seed'] = args.getint('seed', fallback=0) & 0xFFFFFFFF
    return FlockParams(**kw)


def _label(expert):
    """(B,N,nA) -> (B,1,nA,N)   (reference gnn_dagger.py:174-176, batched)"""
    return expert.permute(0, 2, 1).unsqueeze(1).contiguous()


def evaluate(learner, sim, state, n_episodes, steps):
    """Policy-only episodes in lanes of sim.B; returns the list of per-episode reward sums."""
    rewards = []
    while len(rewards) < n_episodes:
        sim.reset(np.random)
        state.reset()
        state.push(sim.network, sim.features)
        # whole episodes in one launch of the episode-resident kernel when the shape is covered (else step by step)
        per_step = torch.zeros((sim.B, steps), device=sim.device, dtype=torch.float64)
        policy_rollout(learner.actor, sim, state, steps, rewards=per_step)
        total = per_step.sum(dim=1)
        rewards += total.cpu().tolist()
    return rewards[:n_episodes]


def train_dagger_vec(args, device, n_envs=64, episode_steps=None):
    """Returns {'mean','std'} like train_dagger.  `n_envs` parallel episodes per rank."""
    device = torch.device(device)
    p = _params_from_args(args)
    N, K, F, n_a = p.n_agents, args.getint('k'), args.getint('n_states'), args.getint('n_actions')
    T = episode_steps or p.max_episode_steps
    learner = DAGGER(device, args)
    # The reference's ring of `buffer_size` (10,000) transitions holds its 20 most recent WHOLE episodes.  Here n_envs
    # episodes advance in lock step, so a ring shorter than one round (n_envs * T transitions) would keep only the last
    # steps of every episode -- the already-flocked states -- and the policy would never see a start-up state.  The ring
    # therefore holds at least one full round (128 KB per transition at N = 100, K = 3: 4 GB of the 288 GB for 64 x 500).
    memory = DeviceReplay(max(args.getint('buffer_size'), n_envs * T), K, F, N, n_a, device)
    sim = VecFlock(n_envs, p, device, with_expert=True)
    state = BatchedDelayState(device, n_envs, K, F, N)
    batch_size = args.getint('batch_size')
    beta_coeff = args.getfloat('beta_coeff')
    updates_per_step = args.getint('updates_per_step')
    n_train_episodes = args.getint('n_train_episodes')
    n_test_episodes = args.getint('n_test_episodes')
    debug = args.getboolean('debug')
    rank, world = parallel.rank(), parallel.world_size()
    rounds = (n_train_episodes + n_envs * world - 1) // (n_envs * world)
    updates = 0
    indexed = None
    for rd in range(rounds):
        e0 = (rd * world + rank) * n_envs
        beta = np.maximum(beta_coeff ** (np.arange(e0, e0 + n_envs) + 1.0), 0.5)
        sim.reset(np.random)
        state.reset()
        state.push(sim.network, sim.features)
        for _ in range(T):
            expert = sim.controller()                                     # by-product of the last sim kernel
            memory.insert_batch(state.delay_state, state.delay_gso, _label(expert))
            with torch.no_grad():
                policy = learner.actor(state.delay_state, state.delay_gso)   # (B,1,nA,N)
            use_expert = torch.from_numpy(np.random.binomial(1, beta).astype(np.bool_)).to(device)
            action = torch.where(use_expert.view(-1, 1, 1), expert, policy[:, 0].permute(0, 2, 1)).contiguous()
            A_dst, X_dst = state.next_slots()
            sim.step(action, A_out=A_dst, feat_out=X_dst)
            state.advance()
        loss_sum = 0.0
        if memory.curr_size > batch_size and IndexedUpdates.supported(learner, batch_size, N):
            if indexed is None:
                indexed = IndexedUpdates(learner, memory, batch_size, updates_per_step * n_envs)
            ids = [random.sample(range(memory.curr_size), batch_size) for _ in range(updates_per_step * n_envs)]
            loss_sum = float(indexed.run(ids).item())
            updates += len(ids)
        elif memory.curr_size > batch_size:
            bufs = learner.graphed_buffers(batch_size, N)        # None: eager / distributed updates
            loss_dev = torch.zeros((1,), device=device)
            for _ in range(updates_per_step * n_envs):
                xs, gs, ys = memory.sample(batch_size, out=bufs)
                if bufs is not None:                              # no host sync per update: losses add up on the device
                    loss_dev += learner.gradient_step_tensors(xs, gs, ys, sync=False)
                else:
                    loss_sum += learner.gradient_step_tensors(xs, gs, ys)
                updates += 1
            loss_sum += float(loss_dev.item())
        if debug and rank == 0:
            print("Round: {}, episodes: {}, updates: {}, policy loss: {}".format(rd, (rd + 1) * n_envs * world,
                                                                                   updates, loss_sum))
    lo, hi = parallel.shard_range(n_test_episodes)
    n_local = max(1, hi - lo) if world > 1 else n_test_episodes
    rewards = parallel.all_gather_floats(evaluate(learner, sim, state, n_local, T))
    if debug and args.get('fname') and rank == 0:            # reference gnn_dagger.py:239-240
        learner.save_model(args.get('env'), suffix=args.get('fname'))
    return {'mean': float(np.mean(rewards)), 'std': float(np.std(rewards)), 'learner': learner, 'updates': updates}
