"""numpy restatement of the DAGGER training LOOP and its replay memory (TEST ORACLE; never imported by the product).

Follows reference learner/gnn_dagger.py:126-243 (`train_dagger`) and learner/replay_buffer.py:6-49 (`ReplayBuffer`):

  * replay memory: list that grows to max_size, then a ring overwriting position; `random.sample`   replay_buffer.py:21-41
  * beta = max(beta * beta_coeff, 0.5), a RUNNING product updated once per episode                  gnn_dagger.py:148
  * per step: expert = env.env.controller(); np.random.binomial(1, beta) > 0 -> expert drives,
    else select_action(state).cpu().numpy(); env.step; next state from prev state                   :156-165
  * label = expert transposed to (1,1,nA,N), stored with the state BEFORE the step                  :174-178
  * after the episode, if curr_size > batch_size: updates_per_step x (sample, gradient_step)        :182-188
  * if i % test_interval == 0 and debug: n_test_episodes policy-only episodes, one printed line     :190-219
  * final n_test_episodes policy-only episodes -> {'mean','std'}; env.close()                       :221-243

Pinned by tests/golden/train_dagger_trace.npz: the reference's own loop recorded in the build container
(tests/golden/gen_golden.py::gen_train_trace) on tests/fake_env.FakeFlockEnv; tests/test_oracle_trace.py replays this
restatement against it.  The arithmetic underneath is oracle/actor.py, state.py, dagger.py (pinned separately).
The reference's default Conv2d initialisation (torch RNG) is not restated: initial weights are an argument.
"""
import random
from collections import namedtuple

import numpy as np

from . import actor as _actor
from . import dagger as _dagger
from . import state as _state

Transition = namedtuple('Transition', ('state', 'action', 'done', 'next_state', 'reward'))     # replay_buffer.py:4


class ReplayBuffer(object):
    """replay_buffer.py:6-49"""

    def __init__(self, max_size=1000):
        self.buffer = []
        self.max_size = max_size
        self.curr_size = 0
        self.position = 0

    def insert(self, sample):
        if self.curr_size < self.max_size:                       # :27-29
            self.buffer.append(None)
            self.curr_size = self.curr_size + 1
        self.buffer[self.position] = Transition(*sample)         # :31
        self.position = (self.position + 1) % self.max_size      # :32

    def sample(self, num_samples):
        return random.sample(self.buffer, num_samples)           # :40

    def clear(self):
        self.buffer = []
        self.curr_size = 0
        self.position = 0


class State(object):
    """state_with_delay.py:6-53 for one episode (B = 1), fp32."""

    def __init__(self, args, env_state, prev_state=None):
        n_states, n_agents, k = args.getint('n_states'), args.getint('n_agents'), args.getint('k')
        values, network = env_state
        assert values.shape == (n_agents, n_states) and network.shape == (n_agents, n_agents)      # :24-25
        v, a = _state.cast_env_state(values, network)
        Gp = prev_state.delay_gso if prev_state is not None else None
        Xp = prev_state.delay_state if prev_state is not None else None
        self.delay_gso, self.delay_state = _state.gso_update(a[0], Gp, v[0], Xp, k)


class DAGGER(object):
    """gnn_dagger.py:18-96 on numpy: fp32 forward / backward / Adam (oracle/actor.py, dagger.py)."""

    def __init__(self, args, weights, biases):
        self.n_agents, self.n_actions = args.getint('n_agents'), args.getint('n_actions')
        self.lr = args.getfloat('actor_lr')
        self.W = [np.array(w, dtype=np.float32) for w in weights]
        self.b = [np.array(b_, dtype=np.float32) for b_ in biases]
        self.m = [np.zeros_like(p) for pair in zip(self.W, self.b) for p in pair]
        self.v = [np.zeros_like(p) for pair in zip(self.W, self.b) for p in pair]
        self.t = 0
        self.ind_agg = 0                                                                         # :43

    def select_action(self, state):
        out = _actor.forward(state.delay_state, state.delay_gso, self.W, self.b, self.ind_agg, dtype=np.float32)
        return _dagger.action_from_output(out).astype(np.float32)                                # :66-68

    def gradient_step(self, batch):
        G = np.concatenate([s.delay_gso for s in batch.state])                                   # :83
        X = np.concatenate([s.delay_state for s in batch.state])                                 # :84
        Y = np.concatenate(batch.action)                                                         # :86
        self.t += 1
        loss, self.W, self.b, self.m, self.v, _ = _dagger.gradient_step(X, G, Y, self.W, self.b, self.ind_agg,
                                                                        self.m, self.v, self.t, self.lr)
        return loss

    def state_dict(self):
        d = {}
        for i, (w, b_) in enumerate(zip(self.W, self.b)):
            d['conv_layers__%d__weight' % i] = w
            d['conv_layers__%d__bias' % i] = b_
        return d


def _test_episodes(env, args, learner, n):
    """gnn_dagger.py:192-203 / :222-232"""
    rewards = []
    for _ in range(n):
        ep_reward = 0
        state = State(args, env.reset(), None)
        done = False
        while not done:
            action = learner.select_action(state)
            next_state, reward, done, _ = env.step(action)
            state = State(args, next_state, state)
            ep_reward += reward
        rewards.append(ep_reward)
    return rewards


def train_dagger(env, args, make_learner, replay_cls=ReplayBuffer):
    """gnn_dagger.py:126-243.  `make_learner()` -> DAGGER-like object (initial weights are the caller's)."""
    debug = args.getboolean('debug')
    memory = replay_cls(max_size=args.getint('buffer_size'))
    learner = make_learner()
    n_a, n_agents, batch_size = args.getint('n_actions'), args.getint('n_agents'), args.getint('batch_size')
    n_train_episodes, beta_coeff = args.getint('n_train_episodes'), args.getfloat('beta_coeff')
    test_interval, n_test_episodes = args.getint('test_interval'), args.getint('n_test_episodes')
    total_numsteps, updates, beta = 0, 0, 1
    stats = {'mean': -1.0 * np.inf, 'std': 0}
    for i in range(n_train_episodes):
        beta = max(beta * beta_coeff, 0.5)                                                       # :148
        state = State(args, env.reset(), None)
        done = False
        policy_loss_sum = 0
        while not done:
            optimal_action = env.env.controller()                                                # :156
            if np.random.binomial(1, beta) > 0:                                                  # :157
                action = optimal_action
            else:
                action = learner.select_action(state)
            next_state, reward, done, _ = env.step(action)                                       # :163
            next_state = State(args, next_state, state)
            total_numsteps += 1
            label = _dagger.label_from_action(np.asarray(optimal_action, dtype=np.float32))      # :174-176
            memory.insert(Transition(state, label, float(not done), next_state, float(reward)))  # :178
            state = next_state
        if memory.curr_size > batch_size:                                                        # :182
            for _ in range(args.getint('updates_per_step')):
                transitions = memory.sample(batch_size)
                batch = Transition(*zip(*transitions))
                policy_loss_sum += learner.gradient_step(batch)
                updates += 1
        if i % test_interval == 0 and debug:                                                     # :190
            mean_reward = np.mean(_test_episodes(env, args, learner, n_test_episodes))
            print("Episode: {}, updates: {}, total numsteps: {}, reward: {}, policy loss: {}".format(
                i, updates, total_numsteps, mean_reward, policy_loss_sum))
    test_rewards = _test_episodes(env, args, learner, n_test_episodes)                           # :221-232
    stats['mean'] = np.mean(test_rewards)
    stats['std'] = np.std(test_rewards)
    env.close()                                                                                  # :242
    return stats
